#!/bin/bash
# The ping-pong GEMM kernels count their vmcnt queue by hand: a register spilled to scratch inside them would put uncounted memory
# operations into that queue.  Compiles gemm_bf16.hip to assembly and fails if any gemm_bf16_pp_kernel instantiation contains a
# scratch access.   bash profiles/scripts/check_pp_scratch.sh
set -e
cd "$(dirname "$0")/../../ip_avsr_amd/csrc"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I../../include --cuda-device-only -S gemm_bf16.hip -o /tmp/gemm_bf16_check.s
awk '/^_ZN3adn19gemm_bf16_pp_kernel.*:/{f=1; name=$1} f && /scratch_|buffer_(load|store)/{print name, $0; bad=1} /s_endpgm/{f=0} END{exit bad}' /tmp/gemm_bf16_check.s && echo "no scratch access in the ping-pong kernels"
