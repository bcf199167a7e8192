#!/bin/bash
# The ping-pong GEMM kernels count their vmcnt queue by hand: a register spilled to scratch inside them would put uncounted memory
# operations into that queue.  Compiles gemm_bf16.hip to assembly and fails if any gemm_bf16_pp_kernel instantiation contains a
# scratch access.   bash profiles/scripts/check_pp_scratch.sh
set -e
cd "$(dirname "$0")/../../ip_avsr_amd/csrc"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I../../include --cuda-device-only -S gemm_bf16.hip -o /tmp/gemm_bf16_check.s
awk '/^_ZN3adn19gemm_bf16_pp_kernel.*:/{f=1; name=$1} f && /scratch_|buffer_(load|store)/{print name, $0; bad=1} /s_endpgm/{f=0} END{exit bad}' /tmp/gemm_bf16_check.s && echo "no scratch access in the ping-pong kernels"
# ... and hipcc must not have put an s_waitcnt vmcnt(0) at the head of the interior K-step loop (it did until the epilogue ended
# with a compiler-visible vmcnt(0): the DMA ring then drained at every step)
awk '/^_ZN3adn19gemm_bf16_pp_kernel.*:/{name=$1} /Inner Loop Header: Depth=2/{f=14} f>0{ if ($0 ~ /s_waitcnt vmcnt\(0\)/) {print name, "vmcnt(0) at the interior loop head"; bad=1}; f--} END{exit bad}' /tmp/gemm_bf16_check.s && echo "no vmcnt(0) at the head of the interior K-step loops"
# the fused-plane kernels (gemm_x3f.hip) count their queue the same way
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I../../include --cuda-device-only -S gemm_x3f.hip -o /tmp/gemm_x3f_check.s
awk '/^_ZN3adn15gemm_x3f_kernel.*:/{f=1; name=$1} f && /scratch_|buffer_(load|store)/{print name, $0; bad=1} /s_endpgm/{f=0} END{exit bad}' /tmp/gemm_x3f_check.s && echo "no scratch access in the fused-plane kernels"
