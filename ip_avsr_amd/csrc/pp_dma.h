// Device helpers shared by the persistent LDS-DMA GEMM kernels (gemm_bf16.hip: gemm_bf16_pp_kernel / gemm_bf16_w4_kernel;
// gemm_x3f.hip: gemm_x3f_kernel): the LDS-DMA wave-instruction in its forms, hand-counted waits, group selection.
#pragma once
#include "gemm_common.h"
#include <cstdint>

namespace adn {

typedef __attribute__((address_space(3))) void lds_void_t;

// One LDS-DMA wave-instruction: lane l fetches 16 bytes from its own global address into LDS byte address
// lds_dst + 16 l (lds_dst wave-uniform, in an SGPR).  Inline asm on purpose: hipcc orders every ds_read behind a
// pending __builtin_amdgcn_global_load_lds with s_waitcnt vmcnt(0), which would drain the stages these kernels keep in
// flight; the waits are counted by hand instead.  M0 (the DMA destination base) is compiler-reserved, hence saved and
// restored (cdna_hip_programming.md, LDS-DMA recipe).
__device__ __forceinline__ void glds16(const void* g, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
// the same with the address split into a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset: a K-step then
// advances ONE scalar per operand instead of a 64-bit VGPR pointer per piece.  LDS destination = lds_dst + IMM.
template <int IMM> __device__ __forceinline__ void glds16_s(unsigned off, const char* base, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %3, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(base), "s"(lds_dst), "n"(IMM) : "memory", "scc");
}

// the four-wave kernel's form: its base may arrive through v_readfirstlane right ahead of the statement (VALU-written SGPR ->
// VMEM reading it as its scalar base: 5 wait states; the two SALU instructions and the nop make them up)
template <int IMM> __device__ __forceinline__ void glds16_su(unsigned off, const char* base, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %3, %4\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(base), "s"(lds_dst), "n"(IMM) : "memory", "scc");
}

// group `g` of a grouped launch by value (explicit selects: a dynamic index into the kernel-argument block makes hipcc
// copy the whole block to scratch, and scratch traffic would sit in the vmcnt queue these kernels count by hand)
__device__ __forceinline__ GemmGroup pick_group(const GemmParams& p, int g) {
    GemmGroup r = p.grp[0];
    if (g == 1) r = p.grp[1];
    if (g == 2) r = p.grp[2];
    if (g == 3) r = p.grp[3];
    return r;
}

// a wave-uniform pointer the compiler may have parked in VGPRs, back in an SGPR pair (asm "s" operands)
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// C (+)= sum of the split-K partial slabs of a persistent-kernel launch (gemm_bf16.hip)
void launch_splitk_reduce(const GemmParams& p, int splits, hipStream_t s);

}  // namespace adn
