// Frame compaction of the encoder path (round 5): a minibatch of B utterances padded to T frames holds sum(len) valid
// frames and B T - sum(len) padding frames whose inputs are zero (utils/datagen.py:104,129-142 pads with zeros).  Every padding frame
// goes through the dense encoder as the SAME row -- enc(0) -- and, the encoder being row-wise, their gradients only ever enter the
// parameter gradients as a sum: dW_l += a_{l-1}(0)^T (sum of their dZ_l rows), and that sum propagates down the layers like one row
// (dZ_{l-1} = act'(a_{l-1}(0)) * (dZ_l W_l^T) is linear in dZ_l for a fixed mask).  So the encoder GEMMs run over
//     Nc = sum(len) + 1 rows:  the valid frames, utterance after utterance, and ONE zero-input row Z = Nc - 1
// instead of B T (65 % of them at lengths ~ U[12, 40]).  What stays padded: everything from the delta layer up (time-major, B rows per
// step).  The delta layer is where the two layouts meet, and its kernels do the conversion on the way (elementwise.hip, round 6): the
// forward kernel reads the compact encoder output through comp_of_full (every padding frame sees row Z), the backward kernel stores
// a valid frame's gradient at its compact row and sums the padding frames' per utterance; compact_pad_finish() adds those sums up
// into row Z.  (Round 5 ran separate expand / compact-and-sum passes over B T rows: 112 us of launches per step.)
//   full row  r = b T + t  ->  compact row  comp_of_full[r] = prefix[b] + t (t < len[b]) | Z
//   compact row c < Z      ->  full row     full_of_comp[c];   full_of_comp[Z] = -1
#include "adn_common.h"
#include <algorithm>

namespace adn {

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void compact_maps_kernel(const int32_t* __restrict__ lens, const int32_t* __restrict__ prefix, int B, int T,
                                                           int Z, int32_t* __restrict__ comp_of_full, int32_t* __restrict__ full_of_comp) {
    const int total = B * T;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < total; r += gridDim.x * 256) {
        const int b = r / T, t = r - b * T;
        const bool valid = t < lens[b];
        const int c = valid ? prefix[b] + t : Z;
        comp_of_full[r] = c;
        if (valid) full_of_comp[c] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) full_of_comp[Z] = -1;
}

// dst[c][0 .. cols16) = src[full_of_comp[c]][...] (16-byte pieces), the zero row for c = Z
__global__ __launch_bounds__(256) void gather_rows16_kernel(const u32x4* __restrict__ src, int ld_src16, u32x4* __restrict__ dst, int ld_dst16,
                                                            const int32_t* __restrict__ full_of_comp, int Nc, int cols16) {
    const int64_t total = (int64_t)Nc * cols16;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e / cols16), q = (int)(e - (int64_t)c * cols16);
        const int r = full_of_comp[c];
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r >= 0) v = src[(size_t)r * ld_src16 + q];
        dst[(size_t)c * ld_dst16 + q] = v;
    }
}

// the same gather from FLOAT32 rows, converting on the way: dst_hi[c] = bf16(src[full_of_comp[c]]) and -- planes -- dst_lo = bf16(x - hi);
// 8 elements (two float4 in, 16 bytes per plane out) per thread
typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
// (blockIdx.y = matrix: the S stream inputs of a call in one launch)
struct GatherF32Table { const float4* src[4]; cbf16x8* hi[4]; cbf16x8* lo[4]; };
__global__ __launch_bounds__(256) void gather_rows_f32_kernel(const GatherF32Table t, int ld_src4, int ld_dst8, const int32_t* __restrict__ full_of_comp,
                                                              int Nc, int cols8) {
    const float4* __restrict__ src = t.src[0]; cbf16x8* __restrict__ dst_hi = t.hi[0]; cbf16x8* __restrict__ dst_lo = t.lo[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) if ((int)blockIdx.y == k) { src = t.src[k]; dst_hi = t.hi[k]; dst_lo = t.lo[k]; }
    const int64_t total = (int64_t)Nc * cols8;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e / cols8), q = (int)(e - (int64_t)c * cols8);
        const int r = full_of_comp[c];
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (r >= 0) {
            const float4 a = src[(size_t)r * ld_src4 + 2 * q], b = src[(size_t)r * ld_src4 + 2 * q + 1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        }
        cbf16x8 h, l;
#pragma unroll
        for (int k = 0; k < 8; ++k) { h[k] = (__bf16)v[k]; l[k] = (__bf16)(v[k] - (float)h[k]); }
        dst_hi[(size_t)c * ld_dst8 + q] = h;
        if (dst_lo) dst_lo[(size_t)c * ld_dst8 + q] = l;
    }
}
__global__ __launch_bounds__(256) void check_padding32_kernel(const float4* __restrict__ src, int ld_src4, const int32_t* __restrict__ comp_of_full,
                                                              int N, int cols4, int Z, int* __restrict__ flag, int bit) {
    const int64_t total = (int64_t)N * cols4;
    bool bad = false;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / cols4), q = (int)(e - (int64_t)r * cols4);
        if (comp_of_full[r] != Z) continue;
        const float4 v = src[(size_t)r * ld_src4 + q];
        bad = bad || v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f;
    }
    if (bad) atomicOr(flag, bit);
}

// the promise behind an announcement, checked: every 16-byte piece of a PADDING row (comp_of_full[r] == Z) of the 16-bit operand must be
// zero (sign bits aside); a piece that is not raises bit `bit` of *flag
__global__ __launch_bounds__(256) void check_padding16_kernel(const u32x4* __restrict__ src, int ld_src16, const int32_t* __restrict__ comp_of_full,
                                                              int N, int cols16, int Z, int* __restrict__ flag, int bit) {
    const int64_t total = (int64_t)N * cols16;
    unsigned bad = 0u;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / cols16), q = (int)(e - (int64_t)r * cols16);
        if (comp_of_full[r] != Z) continue;
        const u32x4 v = src[(size_t)r * ld_src16 + q];
        bad |= (v[0] | v[1] | v[2] | v[3]) & 0x7FFF7FFFu;
    }
    if (bad) atomicOr(flag, bit);
}

// comp[Z][q] = sum over the utterances' padding sums (pad[b][cols], written by the delta layer's backward kernel: elementwise.hip
// delta_bwd_body), in one fixed order: lane ts adds utterances ts, ts + 32, ... and the 32 lane sums are added in lane order; the
// row's pad columns become zero; the row's 16-bit copies (bf16 copy / hi + lo planes) are written with it.  blockIdx.y = job.
struct PadFinishTable { PadFinishJob j[kMaxPadFinishJobs]; };
__global__ __launch_bounds__(1024) void compact_pad_finish_kernel(const PadFinishTable tab) {
    __shared__ float red[32][33];
    PadFinishJob j = tab.j[0];
    if (blockIdx.y == 1) j = tab.j[1];
    if (blockIdx.y == 2) j = tab.j[2];
    if (blockIdx.y == 3) j = tab.j[3];
    const int fl = threadIdx.x & 31, ts = threadIdx.x >> 5;        // 32 column lanes x 32 utterance lanes
    const int q = blockIdx.x * 32 + fl;
    if ((int)blockIdx.x * 32 >= j.ld) return;
    float s = 0.f;
    if (q < j.cols) for (int k = ts; k < j.nparts; k += 32) s += j.partial[(size_t)k * j.cols + q];
    red[ts][fl] = s;
    __syncthreads();
    if (ts == 0 && q < j.ld) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += red[k][fl];
        const size_t o = (size_t)j.zrow * j.ld + q;
        j.comp[o] = t;
        if (j.c16) {
            const __bf16 h = (__bf16)t;
            reinterpret_cast<__bf16*>(j.c16)[o] = h;
            if (j.c16lo) reinterpret_cast<__bf16*>(j.c16lo)[o] = (__bf16)(t - (float)h);
        }
    }
}

int grid_for_elems(int64_t total) { return (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 16384)); }

}  // namespace

int compact_build_maps(const int32_t* d_lens, const int32_t* d_prefix, int B, int T, int Z, int32_t* comp_of_full, int32_t* full_of_comp,
                       hipStream_t s) {
    hipLaunchKernelGGL(compact_maps_kernel, dim3(grid_for_elems((int64_t)B * T)), dim3(256), 0, s, d_lens, d_prefix, B, T, Z, comp_of_full,
                       full_of_comp);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// the same for up to kMaxGatherJobs matrices of ONE geometry in one launch (blockIdx.y = matrix): the S stream inputs of a call -- and
// their lo planes -- are gathered together
struct GatherJobTable { const u32x4* src[kMaxGatherJobs]; u32x4* dst[kMaxGatherJobs]; };
__global__ __launch_bounds__(256) void gather_rows16_batch_kernel(const GatherJobTable t, int ld_src16, int ld_dst16, const int32_t* __restrict__ full_of_comp,
                                                                  int Nc, int cols16) {
    const u32x4* __restrict__ src = t.src[0]; u32x4* __restrict__ dst = t.dst[0];
#pragma unroll
    for (int k = 1; k < kMaxGatherJobs; ++k) if ((int)blockIdx.y == k) { src = t.src[k]; dst = t.dst[k]; }
    const int64_t total = (int64_t)Nc * cols16;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e / cols16), q = (int)(e - (int64_t)c * cols16);
        const int r = full_of_comp[c];
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r >= 0) v = src[(size_t)r * ld_src16 + q];
        dst[(size_t)c * ld_dst16 + q] = v;
    }
}
int compact_gather_rows16_batch(const void* const* src, void* const* dst, int n, int ld_src, int ld_dst, const int32_t* full_of_comp, int Nc, int cols,
                                hipStream_t s) {
    ADN_CHECK(n >= 1 && n <= kMaxGatherJobs && cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0, ADN_ERR_INVALID,
              "compact_gather_rows16_batch: 1..8 matrices, rows of whole 16-byte pieces");
    if (n == 1) return compact_gather_rows16(src[0], ld_src, dst[0], ld_dst, full_of_comp, Nc, cols, s);
    GatherJobTable t;
    for (int k = 0; k < kMaxGatherJobs; ++k) {
        const int q = k < n ? k : 0;
        ADN_CHECK(((uintptr_t)src[q] % 16) == 0 && ((uintptr_t)dst[q] % 16) == 0, ADN_ERR_INVALID, "compact_gather_rows16_batch: 16-byte aligned matrices");
        t.src[k] = static_cast<const u32x4*>(src[q]); t.dst[k] = static_cast<u32x4*>(dst[q]);
    }
    const int grid = std::max(1, grid_for_elems((int64_t)Nc * (cols / 8)) / 2);
    hipLaunchKernelGGL(gather_rows16_batch_kernel, dim3(grid, n), dim3(256), 0, s, t, ld_src / 8, ld_dst / 8, full_of_comp, Nc, cols / 8);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int compact_gather_rows16(const void* src, int ld_src, void* dst, int ld_dst, const int32_t* full_of_comp, int Nc, int cols, hipStream_t s) {
    ADN_CHECK(cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, ADN_ERR_INVALID,
              "compact_gather_rows16: rows of whole 16-byte pieces");
    hipLaunchKernelGGL(gather_rows16_kernel, dim3(grid_for_elems((int64_t)Nc * (cols / 8))), dim3(256), 0, s, static_cast<const u32x4*>(src),
                       ld_src / 8, static_cast<u32x4*>(dst), ld_dst / 8, full_of_comp, Nc, cols / 8);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int compact_gather_rows_f32_batch(const float* const* src, void* const* dst_hi, void* const* dst_lo, int n, int ld_src, int ld_dst,
                                  const int32_t* full_of_comp, int Nc, int cols, hipStream_t s) {
    ADN_CHECK(n >= 1 && n <= 4 && cols % 8 == 0 && ld_src % 4 == 0 && ld_dst % 8 == 0, ADN_ERR_INVALID,
              "compact_gather_rows_f32: 1..4 matrices, rows of whole 16-byte pieces");
    GatherF32Table t;
    for (int k = 0; k < 4; ++k) {
        const int q = k < n ? k : 0;
        ADN_CHECK(((uintptr_t)src[q] % 16) == 0 && ((uintptr_t)dst_hi[q] % 16) == 0 && ((uintptr_t)dst_lo[q] % 16) == 0 && dst_hi[q] &&
                  (dst_lo[q] == nullptr) == (dst_lo[0] == nullptr), ADN_ERR_INVALID, "compact_gather_rows_f32: 16-byte aligned matrices, planes for all or none");
        t.src[k] = reinterpret_cast<const float4*>(src[q]); t.hi[k] = static_cast<cbf16x8*>(dst_hi[q]); t.lo[k] = static_cast<cbf16x8*>(dst_lo[q]);
    }
    const int grid = std::max(1, grid_for_elems((int64_t)Nc * (cols / 8)) / (n > 1 ? 2 : 1));
    hipLaunchKernelGGL(gather_rows_f32_kernel, dim3(grid, n), dim3(256), 0, s, t, ld_src / 4, ld_dst / 8, full_of_comp, Nc, cols / 8);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}
int compact_gather_rows_f32(const float* src, int ld_src, void* dst_hi, void* dst_lo, int ld_dst, const int32_t* full_of_comp, int Nc, int cols,
                            hipStream_t s) {
    return compact_gather_rows_f32_batch(&src, &dst_hi, &dst_lo, 1, ld_src, ld_dst, full_of_comp, Nc, cols, s);
}

int compact_check_padding32(const float* src, int ld_src, const int32_t* comp_of_full, int N, int cols, int Z, int* flag, int bit, hipStream_t s) {
    ADN_CHECK(cols % 4 == 0 && ld_src % 4 == 0 && ((uintptr_t)src % 16) == 0, ADN_ERR_INVALID, "compact_check_padding32: rows of whole 16-byte pieces");
    hipLaunchKernelGGL(check_padding32_kernel, dim3(grid_for_elems((int64_t)N * (cols / 4))), dim3(256), 0, s, reinterpret_cast<const float4*>(src),
                       ld_src / 4, comp_of_full, N, cols / 4, Z, flag, bit);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int compact_check_padding16(const void* src, int ld_src, const int32_t* comp_of_full, int N, int cols, int Z, int* flag, int bit, hipStream_t s) {
    ADN_CHECK(cols % 8 == 0 && ld_src % 8 == 0 && ((uintptr_t)src % 16) == 0, ADN_ERR_INVALID, "compact_check_padding16: rows of whole 16-byte pieces");
    hipLaunchKernelGGL(check_padding16_kernel, dim3(grid_for_elems((int64_t)N * (cols / 8))), dim3(256), 0, s, static_cast<const u32x4*>(src),
                       ld_src / 8, comp_of_full, N, cols / 8, Z, flag, bit);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int compact_pad_finish(const PadFinishJob* jobs, int n, hipStream_t s) {
    for (int k0 = 0; k0 < n; k0 += kMaxPadFinishJobs) {
        const int nn = std::min(kMaxPadFinishJobs, n - k0);
        PadFinishTable tab;
        int ldmax = 0;
        for (int k = 0; k < kMaxPadFinishJobs; ++k) { tab.j[k] = jobs[k0 + std::min(k, nn - 1)]; if (k < nn) ldmax = std::max(ldmax, tab.j[k].ld); }
        hipLaunchKernelGGL(compact_pad_finish_kernel, dim3(cdiv(ldmax, 32), nn), dim3(1024), 0, s, tab);
    }
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

}  // namespace adn
