// Shared by gemm_f32.hip and gemm_bf16.hip: kernel parameter block, activation helpers and the
// C-tile epilogue (bias, activation, act'(Y) product, accumulate / split-K atomics).
#pragma once
#include "adn_common.h"

namespace adn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// one problem of a grouped launch (gemm_bf16_pp_kernel): same shape, leading dimensions and flags, own buffers
struct GemmGroup {
    const void* A16; const void* B16;
    const void* A16lo; const void* B16lo;          // bf16x3 through planes (GemmParams::kseg > 0): the lo planes, layouts of A16 / B16
    void* C16lo;                                   // ... and the lo plane of the result, written beside C16 (the hi plane)
    float* C; void* C16;
    const void* Y16; const float* Y;
    const float* bias;
    float* colsum;
    // rectifier bit images (ping-pong kernel, 256 x 256 tiles, plain bf16): per tile and thread the 128 bits "C > 0" of the thread's
    // accumulator quads, [tile_m * tiles_n + tile_n][512] uint4.  Cbits: written by a rectify epilogue; Ybits: read by an
    // act'(Y) epilogue of the SAME tile grid instead of 32 eight-byte loads of Y16 per thread and tile
    void* Cbits; const void* Ybits;
};

struct GemmParams {
    int M, N, K;
    const float* A; int lda;
    const float* B; int ldb;
    float* C;       int ldc;
    const void* A16; const void* B16; void* C16;   // optional bf16 shadow copies (same ld / offsets)
    float* colsum; int colsum_ld;                  // optional: per-m-tile column sums of the final C, [tiles_m][colsum_ld]
    const void* Y16;                               // optional bf16 copy of Y (used instead of Y when set)
    const float* bias;
    const float* Y; int ldy;
    int act, act_grad, accumulate, atomic;
    int k_chunk;
    int tiles_m, tiles_n;
    int panel_n;        // tiles are enumerated panel-major: panels of `panel_n` tile columns, m outer / n inner inside
    // ping-pong kernel only
    int ngroups;
    int xcd_slices;     // split-K: K-slice = function of the workgroup's XCD (see the kernel)
    int one_barrier;    // ping-pong kernel: one s_barrier per K-step (halves offset inside the interval) instead of two
    int kseg;           // ping-pong kernel, bf16x3 through hi / lo planes: k per segment (multiple of 32; K = 3 kseg); 0: plain
    int kreal;          // ... and the real K of a segment (the last stage of every segment is masked behind it)
    float* partial;     // split-K partial slabs [group][split][M][ldc]
    GemmGroup grp[kMaxGemmGroups];
};

__device__ __forceinline__ float act_apply(int act, float v) {
    switch (act) {
        case ADN_ACT_RECTIFY: return v > 0.f ? v : 0.f;
        case ADN_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case ADN_ACT_TANH: return tanhf(v);
        case ADN_ACT_LEAKY_RECTIFY: return v > 0.f ? v : 0.01f * v;
        case ADN_ACT_VERY_LEAKY_RECTIFY: return v > 0.f ? v : (1.f / 3.f) * v;
        case ADN_ACT_SCALED_TANH: return 2.4f * tanhf(0.5f * v);
        case ADN_ACT_SCALED_TANH_LECUN: return 1.7159f * tanhf((2.f / 3.f) * v);
        case kActRectifyHalf: return v > 0.f ? v : (v == 0.f ? -0.f : 0.f);       // (the kink leaves as -0.0: adn_common.h)
        default: return v;
    }
}

__device__ __forceinline__ float act_grad_from_output(int act, float y) {
    switch (act) {
        case ADN_ACT_RECTIFY: return y > 0.f ? 1.f : 0.f;
        case ADN_ACT_SIGMOID: return y * (1.f - y);
        case ADN_ACT_TANH: return 1.f - y * y;
        case ADN_ACT_LEAKY_RECTIFY: return y > 0.f ? 1.f : 0.01f;
        case ADN_ACT_VERY_LEAKY_RECTIFY: return y > 0.f ? 1.f : (1.f / 3.f);
        case ADN_ACT_SCALED_TANH: { const float t = y * (1.f / 2.4f); return 1.2f * (1.f - t * t); }
        case ADN_ACT_SCALED_TANH_LECUN: { const float t = y * (1.f / 1.7159f); return (2.f / 3.f) * 1.7159f * (1.f - t * t); }
        case kActRectifyHalf: return y > 0.f ? 1.f : (__float_as_uint(y) == 0x80000000u ? 0.5f : 0.f);
        default: return 1.f;
    }
}


// XCD-aware tile order (cdna_hip_programming.md T1): workgroups are dealt round-robin over the 8 XCDs, so
// block b and b+8 share an L2.  Give every XCD one contiguous chunk of the (m-major, n-fastest) tile list:
// the tiles that are co-resident on an XCD then share their A row-panel and walk the same B panels.
// Bijective for any grid size.  Speed only -- never correctness.
__device__ __forceinline__ int xcd_tile(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// tile index -> (m-tile, n-tile).  Together with xcd_tile() each XCD works on a roughly square block of
// panel_n x (chunk / panel_n) tiles, so the A row-panels AND the B column-panels it touches are shared by many
// of its co-resident workgroups (with plain row-major enumeration an XCD owning 1-2 tile rows re-reads ALL of B:
// measured 4.5x over-fetch on the weight-gradient GEMMs).
__device__ __forceinline__ void tile_coords(const GemmParams& p, int tile, int& tm, int& tn) {
    const int per_panel = p.panel_n * p.tiles_m;
    const int panel = tile / per_panel, within = tile - panel * per_panel;
    const int width = min(p.panel_n, p.tiles_n - panel * p.panel_n);
    tm = within / width;
    tn = panel * p.panel_n + within - tm * width;
}

// epilogue for one 32x32 MFMA accumulator tile whose top-left element is (row0, col0).
// C/D map of the 32x32 MFMAs (all dtypes): col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
__device__ __forceinline__ void store_tile32(const GemmParams& p, const f32x16& acc, int row0, int col0, int lane,
                                             bool first_split) {
    const int col = col0 + (lane & 31);
    if (col >= p.N) return;
    const float bias = (p.bias && first_split) ? p.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= p.M) continue;
        float v = acc[r] + bias;
        float* c = p.C + (size_t)row * p.ldc + col;
        if (p.atomic) {
            if (p.Y) v *= act_grad_from_output(p.act_grad, p.Y[(size_t)row * p.ldy + col]);
            // deterministic mode: the partial tile goes into this (problem, K-slice)'s slab with plain stores; a fixed-order
            // pass adds the slabs (rs_splitk_reduce_kernel)
            if (p.partial) p.partial[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * p.M + row) * p.ldc + col] = v;
            else atomicAdd(c, v);
        } else {
            v = act_apply(p.act, v);
            if (p.Y) v *= act_grad_from_output(p.act_grad, p.Y[(size_t)row * p.ldy + col]);
            if (p.accumulate) v += *c;
            *c = v;
        }
    }
}

// defined in gemm_bf16.hip
void launch_gemm_bf16(const GemmParams& p, int layout, int tile_mode /*0: 64x64, 1: 128x128*/, dim3 grid, hipStream_t s);
// persistent ping-pong LDS-DMA kernel; tile_mode 4: 256x256, 5: 256x128, 6: 128x256; splits > 1: partial slabs + reduce
int split3_cols(const float* src, int ld_src, int rows, int K, int Kp, void* dst, int lo_mask, hipStream_t s);
int split3_transpose(const float* src, int ld_src, int R, int K, int Kp, void* dst, int ld_dst, int lo_mask, hipStream_t s);
int split3_rows(const float* src, int ld_src, int K, int Kp, int cols, void* dst, int ld_dst, int lo_mask, hipStream_t s);
void launch_gemm_bf16_pp(const GemmParams& p, int layout, int tile_mode, int splits, dim3 grid, hipStream_t s, bool reduce = true);
// the tail band of a persistent launch (gemm_f32.hip gemm_pp_try_impl): sums the band's split-K slabs and applies the launch's epilogue
void launch_splitk_tail_epilogue(const GemmParams& p, int splits, int m_off, int cs_row0, hipStream_t s);
constexpr int kTailEpiRows = 8;      // rows per workgroup of that pass = rows per partial column-sum row it writes

}  // namespace adn
