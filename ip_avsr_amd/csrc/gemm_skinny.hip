// Streaming kernels for the train step's SKINNY products: under 20800 (B T) rows, the shapes whose other dimensions are the 50-unit
// bottleneck or the 26 classes.  They are HBM-bound (a few flops per byte) and the register-staged 64 x 64 kernel runs them at
// ~1.9 TB/s: one tile column of 325 workgroups with ONE 8-KB stage in flight each is ~2.6 MB in flight on the whole device, a
// quarter of what 8 TB/s x ~2 us of latency needs.  The kernels here keep 64 - 128 KB per CU in flight instead (VERDICT r4 next #2a):
//
//   skinny_nn_kernel   C [M][N <= 64] = A [M][K <= 512] . B (+ bias, linear / rectify)        forward bottleneck, classifier
//                      B is taken as its k-contiguous transpose [N][K] (the W^T copies the model keeps anyway), staged once per
//                      workgroup in LDS; a wave owns 32 rows and streams its A fragments HBM -> registers, a chunk of 4 k-steps
//                      (up to 16 KB per wave) ahead of the chunk being multiplied: 8 waves x 16 KB = 128 KB in flight per CU.
//                      Wide form (N <= 160, planes only): B^T staged 128 k per pass -- the first LSTMs' input gradient dG W_in^T.
//   skinny_nk_kernel   C [M][N] = A [M][K <= 64] . B^T, B = [N][K] k-contiguous (the weights themselves: dX = dZ W^T), optional
//                      rectify'(Y) mask from the bf16 copy of Y, optional fused column sums (db of the layer below), result as fp32
//                      and / or bf16 hi (+ lo) planes.  Output-bound: a workgroup takes 256 rows x 2 column tiles with every load
//                      requested up front; the column chunks of a row block are dealt to ONE XCD (they share its A fragments).
//   skinny_tn_kernel   C [M <= 1024][N <= 64] (+)= A^T B, A [K][M], B [K][N] both k-strided, K = all frames: 64 x 64 output blocks x
//                      many K-slices (4 workgroups per CU), operands through LDS + transposing reads, two stages of register
//                      prefetch; the slices' partial blocks go to slabs that a fixed-order pass adds (no atomics: deterministic
//                      as it stands).
//
// Every kernel takes PLANES: the operands' bf16 hi / lo planes and three MFMA products per accumulator (a_hi b_hi + a_hi b_lo +
// a_lo b_hi) -- the bf16x3 mode, where these shapes used to run over [hi | hi | lo] split images written per launch (two extra
// passes over the operands) on the same 64 x 64 kernel.
// MFMA: v_mfma_f32_16x16x32_bf16 with the operands swapped (B fragment first) where a lane should end up with four consecutive
// COLUMNS of a row (row-contiguous 16-byte stores).
#include "gemm_common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace adn {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct SkinnyGroup {
    const __bf16 *A, *Alo, *B, *Blo;          // hi / lo planes (lo null without PLANES)
    float* C; __bf16 *C16, *C16lo;
    const float* bias; const __bf16* Y16; float* colsum;
};
struct SkinnyParams {
    int M, N, K, lda, ldb, ldc, ldy, colsum_ld, act, accumulate;
    int rblocks, xcd_map;                     // (NK) row blocks of 256; workgroups dealt so that the column chunks of a row block share an XCD
    int k_chunk, splits;                      // (TN) k per slice, slices
    float* partial;                           // (TN) slabs [group][slice][M][ldc]
    SkinnyGroup g[kMaxGemmGroups];
};

__device__ __forceinline__ SkinnyGroup pick(const SkinnyParams& p, int g) {
    SkinnyGroup r = p.g[0];
    if (g == 1) r = p.g[1];
    if (g == 2) r = p.g[2];
    if (g == 3) r = p.g[3];
    return r;
}
__device__ __forceinline__ bf16x8 zero8() { bf16x8 z; for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f; return z; }
// elements k0 + j >= kmax of a fragment (lane's first element is k0) -> 0
__device__ __forceinline__ bf16x8 mask_k(bf16x8 v, int k0, int kmax) {
    u32x4 w = __builtin_bit_cast(u32x4, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] &= ((k0 + 2 * j < kmax) ? 0x0000FFFFu : 0u) | ((k0 + 2 * j + 1 < kmax) ? 0xFFFF0000u : 0u);
    return __builtin_bit_cast(bf16x8, w);
}
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
    bf16x4 r; r[0] = (__bf16)a; r[1] = (__bf16)b; r[2] = (__bf16)c; r[3] = (__bf16)d;
    return __builtin_bit_cast(uint2, r);
}

// ---------------------------------------------------------------------------------------------------------
// skinny_nn_kernel: N <= 16 NT; B^T is staged KCH k at a time (one pass when K <= KCH).  512 threads = 8 waves x 32 rows.
// (NT = 2 / 4: KCH = 512, the whole of B^T once -- forward bottleneck, classifier.  NT = 10: 128 (planes) / 256 k per pass, the
//  first LSTM's input gradient dfeat = dG W_in^T, N = 150, K = 4H: each pass costs two barriers, the A loads stay in flight across them)
// ---------------------------------------------------------------------------------------------------------
constexpr int kNnKMax = 512;

template <int NT, bool PLANES, int KCH>
__global__ __launch_bounds__(512) void skinny_nn_kernel(const SkinnyParams p) {
    extern __shared__ __attribute__((aligned(16))) __bf16 sk_lds[];
    constexpr int kStride = KCH + 8;                                 // bf16 per LDS row of a B^T plane
    const SkinnyGroup g = pick(p, blockIdx.y);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int ksteps = (p.K + 31) / 32, kp = ksteps * 32;
    __bf16* bs_hi = sk_lds;                                          // [16 NT][kStride]
    __bf16* bs_lo = sk_lds + (size_t)16 * NT * kStride;
    const int row0 = blockIdx.x * 256 + wave * 32;
    // ---- A fragments: chunk = CH k-steps of both row tiles; k-steps past the end read as zero
    constexpr int CH = (NT > 4 && PLANES) ? 2 : 4;                   // (the wide form's accumulators leave room for two)
    constexpr int CPB = KCH / (32 * CH);                             // chunks per staged pass of B^T
    static_assert(KCH % (32 * CH) == 0, "a pass of B^T holds whole chunks");
    size_t aoff[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) aoff[r] = (size_t)min(row0 + 16 * r + i16, p.M - 1) * p.lda + 8 * kq;      // clamped rows feed dropped outputs
    bf16x8 a_hi[2][CH][2], a_lo[2][CH][2];                           // [buffer][k-step of the chunk][row tile]
    auto load_chunk = [&](int buf, int c) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int ks = c * CH + s;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (ks < ksteps) {
                    a_hi[buf][s][r] = *reinterpret_cast<const bf16x8*>(g.A + aoff[r] + 32 * ks);
                    if (PLANES) a_lo[buf][s][r] = *reinterpret_cast<const bf16x8*>(g.Alo + aoff[r] + 32 * ks);
                } else {
                    a_hi[buf][s][r] = zero8();
                    if (PLANES) a_lo[buf][s][r] = zero8();
                }
            }
        }
    };
    load_chunk(0, 0);                                                 // in flight while B^T is staged
    // ---- B^T planes -> LDS, k in [k0, k0 + KCH): rows n < N (others zero), k < K (others zero)
    auto stage_b = [&](int k0) __attribute__((always_inline)) {
        constexpr int chunks_per_row = KCH / 8;
        const int kend = min(kp, k0 + KCH);
        for (int e = tid; e < 16 * NT * chunks_per_row; e += 512) {
            const int n = e / chunks_per_row, c = e % chunks_per_row, k = k0 + 8 * c;
            if (k >= kend) continue;                                  // (never read: the k loop stops at ksteps)
            bf16x8 vh = zero8(), vl = zero8();
            if (n < p.N && k < p.K) {
                vh = *reinterpret_cast<const bf16x8*>(g.B + (size_t)n * p.ldb + k);
                if (PLANES) vl = *reinterpret_cast<const bf16x8*>(g.Blo + (size_t)n * p.ldb + k);
                if (k + 8 > p.K) { vh = mask_k(vh, k, p.K); if (PLANES) vl = mask_k(vl, k, p.K); }
            }
            *reinterpret_cast<bf16x8*>(bs_hi + (size_t)n * kStride + 8 * c) = vh;
            if (PLANES) *reinterpret_cast<bf16x8*>(bs_lo + (size_t)n * kStride + 8 * c) = vl;
        }
    };
    f32x4 acc[2][NT];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nchunks = (ksteps + CH - 1) / CH;
    auto open_pass = [&](int c) __attribute__((always_inline)) {     // chunk c opens a pass of B^T (uniform over the block); called
        if (c % CPB == 0) {                                           // BEFORE the next chunk's A loads are issued, so that the
            if (c) __syncthreads();                                   // pass's own loads are not queued behind them
            stage_b(32 * CH * c);
            __syncthreads();
        }
    };
    auto multiply = [&](int buf, int c) __attribute__((always_inline)) {
        const int kb = 32 * CH * (c % CPB);                           // the chunk's k offset inside the staged pass
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int ks = c * CH + s;
            if (ks >= ksteps) break;
            bf16x8 ah[2], al[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                ah[r] = a_hi[buf][s][r];
                if (PLANES) al[r] = a_lo[buf][s][r];
                if (32 * ks + 32 > p.K) { ah[r] = mask_k(ah[r], 32 * ks + 8 * kq, p.K); if (PLANES) al[r] = mask_k(al[r], 32 * ks + 8 * kq, p.K); }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bs_hi + (size_t)(16 * t + i16) * kStride + kb + 32 * s + 8 * kq);
#pragma unroll
                for (int r = 0; r < 2; ++r) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[r], acc[r][t], 0, 0, 0);
                if (PLANES) {
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(bs_lo + (size_t)(16 * t + i16) * kStride + kb + 32 * s + 8 * kq);
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah[r], acc[r][t], 0, 0, 0);
                        acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[r], acc[r][t], 0, 0, 0);
                    }
                }
            }
        }
    };
    for (int c = 0; c < nchunks; c += 2) {                            // (two chunks per iteration: the buffers are compile-time)
        open_pass(c);
        if (c + 1 < nchunks) load_chunk(1, c + 1);
        multiply(0, c);
        if (c + 1 < nchunks) {
            open_pass(c + 1);
            if (c + 2 < nchunks) load_chunk(0, c + 2);
            multiply(1, c + 1);
        }
    }
    // ---- epilogue: swapped operands -> lane = row i16 of a row tile, columns 16 t + 4 kq .. + 3
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = row0 + 16 * r + i16;
        if (row >= p.M) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = 16 * t + 4 * kq;
            if (col >= p.N) continue;
            float v[4] = {acc[r][t][0], acc[r][t][1], acc[r][t][2], acc[r][t][3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (col + e < p.N) {
                    if (g.bias) v[e] += g.bias[col + e];
                    if (p.act == ADN_ACT_RECTIFY) v[e] = fmaxf(v[e], 0.f);
                }
            }
            const size_t off = (size_t)row * p.ldc + col;
            if (col + 4 <= p.N) {
                if (g.C) *reinterpret_cast<float4*>(g.C + off) = make_float4(v[0], v[1], v[2], v[3]);
                if (g.C16) *reinterpret_cast<uint2*>(g.C16 + off) = pack4(v[0], v[1], v[2], v[3]);
            } else {
                for (int e = 0; e < 4 && col + e < p.N; ++e) {
                    if (g.C) g.C[off + e] = v[e];
                    if (g.C16) g.C16[off + e] = (__bf16)v[e];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// skinny_nk_kernel: K <= 64.  256 threads = 4 waves x 64 rows; a workgroup takes 256 rows x kNkTC column tiles (blockIdx.x = row
// block x column chunks + chunk).  Everything a wave reads -- its A fragments, the chunk's B fragments, the act'(Y) masks -- is
// requested up front (one exposed latency per wave); many small workgroups (61 waves per CU at the bench shape) hide it and the
// store acknowledgements.  (The first form -- a wave walking all N / 16 tiles with a one-tile B prefetch -- paid a memory round
// trip per tile: vmcnt retires in order, so the next tile's B fragments waited for the previous tile's stores: 36 / 80 us where
// this form takes the time of its bytes.  Tried on top, round 5, and withdrawn: pairs of column tiles computed together so that lanes
// kq / kq ^ 1 can swap halves and store 16-byte pieces (64 contiguous bytes per row), 2 or 4 tiles per workgroup -- on one box, against
// this form: three problems per launch 97 -> 98..107 us (bottleneck shape), one problem per launch 19.5 -> 38.8 us (classifier shape):
// profiles/r05/lab_skinny2.txt.)
// ---------------------------------------------------------------------------------------------------------
constexpr int kNkTC = 2;
template <bool PLANES>
__global__ __launch_bounds__(256) void skinny_nk_kernel(const SkinnyParams p, int nchunks) {
    const SkinnyGroup g = pick(p, blockIdx.y);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    // The chunks of a row block all read its A fragments: dealt over the XCDs round-robin (workgroup id mod 8) each of the eight L2s
    // fetched the whole of A -- 264 MB per launch at the bench shape against 78 MB of operands (FETCH_SIZE, profiles/r05/pmc_lab_nk.txt).
    // xcd_map: workgroup id -> (XCD = id mod 8, slot = id / 8); the slots of an XCD walk the chunks of ITS row blocks (8 j + XCD).
    int rblock, chunk;
    if (p.xcd_map) {
        const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
        rblock = (slot / nchunks) * 8 + xcd; chunk = slot - (slot / nchunks) * nchunks;
        if (rblock >= p.rblocks) return;                              // (the grid is padded to whole groups of 8 row blocks)
    } else { rblock = (int)blockIdx.x / nchunks; chunk = (int)blockIdx.x - rblock * nchunks; }
    const int row0 = (rblock * 4 + wave) * 64;
    const int t0 = chunk * kNkTC;
    const int ksteps = (p.K + 31) / 32;                               // 1 or 2
    bf16x8 a_hi[4][2], a_lo[4][2], b_hi[kNkTC][2], b_lo[kNkTC][2];
    uint2 ym[kNkTC][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            a_hi[r][s] = zero8(); a_lo[r][s] = zero8();
            const int row = min(row0 + 16 * r + i16, p.M - 1), k = 32 * s + 8 * kq;
            if (s < ksteps && k < p.K) {
                a_hi[r][s] = *reinterpret_cast<const bf16x8*>(g.A + (size_t)row * p.lda + k);
                if (PLANES) a_lo[r][s] = *reinterpret_cast<const bf16x8*>(g.Alo + (size_t)row * p.lda + k);
            }
        }
#pragma unroll
    for (int c = 0; c < kNkTC; ++c) {
        const int n = min(16 * (t0 + c) + i16, p.N - 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int k = 32 * s + 8 * kq;
            b_hi[c][s] = zero8(); b_lo[c][s] = zero8();
            if (s < ksteps && k < p.K) {                              // (k >= K of a row: masked in A, finite here)
                b_hi[c][s] = *reinterpret_cast<const bf16x8*>(g.B + (size_t)n * p.ldb + k);
                if (PLANES) b_lo[c][s] = *reinterpret_cast<const bf16x8*>(g.Blo + (size_t)n * p.ldb + k);
            }
        }
        const int col = 16 * (t0 + c) + 4 * kq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + 16 * r + i16;
            ym[c][r] = make_uint2(0x3F803F80u, 0x3F803F80u);          // (1.0: keep)
            if (g.Y16 && row < p.M && col < p.N) ym[c][r] = *reinterpret_cast<const uint2*>(g.Y16 + (size_t)row * p.ldy + col);     // (ldy % 4 == 0: a partial float4's pad columns are readable)
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int k = 32 * s + 8 * kq;
            if (k + 8 > p.K) { a_hi[r][s] = mask_k(a_hi[r][s], k, p.K); if (PLANES) a_lo[r][s] = mask_k(a_lo[r][s], k, p.K); }
        }
#pragma unroll
    for (int c = 0; c < kNkTC; ++c) {
        const int t = t0 + c;
        if (16 * t >= p.N) break;                                      // (uniform: the chunk's tiles beyond N)
        f32x4 acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s >= ksteps) break;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_hi[c][s], a_hi[r][s], acc[r], 0, 0, 0);
                if (PLANES) {
                    acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_lo[c][s], a_hi[r][s], acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_hi[c][s], a_lo[r][s], acc[r], 0, 0, 0);
                }
            }
        }
        // lane = row i16 of row tile r, columns 16 t + 4 kq .. + 3
        const int col = 16 * t + 4 * kq;
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool cok = col + 4 <= p.N;                              // the whole float4 inside (N % 4 != 0: the last one is partial)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + 16 * r + i16;
            if (row >= p.M || col >= p.N) continue;
            float4 v = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
            if (g.Y16) {                                              // rectify'(Y) from the bf16 copy of Y
                const bf16x4 y = __builtin_bit_cast(bf16x4, ym[c][r]);
                v.x = (float)y[0] > 0.f ? v.x : 0.f; v.y = (float)y[1] > 0.f ? v.y : 0.f;
                v.z = (float)y[2] > 0.f ? v.z : 0.f; v.w = (float)y[3] > 0.f ? v.w : 0.f;
            }
            const size_t off = (size_t)row * p.ldc + col;
            if (!cok) {                                               // partial float4 at the right edge: element by element (no column sums here)
                const float ve[4] = {v.x, v.y, v.z, v.w};
                for (int e = 0; e < 4 && col + e < p.N; ++e) {
                    float x = ve[e];
                    if (g.C) { if (p.accumulate) x += g.C[off + e]; g.C[off + e] = x; }
                    if (g.C16) { const __bf16 hb = (__bf16)x; g.C16[off + e] = hb; if (g.C16lo) g.C16lo[off + e] = (__bf16)(x - (float)hb); }
                }
                continue;
            }
            if (g.C) {
                if (p.accumulate) { const float4 cc = *reinterpret_cast<const float4*>(g.C + off); v.x += cc.x; v.y += cc.y; v.z += cc.z; v.w += cc.w; }
                *reinterpret_cast<float4*>(g.C + off) = v;
            }
            cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
            if (g.C16) {
                const uint2 h = pack4(v.x, v.y, v.z, v.w);
                *reinterpret_cast<uint2*>(g.C16 + off) = h;
                if (g.C16lo) {
                    const bf16x4 hb = __builtin_bit_cast(bf16x4, h);
                    *reinterpret_cast<uint2*>(g.C16lo + off) = pack4(v.x - (float)hb[0], v.y - (float)hb[1], v.z - (float)hb[2], v.w - (float)hb[3]);
                }
            }
        }
        if (g.colsum) {                                               // sums over this wave's 64 rows -> row (row0 / 64) of the workspace
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                cs.x += __shfl_xor(cs.x, o, 64); cs.y += __shfl_xor(cs.y, o, 64);
                cs.z += __shfl_xor(cs.z, o, 64); cs.w += __shfl_xor(cs.w, o, 64);
            }
            if (i16 == 0 && cok && row0 < p.M) *reinterpret_cast<float4*>(g.colsum + (size_t)(row0 / 64) * p.colsum_ld + col) = cs;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// skinny_tn_kernel: C [M][N <= 64] partial = A^T B over one K-slice; grid (m-blocks of 64, slices, groups); 256 threads: wave w owns
// the 16-row strip w of the 64 x 64 block.  LDS images [64 k][64 + 8] per plane; fragments by ds_read_b64_tr_b16.
// ---------------------------------------------------------------------------------------------------------
constexpr int kTnStride = 64 + 8;
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* lds, int col0, int s, int lane) {
    // 16 columns from col0, k-step s (32 k): lane 4q + p of a 16-lane group addresses k-row q, columns 4p .. 4p + 3 (T10)
    const int q = (lane & 15) >> 2, pcol = (lane & 3) * 4;
    const __bf16* a = lds + (s * 32 + (lane >> 4) * 8 + q) * kTnStride + col0 + pcol;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * kTnStride));
    return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <bool PLANES>
__global__ __launch_bounds__(256) void skinny_tn_kernel(const SkinnyParams p) {
    __shared__ __attribute__((aligned(16))) __bf16 lds[(PLANES ? 4 : 2) * 64 * kTnStride];
    __bf16* As = lds; __bf16* Bs = lds + 64 * kTnStride;
    __bf16* Asl = lds + 2 * 64 * kTnStride; __bf16* Bsl = lds + 3 * 64 * kTnStride;
    const SkinnyGroup g = pick(p, blockIdx.z);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // xcd_map: blockIdx.x = K-slice (grid padded to a multiple of 8), blockIdx.y = 64-row block of C -- the row blocks of a slice, which
    // all read its rows of B, then sit on ONE XCD (workgroup id mod 8 = slice mod 8); otherwise x = row block, y = slice as before
    const int mblk = p.xcd_map ? (int)blockIdx.y : (int)blockIdx.x, slice = p.xcd_map ? (int)blockIdx.x : (int)blockIdx.y;
    if (slice >= p.splits) return;                                    // (uniform; padding slices)
    const int m0 = mblk * 64;
    const int kbeg = slice * p.k_chunk, kend = min(p.K, kbeg + p.k_chunk);
    // staging: thread -> (k-row tid >> 3 (+ 32), 8 columns (tid & 7) * 8); columns beyond M / N read column block 0 (dropped / zero-masked below)
    const int srow = tid >> 3, scol = (tid & 7) * 8;
    const int acol = (m0 + scol + 8 <= p.lda) ? m0 + scol : 0;
    const bool b_in = scol < p.N;                                     // (whole 8-column pieces: ldb >= round_up(N, 8); columns >= N are dropped)
    bf16x8 sa[2][2], sb[2][2], sal[2][2], sbl[2][2];                  // [buffer][half]
    auto request = [&](int buf, int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const int k = k0 + srow + 32 * hlf;
            sa[buf][hlf] = zero8(); sb[buf][hlf] = zero8(); sal[buf][hlf] = zero8(); sbl[buf][hlf] = zero8();
            if (k < kend) {
                sa[buf][hlf] = *reinterpret_cast<const bf16x8*>(g.A + (size_t)k * p.lda + acol);
                if (b_in) sb[buf][hlf] = *reinterpret_cast<const bf16x8*>(g.B + (size_t)k * p.ldb + scol);
                if (PLANES) {
                    sal[buf][hlf] = *reinterpret_cast<const bf16x8*>(g.Alo + (size_t)k * p.lda + acol);
                    if (b_in) sbl[buf][hlf] = *reinterpret_cast<const bf16x8*>(g.Blo + (size_t)k * p.ldb + scol);
                }
            }
        }
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const int o = (srow + 32 * hlf) * kTnStride + scol;
            *reinterpret_cast<bf16x8*>(As + o) = sa[buf][hlf];
            *reinterpret_cast<bf16x8*>(Bs + o) = sb[buf][hlf];
            if (PLANES) { *reinterpret_cast<bf16x8*>(Asl + o) = sal[buf][hlf]; *reinterpret_cast<bf16x8*>(Bsl + o) = sbl[buf][hlf]; }
        }
    };
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto multiply = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 ah = tr_frag(As, 16 * wave, s, lane);
            bf16x8 al = ah;
            if (PLANES) al = tr_frag(Asl, 16 * wave, s, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8 bh = tr_frag(Bs, 16 * t, s, lane);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[t], 0, 0, 0);
                if (PLANES) {
                    const bf16x8 bl = tr_frag(Bsl, 16 * t, s, lane);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[t], 0, 0, 0);
                }
            }
        }
    };
    // two stages of register prefetch ahead of the stage in LDS
    request(0, kbeg);
    request(1, kbeg + 64);
    for (int k0 = kbeg; k0 < kend; k0 += 128) {
        commit(0);
        __syncthreads();
        if (k0 + 128 < kend) request(0, k0 + 128);
        multiply();
        __syncthreads();
        if (k0 + 64 < kend) {
            commit(1);
            __syncthreads();
            if (k0 + 192 < kend) request(1, k0 + 192);
            multiply();
            __syncthreads();
        }
    }
    // partial block -> this (group, slice)'s slab: acc[t]: col = 16 t + (lane & 15), rows m0 + 16 wave + 4 (lane >> 4) + r
    float* slab = p.partial + ((size_t)((int)blockIdx.z * p.splits + slice) * p.M) * p.ldc;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = 16 * t + (lane & 15);
        if (col >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 16 * wave + 4 * (lane >> 4) + r;
            if (row < p.M) slab[(size_t)row * p.ldc + col] = acc[t][r];
        }
    }
}

// (round 6: the 36 - 65 slabs of an element are spread over the workgroup's four thread rows, four loads in flight each -- one
//  thread walking them all was a chain of round trips; the four partial sums meet in LDS and are added in row order: one fixed order)
__global__ __launch_bounds__(256) void skinny_tn_reduce_kernel(const SkinnyParams p) {
    __shared__ float part4[3][64];
    const SkinnyGroup g = pick(p, blockIdx.y);
    const float* part = p.partial + (size_t)blockIdx.y * p.splits * p.M * p.ldc;
    const size_t slab = (size_t)p.M * p.ldc;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6, total = p.M * p.N;
    for (int e0 = blockIdx.x * 64; e0 < total; e0 += gridDim.x * 64) {
        const int e = e0 + lane;
        const bool ok = e < total;
        const size_t i = ok ? (size_t)(e / p.N) * p.ldc + (size_t)(e % p.N) : 0;
        float v = 0.f;
        if (ok) {
            int s = q;
            for (; s + 12 < p.splits; s += 16) {
                const float a = part[(size_t)s * slab + i], b = part[(size_t)(s + 4) * slab + i];
                const float c = part[(size_t)(s + 8) * slab + i], d = part[(size_t)(s + 12) * slab + i];
                v += a; v += b; v += c; v += d;
            }
            for (; s < p.splits; s += 4) v += part[(size_t)s * slab + i];
        }
        if (q > 0) part4[q - 1][lane] = v;
        __syncthreads();
        if (q == 0 && ok) g.C[i] = (p.accumulate ? g.C[i] : 0.f) + v + part4[0][lane] + part4[1][lane] + part4[2][lane];
        __syncthreads();
    }
}

void fill_groups(SkinnyParams& p, const GemmArgs* gs, int n, bool planes, bool kc_b) {
    for (int k = 0; k < n; ++k) {
        SkinnyGroup& q = p.g[k];
        q.A = reinterpret_cast<const __bf16*>(gs[k].A16); q.Alo = planes ? reinterpret_cast<const __bf16*>(gs[k].A16lo) : nullptr;
        q.B = reinterpret_cast<const __bf16*>(kc_b ? gs[k].Bkc16 : gs[k].B16);
        q.Blo = planes ? reinterpret_cast<const __bf16*>(kc_b ? gs[k].Bkc16lo : gs[k].B16lo) : nullptr;
        q.C = gs[k].C; q.C16 = reinterpret_cast<__bf16*>(gs[k].C16); q.C16lo = reinterpret_cast<__bf16*>(gs[k].C16lo);
        q.bias = gs[k].bias; q.Y16 = reinterpret_cast<const __bf16*>(gs[k].Y16); q.colsum = nullptr;
    }
}

}  // namespace

// Runs n same-shape problems through a skinny kernel if one applies; *used says whether.  Operands: bf16 copies (precision bf16)
// or hi / lo planes (precision bf16x3).  dry: only answer.
int gemm_skinny_try(const GemmArgs* gs, int n, hipStream_t stream, bool* used, bool dry) {
    *used = false;
    static const bool off = getenv("ADN_GEMM_NO_SKINNY") != nullptr;
    if (off || n < 1 || n > kMaxGemmGroups) return ADN_OK;
    const GemmArgs& g = gs[0];
    const bool planes = g.precision == ADN_PRECISION_BF16X3;
    if (g.precision != ADN_PRECISION_BF16 && !planes) return ADN_OK;
    if (g.M < 1024 && g.layout != GEMM_TN) return ADN_OK;
    for (int k = 1; k < n; ++k) {
        const GemmArgs& q = gs[k];
        if (q.layout != g.layout || q.M != g.M || q.N != g.N || q.K != g.K || q.lda != g.lda || q.ldb != g.ldb || q.ldc != g.ldc ||
            q.ldy != g.ldy || q.ldbkc != g.ldbkc || q.act != g.act || q.act_grad != g.act_grad || q.accumulate != g.accumulate ||
            q.precision != g.precision || (q.C == nullptr) != (g.C == nullptr) || (q.C16 == nullptr) != (g.C16 == nullptr) ||
            (q.C16lo == nullptr) != (g.C16lo == nullptr) || q.hi_result != g.hi_result || q.hi_product != g.hi_product || (q.bias == nullptr) != (g.bias == nullptr) || (q.Y16 == nullptr) != (g.Y16 == nullptr) ||
            (q.Y == nullptr) != (g.Y == nullptr) || (q.colsum == nullptr) != (g.colsum == nullptr) || (q.Bkc16 == nullptr) != (g.Bkc16 == nullptr))
            return ADN_OK;
    }
    auto aligned = [](const void* p, size_t a) { return ((uintptr_t)p % a) == 0; };
    SkinnyParams p;
    std::memset(static_cast<void*>(&p), 0, sizeof(p));
    p.M = g.M; p.N = g.N; p.K = g.K; p.lda = g.lda; p.ldc = g.ldc; p.ldy = g.ldy; p.act = g.act; p.accumulate = g.accumulate;
    static const bool trace = getenv("ADN_GEMM_TRACE") != nullptr;
    auto say = [&](int kind, int split) {        // (tile codes of profiles/gemm_breakdown.py: 1001 skinny_nn, 1002 skinny_nk, 1003 skinny_tn)
        if (trace)
            fprintf(stderr, "ADN_GEMM %s M=%d N=%d K=%d tile=%d tiles=%d split=%d shadows=1 lean=%d acc=%d groups=%d%s\n",
                    g.layout == GEMM_TN ? "TN" : "NN", g.M, g.N, planes ? 3 * g.K : g.K, kind, 0, split, (int)(g.C == nullptr || g.lean_ok), g.accumulate, n,
                    planes ? " planes=1" : "");
    };
    // ---------------- forward bottleneck / classifier: N <= 64, K <= 512, B as its k-contiguous transpose
    // Where they pay (profiles/r05/lab_skinny.txt, three problems per launch, us): over planes every kernel here -- forward
    // bottleneck 134.9 -> 35.4, classifier 60.0 -> 17.1, bottleneck weight gradient 147.4 -> 55.9, classifier's 74.9 -> 43.3, input
    // gradient behind the bottleneck with act'(Y) and fused sums 118.5 -> 98.1 -- because their alternative is the split-image path.
    // In plain bf16 the register-staged kernels already move these shapes at 2.6 - 3 TB/s (forward bottleneck 23.9 against 24.4
    // here; the K <= 64 input gradient 41.5 against 66.2: its 64-column tiles store 128-byte row pieces through an LDS bounce where the
    // transposed accumulators here store 32-byte pieces): only the 26-column classifier product (14.8 -> 11.6) comes here.
    // (A/B switch: every skinny kernel in plain bf16 too.  The mixed mode's back-propagation -- one bf16 product whose result is still
    //  wanted as planes -- takes them as well: the register-staged kernel would write fp32 + a split pass behind it; 5.66 -> 5.52 ms)
    static const bool all_env = getenv("ADN_GEMM_SKINNY_ALL") != nullptr;
    const bool all_bf16 = all_env || g.hi_product;
    static const bool no_wide = getenv("ADN_GEMM_NO_SKINNY_WIDE") != nullptr;
    const bool nn_narrow = g.N <= 64 && g.K <= kNnKMax && (planes || g.N <= 32 || all_bf16);
    // (wide form, profiles/r05/lab_skinny2.txt, the first LSTMs' input gradient N = 150, K = 1000, two problems per launch: over planes
    //  83.6 us against 145 in the model for the ping-pong kernel's 256-column tiles; in plain bf16 the register-staged kernel's 48.8 us
    //  beats this form's 53.7 -- its staged passes stall the whole workgroup on two barriers each -- so bf16 problems do not come here)
    const bool nn_wide = planes && g.N > 64 && g.N <= 160 && g.K >= 256 && g.M >= 2048 && !no_wide;
    if ((g.layout == GEMM_NN || g.layout == GEMM_NT) && g.Bkc16 && g.K >= 32 && (nn_narrow || nn_wide)) {
        // (Bkc16: B as [N][K] k-contiguous -- for an NT problem B itself, for NN the caller's transposed copy)
        for (int k = 0; k < n; ++k) {
            const GemmArgs& q = gs[k];
            const void* bh = q.Bkc16; const void* bl = q.Bkc16lo;
            if (!q.A16 || !bh || (planes && (!q.A16lo || !bl)) || !q.C || q.Y || q.Y16 || q.colsum || q.accumulate) return ADN_OK;
            if (!aligned(q.A16, 16) || !aligned(bh, 16) || !aligned(q.C, 16) || (planes && (!aligned(q.A16lo, 16) || !aligned(bl, 16)))) return ADN_OK;
        }
        const int ldb = g.ldbkc;
        if (g.act != ADN_ACT_LINEAR && g.act != ADN_ACT_RECTIFY) return ADN_OK;
        if (g.lda % 8 || ldb % 8 || g.ldc % 4 || g.lda < (int)round_up(g.K, 32) || ldb < (int)round_up(g.K, 8)) return ADN_OK;
        if (dry) { *used = true; return ADN_OK; }
        p.ldb = ldb;
        fill_groups(p, gs, n, planes, true);
        for (int k = 0; k < n; ++k) {
            p.g[k].C16 = (gs[k].C16 && g.ldc % 4 == 0) ? reinterpret_cast<__bf16*>(gs[k].C16) : nullptr; p.g[k].C16lo = nullptr;
            if (gs[k].planes_done) *gs[k].planes_done = 0;
            if (gs[k].fp32_skipped) *gs[k].fp32_skipped = 0;
        }
        const int NT = g.N <= 32 ? 2 : g.N <= 64 ? 4 : 10;
        const int kch = NT <= 4 ? kNnKMax : planes ? 128 : 256;
        const size_t lds = (size_t)(planes ? 2 : 1) * 16 * NT * (kch + 8) * 2;
        const dim3 grid((unsigned)cdiv(g.M, 256), (unsigned)n);
        say(1001, 1);
        ProfScope prof(PROF_GEMM_NN, 2.0 * g.M * g.N * (planes ? 3.0 : 1.0) * g.K * n, 4.0 * n * ((double)g.M * g.K + (double)g.K * g.N + (double)g.M * g.N), stream, n);
#define ADN_SK_NN(NTv, PL, KCHv) do { \
        static bool attr_done[64] = {};                /* (a function attribute is per device) */ \
        int dev_ = 0; ADN_HIP_CHECK(hipGetDevice(&dev_)); \
        ADN_CHECK(dev_ >= 0 && dev_ < 64, ADN_ERR_STATE, "gemm_skinny: device ordinal out of range"); \
        if (!attr_done[dev_]) { ADN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_nn_kernel<NTv, PL, KCHv>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_done[dev_] = true; } \
        hipLaunchKernelGGL((skinny_nn_kernel<NTv, PL, KCHv>), grid, dim3(512), lds, stream, p); } while (0)
        if (planes) { if (NT == 2) ADN_SK_NN(2, true, kNnKMax); else if (NT == 4) ADN_SK_NN(4, true, kNnKMax); else ADN_SK_NN(10, true, 128); }
        else { if (NT == 2) ADN_SK_NN(2, false, kNnKMax); else if (NT == 4) ADN_SK_NN(4, false, kNnKMax); else ADN_SK_NN(10, false, 256); }
#undef ADN_SK_NN
        ADN_HIP_CHECK(hipGetLastError());
        *used = true;
        return ADN_OK;
    }
    // ---------------- input gradient behind a narrow layer: K <= 64, B = [N][K] k-contiguous
    if ((g.layout == GEMM_NT || g.layout == GEMM_NN) && g.Bkc16 && g.K <= 64 && g.N >= 64 && (g.N % 4 == 0 || !g.colsum) && (planes || all_bf16)) {
        const int ldb = g.ldbkc;
        if (g.act != ADN_ACT_LINEAR || g.lda % 8 || ldb % 8 || g.ldc % 4 || g.lda < (int)round_up(g.K, 8) || ldb < (int)round_up(g.K, 8)) return ADN_OK;
        if ((g.Y || g.Y16) && g.act_grad != ADN_ACT_LINEAR && (g.act_grad != ADN_ACT_RECTIFY || !g.Y16 || g.ldy % 4)) return ADN_OK;
        const int cs_ld = (int)round_up(g.N, 4);
        bool fused_colsum = g.colsum != nullptr;
        for (int k = 0; k < n; ++k) {
            const GemmArgs& q = gs[k];
            const void* bh = q.Bkc16; const void* bl = q.Bkc16lo;
            if (!q.A16 || !bh || (planes && (!q.A16lo || !bl)) || q.bias || (!q.C && !q.C16)) return ADN_OK;
            if (!aligned(q.A16, 16) || !aligned(bh, 16) || (q.C && !aligned(q.C, 16)) || (q.C16 && !aligned(q.C16, 8)) || (q.Y16 && !aligned(q.Y16, 8))) return ADN_OK;
            if (planes && (!aligned(q.A16lo, 16) || !aligned(bl, 16) || (q.C16lo && !aligned(q.C16lo, 8)))) return ADN_OK;
            if (q.accumulate && !q.C) return ADN_OK;
            if (q.colsum && (!q.colsum_ws || !aligned(q.colsum_ws, 16) || (size_t)cdiv(g.M, 64) * cs_ld > q.colsum_ws_floats)) fused_colsum = false;
        }
        if (g.colsum && !fused_colsum && !g.C) return ADN_OK;          // (lean result without fused sums: not here)
        if (dry) { *used = true; return ADN_OK; }
        p.ldb = ldb; p.colsum_ld = cs_ld;
        fill_groups(p, gs, n, planes, true);
        for (int k = 0; k < n; ++k) {
            SkinnyGroup& q = p.g[k];
            q.Y16 = (g.act_grad == ADN_ACT_RECTIFY) ? reinterpret_cast<const __bf16*>(gs[k].Y16) : nullptr;
            q.colsum = fused_colsum ? gs[k].colsum_ws : nullptr;
            const bool hi_res = gs[k].hi_product && gs[k].hi_result && q.C16 && q.C16lo;      // (the hi plane is the result)
            q.C16lo = ((planes || gs[k].hi_product) && !hi_res) ? q.C16lo : nullptr;
            const bool planes_out = q.C16 && (q.C16lo || hi_res);
            if (gs[k].planes_done) *gs[k].planes_done = planes_out ? 1 : 0;
            if (gs[k].fp32_skipped) *gs[k].fp32_skipped = 0;
            if (planes_out && gs[k].lean_ok && !g.accumulate && (!g.colsum || fused_colsum)) {
                q.C = nullptr;
                if (gs[k].fp32_skipped) *gs[k].fp32_skipped = 1;
            }
            if (gs[k].colsum_done) *gs[k].colsum_done = fused_colsum ? 1 : 0;
        }
        const int nchunks = cdiv(cdiv(g.N, 16), kNkTC);
        static const bool no_xcd = getenv("ADN_GEMM_SKINNY_NO_XCD") != nullptr;        // (A/B)
        p.rblocks = cdiv(g.M, 256); p.xcd_map = no_xcd ? 0 : 1;
        const dim3 grid((unsigned)((p.xcd_map ? (int)round_up(p.rblocks, 8) : p.rblocks) * nchunks), (unsigned)n);
        say(1002, 1);
        {
            ProfScope prof(PROF_GEMM_NN, 2.0 * g.M * g.N * (planes ? 3.0 : 1.0) * g.K * n, 4.0 * n * ((double)g.M * g.K + (double)g.K * g.N + (double)g.M * g.N), stream, n);
            if (planes) hipLaunchKernelGGL((skinny_nk_kernel<true>), grid, dim3(256), 0, stream, p, nchunks);
            else hipLaunchKernelGGL((skinny_nk_kernel<false>), grid, dim3(256), 0, stream, p, nchunks);
            ADN_HIP_CHECK(hipGetLastError());
        }
        for (int k = 0; k < n; ++k)
            if (fused_colsum) {
                if (gs[k].colsum_batch && gs[k].colsum_batch->n < 8) col_sum_batch_add(*gs[k].colsum_batch, gs[k].colsum_ws, cs_ld, cdiv(g.M, 64), g.N, gs[k].colsum);
                else ADN_TRY(col_sum(gs[k].colsum_ws, cs_ld, cdiv(g.M, 64), g.N, gs[k].colsum, 1, stream));
            }
        *used = true;
        return ADN_OK;
    }
    // ---------------- weight gradient of a narrow layer: TN, N <= 64, small M, long K
    // (plain bf16: the register-staged 64 x 64 kernel with its atomic split-K is as fast or faster on these -- 31.7 against 35.6 us
    //  for three 500 x 50 x 20800; over planes its alternative is two split passes + a three times deeper K: 150.7 against 55.4)
    static const bool tn_always = getenv("ADN_GEMM_SKINNY_TN") != nullptr;
    if (g.layout == GEMM_TN && g.N <= 64 && g.M <= 1024 && g.K >= 2048 && (planes || tn_always || all_bf16)) {
        for (int k = 0; k < n; ++k) {
            const GemmArgs& q = gs[k];
            if (!q.A16 || !q.B16 || (planes && (!q.A16lo || !q.B16lo)) || !q.C || q.bias || q.Y || q.Y16 || q.colsum) return ADN_OK;
            if (!aligned(q.A16, 16) || !aligned(q.B16, 16) || (planes && (!aligned(q.A16lo, 16) || !aligned(q.B16lo, 16)))) return ADN_OK;
        }
        if (g.act != ADN_ACT_LINEAR || g.lda % 8 || g.ldb % 8 || g.ldb < (int)round_up(g.N, 8)) return ADN_OK;
        const int mblocks = cdiv(g.M, 64);
        int splits = std::max(1, std::min(1024 / std::max(1, mblocks * n), g.K / 256));
        p.k_chunk = (int)round_up(cdiv(g.K, splits), 64);
        splits = cdiv(g.K, p.k_chunk);
        const size_t need = (size_t)n * splits * g.M * g.ldc;
        if (!g.splitk_ws || need > g.splitk_ws_floats) return ADN_OK;
        if (dry) { *used = true; return ADN_OK; }
        p.ldb = g.ldb; p.splits = splits; p.partial = g.splitk_ws;
        fill_groups(p, gs, n, planes, false);
        say(1003, splits);
        ProfScope prof(PROF_GEMM_TN, 2.0 * g.M * g.N * (planes ? 3.0 : 1.0) * g.K * n, 4.0 * n * ((double)g.M * g.K + (double)g.K * g.N + (double)g.M * g.N), stream, n);
        static const bool no_xcd_tn = getenv("ADN_GEMM_SKINNY_NO_XCD") != nullptr;     // (A/B)
        p.xcd_map = no_xcd_tn ? 0 : 1;
        const dim3 grid = p.xcd_map ? dim3((unsigned)round_up(splits, 8), (unsigned)mblocks, (unsigned)n) : dim3((unsigned)mblocks, (unsigned)splits, (unsigned)n);
        if (planes) hipLaunchKernelGGL((skinny_tn_kernel<true>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((skinny_tn_kernel<false>), grid, dim3(256), 0, stream, p);
        hipLaunchKernelGGL(skinny_tn_reduce_kernel, dim3((unsigned)std::min(1024, cdiv(g.M * g.N, 64)), (unsigned)n), dim3(256), 0, stream, p);
        ADN_HIP_CHECK(hipGetLastError());
        for (int k = 0; k < n; ++k)
            if (gs[k].C16 && g.ldc % 8 == 0) ADN_TRY(to_bf16(gs[k].C, gs[k].C16, (size_t)g.M * g.ldc, stream));
        *used = true;
        return ADN_OK;
    }
    return ADN_OK;
}

}  // namespace adn
