// fp32 GEMM on the gfx950 f32 MFMA pipe (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
//
// Used for every matmul-shaped op of the path that is not inside the recurrence:
//   dense encoder forward      Y  = act(X W + b)            (modelzoo/pretrained_encoder.py:4-9)   NN
//   input gradients            dX = (dY W^T) * act'(Y)                                              NT
//   weight gradients           dW = X^T dY                  (split-K, fp32 atomics)                 TN
//   LSTM input projections     xproj = X W_in + b           (Lasagne precompute_input [upstream])   NN
//
// Structure: 256 threads = 4 waves in a 2x2 arrangement over a BMxBN block tile, BK = 32 per LDS
// stage, register prefetch of the next stage while the current one feeds the MFMAs.  Each operand
// is staged in LDS in the orientation it has in HBM ("k-contiguous" rows read back with one
// ds_read_b128 per four k-steps, "k-strided" rows with four conflict-free ds_read_b32), so no
// transposing stores are needed for any of the three layouts.  The f32 MFMA issues once per 64
// cycles per SIMD, so one LDS fragment read per MFMA keeps the kernel MFMA-bound.
#include "gemm_common.h"
#include <cstdio>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

namespace adn {

constexpr int BK = 32;
constexpr int PAD = 4;

// One operand tile.  KC (k-contiguous): global [R rows][k], LDS [R][BK+PAD].
//                    !KC (k-strided):   global [k][R cols], LDS [BK][R+PAD].
template <int R, bool KC>
struct OperandTile {
    static constexpr int kVecPerThread = R * BK / 4 / 256;   // float4 per thread per stage
    static constexpr int kLdsFloats = KC ? R * (BK + PAD) : BK * (R + PAD);

    // r0: first row (KC) / first column (!KC) of the tile, rmax: logical extent in that dimension
    __device__ __forceinline__ static void load(float4 (&v)[kVecPerThread], const float* __restrict__ g, int ld,
                                                int r0, int rmax, int k0, int kend, int tid) {
#pragma unroll
        for (int i = 0; i < kVecPerThread; ++i) {
            const int f = tid + 256 * i;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (KC) {
                const int row = r0 + (f >> 3);
                const int k = k0 + ((f & 7) << 2);
                if (row < rmax && k < kend) {
                    x = *reinterpret_cast<const float4*>(g + (size_t)row * ld + k);
                    if (k + 1 >= kend) x.y = 0.f;
                    if (k + 2 >= kend) x.z = 0.f;
                    if (k + 3 >= kend) x.w = 0.f;
                }
            } else {
                constexpr int V = R / 4;       // float4 per k-row
                const int k = k0 + f / V;
                const int c = r0 + ((f % V) << 2);
                if (k < kend && c < rmax) {
                    x = *reinterpret_cast<const float4*>(g + (size_t)k * ld + c);
                    if (c + 1 >= rmax) x.y = 0.f;
                    if (c + 2 >= rmax) x.z = 0.f;
                    if (c + 3 >= rmax) x.w = 0.f;
                }
            }
            v[i] = x;
        }
    }

    __device__ __forceinline__ static void store(const float4 (&v)[kVecPerThread], float* lds, int tid) {
#pragma unroll
        for (int i = 0; i < kVecPerThread; ++i) {
            const int f = tid + 256 * i;
            if (KC) {
                *reinterpret_cast<float4*>(lds + (f >> 3) * (BK + PAD) + ((f & 7) << 2)) = v[i];
            } else {
                constexpr int V = R / 4;
                *reinterpret_cast<float4*>(lds + (f / V) * (R + PAD) + ((f % V) << 2)) = v[i];
            }
        }
    }

    // the four k-values {8*s + 4*h + q, q=0..3} of tile row/column `r` for this lane half h
    __device__ __forceinline__ static float4 frag(const float* lds, int r, int s, int h) {
        if (KC) {
            return *reinterpret_cast<const float4*>(lds + r * (BK + PAD) + 8 * s + 4 * h);
        } else {
            const float* p = lds + (8 * s + 4 * h) * (R + PAD) + r;
            return make_float4(p[0], p[R + PAD], p[2 * (R + PAD)], p[3 * (R + PAD)]);
        }
    }
};

template <int BM, int BN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmParams p) {
    using TA = OperandTile<BM, A_KC>;
    using TB = OperandTile<BN, B_KC>;
    constexpr int WTM = BM / 2, WTN = BN / 2;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    __shared__ __attribute__((aligned(16))) float smem[TA::kLdsFloats + TB::kLdsFloats];
    float* As = smem;
    float* Bs = smem + TA::kLdsFloats;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    int tile_m, tile_n;
    tile_coords(p, xcd_tile(blockIdx.x, gridDim.x), tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = blockIdx.y * p.k_chunk;
    const int kend = min(p.K, kbeg + p.k_chunk);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra[TA::kVecPerThread], rb[TB::kVecPerThread];
    TA::load(ra, p.A, p.lda, m0, p.M, kbeg, kend, tid);
    TB::load(rb, p.B, p.ldb, n0, p.N, kbeg, kend, tid);

    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        TA::store(ra, As, tid);
        TB::store(rb, Bs, tid);
        __syncthreads();
        if (k0 + BK < kend) {
            TA::load(ra, p.A, p.lda, m0, p.M, k0 + BK, kend, tid);
            TB::load(rb, p.B, p.ldb, n0, p.N, k0 + BK, kend, tid);
        }
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = TA::frag(As, wm * WTM + a * 32 + i, s, h);
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[b] = TB::frag(Bs, wn * WTN + b * 32 + i, s, h);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[b].w, acc[a][b], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    const bool first_split = blockIdx.y == 0;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
            store_tile32(p, acc[a][b], m0 + wm * WTM + a * 32, n0 + wn * WTN + b * 32, lane, first_split);
}

// deterministic split-K of the register-staged kernels: C (+)= the K-slices' slabs ([problem][slice][M][ldc]) in slice order
struct RsReduceArgs { float* C[kMaxGemmGroups]; };
__global__ __launch_bounds__(256) void rs_splitk_reduce_kernel(const RsReduceArgs a, const float* __restrict__ partial, int M, int N, int ldc,
                                                               int splits, int accumulate) {
    const int g = blockIdx.z;
    float* C = g == 0 ? a.C[0] : g == 1 ? a.C[1] : g == 2 ? a.C[2] : a.C[3];
    const float* part = partial + (size_t)g * splits * M * ldc;
    const size_t slab = (size_t)M * ldc;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)M * N; e += (int64_t)gridDim.x * 256) {
        const size_t i = (size_t)(e / N) * ldc + (size_t)(e % N);
        float v = accumulate ? C[i] : 0.f;
        for (int s = 0; s < splits; ++s) v += part[(size_t)s * slab + i];
        C[i] = v;
    }
}

template <int BM, int BN>
static void launch(const GemmParams& p, int layout, dim3 grid, hipStream_t s) {
    switch (layout) {
        case GEMM_NN: hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, false>), grid, dim3(256), 0, s, p); break;
        case GEMM_NT: hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, true>), grid, dim3(256), 0, s, p); break;
        default:      hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, false>), grid, dim3(256), 0, s, p); break;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Persistent ping-pong kernel (gemm_bf16.hip): applicability, tile shape, split-K and grouping
// ---------------------------------------------------------------------------------------------------------
static int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    return cus;
}

// Runs `n` problems of identical shape / layout / flags through the ping-pong kernel if it applies; *used says whether.
// dry: only answer whether the kernel WOULD take the launch (no side effects)
constexpr int kDefaultPpMode = 4;       // tile mode the selection uses unless ADN_GEMM_PP forces one (4: eight waves, 7: four waves)
static int gemm_pp_try_impl(const GemmArgs* gs, int n, const GemmArgs& g, int kseg, hipStream_t stream, bool* used, bool dry);
// bf16x3 over planes: the fused-plane kernel (gemm_x3f.hip: the four planes of a 16-k unit staged once, three products from four
// fragment sets) instead of the ping-pong kernel's three K-segments.  Measured (profiles/r05/lab_x3f.txt, MI355X): a third less
// LDS-DMA and fragment traffic per flop buys nothing in wall time on the NN shapes (0.97 - 1.00 x: the chip holds ~1.5 GHz under
// either kernel -- both are bound by the energy of their MFMAs, and the 32x32x16 shape the 16-k unit needs holds a lower clock
// than 16x16x32, MI355X_MICROARCH.md 'DVFS give-back' item 7) and 2 - 7 % on the weight gradients (TN), where it also takes any K
// (the three-segment walk needs K % 32 == 0).  In the train step itself (profiles/r05/mode_costs.txt, one box, bf16x3 ms per step):
// never 7.717, TN only 7.662, every launch over planes 7.582 -- the default.  ADN_GEMM_NO_X3F=1: never; ADN_GEMM_X3F=tn: TN only
static bool x3f_enabled() {
    static const bool off = getenv("ADN_GEMM_NO_X3F") != nullptr;
    return !off;
}
// the tile mode a product over planes runs in: 8 = fused-plane kernel, otherwise the ping-pong kernel's (forced or default) mode
static int planes_tile_mode(int layout, int mode_env, int default_mode) {
    static const bool tn_only = getenv("ADN_GEMM_X3F") && !strcmp(getenv("ADN_GEMM_X3F"), "tn");
    if (mode_env >= 4) return (mode_env == 8 && !x3f_enabled()) ? default_mode : mode_env;
    return (x3f_enabled() && (!tn_only || layout == GEMM_TN)) ? 8 : default_mode;
}
void launch_gemm_x3f(const GemmParams& p, int layout, bool planes, int splits, dim3 grid, hipStream_t s, bool reduce = true);      // gemm_x3f.hip
static int gemm_pp_try(const GemmArgs* gs0, int n, hipStream_t stream, bool* used, bool dry = false) {
    *used = false;
    if (n < 1 || n > kMaxGemmGroups) return ADN_OK;
    // an output width that is not a multiple of 4 (the 150-column input gradient of the stream LSTMs): the kernel's epilogue works
    // in float4 -- widened into C's pad columns where B's pad columns are known to be zero (the transposed weight copies)
    GemmArgs widened[kMaxGemmGroups];
    const GemmArgs* gs = gs0;
    if (gs0[0].N % 4 && gs0[0].layout == GEMM_NN) {
        const int nr = (int)round_up(gs0[0].N, 4);
        for (int k = 0; k < n; ++k) {
            const GemmArgs& q = gs0[k];
            if (!q.b_pad_zero || q.N != gs0[0].N || q.ldc < nr || q.ldb < nr || q.bias || q.Y || q.Y16 || q.colsum || q.accumulate) return ADN_OK;
            widened[k] = q; widened[k].N = nr;
        }
        gs = widened;
    }
    const GemmArgs& g = gs[0];
    static const int mode_env = getenv("ADN_GEMM_PP") ? atoi(getenv("ADN_GEMM_PP")) : -1;   // 0: off, 4/5/6: force a tile shape
    if (mode_env == 0 || n > kMaxGemmGroups) return ADN_OK;
    if (g.precision != ADN_PRECISION_BF16 || g.layout == GEMM_NT) return ADN_OK;
    // bf16x3 through hi / lo planes (see GemmArgs::A16lo): three K-segments of kseg = K rounded up to the kernel's K-step.
    // The pad k of a k-contiguous A are its zero pad columns (lda >= kseg); k-strided operands have no pad: K % 32 == 0 then.
    const bool planes = g.A16lo != nullptr || g.B16lo != nullptr;
    const int kseg = planes ? (int)round_up(g.K, 32) : 0;
    if (planes) {
        if (!g.A16lo || !g.B16lo || !g.A16 || !g.B16) return ADN_OK;
        // (NN: the partial last stage of a segment is masked in the A fragments -- the columns of A behind K may hold anything;
        //  the k-strided B is then read up to 31 rows past K: finite values of the next tensor or zero slack, times zero.
        //  TN: both operands are k-strided and neither is masked: K % 32 == 0 there, or the image path)
        // (the fused-plane kernel masks its A fragments in both layouts and clamps the k-rows it fetches: any K)
        if (g.layout != GEMM_NN && g.K % 32 != 0 && planes_tile_mode(g.layout, mode_env, kDefaultPpMode) != 8) return ADN_OK;
        if (mode_env == 7) return ADN_OK;         // (the experimental four-wave kernel has no segment cursors)
    }
    GemmArgs gv = g;                              // the problem the tile / split heuristics see: K = all three segments
    if (planes) gv.K = 3 * kseg;
    return gemm_pp_try_impl(gs, n, gv, kseg, stream, used, dry);
}

static int gemm_pp_try_impl(const GemmArgs* gs, int n, const GemmArgs& g, int kseg, hipStream_t stream, bool* used, bool dry) {
    static const int mode_env = getenv("ADN_GEMM_PP") ? atoi(getenv("ADN_GEMM_PP")) : -1;
    // (one output of < 256 rows wastes too much of its 256-row tiles; several of them in one launch still win: three
    //  250 x 1000 x 20800 weight gradients 66 against 3 x 33 us, three 150 x 1000 62 against 3 x 30)
    // (narrow outputs: a 150-column input gradient fills 59 % of its one tile column -- over planes still 2.6x the register-staged
    //  kernel's rate on split images, 232 -> 9x us for the three streams' 20800 x 150 x 1000)
    static const int min_n_env = getenv("ADN_GEMM_PP_MIN_N") ? atoi(getenv("ADN_GEMM_PP_MIN_N")) : 0;
    const int min_n = g.pp_force ? 96 : (min_n_env ? min_n_env : (kseg ? 128 : 256));
    if (g.M < (n >= 2 ? 128 : 256) || g.N < min_n || g.K < 256) return ADN_OK;
    if (g.N % 4 || g.ldc % 4 || g.lda % 8 || g.ldb % 8) return ADN_OK;
    if (g.layout == GEMM_NN && g.K % 8 && g.lda < round_up(g.K, 8)) return ADN_OK;
    // epilogue forms of the kernel: linear / rectify output; optional rectify'(Y) from the bf16 copy of Y
    if (g.act != ADN_ACT_LINEAR && g.act != ADN_ACT_RECTIFY) return ADN_OK;
    if ((g.Y || g.Y16) && g.act_grad != ADN_ACT_LINEAR && (g.act_grad != ADN_ACT_RECTIFY || !g.Y16 || g.ldy % 4)) return ADN_OK;
    for (int k = 0; k < n; ++k) {
        const GemmArgs& q = gs[k];
        if (!q.A16 || !q.B16 || ((uintptr_t)q.A16 % 16) || ((uintptr_t)q.B16 % 16)) return ADN_OK;
        if (!q.C && !q.C16) return ADN_OK;
        if ((q.C && ((uintptr_t)q.C % 16)) || (q.C16 && (((uintptr_t)q.C16 % 16) || g.ldc % 8))) return ADN_OK;
        if ((q.Y && ((uintptr_t)q.Y % 16)) || (q.Y16 && ((uintptr_t)q.Y16 % 8)) || (q.bias && ((uintptr_t)q.bias % 16))) return ADN_OK;
        if ((q.A16lo == nullptr) != (gs[0].A16lo == nullptr) || (q.B16lo == nullptr) != (gs[0].B16lo == nullptr)) return ADN_OK;
        if (q.M != g.M || q.N != g.N || q.K != gs[0].K || q.lda != g.lda || q.ldb != g.ldb || q.ldc != g.ldc || q.ldy != g.ldy ||
            q.layout != g.layout || q.act != g.act || q.act_grad != g.act_grad || q.accumulate != g.accumulate || q.hi_product != g.hi_product ||
            q.no_split != g.no_split || q.pp_force != g.pp_force || (q.C == nullptr) != (g.C == nullptr) || (q.C16 == nullptr) != (g.C16 == nullptr) ||
            (q.bias == nullptr) != (g.bias == nullptr) || (q.Y16 == nullptr) != (g.Y16 == nullptr) ||
            (q.Y == nullptr) != (g.Y == nullptr) || (q.colsum == nullptr) != (g.colsum == nullptr))
            return ADN_OK;
    }
    const int cus = device_cus();
    const bool lean_c = !g.C;
    const bool can_split = g.act == ADN_ACT_LINEAR && !lean_c && !g.no_split && !g.bias && !g.Y && !g.Y16 && !g.colsum;
    // ---- where it pays (profiles/r02/gemm_lab_pp.txt, MI355X, against the register-staged 128 x 128 kernel):
    //   * split-K weight gradients with outputs of >= 0.4 M elements: dW fc1 / fc2 125 / 102 us against 178 / 139 (800 - 818
    //     TFLOP/s; partial slabs + reduce instead of 64 MB of float atomics), 768 x 1000 (the aggregation LSTMs' input
    //     weights) 55 against 81, 1000 x 500 50 against 54
    //   * forward GEMMs with a plain epilogue (bias / rectify, no act'(Y) loads, no fused column sums) whose tiles fill
    //     >= 80 % of their rounds of 256 CUs -- in practice the input streams' encoder layers as ONE grouped launch:
    //     110 against 134 us per stream for 20800 x 1000 x 2000 (three streams fill 3.8 rounds where one fills 1.3 and
    //     loses the difference to its last round), 142 against 161 us for 20800 x 2000 x 1200
    //   * NOT the input-gradient GEMMs: the transposed-accumulator epilogue reads the act'(Y) mask in 32-byte
    //     row pieces (205 against 146 us), and not 128-wide tiles (445 TFLOP/s: B-fragment reads per flop double)
    struct Cand { int mode, bm, bn; double rate; int wave_rows; };
    // mode 7: the four-wave kernel (128 x 128 per wave); ADN_GEMM_PP=4 forces the eight-wave kernel
    // mode 8: the fused-plane kernel (gemm_x3f.hip) -- the default over planes; ADN_GEMM_PP=8 also sends plain bf16 products to it
    static const Cand cands[5] = {{7, 256, 256, 1.0, 2}, {4, 256, 256, 1.0, 4}, {5, 256, 128, 0.70, 4}, {6, 128, 256, 0.70, 4}, {8, 256, 256, 1.0, 4}};
    const int want_mode = kseg ? planes_tile_mode(g.layout, mode_env, kDefaultPpMode) : (mode_env >= 4 ? mode_env : kDefaultPpMode);
    int best = -1, splits = 1; double best_cost = 0;
    for (int c = 0; c < 5; ++c) {
        if (cands[c].mode != want_mode) continue;
        const int64_t tiles = (int64_t)cdiv(g.M, cands[c].bm) * cdiv(g.N, cands[c].bn) * n;
        int sp = 1;
        if (tiles * 2 <= cus && can_split && g.K >= 2048) {
            if (cands[c].mode != 4 && cands[c].mode != 7 && cands[c].mode != 8) continue;
            sp = (int)std::min<int64_t>(cus / tiles, g.K / 512);
        }
        const int64_t rounds = (tiles * sp + cus - 1) / cus;
        const double cost = (double)rounds * cands[c].bm * cands[c].bn / sp / cands[c].rate;
        if (best < 0 || cost < best_cost) { best = c; best_cost = cost; splits = sp; }
    }
    if (best < 0) return ADN_OK;
    if (mode_env < 4 && !g.pp_force) {
        const double fill = ((double)g.M * g.N * n / cus) / best_cost;       // useful share of the tile-rounds
        const bool plain = !g.Y && !g.Y16 && !g.colsum;
        // (several weight gradients in one launch only when each alone has too few tiles to split well: three 1000 x 500
        //  162 -> 89 us, but three 2000 x 1000 332 against 3 x 97)
        const int64_t per_group_tiles = (int64_t)cdiv(g.M, cands[best].bm) * cdiv(g.N, cands[best].bn);
        // (... or when the joint launch still fills three quarters of the device: three 1200 x 2000 take 359 us grouped with 2
        //  slabs each against 3 x 120 with 6 slabs each -- the same GEMM time and a third of the reduce traffic; three
        //  2000 x 1000 (192 workgroups, 2 slabs) 223 us against 3 x 72 with 8 slabs: 0.015 ms less per step with the reduce)
        const bool wgrad = splits > 1 && (int64_t)g.M * g.N * n >= 400000 &&
                           (n == 1 || per_group_tiles <= 16 || per_group_tiles * n * splits * 4 >= (int64_t)cus * 3);
        // (accumulate: the epilogue would read C back -- bf16 mode leaves those to the register-staged kernel; over planes the
        //  alternative is a split pass per operand, so the ping-pong kernel takes them)
        static const double fill_env = getenv("ADN_GEMM_PP_FILL") ? atof(getenv("ADN_GEMM_PP_FILL")) : 0.0;
        const bool fwd_group = splits == 1 && plain && (!g.accumulate || kseg) && fill >= (fill_env > 0 ? fill_env : kseg ? 0.50 : 0.80);
        // input-gradient GEMMs of several streams (act'(Y) mask from the bf16 copy, fused column sums): 445 against 3 x 153 us
        // for 20800 x 2000 x 1000, 178 against 3 x 59 for 20800 x 1000 x 500; one alone is no faster than the 128 x 128 kernel
        // (0.85 when the rule was measured at B T = 20800 rows -- fill 0.93; the compacted step's 13 730 rows fill 0.82 of their 6 / 3
        //  rounds and the register-staged alternative costs 280 us for 13730 x 2000 x 1000 x 3: 0.80.  ADN_GEMM_PP_BWD_FILL overrides)
        static const double bwd_fill_env = getenv("ADN_GEMM_PP_BWD_FILL") ? atof(getenv("ADN_GEMM_PP_BWD_FILL")) : 0.0;
        const bool bwd_group = splits == 1 && n >= 2 && g.Y16 && fill >= (bwd_fill_env > 0 ? bwd_fill_env : 0.80);
        if (!wgrad && !fwd_group && !bwd_group) return ADN_OK;
    }
    const Cand& cd = cands[best];

    GemmParams p;
    std::memset(static_cast<void*>(&p), 0, sizeof(p));
    p.M = g.M; p.N = g.N; p.K = g.K; p.lda = g.lda; p.ldb = g.ldb; p.ldc = g.ldc; p.ldy = g.ldy;
    p.act = g.act; p.act_grad = g.act_grad; p.accumulate = g.accumulate;
    p.one_barrier = 1;
    p.kseg = kseg; p.kreal = gs[0].K;
    p.tiles_m = cdiv(g.M, cd.bm); p.tiles_n = cdiv(g.N, cd.bn);
    const bool fused = cd.mode == 8;
    if (fused) { p.K = kseg ? gs[0].K : g.K; p.kseg = 0; }       // the fused kernel walks the real K once
    p.k_chunk = (int)round_up(cdiv(p.K, splits), 32);
    splits = cdiv(p.K, p.k_chunk);
    p.ngroups = n;
    const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n * n;
    const int cs_ld = (int)round_up(g.N, 4);
    if (splits > 1) {
        const size_t need = (size_t)n * splits * g.M * g.ldc;
        if (!g.splitk_ws || need > g.splitk_ws_floats || ((uintptr_t)g.splitk_ws % 16)) return ADN_OK;
        p.partial = g.splitk_ws;
    }
    if (dry) { *used = true; return ADN_OK; }
    // ---- tail band (round 6).  The tile list of an NN launch is walked by 256 persistent workgroups in rounds; a list that ends
    // just behind a round boundary -- the compacted bench step: 54 row tiles x 8 x 3 streams = 1296 tiles = 5.06 rounds for the
    // first layer and for its input gradient -- pays a whole round for a handful of tiles.  Then the last row tiles are taken off
    // the list (the main launch ends on full rounds) and their rows run K-SPLIT over the whole device: S slices per tile into partial
    // slabs, and splitk_tail_epilogue_kernel sums the slabs in slice order and applies this launch's epilogue (gemm_bf16.hip).  Same
    // products; the band's rows are summed in another order than the main rows (fixed: reproducible).
    // MEASURED, round 6, one box, alternating runs (profiles/r06/ab_tail_split.txt): bf16 3.162 / 3.163 ms per step with it against
    // 3.170 / 3.077 without, bf16x3 6.040 / 6.035 against 6.007 / 5.998, mixed 4.390 / 4.395 against 4.323 / 4.333 -- it does NOT pay:
    // the sixth "round" of 16 tiles costs less than a round (those workgroups run alone: full clocks, the whole L2 and HBM), and the
    // band's slabs + two more launches cost about what it saves.  Kept as an experiment switch: ADN_GEMM_TAIL_SPLIT=1 turns it on.
    int band_rt = 0, band_rows = 0, band_S = 1;
    {
        static const bool no_tail = getenv("ADN_GEMM_TAIL_SPLIT") == nullptr;
        const int per_row = p.tiles_n * n;
        const int64_t full = tiles / cus, rem = tiles - full * cus;
        if (!no_tail && !dry && splits == 1 && g.layout == GEMM_NN && !g.accumulate && cd.bm == 256 && (cd.mode == 4 || cd.mode == 8) &&
            (fused || !kseg) && full >= 2 && rem > 0 && rem * 4 <= cus && per_row > 0) {
            const int rt = (int)((rem + per_row - 1) / per_row);           // row tiles that leave the main launch
            const int64_t bt = (int64_t)rt * per_row;
            int S = (int)(cus / std::max<int64_t>(1, bt));
            S = S >= 8 ? 8 : (S >= 4 ? 4 : 1);
            const int rows = g.M - (p.tiles_m - rt) * cd.bm;
            const int cs_rows = (p.tiles_m - rt) * cd.wave_rows + cdiv(rows, kTailEpiRows);
            bool ok = S > 1 && rt < p.tiles_m && rows > 0 && p.K / S >= 64 && g.splitk_ws && ((uintptr_t)g.splitk_ws % 16) == 0 &&
                      (size_t)n * S * rows * g.ldc <= g.splitk_ws_floats;
            if (ok && g.colsum)                                             // (fused sums: the band's partial rows must fit behind the main launch's)
                for (int k = 0; k < n; ++k)
                    ok = ok && (!gs[k].colsum_ws || (size_t)cs_rows * cs_ld <= gs[k].colsum_ws_floats);
            if (ok) { band_rt = rt; band_rows = rows; band_S = S; }
        }
    }
    const int main_tiles_m = p.tiles_m - band_rt;
    bool fused_colsum = false;
    if (g.colsum && splits == 1) {
        fused_colsum = true;
        for (int k = 0; k < n; ++k)
            if (!gs[k].colsum_ws || ((uintptr_t)gs[k].colsum_ws % 16) || (size_t)p.tiles_m * cd.wave_rows * cs_ld > gs[k].colsum_ws_floats) fused_colsum = false;
    }
    p.colsum_ld = cs_ld;
    for (int k = 0; k < n; ++k) {
        GemmGroup& q = p.grp[k];
        q.A16 = gs[k].A16; q.B16 = gs[k].B16; q.C = gs[k].C; q.C16 = splits > 1 ? nullptr : gs[k].C16;
        q.A16lo = kseg ? gs[k].A16lo : nullptr; q.B16lo = kseg ? gs[k].B16lo : nullptr;
        const bool hi_res = gs[k].hi_product && gs[k].hi_result && splits == 1 && q.C16 && gs[k].C16lo;      // (the hi plane is the result)
        q.C16lo = ((kseg || gs[k].hi_product) && splits == 1 && q.C16 && !hi_res) ? gs[k].C16lo : nullptr;
        const bool planes_out = q.C16lo || hi_res;
        if (gs[k].planes_done) *gs[k].planes_done = planes_out ? 1 : 0;
        // (planes: the fp32 copy of a result every reader takes from its planes is not written -- unless the bias gradient
        //  that rides on this launch could not be fused and will be summed from the fp32 values)
        if (gs[k].fp32_skipped) *gs[k].fp32_skipped = 0;
        if (planes_out && gs[k].lean_ok && !g.accumulate && (!g.colsum || fused_colsum)) {
            q.C = nullptr;
            if (gs[k].fp32_skipped) *gs[k].fp32_skipped = 1;
        }
        q.Y16 = (g.act_grad == ADN_ACT_RECTIFY) ? gs[k].Y16 : nullptr; q.Y = nullptr; q.bias = gs[k].bias;
        q.colsum = fused_colsum ? gs[k].colsum_ws : nullptr;
        if (gs[k].colsum_done) *gs[k].colsum_done = fused_colsum ? 1 : 0;
        // rectifier bit images: this kernel, whole 256 x 256 tile grid in one launch, one product per element
        static const bool no_bits = getenv("ADN_NO_RELU_BITS") != nullptr;
        // (the key names kernel AND tile grid: the two kernels number their accumulator quads differently, and the mixed arithmetic
        //  runs its forward pass on one and its back-propagation on the other -- there the mask stays the bf16 activation's)
        const bool bits_ok = !no_bits && (cd.mode == 4 || (cd.mode == 8 && kseg)) && cd.bm == 256 && cd.bn == 256 && splits == 1 && band_rt == 0 &&
                             g.layout == GEMM_NN;
        const int tiles_key = (cd.mode << 26) | (p.tiles_m << 13) | p.tiles_n;
        q.Cbits = (bits_ok && g.act == ADN_ACT_RECTIFY && !g.accumulate && ((uintptr_t)gs[k].Cbits % 16) == 0) ? gs[k].Cbits : nullptr;
        if (gs[k].bits_done) *gs[k].bits_done = q.Cbits ? tiles_key : 0;
        q.Ybits = (bits_ok && q.Y16 && gs[k].Ybits && gs[k].Ybits_tiles == tiles_key && ((uintptr_t)gs[k].Ybits % 16) == 0) ? gs[k].Ybits : nullptr;
    }
    static const bool trace = getenv("ADN_GEMM_TRACE") != nullptr;
    auto say = [&](int M_, long long tiles_, int split_) {
        if (trace)
            fprintf(stderr, "ADN_GEMM %s M=%d N=%d K=%d tile=%d tiles=%lld split=%d shadows=1 lean=%d acc=%d groups=%d%s\n",
                    g.layout == GEMM_NN ? "NN" : "TN", M_, g.N, g.K, cd.bm * 1000 + cd.bn, tiles_, split_, (int)lean_c,
                    g.accumulate, n, kseg ? (fused ? " planes=1 fused=1" : " planes=1") : (fused ? " fused=1" : ""));
    };
    auto launch = [&](const GemmParams& q, int sp, dim3 grid, bool reduce) {
        if (fused) launch_gemm_x3f(q, g.layout, kseg != 0, sp, grid, stream, reduce);
        else launch_gemm_bf16_pp(q, g.layout, cd.mode, sp, grid, stream, reduce);
    };
    auto set_panel = [](GemmParams& q) {   // square-ish per-XCD tile blocks (per group)
        const int chunk = (int)std::max<int64_t>(1, (int64_t)q.tiles_m * q.tiles_n / 8);
        q.panel_n = std::max(1, std::min((int)std::lround(std::sqrt((double)chunk)), q.tiles_n));
    };
    {
        ProfScope prof(PROF_GEMM_NN + g.layout, 2.0 * g.M * g.N * g.K * n,
                       4.0 * n * ((double)g.M * g.K + (double)g.K * g.N + (double)g.M * g.N), stream, n);
        if (band_rt > 0) {
            // main launch: the row tiles that fill whole rounds
            GemmParams pm = p;
            pm.M = main_tiles_m * cd.bm; pm.tiles_m = main_tiles_m;
            set_panel(pm);
            const int64_t tiles_main = (int64_t)pm.tiles_m * pm.tiles_n * n;
            int gxm = (int)std::min<int64_t>(tiles_main, cus);
            if (gxm >= 8) gxm = gxm / 8 * 8;
            say(pm.M, (long long)tiles_main, 1);
            launch(pm, 1, dim3((unsigned)gxm, 1u), true);
            // the band, K-split into slabs (no epilogue in the kernel) ...
            GemmParams pt = p;
            pt.M = band_rows; pt.tiles_m = band_rt;
            set_panel(pt);
            pt.partial = g.splitk_ws;
            pt.k_chunk = (int)round_up(cdiv(pt.K, band_S), 32);
            const int S = cdiv(pt.K, pt.k_chunk);
            for (int k = 0; k < n; ++k) {
                GemmGroup& q = pt.grp[k];
                q.A16 = static_cast<const char*>(q.A16) + (size_t)pm.M * g.lda * 2;
                if (q.A16lo) q.A16lo = static_cast<const char*>(q.A16lo) + (size_t)pm.M * g.lda * 2;
                q.C16 = nullptr; q.C16lo = nullptr; q.Y16 = nullptr; q.bias = nullptr; q.colsum = nullptr;
            }
            const int gxt = band_rt * pt.tiles_n * n;
            static const bool no_xcd = getenv("ADN_GEMM_NO_XCD_SLICES") != nullptr;
            pt.xcd_slices = !no_xcd && (S % 8 == 0 || 8 % S == 0) && ((int64_t)gxt * S) % 8 == 0;
            say(band_rows, (long long)gxt, S);
            launch(pt, S, dim3((unsigned)gxt, (unsigned)S), false);
            // ... and its epilogue over the summed slabs, at rows pm.M .. of the problems' matrices
            GemmParams pe = p;
            pe.M = band_rows; pe.partial = g.splitk_ws;
            launch_splitk_tail_epilogue(pe, S, pm.M, main_tiles_m * cd.wave_rows, stream);
        } else {
            set_panel(p);
            int gx = (int)std::min<int64_t>(tiles, cus / splits);
            if (splits > 1) {
                gx = (int)tiles;                      // one (tile, slice) per workgroup
                static const bool no_xcd = getenv("ADN_GEMM_NO_XCD_SLICES") != nullptr;
                p.xcd_slices = !no_xcd && (splits % 8 == 0 || 8 % splits == 0) && ((int64_t)gx * splits) % 8 == 0;
            } else if (gx >= 8) gx = gx / 8 * 8;      // a workgroup's tiles then stay on its own XCD's chunk of the tile list
            say(g.M, (long long)tiles, splits);
            launch(p, splits, dim3((unsigned)gx, (unsigned)splits), true);
        }
        ADN_HIP_CHECK(hipGetLastError());
    }
    const int cs_part_rows = band_rt > 0 ? main_tiles_m * cd.wave_rows + cdiv(band_rows, kTailEpiRows) : p.tiles_m * cd.wave_rows;
    for (int k = 0; k < n; ++k) {
        if (fused_colsum) {
            if (gs[k].colsum_batch && gs[k].colsum_batch->n < 8)
                col_sum_batch_add(*gs[k].colsum_batch, gs[k].colsum_ws, cs_ld, cs_part_rows, g.N, gs[k].colsum);
            else ADN_TRY(col_sum(gs[k].colsum_ws, cs_ld, cs_part_rows, g.N, gs[k].colsum, 1, stream));
        }
        if (gs[k].C16 && splits > 1) {            // split-K result: refresh the bf16 shadow of whole rows
            ADN_CHECK(g.ldc % 8 == 0, ADN_ERR_INVALID, "gemm: bf16 shadow of C needs ldc % 8 == 0");
            ADN_TRY(to_bf16(gs[k].C, gs[k].C16, (size_t)g.M * g.ldc, stream));
        }
    }
    *used = true;
    return ADN_OK;
}

static bool x3_takes(const GemmArgs& g);
static int gemm_bf16x3_images(const GemmArgs& g, hipStream_t stream);
static int x3_try_planes(const GemmArgs* gs, int n, hipStream_t stream, bool* used, bool dry);
static int gemm_grouped_bf16x3(const GemmArgs* gs, int n, hipStream_t stream);
static int gemm_rs(const GemmArgs* gs, int n, hipStream_t stream);
static bool rs_groupable(const GemmArgs* gs, int n);

// would gemm_grouped() multiply these n problems over their hi / lo planes (bf16x3 mode)?  model.hip asks before it builds
// operands that exist only as planes (the materialised concat)
bool gemm_planes_would_run(const GemmArgs* gs, int n) {
    bool used = false;
    if (n < 1 || n > kMaxGemmGroups) return false;
    if (gs[0].precision == ADN_PRECISION_BF16X3 && gemm_skinny_try(gs, n, nullptr, &used, /*dry=*/true) == ADN_OK && used) return true;
    if (x3_try_planes(gs, n, nullptr, &used, /*dry=*/true) != ADN_OK) return false;
    return used;
}

int gemm_grouped(const GemmArgs* gs, int n, hipStream_t stream) {
    if (n <= 0) return ADN_OK;
    if (gs[0].M <= 0 || gs[0].N <= 0) return ADN_OK;
    if (n <= kMaxGemmGroups) {                     // the skinny shapes: streaming kernels (gemm_skinny.hip)
        bool used = false;
        ADN_TRY(gemm_skinny_try(gs, n, stream, &used, false));
        if (used) return ADN_OK;
    }
    if (n >= 1 && n <= kMaxGemmGroups && gs[0].precision == ADN_PRECISION_BF16X3) {       // over the operands' planes first
        bool used = false;
        ADN_TRY(x3_try_planes(gs, n, stream, &used, false));
        if (used) return ADN_OK;
    }
    if (n > 1 && n <= kMaxGemmGroups && gs[0].precision == ADN_PRECISION_BF16X3 && !getenv("ADN_X3_NO_GROUPS")) {
        bool all = true;
        for (int k = 0; k < n; ++k)
            all = all && x3_takes(gs[k]) && gs[k].precision == ADN_PRECISION_BF16X3 && gs[k].layout == gs[0].layout &&
                  gs[k].M == gs[0].M && gs[k].N == gs[0].N && gs[k].K == gs[0].K;
        if (all) return gemm_grouped_bf16x3(gs, n, stream);
    }
    bool used = false;
    if (n > 1) ADN_TRY(gemm_pp_try(gs, n, stream, &used));
    if (used) return ADN_OK;
    // (a problem the ping-pong kernel takes alone -- the 2000 x 1000 weight gradients: 3 x 80 us against 450 us as one
    //  register-staged launch -- goes out alone)
    if (n > 1) ADN_TRY(gemm_pp_try(gs, 1, stream, &used, /*dry=*/true));
    if (!used && rs_groupable(gs, n)) return gemm_rs(gs, n, stream);
    for (int k = 0; k < n; ++k) ADN_TRY(gemm(gs[k], stream));
    return ADN_OK;
}

// ---- bf16x3 (ADN_PRECISION_BF16X3): split both operands into [hi | hi | lo] / [hi ; lo ; hi] images along k (gemm_bf16.hip)
// and run the bf16 kernels on a three times deeper K.  The images live in a workspace that belongs to the STREAM the GEMM is
// enqueued on (consecutive GEMMs of a stream reuse it in order; two streams never share one); it grows on demand and is
// kept for the life of the process.
namespace {
struct X3Workspace { void* ptr = nullptr; size_t bytes = 0; };
std::mutex g_x3_mutex;
std::map<std::pair<int, hipStream_t>, X3Workspace> g_x3_ws;

int x3_workspace(hipStream_t stream, size_t bytes, void** out) {
    int dev = 0;
    ADN_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_x3_mutex);
    X3Workspace& w = g_x3_ws[std::make_pair(dev, stream)];
    if (w.bytes < bytes) {
        if (w.ptr) { ADN_HIP_CHECK(hipStreamSynchronize(stream)); ADN_HIP_CHECK(hipFree(w.ptr)); w.ptr = nullptr; w.bytes = 0; }
        const size_t want = std::max(bytes + bytes / 8, (size_t)64 << 20);
        ADN_HIP_CHECK(hipMalloc(&w.ptr, want));
        w.bytes = want;
    }
    *out = w.ptr;
    return ADN_OK;
}
}  // namespace

// operand images of one product: sizes, then the split passes + the bf16 problem that multiplies them
struct X3Plan {
    int Kp, lda3, ldb3;
    bool a_rows, b_transpose;
    size_t a_bytes, b_bytes;
};

static X3Plan x3_plan(const GemmArgs& g) {
    X3Plan p;
    p.Kp = (int)round_up(g.K, 8);
    p.a_rows = g.layout == GEMM_TN;                              // A given as [K][M]: k along its rows
    // B given as [N][K] (NT: the input-gradient GEMMs dZ W^T) is written TRANSPOSED into the k-strided form, so that the
    // product runs through the NN kernels (both operands k-contiguous is the slow form of the bf16 kernels: 3.4 -> 2.3 ms
    // per train step at the bench geometry)
    p.b_transpose = g.layout == GEMM_NT;
    p.lda3 = p.a_rows ? (int)round_up(g.M, 64) : 3 * p.Kp;
    p.ldb3 = (int)round_up(g.N, 64);
    const size_t a_elems = p.a_rows ? (size_t)3 * p.Kp * p.lda3 : (size_t)g.M * p.lda3;
    p.a_bytes = (size_t)round_up((int64_t)a_elems * 2, 256);
    p.b_bytes = (size_t)round_up((int64_t)3 * p.Kp * p.ldb3 * 2, 256);
    return p;
}

// (split_a / split_b false: the image at A3 / B3 is already there -- an operand shared with an earlier problem of the group)
static int x3_split_operands(const GemmArgs& g, const X3Plan& p, void* A3, void* B3, GemmArgs* h, hipStream_t stream,
                             bool split_a = true, bool split_b = true) {
    if (split_a) {
        if (p.a_rows) ADN_TRY(split3_rows(g.A, g.lda, g.K, p.Kp, g.M, A3, p.lda3, 0b100, stream));
        else ADN_TRY(split3_cols(g.A, g.lda, g.M, g.K, p.Kp, A3, 0b100, stream));
    }
    if (split_b) {
        if (p.b_transpose) ADN_TRY(split3_transpose(g.B, g.ldb, g.N, g.K, p.Kp, B3, p.ldb3, 0b010, stream));
        else ADN_TRY(split3_rows(g.B, g.ldb, g.K, p.Kp, g.N, B3, p.ldb3, 0b010, stream));
    }
    *h = g;
    h->precision = ADN_PRECISION_BF16;
    if (p.b_transpose) h->layout = GEMM_NN;
    h->A16 = A3; h->B16 = B3; h->lda = p.lda3; h->ldb = p.ldb3; h->K = 3 * p.Kp;
    h->A16lo = nullptr; h->B16lo = nullptr;
    // fp32 outputs; a bf16 copy of C / a bf16 mask only where the caller asked for them (rectifier masks: model.hip)
    h->C16 = g.C16; h->Y16 = (g.act_grad == ADN_ACT_RECTIFY) ? g.Y16 : nullptr;
    return ADN_OK;
}

static bool x3_takes(const GemmArgs& g) {
    // small problems (launch-bound; the split passes would cost more than they save) stay on the fp32 MFMA kernels
    // (ADN_X3_MIN_WORK: test hook -- 0 sends every shape through the split path)
    static const double min_work = getenv("ADN_X3_MIN_WORK") ? atof(getenv("ADN_X3_MIN_WORK")) : 6.4e7;
    return (double)g.M * g.N * g.K >= min_work && (g.K >= 32 || min_work == 0) && g.A && g.B && g.C && g.lda % 4 == 0 &&
           g.ldb % 4 == 0 && ((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0;
}

// the product over the operands' hi / lo planes (model.hip keeps them for every GEMM operand in this mode) where the ping-pong
// kernel takes the shape; *used says whether.  No split passes, no workspace.
static int x3_try_planes(const GemmArgs* gs, int n, hipStream_t stream, bool* used, bool dry) {
    *used = false;
    static const bool off = getenv("ADN_X3_NO_PLANES") != nullptr;
    if (off) return ADN_OK;
    GemmArgs h[kMaxGemmGroups];
    for (int k = 0; k < n; ++k) {
        h[k] = gs[k];
        if (h[k].layout == GEMM_NT && h[k].BT16 && h[k].BT16lo) {        // dZ W^T over the planes of W^T
            h[k].layout = GEMM_NN; h[k].B16 = h[k].BT16; h[k].B16lo = h[k].BT16lo; h[k].ldb = h[k].ldbT;
            h[k].b_pad_zero = 1;                  // (transposed weight planes: written for the valid columns only, zero elsewhere)
        }
        if (!h[k].A16 || !h[k].A16lo || !h[k].B16 || !h[k].B16lo) return ADN_OK;
        h[k].precision = ADN_PRECISION_BF16;
        h[k].Y16 = (gs[k].act_grad == ADN_ACT_RECTIFY) ? gs[k].Y16 : nullptr;
    }
    return gemm_pp_try(h, n, stream, used, dry);
}

// the product over split images of the fp32 operands, made for this launch in the stream's workspace
static int gemm_bf16x3_images(const GemmArgs& g, hipStream_t stream) {
    const X3Plan p = x3_plan(g);
    void* ws = nullptr;
    ADN_TRY(x3_workspace(stream, p.a_bytes + p.b_bytes, &ws));
    GemmArgs h;
    ADN_TRY(x3_split_operands(g, p, ws, static_cast<char*>(ws) + p.a_bytes, &h, stream));
    return gemm(h, stream);
}

// n problems of one shape: every problem's images side by side in the stream's workspace, then ONE grouped bf16 launch
// (the ping-pong kernel fills its rounds with the tiles of all of them) -- or, if that kernel declines, one launch each
static int gemm_grouped_bf16x3(const GemmArgs* gs, int n, hipStream_t stream) {
    {
        bool used = false;
        ADN_TRY(x3_try_planes(gs, n, stream, &used, false));
        if (used) return ADN_OK;
        // (a shape the ping-pong kernel takes alone but not as a group -- the 2000 x 1000 weight gradients: one by one over the planes)
        ADN_TRY(x3_try_planes(gs, 1, stream, &used, /*dry=*/true));
        if (used) {
            for (int k = 0; k < n; ++k) {
                ADN_TRY(x3_try_planes(gs + k, 1, stream, &used, false));
                if (!used) ADN_TRY(gemm_bf16x3_images(gs[k], stream));
            }
            return ADN_OK;
        }
    }
    const X3Plan p = x3_plan(gs[0]);
    void* ws = nullptr;
    ADN_TRY(x3_workspace(stream, (size_t)n * (p.a_bytes + p.b_bytes), &ws));
    GemmArgs h[kMaxGemmGroups];
    for (int k = 0; k < n; ++k) {
        char* base = static_cast<char*>(ws) + (size_t)k * (p.a_bytes + p.b_bytes);
        void* A3 = base; void* B3 = base + p.a_bytes;
        bool split_a = true, split_b = true;
        for (int j = 0; j < k; ++j) {          // an operand the group shares (the dG of one LSTM under its weight gradients) is split once
            if (split_a && gs[j].A == gs[k].A && gs[j].lda == gs[k].lda) { A3 = const_cast<void*>(h[j].A16); split_a = false; }
            if (split_b && gs[j].B == gs[k].B && gs[j].ldb == gs[k].ldb) { B3 = const_cast<void*>(h[j].B16); split_b = false; }
        }
        ADN_TRY(x3_split_operands(gs[k], p, A3, B3, &h[k], stream, split_a, split_b));
    }
    bool used = false;
    ADN_TRY(gemm_pp_try(h, n, stream, &used));
    if (used) return ADN_OK;
    ADN_TRY(gemm_pp_try(h, 1, stream, &used, /*dry=*/true));
    if (!used && rs_groupable(h, n)) return gemm_rs(h, n, stream);
    for (int k = 0; k < n; ++k) ADN_TRY(gemm(h[k], stream));
    return ADN_OK;
}

// The register-staged kernels for n >= 1 problems of ONE configuration (identical shapes, strides, flags and pointer
// null-ness -- the caller checked): n > 1 goes out as ONE launch whose blockIdx.z picks the problem (bf16 kernels reading
// bf16 operand copies only).  What a small-batch step gains from it: its GEMMs are latency-bound (M = B T = 1040 rows at the
// reference's minibatch: 6-27 us per launch whatever the flops), and three streams' layers in one launch cost one latency.
static int gemm_rs(const GemmArgs* gs, int n, hipStream_t stream) {
    const GemmArgs& g = gs[0];
    const bool lean_c = !g.C && g.C16 && g.precision == ADN_PRECISION_BF16 && g.A16 && g.B16 && !g.accumulate;
    for (int k = 0; k < n; ++k) {
        const GemmArgs& q = gs[k];
        ADN_CHECK(q.A && q.B && (q.C || lean_c), ADN_ERR_INVALID, "gemm: null operand");
        ADN_CHECK(((uintptr_t)q.A % 16) == 0 && ((uintptr_t)q.B % 16) == 0, ADN_ERR_INVALID, "gemm: A and B must be 16-byte aligned");
    }
    ADN_CHECK(g.lda % 4 == 0 && g.ldb % 4 == 0, ADN_ERR_INVALID, "gemm: lda/ldb must be multiples of 4 floats");
    ADN_CHECK(g.precision == ADN_PRECISION_F32 || g.precision == ADN_PRECISION_BF16, ADN_ERR_INVALID,
              "gemm: unsupported precision");
    // a short-K NT product into a bf16-only matrix under very many rows: the A-stationary kernel (gemm_bf16.hip)
    if (n == 1 && lean_c && g.layout == GEMM_NT && !g.bias && !g.Y && !g.Y16 && !g.colsum && g.act == ADN_ACT_LINEAR &&
        gemm_nt_astat_takes(g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.A16, g.B16, g.C16)) {
        ProfScope prof(PROF_GEMM_NN + g.layout, 2.0 * g.M * g.N * g.K, 2.0 * ((double)g.M * g.K + (double)g.K * g.N + (double)g.M * g.N), stream);
        static const bool trace = getenv("ADN_GEMM_TRACE") != nullptr;
        if (trace) fprintf(stderr, "ADN_GEMM NT M=%d N=%d K=%d tile=1 tiles=%d split=1 shadows=1 lean=1 acc=0\n", g.M, g.N, g.K, cdiv(g.M, 128));
        return gemm_nt_astat(g.A16, g.lda, g.B16, g.ldb, g.C16, g.ldc, g.M, g.N, g.K, stream);
    }
    GemmParams p;
    std::memset(static_cast<void*>(&p), 0, sizeof(p));
    p.M = g.M; p.N = g.N; p.K = g.K;
    p.A = g.A; p.lda = g.lda; p.B = g.B; p.ldb = g.ldb; p.C = g.C; p.ldc = g.ldc;
    p.bias = g.bias; p.Y = g.Y; p.ldy = g.ldy;
    p.A16 = g.A16; p.B16 = g.B16; p.C16 = g.C16;
    p.colsum = nullptr; p.colsum_ld = 0;
    p.Y16 = (g.precision == ADN_PRECISION_BF16 && g.A16 && g.B16) ? g.Y16 : nullptr;
    for (int k = 0; k < n; ++k) if (gs[k].colsum_done) *gs[k].colsum_done = 0;
    p.act = g.act; p.act_grad = g.act_grad; p.accumulate = g.accumulate;

    // (tile counts of the whole launch: a group fills the device where one of its problems would not)
    const int64_t t128 = (int64_t)cdiv(g.M, 128) * cdiv(g.N, 128) * n;
    const int64_t t64 = (int64_t)cdiv(g.M, 64) * cdiv(g.N, 64) * n;
    // (the split-K epilogue adds bare partial sums: no bias, no act'(Y) mask)
    const bool can_split = g.act == ADN_ACT_LINEAR && !g.bias && !g.Y && !g.Y16;
    // 128x128 tiles (4 MFMA tiles per wave, half the LDS fragment reads per MFMA of the 64x64 shape) whenever
    // the grid can still fill 256 CUs: either by tile count alone or together with split-K (weight gradients:
    // K = all frames of the batch)
    static const int force_tile = getenv("ADN_GEMM_TILE") ? atoi(getenv("ADN_GEMM_TILE")) : 0;   // experiments only
    // ... and K is deep enough to amortise a 128x128 tile's prologue + epilogue latency (3 workgroups per CU
    // cannot hide it: at K <= 512 the 64x64 shape, 8 workgroups per CU, is 1.3-2.2x faster; equal at K = 1024)
    // ... and the 128-wide tiles do not waste much more of their area on the matrix edges than 64-wide ones would
    // (N = 152: 59 % vs 79 % useful)
    const double fill128 = (double)g.M * g.N * n / ((double)t128 * 128 * 128), fill64 = (double)g.M * g.N * n / ((double)t64 * 64 * 64);
    const bool big = force_tile ? force_tile == 128
                                : (g.K >= 768 && fill128 >= 0.85 * fill64 &&
                                   (t128 >= 384 || (can_split && t128 >= 24 && g.K >= 2048)));
    // 256 x 64 tiles (bf16 kernels): a narrow output under many rows -- the B panel is re-read once per 256 rows
    // instead of once per 64, at the per-wave tile of the 128 x 128 shape
    const int64_t t_tall = (int64_t)cdiv(g.M, 256) * cdiv(g.N, 64) * n;
    // (measured, profiles/r02/gemm_lab_tall.txt: 129024 x 152 x 2504 -- the conv auto-encoder's second convolution at batch
    //  1024 -- 266 us against 331 (64 x 64) / 293 (128 x 128); N = 104 / 150 / 200 and the 20800-row shapes: no gain)
    const bool tall = g.precision == ADN_PRECISION_BF16 &&
                      (force_tile ? force_tile == 256 : (!big && g.N > 128 && g.N <= 192 && g.M >= 65536 && g.K >= 1024));
    const int64_t tiles = tall ? t_tall : (big ? t128 : t64);    // of the whole launch
    int split = 1;
    // split-K: enough workgroups for two per CU (measured on the weight-gradient shapes: 512 beats 768 / 1024 by 0-12 %,
    // fewer partial sums to add atomically; 256 leaves CUs idle on the large ones)
    static const int split_target = getenv("ADN_GEMM_SPLIT_TARGET") ? atoi(getenv("ADN_GEMM_SPLIT_TARGET")) : 512;
    // (partial sums meet in float atomics -- or, in deterministic mode, in per-slice slabs that a fixed-order pass adds: the
    //  mode then keeps the split where the caller's slab workspace holds the launch's slices, and runs unsplit otherwise)
    if (tiles < 384 && g.K >= 512 && can_split && !lean_c && !g.no_split) {
        split = (int)((split_target + tiles - 1) / tiles);
        split = std::min(split, g.K / 128);
        split = std::max(1, std::min(split, 128));
    }
    p.k_chunk = (int)round_up(cdiv(g.K, split), BK);
    split = cdiv(g.K, p.k_chunk);
    p.partial = nullptr;
    if (split > 1 && deterministic()) {
        if (g.splitk_ws && (size_t)n * split * g.M * g.ldc <= g.splitk_ws_floats) p.partial = g.splitk_ws;
        else { split = 1; p.k_chunk = (int)round_up(g.K, BK); }
    }
    p.atomic = split > 1;
    if (p.atomic) p.C16 = nullptr;            // partial sums: the bf16 copy is made after the kernel (below)
    if (g.precision == ADN_PRECISION_BF16 && p.A16 && p.B16)
        ADN_CHECK(g.lda % 8 == 0 && g.ldb % 8 == 0, ADN_ERR_INVALID, "gemm: bf16 shadows need lda/ldb % 8 == 0");
    ProfScope prof(PROF_GEMM_NN + g.layout, 2.0 * g.M * g.N * g.K * n,
                   4.0 * n * ((double)g.M * g.K + (double)g.K * g.N + (double)g.M * g.N), stream);
    if (p.atomic && !g.accumulate && !p.partial)
        for (int k = 0; k < n; ++k)
            ADN_HIP_CHECK(hipMemset2DAsync(gs[k].C, (size_t)g.ldc * 4, 0, (size_t)g.N * 4, g.M, stream));
    if (lean_c)
        for (int k = 0; k < n; ++k)
            ADN_CHECK(g.ldc % 4 == 0 && g.N % 4 == 0 && ((uintptr_t)gs[k].C16 % 8) == 0, ADN_ERR_INVALID,
                      "gemm: bf16-only output needs N and ldc to be multiples of 4");
    const int tsz = big ? 128 : 64;
    p.tiles_m = cdiv(g.M, tall ? 256 : tsz); p.tiles_n = cdiv(g.N, tsz);
    const int cs_ld = (int)round_up(g.N, 4);
    bool fused_colsum = g.colsum && g.precision == ADN_PRECISION_BF16 && !p.atomic && g.ldc % 4 == 0 &&
                        (!g.Y || g.ldy % 4 == 0) && g.N % 4 == 0;
    for (int k = 0; k < n && fused_colsum; ++k) {
        const GemmArgs& q = gs[k];
        fused_colsum = ((uintptr_t)q.C % 16) == 0 && (!q.Y || ((uintptr_t)q.Y % 16) == 0) && q.colsum_ws &&
                       ((uintptr_t)q.colsum_ws % 16) == 0 && (size_t)p.tiles_m * cs_ld <= q.colsum_ws_floats;
    }
    if (fused_colsum) {
        p.colsum = g.colsum_ws; p.colsum_ld = cs_ld;   // the vectorised epilogue path is guaranteed for every element
        for (int k = 0; k < n; ++k) if (gs[k].colsum_done) *gs[k].colsum_done = 1;
    }
    const int64_t tiles_one = tiles / n;      // blockIdx.x walks ONE problem's tiles, blockIdx.z the problems
    {   // square-ish per-XCD tile blocks: panel width ~ sqrt(tiles per XCD)
        const int chunk = std::max<int64_t>(1, tiles_one / 8);
        int bn = (int)std::lround(std::sqrt((double)chunk));
        p.panel_n = std::max(1, std::min(bn, p.tiles_n));
    }
    if (n > 1) {
        p.ngroups = n;
        for (int k = 0; k < n; ++k) {
            GemmGroup& q = p.grp[k];
            q.A16 = gs[k].A16; q.B16 = gs[k].B16; q.C = gs[k].C; q.C16 = p.atomic ? nullptr : gs[k].C16;
            q.Y16 = p.Y16 ? gs[k].Y16 : nullptr; q.Y = gs[k].Y; q.bias = gs[k].bias;
            q.colsum = fused_colsum ? gs[k].colsum_ws : nullptr;
        }
    }
    const dim3 grid((unsigned)tiles_one, split, n);
    static const bool trace = getenv("ADN_GEMM_TRACE") != nullptr;       // one line per launch, pairs with a kernel trace
    if (trace)
        fprintf(stderr, "ADN_GEMM %s M=%d N=%d K=%d tile=%d tiles=%lld split=%d shadows=%d lean=%d acc=%d groups=%d\n",
                g.layout == GEMM_NN ? "NN" : (g.layout == GEMM_NT ? "NT" : "TN"), g.M, g.N, g.K, tall ? 25664 : tsz,
                (long long)tiles, split, (int)(p.A16 && p.B16), (int)lean_c, g.accumulate, n);      // (groups: problems of this launch, blockIdx.z)
    if (g.precision == ADN_PRECISION_BF16) launch_gemm_bf16(p, g.layout, tall ? 2 : (big ? 1 : 0), grid, stream);
    else if (big) launch<128, 128>(p, g.layout, grid, stream);
    else launch<64, 64>(p, g.layout, grid, stream);
    ADN_HIP_CHECK(hipGetLastError());
    if (p.partial) {                              // deterministic split-K: C (+)= slab 0 + slab 1 + ... in that order
        RsReduceArgs ra;
        for (int k = 0; k < n; ++k) ra.C[k] = gs[k].C;
        const int64_t work = (int64_t)g.M * g.N;
        hipLaunchKernelGGL(rs_splitk_reduce_kernel, dim3((unsigned)std::min<int64_t>(2048, (work + 255) / 256), 1, (unsigned)n), dim3(256), 0, stream,
                           ra, p.partial, g.M, g.N, g.ldc, split, g.accumulate);
        ADN_HIP_CHECK(hipGetLastError());
    }
    for (int k = 0; k < n; ++k) {
        const GemmArgs& q = gs[k];
        if (fused_colsum) {
            if (q.colsum_batch && q.colsum_batch->n < 8) col_sum_batch_add(*q.colsum_batch, q.colsum_ws, p.colsum_ld, p.tiles_m, g.N, q.colsum);
            else ADN_TRY(col_sum(q.colsum_ws, p.colsum_ld, p.tiles_m, g.N, q.colsum, 1, stream));
        }
        if (q.C16 && p.atomic) {                  // split-K result: refresh the bf16 shadow of whole rows
            ADN_CHECK(g.ldc % 8 == 0, ADN_ERR_INVALID, "gemm: bf16 shadow of C needs ldc % 8 == 0");
            ADN_TRY(to_bf16(q.C, q.C16, (size_t)g.M * g.ldc, stream));
        }
    }
    return ADN_OK;
}

// may these problems share one launch of the register-staged bf16 kernels?
static bool rs_groupable(const GemmArgs* gs, int n) {
    const GemmArgs& g = gs[0];
    if (n < 2 || n > kMaxGemmGroups || g.precision != ADN_PRECISION_BF16 || getenv("ADN_GEMM_NO_RS_GROUPS")) return false;
    for (int k = 0; k < n; ++k) {
        const GemmArgs& q = gs[k];
        if (!q.A16 || !q.B16 || q.precision != g.precision || q.layout != g.layout || q.M != g.M || q.N != g.N || q.K != g.K ||
            q.lda != g.lda || q.ldb != g.ldb || q.ldc != g.ldc || q.ldy != g.ldy || q.act != g.act || q.act_grad != g.act_grad ||
            q.accumulate != g.accumulate || q.no_split != g.no_split || (q.C == nullptr) != (g.C == nullptr) ||
            (q.C16 == nullptr) != (g.C16 == nullptr) || (q.bias == nullptr) != (g.bias == nullptr) ||
            (q.Y == nullptr) != (g.Y == nullptr) || (q.Y16 == nullptr) != (g.Y16 == nullptr) ||
            (q.colsum == nullptr) != (g.colsum == nullptr))
            return false;
    }
    return true;
}

int gemm(const GemmArgs& g, hipStream_t stream) {
    ADN_CHECK(g.layout >= GEMM_NN && g.layout <= GEMM_TN, ADN_ERR_INVALID, "gemm: bad layout");
    if (g.M <= 0 || g.N <= 0) return ADN_OK;
    ADN_CHECK(g.K > 0, ADN_ERR_INVALID, "gemm: K must be positive");
    {
        bool used = false;
        ADN_TRY(gemm_skinny_try(&g, 1, stream, &used, false));
        if (used) return ADN_OK;
    }
    if (g.precision == ADN_PRECISION_BF16X3) {
        {                                      // over the operands' planes wherever the ping-pong kernel takes the shape ...
            bool used = false;
            ADN_TRY(x3_try_planes(&g, 1, stream, &used, false));
            if (used) return ADN_OK;
        }
        if (x3_takes(g)) return gemm_bf16x3_images(g, stream);      // ... over split images of the fp32 operands ... (else fp32 MFMA)
        GemmArgs h = g;
        h.precision = ADN_PRECISION_F32;
        h.C16 = nullptr; h.Y16 = nullptr;
        ADN_TRY(gemm(h, stream));
        if (g.C16 && g.C)                      // (the fp32 kernels do not write bf16 copies: a requested one is made here)
            ADN_TRY(to_bf16(g.C, g.C16, (size_t)round_up((int64_t)g.M * g.ldc, 8), stream));
        return ADN_OK;
    }
    ADN_CHECK(g.precision == ADN_PRECISION_F32 || g.precision == ADN_PRECISION_BF16, ADN_ERR_INVALID,
              "gemm: unsupported precision");
    {
        bool used = false;
        ADN_TRY(gemm_pp_try(&g, 1, stream, &used));
        if (used) return ADN_OK;
    }
    return gemm_rs(&g, 1, stream);
}

}  // namespace adn
