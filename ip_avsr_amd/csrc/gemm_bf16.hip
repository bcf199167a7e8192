// bf16 MFMA GEMM with fp32 operands in HBM: v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//
// Same interface, grid and epilogue semantics as gemm_f32.hip; operands stay fp32 in HBM (master weights,
// activations and gradients are all fp32) and are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) on their way into
// LDS, so this kernel is a drop-in for every GEMM of the path.
//
// Each operand is staged in LDS in the orientation it has in HBM -- global loads are always float4 along the
// contiguous dimension, no gathers, no transposing stores:
//   * k-contiguous operand ([R][K] in HBM)  -> LDS image [R][64 k] (+8 pad); a 16x16x32 fragment (lane l:
//     row l&15, k = 8*(l>>4) .. +7) is one ds_read_b128;
//   * k-strided operand ([K][R] in HBM)     -> LDS image [64 k][R] (+16 pad); the same fragment is two
//     ds_read_b64_tr_b16 (the CDNA4 transposing LDS read: 4 k-rows x 16 columns per 16-lane group, delivered
//     column-major).  This is what makes X^T dY (weight gradients) and X W (forward, W stored [in][out])
//     run without ever materialising a transposed copy.
// The main loop is unguarded in k (rows/columns outside the matrix only ever feed accumulator rows/columns
// that the epilogue drops, so their loads are merely clamped in-bounds); only the last partial k-stage pays
// for zero-filling.  Register prefetch of stage s+1 overlaps the MFMAs of stage s.
#include "gemm_common.h"
#include <algorithm>
#include <type_traits>

namespace adn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef ADN_BKH
#define ADN_BKH 64
#endif
constexpr int BKH = ADN_BKH;    // k per LDS stage

__device__ __forceinline__ bf16x4 cvt4(const float4 v) {
    bf16x4 r;
    r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
    return r;
}

__device__ __forceinline__ bf16x8 join(const bf16x4 lo, const bf16x4 hi) {
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ---- k-contiguous operand: global [R][K] fp32 -------------------------------------------------------
template <int R, typename T>
struct StageKC;

template <int R>
struct StageKC<R, float> {
    static constexpr int kStride = BKH + 16;              // bf16 per LDS row (160 B = 40 banks: conflict-free b128 fragment reads)
    static constexpr int kLds = R * kStride;
    static constexpr int kCPR = BKH / 8, kRPP = 256 / kCPR; // 8-k chunks per row, rows covered per pass
    static constexpr int kIter = R / kRPP;                // row groups per thread; 2 float4 (8 k) each
    const float* ptr[kIter];
    float4 v[kIter][2];
    int kc8;                                              // this thread's k offset inside a stage (0,8,..,56)

    __device__ __forceinline__ void init(const float* g, int ld, int r0, int rmax, int kbeg, int tid) {
        kc8 = (tid % kCPR) * 8;
#pragma unroll
        for (int i = 0; i < kIter; ++i) {
            const int row = min(r0 + (tid / kCPR) + kRPP * i, rmax - 1);      // clamped: feeds dropped outputs only
            ptr[i] = g + (size_t)row * ld + kbeg + kc8;
        }
    }
    __device__ __forceinline__ void load_full() {
#pragma unroll
        for (int i = 0; i < kIter; ++i) {
            v[i][0] = *reinterpret_cast<const float4*>(ptr[i]);
            v[i][1] = *reinterpret_cast<const float4*>(ptr[i] + 4);
            ptr[i] += BKH;
        }
    }
    __device__ __forceinline__ void load_tail(int k0, int kend) {           // k0 = first k of this stage
        const int k = k0 + kc8;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < kIter; ++i) {
            float4 a = z, b = z;
            if (k < kend) a = *reinterpret_cast<const float4*>(ptr[i]);
            if (k + 4 < kend) b = *reinterpret_cast<const float4*>(ptr[i] + 4);
            if (k + 1 >= kend) a.y = 0.f;
            if (k + 2 >= kend) a.z = 0.f;
            if (k + 3 >= kend) a.w = 0.f;
            if (k + 5 >= kend) b.y = 0.f;
            if (k + 6 >= kend) b.z = 0.f;
            if (k + 7 >= kend) b.w = 0.f;
            v[i][0] = a; v[i][1] = b;
        }
    }
    __device__ __forceinline__ void store(__bf16* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < kIter; ++i) {
            *reinterpret_cast<bf16x8*>(lds + ((tid / kCPR) + kRPP * i) * kStride + kc8) =
                join(cvt4(v[i][0]), cvt4(v[i][1]));
        }
    }
    // fragment of the 16 rows starting at `row0` for k-step s (32 k each)
    __device__ __forceinline__ static bf16x8 frag(const __bf16* lds, int row0, int s, int lane) {
        return *reinterpret_cast<const bf16x8*>(lds + (row0 + (lane & 15)) * kStride + s * 32 + (lane >> 4) * 8);
    }
};

// ---- k-contiguous operand already in bf16 (shadow copy): one 16-byte load = 8 k ---------------------
template <int R>
struct StageKC<R, __bf16> {
    static constexpr int kStride = BKH + 16;
    static constexpr int kLds = R * kStride;
    static constexpr int kCPR = BKH / 8, kRPP = 256 / kCPR;
    static constexpr int kIter = R / kRPP;
    const __bf16* ptr[kIter];
    bf16x8 v[kIter];
    int kc8;

    __device__ __forceinline__ void init(const __bf16* g, int ld, int r0, int rmax, int kbeg, int tid) {
        kc8 = (tid % kCPR) * 8;
#pragma unroll
        for (int i = 0; i < kIter; ++i) {
            const int row = min(r0 + (tid / kCPR) + kRPP * i, rmax - 1);
            ptr[i] = g + (size_t)row * ld + kbeg + kc8;
        }
    }
    __device__ __forceinline__ void load_full() {
#pragma unroll
        for (int i = 0; i < kIter; ++i) {
            v[i] = *reinterpret_cast<const bf16x8*>(ptr[i]);
            ptr[i] += BKH;
        }
    }
    __device__ __forceinline__ void load_tail(int k0, int kend) {
        const int k = k0 + kc8;
#pragma unroll
        for (int i = 0; i < kIter; ++i) {
            bf16x8 w;
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = (__bf16)0.f;
            if (k < kend) {
                w = *reinterpret_cast<const bf16x8*>(ptr[i]);
#pragma unroll
                for (int j = 1; j < 8; ++j)
                    if (k + j >= kend) w[j] = (__bf16)0.f;
            }
            v[i] = w;
        }
    }
    __device__ __forceinline__ void store(__bf16* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < kIter; ++i)
            *reinterpret_cast<bf16x8*>(lds + ((tid / kCPR) + kRPP * i) * kStride + kc8) = v[i];
    }
    __device__ __forceinline__ static bf16x8 frag(const __bf16* lds, int row0, int s, int lane) {
        return *reinterpret_cast<const bf16x8*>(lds + (row0 + (lane & 15)) * kStride + s * 32 + (lane >> 4) * 8);
    }
};

// transposing fragment read shared by both k-strided stagers: 16 columns from `col0`, k-step s
template <int STRIDE>
__device__ __forceinline__ bf16x8 frag_tr(const __bf16* lds, int col0, int s, int lane) {
    // Lane 4q+p of a 16-lane group supplies the address of k-row q, columns 4p..4p+3 and receives column
    // (lane&15) of the 4 rows (cdna_hip_programming.md T10); two reads cover the 8 k of this lane group.
    const int q = (lane & 15) >> 2, pcol = (lane & 3) * 4;
    const __bf16* a = lds + (s * 32 + (lane >> 4) * 8 + q) * STRIDE + col0 + pcol;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * STRIDE));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 w = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, w);
}

// ---- k-strided operand: global [K][R] fp32 ----------------------------------------------------------
template <int R, typename T>
struct StageKS;

template <int R>
struct StageKS<R, float> {
    static constexpr int kStride = R + 16;                // bf16 per LDS k-row
    static constexpr int kLds = BKH * kStride;
    static constexpr int kVecRow = R / 4;                 // float4 per k-row
    static constexpr int kRowsPerPass = 256 / kVecRow;    // k-rows covered by the 256 threads at once
    static constexpr int kIter = BKH / kRowsPerPass;
    const float* ptr;
    size_t step;                                          // floats between this thread's consecutive k-rows
    float4 v[kIter];
    int c4, krow;

    __device__ __forceinline__ void init(const float* g, int ld, int c0, int cmax, int kbeg, int tid) {
        c4 = (tid % kVecRow) * 4;
        krow = tid / kVecRow;
        const int col = (c0 + c4 < cmax) ? c0 + c4 : 0;   // clamped: feeds dropped outputs only
        ptr = g + (size_t)(kbeg + krow) * ld + col;
        step = (size_t)kRowsPerPass * ld;
    }
    __device__ __forceinline__ void load_full() {
#pragma unroll
        for (int i = 0; i < kIter; ++i) {
            v[i] = *reinterpret_cast<const float4*>(ptr);
            ptr += step;
        }
    }
    __device__ __forceinline__ void load_tail(int k0, int kend) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < kIter; ++i) {
            const int k = k0 + krow + i * kRowsPerPass;
            v[i] = (k < kend) ? *reinterpret_cast<const float4*>(ptr + i * step) : z;
        }
    }
    __device__ __forceinline__ void store(__bf16* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < kIter; ++i)
            *reinterpret_cast<bf16x4*>(lds + (krow + i * kRowsPerPass) * kStride + c4) = cvt4(v[i]);
    }
    __device__ __forceinline__ static bf16x8 frag(const __bf16* lds, int col0, int s, int lane) {
        return frag_tr<kStride>(lds, col0, s, lane);
    }
};

// ---- k-strided operand already in bf16: one 16-byte load = 8 columns of one k-row -------------------
template <int R>
struct StageKS<R, __bf16> {
    static constexpr int kStride = R + 16;
    static constexpr int kLds = BKH * kStride;
    static constexpr int kVecRow = R / 8;                 // 16-byte chunks per k-row
    static constexpr int kRowsPerPass = 256 / kVecRow;
    static constexpr int kIter = BKH / kRowsPerPass;
    const __bf16* ptr;
    size_t step;
    bf16x8 v[kIter];
    int c8, krow;

    __device__ __forceinline__ void init(const __bf16* g, int ld, int c0, int cmax, int kbeg, int tid) {
        c8 = (tid % kVecRow) * 8;
        krow = tid / kVecRow;
        const int col = (c0 + c8 < cmax) ? c0 + c8 : 0;
        ptr = g + (size_t)(kbeg + krow) * ld + col;
        step = (size_t)kRowsPerPass * ld;
    }
    __device__ __forceinline__ void load_full() {
#pragma unroll
        for (int i = 0; i < kIter; ++i) {
            v[i] = *reinterpret_cast<const bf16x8*>(ptr);
            ptr += step;
        }
    }
    __device__ __forceinline__ void load_tail(int k0, int kend) {
#pragma unroll
        for (int i = 0; i < kIter; ++i) {
            const int k = k0 + krow + i * kRowsPerPass;
            bf16x8 w;
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = (__bf16)0.f;
            if (k < kend) w = *reinterpret_cast<const bf16x8*>(ptr + i * step);
            v[i] = w;
        }
    }
    __device__ __forceinline__ void store(__bf16* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < kIter; ++i)
            *reinterpret_cast<bf16x8*>(lds + (krow + i * kRowsPerPass) * kStride + c8) = v[i];
    }
    __device__ __forceinline__ static bf16x8 frag(const __bf16* lds, int col0, int s, int lane) {
        return frag_tr<kStride>(lds, col0, s, lane);
    }
};

template <int R, bool KC, typename T> struct StageSel { typedef StageKC<R, T> type; };
template <int R, typename T> struct StageSel<R, false, T> { typedef StageKS<R, T> type; };

// epilogue for one 16x16 accumulator tile: col = lane&15, row = 4*(lane>>4) + reg
__device__ __forceinline__ void store_tile16(const GemmParams& p, const f32x4& acc, int row0, int col0, int lane,
                                             bool first_split) {
    const int col = col0 + (lane & 15);
    if (col >= p.N) return;
    const float bias = (p.bias && first_split) ? p.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * (lane >> 4) + r;
        if (row >= p.M) continue;
        float v = acc[r] + bias;
        float* c = p.C + (size_t)row * p.ldc + col;
        if (p.atomic) {
            if (p.Y) v *= act_grad_from_output(p.act_grad, p.Y[(size_t)row * p.ldy + col]);
            atomicAdd(c, v);
        } else {
            v = act_apply(p.act, v);
            if (p.Y) v *= act_grad_from_output(p.act_grad, p.Y[(size_t)row * p.ldy + col]);
            if (p.accumulate) v += *c;
            *c = v;
            if (p.C16) reinterpret_cast<__bf16*>(p.C16)[(size_t)row * p.ldc + col] = (__bf16)v;
        }
    }
}

template <int BM, int BN, bool A_KC, bool B_KC, typename T>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const GemmParams p) {
    typedef typename StageSel<BM, A_KC, T>::type SA;
    typedef typename StageSel<BN, B_KC, T>::type SB;
    constexpr int WTM = BM / 2, WTN = BN / 2;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int kEpiElems = 4 * 32 * (WTN + 4) * 2;        // epilogue bounce buffer, in bf16 units
    constexpr int kSmemElems = (SA::kLds + SB::kLds) > kEpiElems ? (SA::kLds + SB::kLds) : kEpiElems;
    __shared__ __attribute__((aligned(16))) __bf16 smem[kSmemElems];
    __bf16* As = smem;
    __bf16* Bs = smem + SA::kLds;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n;
    tile_coords(p, xcd_tile(blockIdx.x, gridDim.x), tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = blockIdx.y * p.k_chunk;
    const int kend = min(p.K, kbeg + p.k_chunk);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    SA sa; SB sb;
    sa.init(reinterpret_cast<const T*>(sizeof(T) == 2 ? p.A16 : (const void*)p.A), p.lda, m0, p.M, kbeg, tid);
    sb.init(reinterpret_cast<const T*>(sizeof(T) == 2 ? p.B16 : (const void*)p.B), p.ldb, n0, p.N, kbeg, tid);
    if (kbeg + BKH <= kend) { sa.load_full(); sb.load_full(); }
    else { sa.load_tail(kbeg, kend); sb.load_tail(kbeg, kend); }

    for (int k0 = kbeg; k0 < kend; k0 += BKH) {
        sa.store(As, tid);
        sb.store(Bs, tid);
        __syncthreads();
        const int kn = k0 + BKH;
        if (kn + BKH <= kend) { sa.load_full(); sb.load_full(); }
        else if (kn < kend) { sa.load_tail(kn, kend); sb.load_tail(kn, kend); }
#pragma unroll
        for (int s = 0; s < BKH / 32; ++s) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = SA::frag(As, wm * WTM + a * 16, s, lane);
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[b] = SB::frag(Bs, wn * WTN + b * 16, s, lane);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }

    const bool first_split = blockIdx.y == 0;
    if (p.atomic) {                                   // split-K: fp32 atomics straight from the accumulators
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
                store_tile16(p, acc[a][b], m0 + wm * WTM + a * 16, n0 + wn * WTN + b * 16, lane, first_split);
        return;
    }
    // Coalesced epilogue: the accumulators of a wave (column on the lane, 4 rows per register quad) are bounced
    // through LDS, 32 rows at a time, so that every lane then owns 4 CONSECUTIVE columns of one row: bias /
    // activation / act'(Y) / accumulate run on float4, C is written with 16-byte stores (256 contiguous bytes per
    // row per wave instruction) and the bf16 shadow with 8-byte stores.  (Per-lane scalar stores of the raw MFMA
    // layout touch 4 rows x 64 B per instruction and made the store tail as long as the main loop for N ~ K.)
    constexpr int LW = WTN + 4;                       // floats per LDS row (16-byte aligned, bank-staggered)
    constexpr int LPR = WTN / 4;                      // lanes per row in the read phase
    constexpr int RPI = 64 / LPR;                     // rows per read instruction
    float* wl = reinterpret_cast<float*>(smem) + wave * 32 * LW;
    const int i16 = lane & 15, kq4 = (lane >> 4) * 4;
    const bool vec_ok = (p.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                        (!p.Y || (p.ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(p.Y) & 15) == 0));
    const __bf16* y16 = reinterpret_cast<const __bf16*>(p.Y16);
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);    // column sums of this lane's 4 columns (p.colsum)
    // (compile-time pass index: a run-time indexed accumulator array would be placed in scratch)
    auto do_pass = [&](auto pass_c) {
        constexpr int pass = decltype(pass_c)::value;
        __syncthreads();
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) wl[(a2 * 16 + kq4 + r) * LW + b * 16 + i16] = acc[pass * 2 + a2][b][r];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32 / RPI; ++j) {
            const int lr = j * RPI + lane / LPR, lc = (lane % LPR) * 4;
            const int row = m0 + wm * WTM + pass * 32 + lr, col = n0 + wn * WTN + lc;
            if (row >= p.M || col >= p.N) continue;
            float4 v = *reinterpret_cast<const float4*>(wl + lr * LW + lc);
            float* vv = reinterpret_cast<float*>(&v);
            const size_t off = (size_t)row * p.ldc + col;
            if (vec_ok && col + 3 < p.N) {
                if (p.bias) { v.x += p.bias[col]; v.y += p.bias[col + 1]; v.z += p.bias[col + 2]; v.w += p.bias[col + 3]; }
#pragma unroll
                for (int e = 0; e < 4; ++e) vv[e] = act_apply(p.act, vv[e]);
                if (y16) {
                    const bf16x4 y = *reinterpret_cast<const bf16x4*>(y16 + (size_t)row * p.ldy + col);
                    v.x *= act_grad_from_output(p.act_grad, (float)y[0]); v.y *= act_grad_from_output(p.act_grad, (float)y[1]);
                    v.z *= act_grad_from_output(p.act_grad, (float)y[2]); v.w *= act_grad_from_output(p.act_grad, (float)y[3]);
                } else if (p.Y) {
                    const float4 y = *reinterpret_cast<const float4*>(p.Y + (size_t)row * p.ldy + col);
                    v.x *= act_grad_from_output(p.act_grad, y.x); v.y *= act_grad_from_output(p.act_grad, y.y);
                    v.z *= act_grad_from_output(p.act_grad, y.z); v.w *= act_grad_from_output(p.act_grad, y.w);
                }
                if (p.accumulate) {
                    const float4 c = *reinterpret_cast<const float4*>(p.C + off);
                    v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w;
                }
                if (p.C) *reinterpret_cast<float4*>(p.C + off) = v;
                if (p.C16) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.C16) + off) = cvt4(v);
                csum.x += v.x; csum.y += v.y; csum.z += v.z; csum.w += v.w;
            } else {
                for (int e = 0; e < 4 && col + e < p.N; ++e) {
                    float x = vv[e] + (p.bias ? p.bias[col + e] : 0.f);
                    x = act_apply(p.act, x);
                    if (p.Y) x *= act_grad_from_output(p.act_grad, p.Y[(size_t)row * p.ldy + col + e]);
                    if (p.accumulate) x += p.C[off + e];
                    p.C[off + e] = x;
                    if (p.C16) reinterpret_cast<__bf16*>(p.C16)[off + e] = (__bf16)x;
                }
            }
        }
    };
    do_pass(std::integral_constant<int, 0>{});
    if constexpr (TM / 2 > 1) do_pass(std::integral_constant<int, 1>{});
    if constexpr (TM / 2 > 2) do_pass(std::integral_constant<int, 2>{});
    if constexpr (TM / 2 > 3) do_pass(std::integral_constant<int, 3>{});
    static_assert(TM / 2 <= 4, "epilogue passes");
    if (p.colsum) {                                   // lanes that differ only in their row share the columns
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
            csum.x += __shfl_xor(csum.x, o, 64); csum.y += __shfl_xor(csum.y, o, 64);
            csum.z += __shfl_xor(csum.z, o, 64); csum.w += __shfl_xor(csum.w, o, 64);
        }
        // the two waves stacked in m combine through LDS; one plain float4 store per 4 columns and tile
        float4* cs = reinterpret_cast<float4*>(smem);
        __syncthreads();
        if (wm == 1 && lane < LPR) cs[wn * LPR + lane] = csum;
        __syncthreads();
        const int col = n0 + wn * WTN + lane * 4;
        if (wm == 0 && lane < LPR && col + 3 < p.N) {
            const float4 o = cs[wn * LPR + lane];
            csum.x += o.x; csum.y += o.y; csum.z += o.z; csum.w += o.w;
            *reinterpret_cast<float4*>(p.colsum + (size_t)tile_m * p.colsum_ld + col) = csum;
        }
    }
}

template <int BM, int BN, typename T>
static void launch_bf16_t(const GemmParams& p, int layout, dim3 grid, hipStream_t s) {
    switch (layout) {
        case GEMM_NN: hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, true, false, T>), grid, dim3(256), 0, s, p); break;
        case GEMM_NT: hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, true, true, T>), grid, dim3(256), 0, s, p); break;
        default:      hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, false, false, T>), grid, dim3(256), 0, s, p); break;
    }
}

void launch_gemm_bf16(const GemmParams& p, int layout, int tile_mode, dim3 grid, hipStream_t s) {
    const bool shadows = p.A16 && p.B16;      // operands already available as bf16 copies
    const bool big = tile_mode >= 1;
    if (shadows) {
        if (tile_mode == 2) launch_bf16_t<256, 128, __bf16>(p, layout, grid, s);
        else if (big) launch_bf16_t<128, 128, __bf16>(p, layout, grid, s);
        else launch_bf16_t<64, 64, __bf16>(p, layout, grid, s);
    } else {
        if (big) launch_bf16_t<128, 128, float>(p, layout, grid, s);
        else launch_bf16_t<64, 64, float>(p, layout, grid, s);
    }
}

// fp32 -> bf16 shadow copy (RNE), 8 elements per lane
__global__ __launch_bounds__(256) void to_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
        reinterpret_cast<bf16x8*>(dst)[i] = join(cvt4(a), cvt4(b));
    }
}

// bf16 transpose through a padded 32x32 LDS tile: coalesced fp32 reads along c, coalesced bf16 writes along r
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const float* __restrict__ W, int rows, int cols, int ld,
                                                             __bf16* __restrict__ out, int ldT) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < rows && c < cols) ? W[(size_t)r * ld + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < cols && r < rows) out[(size_t)c * ldT + r] = (__bf16)tile[tx][j];
    }
}

int transpose_to_bf16(const float* W, int rows, int cols, int ld, void* out, int ldT, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return ADN_OK;
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, s, W, rows, cols, ld,
                       reinterpret_cast<__bf16*>(out), ldT);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int to_bf16(const float* src, void* dst, size_t n, hipStream_t s) {
    ADN_CHECK(n % 8 == 0, ADN_ERR_INVALID, "to_bf16: element count must be a multiple of 8");
    if (!n) return ADN_OK;
    const size_t n8 = n / 8;
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((n8 + 255) / 256, 4096));
    hipLaunchKernelGGL(to_bf16_kernel, dim3(grid), dim3(256), 0, s, src, reinterpret_cast<__bf16*>(dst), n8);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

}  // namespace adn
