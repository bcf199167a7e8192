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
#include "pp_dma.h"
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

// x = hi + lo with hi = bf16(x), lo = bf16(x - hi)  (bf16x3 mode)
__device__ __forceinline__ void split4(const float4& v, bf16x4& hi, bf16x4& lo) {
    hi = cvt4(v);
    const float4 r = make_float4(v.x - (float)hi[0], v.y - (float)hi[1], v.z - (float)hi[2], v.w - (float)hi[3]);
    lo = cvt4(r);
}

// ---- k-contiguous operand: global [R][K] fp32 -------------------------------------------------------
template <int R, typename T, int NT>
struct StageKC;

template <int R, int NT>
struct StageKC<R, float, NT> {
    static constexpr int kStride = BKH + 16;              // bf16 per LDS row (160 B = 40 banks: conflict-free b128 fragment reads)
    static constexpr int kLds = R * kStride;
    static constexpr int kCPR = BKH / 8, kRPP = NT / kCPR; // 8-k chunks per row, rows covered per pass
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
template <int R, int NT>
struct StageKC<R, __bf16, NT> {
    static constexpr int kStride = BKH + 16;
    static constexpr int kLds = R * kStride;
    static constexpr int kCPR = BKH / 8, kRPP = NT / kCPR;
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
template <int R, typename T, int NT>
struct StageKS;

template <int R, int NT>
struct StageKS<R, float, NT> {
    static constexpr int kStride = R + 16;                // bf16 per LDS k-row
    static constexpr int kLds = BKH * kStride;
    static constexpr int kVecRow = R / 4;                 // float4 per k-row
    static constexpr int kRowsPerPass = NT / kVecRow;     // k-rows covered by the NT threads at once
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
template <int R, int NT>
struct StageKS<R, __bf16, NT> {
    static constexpr int kStride = R + 16;
    static constexpr int kLds = BKH * kStride;
    static constexpr int kVecRow = R / 8;                 // 16-byte chunks per k-row
    static constexpr int kRowsPerPass = NT / kVecRow;
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

template <int R, bool KC, typename T, int NT> struct StageSel { typedef StageKC<R, T, NT> type; };
template <int R, typename T, int NT> struct StageSel<R, false, T, NT> { typedef StageKS<R, T, NT> type; };

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
            if (p.partial) p.partial[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * p.M + row) * p.ldc + col] = v;
            else atomicAdd(c, v);
        } else {
            v = act_apply(p.act, v);
            if (p.Y) v *= act_grad_from_output(p.act_grad, p.Y[(size_t)row * p.ldy + col]);
            if (p.accumulate) v += *c;
            *c = v;
            if (p.C16) reinterpret_cast<__bf16*>(p.C16)[(size_t)row * p.ldc + col] = (__bf16)v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Epilogue shared by every bf16 kernel.
//
// A wave holds a (16 TM) x (16 TN) block of the tile in 16x16 accumulators (col = lane&15, row = 4 (lane>>4) + reg).
// One 16-row slice at a time is bounced through a wave-PRIVATE LDS strip (no workgroup barrier in here), after which
// a lane owns 4 consecutive columns of a row: bias / activation / act'(Y) / accumulate run on float4, C is written
// with 16-byte stores and the bf16 shadow with 8-byte stores.  (Per-lane scalar stores of the raw MFMA layout touch
// 4 rows x 64 B per instruction and made the store tail as long as the main loop for N ~ K.)
//
// The per-element work is specialised at compile time on (activation, act'(Y) form) for the combinations the
// training step uses -- linear / rectify output, rectify gradient from the bf16 copy of Y -- and selected ONCE per
// tile; the generic form with run-time switches stays as the fallback.  With the switches inside the element loop
// the epilogue was ~40k instructions of mostly skipped code per kernel and cost 10-15 us per tile (instruction
// fetch + scalar branches), more than the K-loop of a K <= 512 tile.
// ---------------------------------------------------------------------------------------------------------
constexpr int EPI_RUNTIME = -1;

template <int ACT, int YM>          // ACT: 0 linear, 1 rectify, -1 run-time; YM: 0 none, 1 rectify' from Y16, -1 run-time
__device__ __forceinline__ void epi_vec4(const GemmParams& p, float4 v, const float4& bias4, int row, int col, float4& csum) {
    const size_t off = (size_t)row * p.ldc + col;
    v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
    if (ACT == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    else if (ACT == EPI_RUNTIME) {
        v.x = act_apply(p.act, v.x); v.y = act_apply(p.act, v.y); v.z = act_apply(p.act, v.z); v.w = act_apply(p.act, v.w);
    }
    if (YM == 1) {
        const bf16x4 y = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.Y16) + (size_t)row * p.ldy + col);
        v.x = (float)y[0] > 0.f ? v.x : 0.f; v.y = (float)y[1] > 0.f ? v.y : 0.f;
        v.z = (float)y[2] > 0.f ? v.z : 0.f; v.w = (float)y[3] > 0.f ? v.w : 0.f;
    } else if (YM == EPI_RUNTIME) {
        if (p.Y16) {
            const bf16x4 y = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.Y16) + (size_t)row * p.ldy + col);
            v.x *= act_grad_from_output(p.act_grad, (float)y[0]); v.y *= act_grad_from_output(p.act_grad, (float)y[1]);
            v.z *= act_grad_from_output(p.act_grad, (float)y[2]); v.w *= act_grad_from_output(p.act_grad, (float)y[3]);
        } else if (p.Y) {
            const float4 y = *reinterpret_cast<const float4*>(p.Y + (size_t)row * p.ldy + col);
            v.x *= act_grad_from_output(p.act_grad, y.x); v.y *= act_grad_from_output(p.act_grad, y.y);
            v.z *= act_grad_from_output(p.act_grad, y.z); v.w *= act_grad_from_output(p.act_grad, y.w);
        }
    }
    if (p.accumulate) {
        const float4 c = *reinterpret_cast<const float4*>(p.C + off);
        v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w;
    }
    if (p.C) *reinterpret_cast<float4*>(p.C + off) = v;
    if (p.C16) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.C16) + off) = cvt4(v);
    csum.x += v.x; csum.y += v.y; csum.z += v.z; csum.w += v.w;
}

// element-wise fallback for rows that cannot be vectorised (unaligned C / Y, N % 4 != 0 edge)
__device__ __forceinline__ void epi_scalar(const GemmParams& p, const float* vv, int row, int col) {
    const size_t off = (size_t)row * p.ldc + col;
    for (int e = 0; e < 4 && col + e < p.N; ++e) {
        float x = vv[e] + (p.bias ? p.bias[col + e] : 0.f);
        x = act_apply(p.act, x);
        if (p.Y) x *= act_grad_from_output(p.act_grad, p.Y[(size_t)row * p.ldy + col + e]);
        if (p.accumulate) x += p.C[off + e];
        p.C[off + e] = x;
        if (p.C16) reinterpret_cast<__bf16*>(p.C16)[off + e] = (__bf16)x;
    }
}

// strip: 16 rows x (16 TN + 4) floats of LDS private to the calling wave; (row_base, col_base): the wave's block.
// csum accumulates this lane's 4 column sums (fused bias gradients).
template <int TM, int TN>
__device__ __forceinline__ void tile_epilogue(const GemmParams& p, f32x4 (&acc)[TM][TN], float* strip, int row_base,
                                              int col_base, int lane, float4& csum) {
    constexpr int LW = TN * 16 + 4;                   // floats per strip row (16-byte aligned, bank-staggered)
    constexpr int LPR = TN * 4;                       // lanes per row in the read phase (4 columns each)
    constexpr int RPI = 64 / LPR;                     // rows per read instruction
    const int i16 = lane & 15, kq4 = (lane >> 4) * 4;
    const int lc = (lane % LPR) * 4, col = col_base + lc;
    const bool vec_all = (p.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) && (p.N % 4 == 0) &&
                         (!p.Y || (p.ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(p.Y) & 15) == 0));
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && col + 3 < p.N) bias4 = *reinterpret_cast<const float4*>(p.bias + col);   // bias rows are 16-byte aligned
    auto run = [&](auto act_c, auto ym_c) {
        constexpr int ACT = decltype(act_c)::value, YM = decltype(ym_c)::value;
        auto slice = [&](auto a_c) {
            constexpr int a = decltype(a_c)::value;
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) strip[(kq4 + r) * LW + b * 16 + i16] = acc[a][b][r];
#pragma unroll
            for (int j = 0; j < 16 / RPI; ++j) {
                const int lr = j * RPI + lane / LPR;
                const int row = row_base + a * 16 + lr;
                if (row >= p.M || col >= p.N) continue;
                const float4 v = *reinterpret_cast<const float4*>(strip + lr * LW + lc);
                if (ACT != EPI_RUNTIME || (vec_all && col + 3 < p.N)) epi_vec4<ACT, YM>(p, v, bias4, row, col, csum);
                else epi_scalar(p, reinterpret_cast<const float*>(&v), row, col);
            }
        };
        slice(std::integral_constant<int, 0>{});
        if constexpr (TM > 1) slice(std::integral_constant<int, 1>{});
        if constexpr (TM > 2) slice(std::integral_constant<int, 2>{});
        if constexpr (TM > 3) slice(std::integral_constant<int, 3>{});
        if constexpr (TM > 4) slice(std::integral_constant<int, 4>{});
        if constexpr (TM > 5) slice(std::integral_constant<int, 5>{});
        if constexpr (TM > 6) slice(std::integral_constant<int, 6>{});
        if constexpr (TM > 7) slice(std::integral_constant<int, 7>{});
        static_assert(TM <= 8, "tile_epilogue: at most 8 accumulator rows");
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, EPI_RUNTIME> IR;
    const bool y_none = (!p.Y && !p.Y16) || p.act_grad == ADN_ACT_LINEAR;
    const bool y_relu16 = p.Y16 && p.act_grad == ADN_ACT_RECTIFY;
    if (vec_all && p.act == ADN_ACT_LINEAR && y_none) run(I0{}, I0{});
    else if (vec_all && p.act == ADN_ACT_RECTIFY && y_none) run(I1{}, I0{});
    else if (vec_all && p.act == ADN_ACT_LINEAR && y_relu16) run(I0{}, I1{});
    else run(IR{}, IR{});
}

// split-K form: the partial tile is ADDED to C.  Float atomics execute at the memory side, one request per 64 bytes,
// and run at full rate only for row-contiguous wave-instructions (MI355X_MICROARCH.md, Global float atomics): the
// slice goes through the same strip and every instruction then adds 64 (TN = 4) or 2 x 32 (TN = 2) consecutive
// floats of a row instead of the accumulator's 4 rows x 16 columns.  (Split-K launches carry no bias / activation.)
template <int TM, int TN>
__device__ __forceinline__ void tile_epilogue_atomic(const GemmParams& p, f32x4 (&acc)[TM][TN], float* strip, int row_base,
                                                     int col_base, int lane) {
    constexpr int LW = TN * 16 + 4, WTN = TN * 16;
    const int i16 = lane & 15, kq4 = (lane >> 4) * 4;
    auto slice = [&](auto a_c) {
        constexpr int a = decltype(a_c)::value;
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) strip[(kq4 + r) * LW + b * 16 + i16] = acc[a][b][r];
#pragma unroll
        for (int j = 0; j < 16 * WTN / 64; ++j) {     // 64 consecutive floats of the slice per wave-instruction
            const int e = j * 64 + lane, lr = e / WTN, c = e % WTN;
            const int row = row_base + a * 16 + lr, col = col_base + c;
            float v = strip[lr * LW + c];
            if (row >= p.M || col >= p.N) continue;
            if (p.Y) v *= act_grad_from_output(p.act_grad, p.Y[(size_t)row * p.ldy + col]);
            // (deterministic mode: this (problem, K-slice)'s slab, plain stores; added in slice order by rs_splitk_reduce_kernel.
            //  `p` is the group's view here: p.C was re-pointed, p.partial is the launch's)
            if (p.partial) p.partial[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * p.M + row) * p.ldc + col] = v;
            else atomicAdd(p.C + (size_t)row * p.ldc + col, v);
        }
    };
    slice(std::integral_constant<int, 0>{});
    if constexpr (TM > 1) slice(std::integral_constant<int, 1>{});
    if constexpr (TM > 2) slice(std::integral_constant<int, 2>{});
    if constexpr (TM > 3) slice(std::integral_constant<int, 3>{});
    static_assert(TM <= 4, "tile_epilogue_atomic: at most 4 accumulator rows");
}

// non-persistent kernels: WM x 2 waves; strips at the start of the (now free) LDS array; the waves stacked in m
// combine their column sums through LDS behind the strips
template <int WM, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[TM][TN], __bf16* smem, int m0, int n0,
                                              int tile_m, int wave, int lane) {
    constexpr int WTM = TM * 16, WTN = TN * 16;
    constexpr int LPR = TN * 4;
    constexpr int kStrip = 16 * (WTN + 4);            // floats per wave
    const int wm = wave >> 1, wn = wave & 1;
    __syncthreads();                                  // every wave is done with the operand images
    float* strips = reinterpret_cast<float*>(smem);
    if (p.atomic) {                                   // split-K partial sums
        tile_epilogue_atomic<TM, TN>(p, acc, strips + wave * kStrip, m0 + wm * WTM, n0 + wn * WTN, lane);
        return;
    }
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    tile_epilogue<TM, TN>(p, acc, strips + wave * kStrip, m0 + wm * WTM, n0 + wn * WTN, lane, csum);
    if (p.colsum) {                                   // lanes that differ only in their row share the columns
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
            csum.x += __shfl_xor(csum.x, o, 64); csum.y += __shfl_xor(csum.y, o, 64);
            csum.z += __shfl_xor(csum.z, o, 64); csum.w += __shfl_xor(csum.w, o, 64);
        }
        float4* cs = reinterpret_cast<float4*>(strips + WM * 2 * kStrip);
        if (wm > 0 && lane < LPR) cs[((wm - 1) * 2 + wn) * LPR + lane] = csum;
        __syncthreads();
        const int col = n0 + wn * WTN + lane * 4;
        if (wm == 0 && lane < LPR && col + 3 < p.N) {
#pragma unroll
            for (int w = 1; w < WM; ++w) {
                const float4 o = cs[((w - 1) * 2 + wn) * LPR + lane];
                csum.x += o.x; csum.y += o.y; csum.z += o.z; csum.w += o.w;
            }
            *reinterpret_cast<float4*>(p.colsum + (size_t)tile_m * p.colsum_ld + col) = csum;
        }
    }
}

// WM x 2 waves per workgroup, each owning a (BM/WM) x (BN/2) block of the tile
template <int BM, int BN, int WM, bool A_KC, bool B_KC, typename T>
__global__ __launch_bounds__(WM * 128) void gemm_bf16_kernel(const GemmParams pin) {
    // grouped launch (host: gemm_rs, bf16 operand copies only): blockIdx.z picks the problem's buffers
    GemmParams p = pin;
    if (pin.ngroups > 1) {
        const GemmGroup& g = pin.grp[blockIdx.z];
        p.A16 = g.A16; p.B16 = g.B16; p.C = g.C; p.C16 = g.C16; p.Y16 = g.Y16; p.Y = g.Y; p.bias = g.bias; p.colsum = g.colsum;
    }
    constexpr int NT = WM * 128;
    typedef typename StageSel<BM, A_KC, T, NT>::type SA;
    typedef typename StageSel<BN, B_KC, T, NT>::type SB;
    constexpr int WTM = BM / WM, WTN = BN / 2;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int kEpiElems = (WM * 2 * 16 * (WTN + 4) + (WM - 1) * 2 * WTN) * 2;   // epilogue strips + column sums, in bf16 units
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

    gemm_epilogue<WM, TM, TN>(p, acc, smem, m0, n0, tile_m, wave, lane);
}

// ---------------------------------------------------------------------------------------------------------
// Ping-pong LDS-DMA kernel for the large GEMMs (bf16 shadow operands; NN and TN).
//
// Persistent: ONE 512-thread workgroup per CU walks a list of BM x BN output tiles (256 x 256, 256 x 128 or
// 128 x 256); 8 waves as 4 (m) x 2 (n).  Operand tiles go L2 -> LDS with global_load_lds_dwordx4 (no VGPR staging, no
// ds_write) into a ring of 4 stage buffers of BK = 32, three stages ahead of the one being multiplied, and the ring
// keeps running across tile boundaries (the next tile's first stages land while this tile's epilogue stores).
//
// The two waves that share a SIMD (w and w + 4: the upper and the lower half of the tile's rows) run the same program
// HALF A K-STEP APART.  A K-step is two segments:
//     L(s): fragment reads of stage s into registers + DMA issue for stage s + 3        (LDS / address traffic)
//     C(s): TM x TN MFMAs on those registers (+ the tile epilogue after the last K-step)  (matrix pipe)
// and ONE s_barrier per K-step, placed behind C by waves 0-3 and behind L by waves 4-7:
//     waves 0-3:   L(0) C(0) | L(1) C(1) | L(2) C(2) | ...
//     waves 4-7:   L(0)      | C(0) L(1) | C(1) L(2) | ...          ( | = the barrier, the same event for all 8 waves)
// so that inside every interval one wave of a SIMD feeds the matrix pipe while its partner does the LDS reads and the DMA
// issue, and they swap in the middle -- instead of all eight waves running wait / barrier / DMA issue / reads / MFMA in
// lock-step (in-kernel stamps of the lock-step predecessor: 45 % of a K-step in reads + MFMA, 23 % DMA issue, 32 % waits;
// MFMA pipe 29 % busy).  (Round 2's first version separated L and C by a barrier each -- waves 0-3: L | C | L | C, waves 4-7
// one barrier behind: a K-step then costs 2 max(L, C) + two barriers, and L is the longer segment; with one barrier it is
// L + C + one barrier: 7 - 16 % faster on every shape, profiles/r02/pp_one_barrier.txt; the old schedule was removed
// in round 3; GemmParams::one_barrier stays set.)
//
// LDS hazards are settled by counted waits and the barrier sequence alone (interval s = between barrier #s-1 and #s):
//   RAW  stage s + 1 is first read in interval s + 1.  Every wave waits for ITS pieces of stage s + 1 (s_waitcnt vmcnt(N),
//        N = pieces of the younger stages it has issued) at the end of its L(s), which lies in interval s for both halves.
//        (One place for both halves on purpose: a wave-half test around the wait is a long-lived boolean that hipcc kept
//        in a VGPR, spilled, and reloaded behind s_waitcnt vmcnt(0) -- draining the ring.)
//   WAR  stage s + 3 overwrites the slot of stage s - 1 and is issued in L(s), i.e. in interval s; the last reads of that
//        slot are the L(s - 1) segments, both in interval s - 1 and each closed by s_waitcnt lgkmcnt(0) before barrier #s-1.
// vmcnt counts stores too and retires in order, so the epilogue's stores sit between the DMA groups of a wave's queue:
// the first wait after an epilogue leaves only the youngest stage in flight (it also waits for the store
// acknowledgements; once per tile).
//
// A DMA wave-instruction writes 1 KiB of LDS linearly (lane l -> base + 16 l), so the stage images are unpadded and
// bank conflicts are removed by permuting, per row, WHICH 16-byte chunk of global memory each lane fetches
// (cdna_hip_programming.md rule 21); the fragment reads apply the same permutation:
//   k-contiguous operand  image [R][32 k] (64-B rows):   chunk c of row r lives at slot c ^ f((r >> 2) & 3), f = {0,2,3,1}
//   k-strided operand     image [32 k][R] (2R-B rows):   chunk c of k-row kr lives at slot c ^ (2 g(kr)),
//                         g(kr) = (kr & 3) | ((kr >> 3) & 1) << 2   (the 8 k-rows one ds_read_b64_tr_b16 lane group
//                         touches get 8 different 32-byte slots of a 256-byte bank row)
// Rows / columns outside the matrix are clamped in-bounds (they only feed dropped outputs).  A partial last K-stage is
// fetched from clamped addresses and the k >= K elements of the A fragments are masked to zero in registers (one
// operand suffices: the other one is finite data).
//
// The MFMA operands are passed swapped (B fragment first), which leaves C transposed in the accumulators: a lane holds
// FOUR CONSECUTIVE COLUMNS of one row, so the epilogue runs straight from registers with 16-byte (fp32) / 8-byte (bf16)
// accesses -- no LDS strips, hence nothing that would collide with the running ring.
//
// Split-K (weight gradients: K = all frames of the batch): every (tile, K-slice) workgroup stores its partial tile
// into a slab of its own with plain stores and splitk_reduce_kernel adds the slabs into C.  (Float atomics run at
// ~1.3 TB/s chip-wide; a 256 x 256 tile per CU is 64 MB of them per GEMM = a 50 us tail.)
//
// Grouped launches: up to kMaxGemmGroups problems of identical shape and flags (the S input streams' encoder GEMMs)
// share one tile list, which fills the last round of 256 CUs far better than each alone (656 tiles = 2.56 rounds;
// 3 x 656 = 7.7).
// ---------------------------------------------------------------------------------------------------------
// ring depth: 4 stages (128 KB of LDS), three in flight.  -DADN_PP_NS=5 (all 160 KB, four in flight) measured no faster on any
// shape of profiles/gemm_lab: the K-step is not bound by the DMA latency
#ifndef ADN_PP_NS
#define ADN_PP_NS 4
#endif
constexpr int kPpBK = 32, kPpNS = ADN_PP_NS, kPpD = ADN_PP_NS - 1;

__device__ __forceinline__ int swz_g(int kr) { return (kr & 3) | (((kr >> 3) & 1) << 2); }
__device__ __forceinline__ int swz_f(int q) { return (0x1320 >> (4 * q)) & 3; }     // {0, 2, 3, 1}

// one float4 of the transposed-accumulator epilogue: 4 consecutive columns of one row.  ONE body for every flag
// combination the host routes here (linear / rectify output, optional rectify'(Y) mask from the bf16 copy of Y,
// optional accumulate), with wave-uniform branches: compile-time variants of this fully unrolled epilogue made the
// kernel ~50 k instructions, the K-loop then straddled that block and the kernel ran instruction-fetch bound (SQ_WAIT_INST_ANY
// 83 % of the wave cycles, 6.7x the L2 requests: 90 TFLOP/s instead of 700).
// The final values of one float4 of the tile (bias, activation, act'(Y) mask, accumulate) and its fp32 store; the bf16 copy
// is stored by the caller after the lane exchange (see the epilogue).  `ok`: this lane's row / columns are inside the matrix.
template <bool SPLIT>
__device__ __forceinline__ float4 pp_epi4(const GemmParams& p, const GemmGroup& gp, float* Cg, f32x4 a, const float4& bias4,
                                          const bf16x4& y, int row, int col, bool ok, float4& csum) {
    const size_t off = (size_t)row * p.ldc + col;
    float4 v = make_float4(a[0], a[1], a[2], a[3]);
    if (SPLIT) { if (ok) *reinterpret_cast<float4*>(Cg + off) = v; return v; }
    v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
    if (p.act == ADN_ACT_RECTIFY) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (gp.Y16) {                                                  // rectify'(Y) from the mask values fetched ahead of the block
        v.x = (float)y[0] > 0.f ? v.x : 0.f; v.y = (float)y[1] > 0.f ? v.y : 0.f;
        v.z = (float)y[2] > 0.f ? v.z : 0.f; v.w = (float)y[3] > 0.f ? v.w : 0.f;
    }
    if (p.accumulate && ok) {
        const float4 c = *reinterpret_cast<const float4*>(Cg + off);
        v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w;
    }
    if (Cg && ok) *reinterpret_cast<float4*>(Cg + off) = v;
    if (ok) { csum.x += v.x; csum.y += v.y; csum.z += v.z; csum.w += v.w; }
    return v;
}

// optional phase timing (build with -DADN_GEMM_STAMPS; read with adn_debug_gemm_stamps): shader-clock cycles that wave 0
// (early half) and wave 4 (late half) of workgroup 3 spend in [0] fragment-read issue, [1] DMA issue, [2] lgkmcnt wait,
// [3] vmcnt wait, [4] barrier behind L, [5] MFMAs, [6] epilogue, [7] barrier behind C; slots 8.. the same for wave 4
#ifdef ADN_GEMM_STAMPS
__device__ unsigned long long g_gstamps[16];
// wave-uniform accumulators (s_memtime is a scalar instruction: they live in SGPRs); written once at the end
#define GSTAMP(k) do { if (stamping) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); gacc_[k] += n_ - gl_; gl_ = n_; } } while (0)
#define GSTAMP_INIT const bool stamping = blockIdx.x == 3 && blockIdx.y == 0 && (wave == 0 || wave == 4); \
    unsigned long long gacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long gl_ = __builtin_amdgcn_s_memtime();
#define GSTAMP_FLUSH do { if (stamping && lane == 0) for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&g_gstamps[k_ + (late ? 8 : 0)], gacc_[k_]); } while (0)
#else
#define GSTAMP(k) do {} while (0)
#define GSTAMP_INIT
#define GSTAMP_FLUSH do {} while (0)
#endif

// PLANES: the launch may run over hi / lo planes (kseg_p > 0); false compiles the segment cursors out of the K-loop
template <int BM, int BN, bool A_KC, bool SPLIT, bool PLANES>
__global__ __launch_bounds__(512) void gemm_bf16_pp_kernel(const GemmParams p) {
    const int kseg_p = PLANES ? p.kseg : 0;
    constexpr int BK = kPpBK, NS = kPpNS, D = kPpD;
    constexpr int WTM = BM / 4, WTN = BN / 2, TM = WTM / 16, TN = WTN / 16;
    constexpr int kAElems = BM * BK, kBElems = BK * BN, kStageElems = kAElems + kBElems;
    constexpr int APW = BM / 128, BPW = BN / 128, PW = APW + BPW;      // 1-KiB DMA pieces per wave and stage
    __shared__ __attribute__((aligned(1024))) __bf16 smem[NS * kStageElems];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bool late = wave >= 4;                                   // the half that runs one barrier behind
    const int per_group = p.tiles_m * p.tiles_n;
    const int ntiles = per_group * p.ngroups;
    const int G = (int)gridDim.x;
    // Split-K: which (tile, K-slice) this workgroup takes.  p.xcd_slices: the K-slice is a function of the workgroup's XCD
    // (workgroups are dealt round-robin over the 8 XCDs in launch order, x fastest), so that the workgroups of one XCD run
    // through the SAME K-slice in step and share one 32-row window of A and B in its L2.  Measured on the 2000 x 1000 x
    // 20800 weight gradient (profiles/r02/pp_xcd_slices.txt): L2 hit rate 50 % -> 76 %, bytes fetched from beyond L2 333 MB
    // -> 125 MB (= the operands once); the launch time does not change (not bound there), the HBM traffic does.
    int bid = (int)blockIdx.x, slice = (int)blockIdx.y;
    if (SPLIT && p.xcd_slices) {
        const int S = (int)gridDim.y, lin = (int)blockIdx.y * G + (int)blockIdx.x, xcd = lin & 7, q = lin >> 3;
        if (S >= 8) { const int m_ = S >> 3; slice = xcd * m_ + q % m_; bid = q / m_; }
        else { const int d_ = 8 / S; slice = xcd % S; bid = q * d_ + xcd / S; }
    }
    const int my_tiles = (ntiles - bid + G - 1) / G;
    const int kbeg = slice * p.k_chunk;
    const int kend = min(p.K, kbeg + p.k_chunk);
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int ktail = (kend - kbeg) - (nk - 1) * BK;               // valid k of the last stage (1..32)
    const bool has_tail = ktail < BK;
    const int total = my_tiles * nk;                               // K-steps of this workgroup's whole tile list
    if (total <= 0) return;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_void_t*)smem;

    // ---- DMA side ------------------------------------------------------------------------------------
    // k-strided operand of width R: CPR = R / 8 chunks per k-row, one piece = 64 / CPR k-rows
    constexpr int A_CPR = BM / 8, B_CPR = BN / 8;
    // source of a piece = wave-uniform base (the operand at the first k of the stage being issued) + a per-lane byte offset
    const char* baseA = nullptr; const char* baseB = nullptr;
    unsigned offA[APW], offB[BPW];
    int a_aux[APW], b_row[BPW];      // A_KC: the lane's k offset inside a stage; k-strided: its k-row inside a stage
#pragma unroll
    for (int t = 0; t < APW; ++t)
        a_aux[t] = A_KC ? ((lane & 3) ^ swz_f((lane >> 4) & 3)) * 8 : (64 / A_CPR) * (wave * APW + t) + lane / A_CPR;
#pragma unroll
    for (int t = 0; t < BPW; ++t) b_row[t] = (64 / B_CPR) * (wave * BPW + t) + lane / B_CPR;
    auto tile_of = [&](int ord, int& grp, int& tm, int& tn) {      // ord-th tile of this workgroup
        const int q = xcd_tile(bid + ord * G, ntiles);
        grp = q / per_group;
        tile_coords(p, q - grp * per_group, tm, tn);
    };
    // bf16x3 through hi / lo PLANES (kseg_p > 0): K runs over three segments of kseg_p (a multiple of BK) -- A_hi B_hi, A_hi B_lo,
    // A_lo B_hi -- read from the two bf16 planes of each operand (same layout, written once by the operand's producer) instead of
    // from [hi | hi | lo] / [hi ; lo ; hi] images made per use.  Only the DMA cursors know: a tile keeps its four plane bases
    // and counts the stages to the next segment boundary.
    const int kseg_steps = kseg_p / BK;
    const char *segA_hi = nullptr, *segA_lo = nullptr, *segB_hi = nullptr, *segB_lo = nullptr;
    int seg_cur = 0, seg_left = 0;
    auto setup_src = [&](int ord) {
        int grp, tm, tn;
        tile_of(ord, grp, tm, tn);
        const GemmGroup sg = pick_group(p, grp);
        const char* A16 = reinterpret_cast<const char*>(sg.A16);
        const char* B16 = reinterpret_cast<const char*>(sg.B16);
        const int m0 = tm * BM, n0 = tn * BN;
        int kin = kbeg;
        if (kseg_p) {
            seg_cur = kbeg / kseg_p; kin = kbeg - seg_cur * kseg_p; seg_left = kseg_steps - kin / BK;
            segA_hi = A_KC ? A16 + (size_t)m0 * p.lda * 2 : A16;
            segA_lo = reinterpret_cast<const char*>(sg.A16lo) + (segA_hi - A16);
            segB_hi = B16; segB_lo = reinterpret_cast<const char*>(sg.B16lo);
            A16 = seg_cur == 2 ? reinterpret_cast<const char*>(sg.A16lo) : A16;
            B16 = seg_cur == 1 ? segB_lo : B16;
        }
        baseA = A_KC ? A16 + ((size_t)m0 * p.lda + kin) * 2 : A16 + (size_t)kin * p.lda * 2;
        baseB = B16 + (size_t)kin * p.ldb * 2;
#pragma unroll
        for (int t = 0; t < APW; ++t) {
            if (A_KC) {
                const int row = min(m0 + 16 * (wave * APW + t) + (lane >> 2), p.M - 1) - m0;
                offA[t] = (unsigned)(row * p.lda + a_aux[t]) * 2u;
            } else {
                int col = m0 + (((lane % A_CPR) ^ (swz_g(a_aux[t]) << 1)) * 8);
                if (col + 8 > p.lda) col = 0;
                offA[t] = (unsigned)(a_aux[t] * p.lda + col) * 2u;
            }
        }
#pragma unroll
        for (int t = 0; t < BPW; ++t) {
            int col = n0 + (((lane % B_CPR) ^ (swz_g(b_row[t]) << 1)) * 8);
            if (col + 8 > p.ldb) col = 0;
            offB[t] = (unsigned)(b_row[t] * p.ldb + col) * 2u;
        }
    };
    const size_t a_step = A_KC ? (size_t)BK * 2 : (size_t)BK * p.lda * 2;      // bytes per K-step
    const size_t b_step = (size_t)BK * p.ldb * 2;
    // LDS destination of this wave's first piece of a stage in ring slot 0
    const unsigned dstA_w = __builtin_amdgcn_readfirstlane(lds_base + wave * APW * 1024);
    const unsigned dstB_w = __builtin_amdgcn_readfirstlane(lds_base + NS * kAElems * 2 + wave * BPW * 1024);
    int is_k = 0, is_ord = 0, is_slot = 0;                         // next stage to issue: K-step inside its tile, tile, ring slot
    auto issue_one = [&](auto tail_c) {
        constexpr bool tail = decltype(tail_c)::value;
        const unsigned dA = dstA_w + (unsigned)(is_slot * kAElems * 2);
        const unsigned dB = dstB_w + (unsigned)(is_slot * kBElems * 2);
        const int k0 = kbeg + is_k * BK;
#pragma unroll
        for (int t = 0; t < APW; ++t) {
            if (tail) {                                            // k >= kend: re-read the last valid k (masked in registers)
                const char* g = baseA + offA[t];
                if (A_KC) { if (k0 + a_aux[t] >= kend) g -= (size_t)(k0 + a_aux[t] - (kend - 8)) * 2; }
                else if (k0 + a_aux[t] >= kend) g -= (size_t)(k0 + a_aux[t] - (kend - 1)) * p.lda * 2;
                glds16(g, __builtin_amdgcn_readfirstlane(dA + t * 1024));
            } else {
                glds16_s<0>(offA[t], baseA, __builtin_amdgcn_readfirstlane(dA + t * 1024));
            }
        }
#pragma unroll
        for (int t = 0; t < BPW; ++t) {
            if (tail) {
                const char* g = baseB + offB[t];
                if (k0 + b_row[t] >= kend) g -= (size_t)(k0 + b_row[t] - (kend - 1)) * p.ldb * 2;
                glds16(g, __builtin_amdgcn_readfirstlane(dB + t * 1024));
            } else {
                glds16_s<0>(offB[t], baseB, __builtin_amdgcn_readfirstlane(dB + t * 1024));
            }
        }
        baseA += a_step; baseB += b_step;
        if (kseg_p && --seg_left == 0) {                           // next stage opens the next segment
            ++seg_cur; seg_left = kseg_steps;
            baseA = seg_cur == 2 ? segA_lo : segA_hi;
            baseB = seg_cur == 1 ? segB_lo : segB_hi;
        }
    };
    auto issue_next = [&]() {
        if (has_tail && is_k == nk - 1) issue_one(std::true_type{});
        else issue_one(std::false_type{});
        if (++is_slot == NS) is_slot = 0;
        if (++is_k == nk) {
            is_k = 0;
            if (++is_ord < my_tiles) setup_src(is_ord);
        }
    };

    // ---- per-lane fragment addresses (LDS byte addresses inside ring slot 0) ----------------------------
    const int q = (lane & 15) >> 2, pp = lane & 3, hi = lane >> 4;
    const int g_lane = q | ((hi & 1) << 2);                        // swz_g of every k-row this lane reads
    unsigned a_off[TM], b_off[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        if (A_KC) a_off[a] = lds_base + (wm * WTM + a * 16 + (lane & 15)) * 64 + ((hi ^ swz_f(q)) << 4);
        else a_off[a] = lds_base + (hi * 8 + q) * (BM * 2) + ((((wm * WTM + a * 16) / 8 + (pp >> 1)) ^ (g_lane << 1)) << 4) + (pp & 1) * 8;
    }
#pragma unroll
    for (int b = 0; b < TN; ++b)         // (the B stages follow the NS A stages)
        b_off[b] = lds_base + NS * kAElems * 2 + (hi * 8 + q) * (BN * 2) + ((((wn * WTN + b * 16) / 8 + (pp >> 1)) ^ (g_lane << 1)) << 4) + (pp & 1) * 8;
    // k >= K mask of an A fragment in the last stage: element j of this lane is k = 8 hi + j
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // (planes: every segment ends with the same partial stage when the real K is not a multiple of BK -- the A operand's
    //  columns behind K may hold anything, e.g. a wider earlier use of a reused gradient buffer)
    const int ktail_m = kseg_p ? p.kreal - (kseg_p - BK) : ktail;
    const bool seg_tail = kseg_p && ktail_m < BK;
    u32x4 amask;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        amask[j] = ((8 * hi + 2 * j < ktail_m) ? 0x0000FFFFu : 0u) | ((8 * hi + 2 * j + 1 < ktail_m) ? 0xFFFF0000u : 0u);

    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // wait until this wave's pieces of stage s + 1 have landed; `fresh_epi`: an epilogue's stores are in the queue.
    // (vmcnt counts loads, stores and LDS-DMA in issue order, so the stores of a tile's epilogue could be left outstanding
    //  under a counted wait -- vmcnt <= stages + stores -- for the first D - 1 K-steps of the next tile instead of being
    //  drained here.  Built and measured in round 3: no gain -- the DMA issued behind the stores cannot be seen complete
    //  before them, so the drain only moves D - 1 K-steps later -- and the extra scalar state cost the K-loop 10 %.)
    auto wait_next = [&](int s, bool fresh_epi) {
        int younger = min(D - 1, total - 2 - s);                   // younger stages this wave has issued
        if (younger < 0) return;                                   // there is no stage s + 1
        if (fresh_epi) younger = min(younger, 1);
        if (D >= 4 && younger >= 3) wait_vmcnt<(D >= 4 ? 3 : 2) * PW>();
        else if (younger >= 2) wait_vmcnt<2 * PW>();
        else if (younger == 1) wait_vmcnt<PW>();
        else wait_vmcnt<0>();
    };

    // ---- prologue: D stages in flight, stage 0 landed for everyone -----------------------------------
    setup_src(0);
    for (int s = 0; s < D && s < total; ++s) issue_next();
    wait_next(-1, false);
    __builtin_amdgcn_s_barrier();                                  // #0
    constexpr bool one = true;                                     // (the two-barrier schedule of round 2 is gone: 7 - 16 % slower)
    if (late && !one) __builtin_amdgcn_s_barrier();                // the lower half starts one segment later

    int kt = 0, ord = 0;
    int grp, tile_m, tile_n;
    tile_of(0, grp, tile_m, tile_n);
    const int cseg_first = kseg_p ? kseg_steps - (kbeg % kseg_p) / BK : 0;      // K-steps to the first segment boundary of a tile
    int cseg_left = cseg_first;
    GSTAMP_INIT
    typedef __attribute__((address_space(3))) bf16x8 lds_bf16x8;
    bf16x8 fa[TM], fb[TN];
    // fragments of the stage in the ring slot at byte offsets (sa, sb) of the A / B regions -> registers
    auto read_frags = [&](unsigned sa, unsigned sb) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            if (A_KC) {
                fa[a] = *(lds_bf16x8*)(uintptr_t)(a_off[a] + sa);
            } else {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(a_off[a] + sa));
                const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(a_off[a] + sa + 4 * (BM * 2)));
                fa[a] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(b_off[b] + sb));
            const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(b_off[b] + sb + 4 * (BN * 2)));
            fb[b] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };
    auto mfmas = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[b], fa[a], acc[a][b], 0, 0, 0);
    };
    int rd_slot = 0;
    // The epilogue of a finished tile runs at the TOP of the next iteration (or of one extra, epilogue-only pass behind the last
    // K-step), not behind the tile's last MFMA block.  For waves 4-7 nothing moves (C(last), epilogue, L(next)); waves 0-3, whose
    // barrier sits behind C, used to run it inside the interval of their last K-step -- the other half then waited out that whole
    // epilogue at its barrier, and one interval later the roles swapped: a tile paid for BOTH halves' epilogues back to back
    // (stamps: epilogue 17-28 % + partner's barrier wait 19-25 % of a launch).  Deferred, the two halves' epilogues fall
    // into the same interval and overlap.
    for (int s = 0;;) {
        if (kt == 0 && s > 0) {
        {
        // ---- epilogue of tile `ord`, straight from the (transposed) accumulators: lane = row (lane & 15),
        //      columns 4 hi .. 4 hi + 3 of every 16-column block
        const GemmGroup gp = pick_group(p, grp);
        float* Cg = SPLIT ? p.partial + ((size_t)(grp * (int)gridDim.y + slice) * p.M) * p.ldc : gp.C;
        constexpr int TNH = 2;                               // 16-column blocks per burst (TM x 2 loads in flight per lane)
        // Addressing: ONE 32-bit element offset per lane from the matrix start (the rows of every matrix span < 4 GB) plus
        // compile-time constants and wave-uniform strides; bounds as three lane predicates per pair of blocks; flag tests
        // once per pair.  (The first form of this epilogue recomputed a 64-bit address with clamps per access and went
        // through a dozen wave-uniform branches per float4: ~80 instructions per float4, 5000 per wave and tile, which two
        // waves per SIMD took 18 us to issue -- a quarter to a half of a launch by the stamps.)
        {
            const int wr0 = tile_m * BM + wm * WTM, wc0 = tile_n * BN + wn * WTN;                  // the wave's corner (uniform)
            const int rl = (lane & 15), cl = 4 * hi;
            const int rlim = p.M - wr0, clim = p.N - wc0;                                        // rows / columns of it inside the matrix
            const unsigned lo32 = (unsigned)(wr0 + rl) * (unsigned)p.ldc + (unsigned)(wc0 + cl);
            const unsigned ly32 = (unsigned)(wr0 + rl) * (unsigned)p.ldy + (unsigned)(wc0 + cl);
            const unsigned rstep = 16u * (unsigned)p.ldc, ystep = 16u * (unsigned)p.ldy;       // one row block further (uniform)
            __bf16* C16m = (!SPLIT && gp.C16) ? reinterpret_cast<__bf16*>(gp.C16) : nullptr;
            __bf16* C16lm = (!SPLIT && gp.C16lo) ? reinterpret_cast<__bf16*>(gp.C16lo) : nullptr;
            // rectifier bit images (GemmGroup::Cbits / Ybits): one uint4 per thread and tile, bit 4 q + j = column j of quad q = a TN + b
            const uint4* Ybm = (!SPLIT && gp.Ybits) ? reinterpret_cast<const uint4*>(gp.Ybits) : nullptr;
            uint4* Cbm = (!SPLIT && gp.Cbits) ? reinterpret_cast<uint4*>(gp.Cbits) : nullptr;
            const size_t bits_at = ((size_t)tile_m * p.tiles_n + tile_n) * 512 + tid;
            unsigned yb[4] = {0u, 0u, 0u, 0u}, cb[4] = {0u, 0u, 0u, 0u};
            if (Ybm) { const uint4 t4 = Ybm[bits_at]; yb[0] = t4.x; yb[1] = t4.y; yb[2] = t4.z; yb[3] = t4.w; }
            const __bf16* Ym = (!SPLIT && gp.Y16 && !Ybm) ? reinterpret_cast<const __bf16*>(gp.Y16) : nullptr;
            const float* biasm = (!SPLIT && gp.bias) ? gp.bias : nullptr;
            const float lower = (!SPLIT && p.act == ADN_ACT_RECTIFY) ? 0.f : -3.0e38f;
            const bool odd = hi & 1;
            // What the epilogue READS is requested in two bursts per tile, each ahead of its half's first store: while an LDS-DMA
            // is in flight hipcc waits vmcnt(0) -- not a counted vmcnt -- at the first use of an ordinary load's result, which
            // also waits for every store issued before that point; with the requests at the top of each pair of blocks every
            // pair sat out the write acknowledgements of the pair before (4 drains per tile, 2 us each).  One register array
            // serves both kinds (a launch has a bias or an act'(Y) mask, never both): 4 float4 of bias, or TM x 4 8-byte masks.
            // blocks per request burst: two bursts per tile (all of it at once spills); four in the k-strided-A form, whose fragment
            // addressing leaves fewer registers (2 VGPRs went to scratch there -- and scratch traffic sits in the vmcnt queue
            // this kernel counts by hand)
            constexpr int QB = (A_KC && !PLANES) ? TN / 2 : TN / 4;       // (... and over planes: the segment cursors take their share)
            uint2 pre[TM * QB];
#pragma clang loop unroll(full)
            for (int half = 0; half < TN / TNH; ++half) {
                const int b0 = half * TNH;
                const int q0 = (b0 / QB) * QB;                 // first block of this burst
                if (b0 == q0) {
                    if (Ym) {
#pragma clang loop unroll(full)
                        for (int b = 0; b < QB; ++b)
#pragma clang loop unroll(full)
                            for (int a = 0; a < TM; ++a) {
                                const bool ok = rl + a * 16 < rlim && cl + (q0 + b) * 16 < clim;
                                pre[a * QB + b] = *reinterpret_cast<const uint2*>(Ym + (ok ? ly32 + a * ystep + (q0 + b) * 16 : 0u));
                            }
                    } else if (biasm) {
#pragma clang loop unroll(full)
                        for (int b = 0; b < QB; ++b) {
                            const float4 bv = *reinterpret_cast<const float4*>(biasm + (cl + (q0 + b) * 16 < clim ? wc0 + cl + (q0 + b) * 16 : 0));
                            pre[2 * b] = make_uint2(__builtin_bit_cast(unsigned, bv.x), __builtin_bit_cast(unsigned, bv.y));
                            pre[2 * b + 1] = make_uint2(__builtin_bit_cast(unsigned, bv.z), __builtin_bit_cast(unsigned, bv.w));
                        }
                    } else {
#pragma clang loop unroll(full)
                        for (int k = 0; k < 2 * QB; ++k) pre[k] = make_uint2(0u, 0u);
                    }
                }
                const int bq = b0 - q0;                        // block index inside the burst
                const bool cA = cl + b0 * 16 < clim, cB = cl + b0 * 16 + 16 < clim;          // this lane's 4 columns of block A / B inside
                // after the exchange a lane holds 8 consecutive columns: even lanes A's cl .. cl+7, odd lanes cl+12 .. cl+19 (= B's cl-4 .. cl+3)
                const int cs = cl + (odd ? 12 : 0) + b0 * 16;
                const bool c16 = cs + 8 <= clim;                                                 // all 8 inside
                const bool c8 = cs + 4 <= clim;                                                  // the first 4 only (right edge)
                const unsigned l16 = lo32 + (odd ? 12u : 0u) + (unsigned)(b0 * 16);
                float4 biasA = make_float4(0.f, 0.f, 0.f, 0.f), biasB = biasA, csumA = biasA, csumB = biasA;
                if (!Ym) {                                   // (zeros without a bias)
                    biasA = make_float4(__builtin_bit_cast(float, pre[2 * bq].x), __builtin_bit_cast(float, pre[2 * bq].y),
                                        __builtin_bit_cast(float, pre[2 * bq + 1].x), __builtin_bit_cast(float, pre[2 * bq + 1].y));
                    biasB = make_float4(__builtin_bit_cast(float, pre[2 * bq + 2].x), __builtin_bit_cast(float, pre[2 * bq + 2].y),
                                        __builtin_bit_cast(float, pre[2 * bq + 3].x), __builtin_bit_cast(float, pre[2 * bq + 3].y));
                }
#pragma clang loop unroll(full)
                for (int a = 0; a < TM; ++a) {
                    __builtin_amdgcn_sched_barrier(0);         // (keeps the scheduler from interleaving all 32 block pairs: spills)
                    const bool rok = rl + a * 16 < rlim;
                    const bool okA = rok && cA, okB = rok && cB;
                    float4 vA = make_float4(acc[a][b0][0], acc[a][b0][1], acc[a][b0][2], acc[a][b0][3]);
                    float4 vB = make_float4(acc[a][b0 + 1][0], acc[a][b0 + 1][1], acc[a][b0 + 1][2], acc[a][b0 + 1][3]);
                    const unsigned oA = lo32 + a * rstep + (unsigned)(b0 * 16), oB = oA + 16u;
                    if (!SPLIT) {
                        vA.x = fmaxf(vA.x + biasA.x, lower); vA.y = fmaxf(vA.y + biasA.y, lower); vA.z = fmaxf(vA.z + biasA.z, lower); vA.w = fmaxf(vA.w + biasA.w, lower);
                        vB.x = fmaxf(vB.x + biasB.x, lower); vB.y = fmaxf(vB.y + biasB.y, lower); vB.z = fmaxf(vB.z + biasB.z, lower); vB.w = fmaxf(vB.w + biasB.w, lower);
                        if (Ym) {
                            const bf16x4 yA = __builtin_bit_cast(bf16x4, pre[a * QB + bq]), yB = __builtin_bit_cast(bf16x4, pre[a * QB + bq + 1]);
                            vA.x = (float)yA[0] > 0.f ? vA.x : 0.f; vA.y = (float)yA[1] > 0.f ? vA.y : 0.f;
                            vA.z = (float)yA[2] > 0.f ? vA.z : 0.f; vA.w = (float)yA[3] > 0.f ? vA.w : 0.f;
                            vB.x = (float)yB[0] > 0.f ? vB.x : 0.f; vB.y = (float)yB[1] > 0.f ? vB.y : 0.f;
                            vB.z = (float)yB[2] > 0.f ? vB.z : 0.f; vB.w = (float)yB[3] > 0.f ? vB.w : 0.f;
                        } else if (Ybm) {
                            const int qA = a * TN + b0, qB = qA + 1;              // (constants once the loops are unrolled: the yb / cb words stay in registers)
                            const unsigned mA = yb[qA >> 3] >> ((qA & 7) * 4), mB = yb[qB >> 3] >> ((qB & 7) * 4);
                            vA.x = (mA & 1u) ? vA.x : 0.f; vA.y = (mA & 2u) ? vA.y : 0.f; vA.z = (mA & 4u) ? vA.z : 0.f; vA.w = (mA & 8u) ? vA.w : 0.f;
                            vB.x = (mB & 1u) ? vB.x : 0.f; vB.y = (mB & 2u) ? vB.y : 0.f; vB.z = (mB & 4u) ? vB.z : 0.f; vB.w = (mB & 8u) ? vB.w : 0.f;
                        }
                        if (Cbm) {
                            const int qA = a * TN + b0, qB = qA + 1;
                            const unsigned mA = (vA.x > 0.f ? 1u : 0u) | (vA.y > 0.f ? 2u : 0u) | (vA.z > 0.f ? 4u : 0u) | (vA.w > 0.f ? 8u : 0u);
                            const unsigned mB = (vB.x > 0.f ? 1u : 0u) | (vB.y > 0.f ? 2u : 0u) | (vB.z > 0.f ? 4u : 0u) | (vB.w > 0.f ? 8u : 0u);
                            cb[qA >> 3] |= mA << ((qA & 7) * 4); cb[qB >> 3] |= mB << ((qB & 7) * 4);
                        }
                        if (p.accumulate) {
                            if (okA) { const float4 c = *reinterpret_cast<const float4*>(Cg + oA); vA.x += c.x; vA.y += c.y; vA.z += c.z; vA.w += c.w; }
                            if (okB) { const float4 c = *reinterpret_cast<const float4*>(Cg + oB); vB.x += c.x; vB.y += c.y; vB.z += c.z; vB.w += c.w; }
                        }
                    }
                    if (Cg) {
                        if (okA) *reinterpret_cast<float4*>(Cg + oA) = vA;
                        if (okB) *reinterpret_cast<float4*>(Cg + oB) = vB;
                    }
                    if (!SPLIT) {
                        if (okA) { csumA.x += vA.x; csumA.y += vA.y; csumA.z += vA.z; csumA.w += vA.w; }
                        if (okB) { csumB.x += vB.x; csumB.y += vB.y; csumB.z += vB.z; csumB.w += vB.w; }
                    }
                    if (C16m) {
                        const uint2 pA = __builtin_bit_cast(uint2, cvt4(vA)), pB = __builtin_bit_cast(uint2, cvt4(vB));
                        const uint2 give = odd ? pA : pB;        // even hi keeps block A and takes the partner's half of it
                        uint2 take;
                        take.x = __shfl_xor(give.x, 16, 64); take.y = __shfl_xor(give.y, 16, 64);
                        const uint4 out = odd ? make_uint4(take.x, take.y, pB.x, pB.y) : make_uint4(pA.x, pA.y, take.x, take.y);
                        __bf16* dst = C16m + (l16 + a * rstep);
                        if (rok && c16) *reinterpret_cast<uint4*>(dst) = out;
                        else if (rok && c8) *reinterpret_cast<uint2*>(dst) = make_uint2(out.x, out.y);
                        if (C16lm) {                             // bf16x3: the lo plane bf16(v - bf16(v)) beside it, same exchange
                            const bf16x4 hA = __builtin_bit_cast(bf16x4, pA), hB = __builtin_bit_cast(bf16x4, pB);
                            const float4 rA = make_float4(vA.x - (float)hA[0], vA.y - (float)hA[1], vA.z - (float)hA[2], vA.w - (float)hA[3]);
                            const float4 rB = make_float4(vB.x - (float)hB[0], vB.y - (float)hB[1], vB.z - (float)hB[2], vB.w - (float)hB[3]);
                            const uint2 qA = __builtin_bit_cast(uint2, cvt4(rA)), qB = __builtin_bit_cast(uint2, cvt4(rB));
                            const uint2 giv = odd ? qA : qB;
                            uint2 tk;
                            tk.x = __shfl_xor(giv.x, 16, 64); tk.y = __shfl_xor(giv.y, 16, 64);
                            const uint4 outl = odd ? make_uint4(tk.x, tk.y, qB.x, qB.y) : make_uint4(qA.x, qA.y, tk.x, tk.y);
                            __bf16* dl = C16lm + (l16 + a * rstep);
                            if (rok && c16) *reinterpret_cast<uint4*>(dl) = outl;
                            else if (rok && c8) *reinterpret_cast<uint2*>(dl) = make_uint2(outl.x, outl.y);
                        }
                    }
                }
                if (Cbm && half == TN / TNH - 1) Cbm[bits_at] = make_uint4(cb[0], cb[1], cb[2], cb[3]);
                if (!SPLIT && gp.colsum) {                     // column sums over this wave's WTM rows
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) {
                        csumA.x += __shfl_xor(csumA.x, o, 64); csumA.y += __shfl_xor(csumA.y, o, 64);
                        csumA.z += __shfl_xor(csumA.z, o, 64); csumA.w += __shfl_xor(csumA.w, o, 64);
                        csumB.x += __shfl_xor(csumB.x, o, 64); csumB.y += __shfl_xor(csumB.y, o, 64);
                        csumB.z += __shfl_xor(csumB.z, o, 64); csumB.w += __shfl_xor(csumB.w, o, 64);
                    }
                    if ((lane & 15) == 0) {
                        float* cs_row = gp.colsum + (size_t)(tile_m * 4 + wm) * p.colsum_ld + wc0 + cl + b0 * 16;
                        if (cA) *reinterpret_cast<float4*>(cs_row) = csumA;
                        if (cB) *reinterpret_cast<float4*>(cs_row + 16) = csumB;
                    }
                }
            }
        }
        }
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (++ord < my_tiles) tile_of(ord, grp, tile_m, tile_n);
        // A compiler-visible vmcnt(0) behind the epilogue: its loads and stores are the only vector-memory operations hipcc
        // knows of in this kernel (the DMA and its waits are inline asm), and with any of them possibly pending it put an
        // s_waitcnt vmcnt(0) at the head of the interior loop below -- the whole ring drained at every K-step.  Here the wait
        // is free: the stages in flight were issued before the epilogue, and the next wait site drains the stores anyway.
        __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        GSTAMP(6);
        if (s == total) break;
        // ---------------- the interior steps of a tile, stripped of every test the general step carries: the stage issued
        //      (s + D) lies inside this tile and ahead of its masked last one (kt + D < nk - 1), this step is neither the first
        //      behind an epilogue (kt >= 1) nor masked, D - 1 younger stages are in flight.  Same L / wait / barrier / C order.
        {
            // (over planes: neither the stage issued nor the step consumed may touch a segment boundary)
            while (kt >= 1 && kt + D < nk - 1 && (!PLANES || !kseg_p || (seg_left > 1 && cseg_left > 1))) {
                read_frags(rd_slot * (kAElems * 2), rd_slot * (kBElems * 2));
                if (++rd_slot == NS) rd_slot = 0;
                issue_one(std::false_type{});
                if (++is_slot == NS) is_slot = 0;
                ++is_k;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                wait_vmcnt<(D - 1) * PW>();
                if (late) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                mfmas();
                ++kt;
                if (PLANES && kseg_p) --cseg_left;
                __builtin_amdgcn_sched_barrier(0);
                if (!late) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                ++s;
            }
        }
        // ---------------- the general step.  L(s): fragments of stage s -> registers, DMA for stage s + D
        read_frags(rd_slot * (kAElems * 2), rd_slot * (kBElems * 2));
        if (++rd_slot == NS) rd_slot = 0;
        GSTAMP(0);
        if (s + D < total) issue_next();
        GSTAMP(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        GSTAMP(2);
        if ((has_tail && kt == nk - 1) || (seg_tail && cseg_left == 1)) {
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, fa[a]) & amask);
        }
        if (kseg_p && --cseg_left == 0) cseg_left = kseg_steps;
        wait_next(s, kt == 0 && s > 0);
        GSTAMP(3);
        if (!one || late) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        GSTAMP(4);
        // ---------------- C(s)
        mfmas();
        GSTAMP(5);
        if (__builtin_expect(kt == nk - 1, 0)) {        // the tile is complete: its epilogue opens the next iteration
            kt = 0;
            cseg_left = cseg_first;
        } else {
            ++kt;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!one || !late) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        GSTAMP(7);
        ++s;
    }
    GSTAMP_FLUSH;
    if (!late && !one) __builtin_amdgcn_s_barrier();               // every wave executes the same number of barriers
}

// ---------------------------------------------------------------------------------------------------------
// Four-wave variant of the LDS-DMA kernel (round 3): the same 256 x 256 tile, ring, DMA images, swizzles, tail and
// epilogue conventions as gemm_bf16_pp_kernel above, but ONE wave per SIMD with a 128 x 128 wave tile (8 x 8 MFMA tiles:
// 64 MFMAs per K-step of 32 against 16 A + 16 B fragment registers, i.e. 0.25 (NN) .. 0.375 (TN) LDS reads per MFMA where
// the eight-wave kernel's 64 x 128 wave tile needs 0.625 .. 0.75), the accumulators in the AGPR half of the 512-entry
// register file, and the overlap of LDS traffic with the matrix pipe done INSIDE the wave instead of between two waves of
// a SIMD: the fragments of stage s + 1 are read into a second register set while the MFMAs of stage s issue
// (MI355X_MICROARCH.md: a single wave per SIMD issues 1-2 LDS reads per MFMA gap at <= 3 cycles per gap, whereas two
// waves per SIMD pay ~20 cycles of issue per LDS / DMA instruction while the partner holds the vector issue port with
// MFMAs -- the eight-wave kernel's in-kernel stamps: reads + DMA issue + waits ~1150 cycles next to 540 of MFMA per wave).
//
// One K-step of one wave:
//     wait (own DMA pieces of stage s + 1)  |  s_barrier  |  DMA issue for stage s + 3  |  fragment reads of stage s + 1 ->
//     set (s + 1) & 1, interleaved by the compiler with the 64 MFMAs on set s & 1  |  (tile epilogue after the last K-step)
// Hazards (barrier #s = the one at the top of step s):
//   RAW  stage s + 1 is read after barrier #s; every wave waited for its own pieces of it before that barrier.
//   WAR  stage s + 3 overwrites the slot of stage s - 1, whose fragment reads were issued in step s - 2 and consumed by the
//        MFMAs of step s - 1 -- issued by every wave before it reached barrier #s.
// The epilogue's stores sit in the in-order vmcnt queue behind the DMA groups the next wait needs: the first wait after an
// epilogue is vmcnt(0) (once per tile).  The step body exists twice (even / odd register set).
// ---------------------------------------------------------------------------------------------------------
// MFMAs of the four-wave kernel as inline assembly with the accumulator pinned to the AGPR file ("a" constraint): left to
// hipcc, a 256-register accumulator block next to 128 fragment registers ends up shuttled between VGPRs, AGPRs and scratch
// inside the K-loop (the first build of this kernel: 577 - 790 spilled registers, v_accvgpr_write ahead of every MFMA).
// The statements are volatile, so their order -- and their position relative to the DMA issue and the waits -- is the
// source order; the compiler still places the (non-volatile) LDS fragment reads among them.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma_acc(f32x4& c, const bf16x8& x, const bf16x8& y) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(__builtin_bit_cast(u32x4_t, x)), "v"(__builtin_bit_cast(u32x4_t, y)));
}
// first K-step of a tile: C = 0 -- a fresh definition of the accumulator (the kernel peels that step out of the K-loop, so
// the value never has to be merged with the previous tile's)
__device__ __forceinline__ void mfma_first(f32x4& c, const bf16x8& x, const bf16x8& y) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(__builtin_bit_cast(u32x4_t, x)), "v"(__builtin_bit_cast(u32x4_t, y)));
}

// LDS fragment reads of the four-wave kernel as volatile inline assembly: they stay exactly where the source puts them (between
// the MFMAs) and the compiler's wait-count pass does not see them -- the kernel counts lgkmcnt by hand.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <int OFF> __device__ __forceinline__ void lds_read_b128(bf16x8& dst, unsigned addr) {
    u32x4_t r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    dst = __builtin_bit_cast(bf16x8, r);
}
// one 16 x 32 fragment of a k-strided image: two transposing reads (k-rows 0-3 | 8-11 ... and 4-7 | 12-15 ...: HALF bytes apart)
template <int HALF> __device__ __forceinline__ void lds_read_tr_frag(bf16x8& dst, unsigned addr) {
    u32x2_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(HALF));
    dst = __builtin_bit_cast(bf16x8, (u32x4_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3));
}
template <int N> __device__ __forceinline__ void wait_lgkmcnt() {
    static_assert(N >= 0 && N <= 15, "lgkmcnt immediate");
    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory");
}

template <bool A_KC, bool SPLIT>
__global__ __launch_bounds__(256) void gemm_bf16_w4_kernel(const GemmParams p) {
    constexpr int BM = 256, BN = 256, BK = kPpBK, NS = 4, D = 3;
    constexpr int WTM = 128, WTN = 128, TM = WTM / 16, TN = WTN / 16;
    constexpr int kAElems = BM * BK, kBElems = BK * BN, kStageElems = kAElems + kBElems;
    constexpr int APW = BM / 64, BPW = BN / 64, PW = APW + BPW;          // 1-KiB DMA pieces per wave and stage
    __shared__ __attribute__((aligned(1024))) __bf16 smem[NS * kStageElems];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int per_group = p.tiles_m * p.tiles_n;
    const int ntiles = per_group * p.ngroups;
    const int G = (int)gridDim.x;
    int bid = (int)blockIdx.x, slice = (int)blockIdx.y;
    if (SPLIT && p.xcd_slices) {                                   // (see gemm_bf16_pp_kernel)
        const int S = (int)gridDim.y, lin = (int)blockIdx.y * G + (int)blockIdx.x, xcd = lin & 7, q = lin >> 3;
        if (S >= 8) { const int m_ = S >> 3; slice = xcd * m_ + q % m_; bid = q / m_; }
        else { const int d_ = 8 / S; slice = xcd % S; bid = q * d_ + xcd / S; }
    }
    const int my_tiles = (ntiles - bid + G - 1) / G;
    const int kbeg = slice * p.k_chunk;
    const int kend = min(p.K, kbeg + p.k_chunk);
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int ktail = (kend - kbeg) - (nk - 1) * BK;
    const bool has_tail = ktail < BK;
    const int total = my_tiles * nk;
    if (total <= 0) return;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_void_t*)smem;

    // ---- DMA side: the eight-wave kernel's images, four 1-KiB pieces per operand, wave and stage, all eight issued right
    // behind the barrier.  MEASURED (profiles/r03/gemm_w4_lab.txt, in-kernel stamps): an LDS-DMA piece costs the issuing wave
    // 60 - 130 cycles (MI355X_MICROARCH.md: ~60 among bare MFMAs) during which it issues no MFMA -- 8 pieces = ~580 of a
    // ~2500-cycle K-step with ONE wave per SIMD; spreading the pieces one by one between the MFMA rows made every piece cost
    // more (half of a K-step 560 -> 1070 cycles: 548 against 646 TFLOP/s on 20800 x 2000 x 1200).  The eight-wave kernel pays
    // the same issue cycles but its second wave per SIMD multiplies meanwhile: that is why this kernel ends 8 - 10 % BEHIND it
    // (646 against 713 TFLOP/s) although its K-loop issues 40 % fewer LDS reads -- kept for ADN_GEMM_PP=7 and as the record
    // of that measurement, not selected by default.  The two operands keep their own (tile, K-step, ring slot) cursors.
    constexpr int A_CPR = BM / 8, B_CPR = BN / 8;
    const char* baseA = nullptr; const char* baseB = nullptr;
    unsigned offA[APW], offB[BPW];
    int a_aux[APW], b_row[BPW];
#pragma unroll
    for (int t = 0; t < APW; ++t)
        a_aux[t] = A_KC ? ((lane & 3) ^ swz_f((lane >> 4) & 3)) * 8 : (64 / A_CPR) * (wave * APW + t) + lane / A_CPR;
#pragma unroll
    for (int t = 0; t < BPW; ++t) b_row[t] = (64 / B_CPR) * (wave * BPW + t) + lane / B_CPR;
    auto tile_of = [&](int ord, int& grp, int& tm, int& tn) {
        const int q = xcd_tile(bid + ord * G, ntiles);
        grp = q / per_group;
        tile_coords(p, q - grp * per_group, tm, tn);
    };
    auto setup_a = [&](int ord) {
        int grp, tm, tn;
        tile_of(ord, grp, tm, tn);
        const char* A16 = reinterpret_cast<const char*>(pick_group(p, grp).A16);
        const int m0 = tm * BM;
        baseA = A_KC ? A16 + ((size_t)m0 * p.lda + kbeg) * 2 : A16 + (size_t)kbeg * p.lda * 2;
#pragma unroll
        for (int t = 0; t < APW; ++t) {
            if (A_KC) {
                const int row = min(m0 + 16 * (wave * APW + t) + (lane >> 2), p.M - 1) - m0;
                offA[t] = (unsigned)(row * p.lda + a_aux[t]) * 2u;
            } else {
                int col = m0 + (((lane % A_CPR) ^ (swz_g(a_aux[t]) << 1)) * 8);
                if (col + 8 > p.lda) col = 0;
                offA[t] = (unsigned)(a_aux[t] * p.lda + col) * 2u;
            }
        }
    };
    auto setup_b = [&](int ord) {
        int grp, tm, tn;
        tile_of(ord, grp, tm, tn);
        const char* B16 = reinterpret_cast<const char*>(pick_group(p, grp).B16);
        const int n0 = tn * BN;
        baseB = B16 + (size_t)kbeg * p.ldb * 2;
#pragma unroll
        for (int t = 0; t < BPW; ++t) {
            int col = n0 + (((lane % B_CPR) ^ (swz_g(b_row[t]) << 1)) * 8);
            if (col + 8 > p.ldb) col = 0;
            offB[t] = (unsigned)(b_row[t] * p.ldb + col) * 2u;
        }
    };
    const size_t a_step = A_KC ? (size_t)BK * 2 : (size_t)BK * p.lda * 2;
    const size_t b_step = (size_t)BK * p.ldb * 2;
    const unsigned dstA_w = __builtin_amdgcn_readfirstlane(lds_base + wave * APW * 1024);
    const unsigned dstB_w = __builtin_amdgcn_readfirstlane(lds_base + NS * kAElems * 2 + wave * BPW * 1024);
    int ia_k = 0, ia_ord = 0, ia_slot = 0, ib_k = 0, ib_ord = 0, ib_slot = 0;       // next stage to issue, per operand
    auto issue_a = [&](auto t_c) __attribute__((always_inline)) {                   // piece t of the A part; t == APW - 1 advances
        constexpr int t = decltype(t_c)::value;
        const unsigned dA = __builtin_amdgcn_readfirstlane(dstA_w + (unsigned)(ia_slot * kAElems * 2) + t * 1024);
        if (has_tail && ia_k == nk - 1) {
            const int k0 = kbeg + ia_k * BK;
            const char* g = baseA + offA[t];
            if (A_KC) { if (k0 + a_aux[t] >= kend) g -= (size_t)(k0 + a_aux[t] - (kend - 8)) * 2; }
            else if (k0 + a_aux[t] >= kend) g -= (size_t)(k0 + a_aux[t] - (kend - 1)) * p.lda * 2;
            glds16(g, dA);
        } else {
            glds16_su<0>(offA[t], uniform_ptr(baseA), dA);
        }
        if (t == APW - 1) {
            baseA += a_step;
            if (++ia_slot == NS) ia_slot = 0;
            if (++ia_k == nk) { ia_k = 0; if (++ia_ord < my_tiles) setup_a(ia_ord); }
        }
    };
    auto issue_b = [&](auto t_c) __attribute__((always_inline)) {
        constexpr int t = decltype(t_c)::value;
        const unsigned dB = __builtin_amdgcn_readfirstlane(dstB_w + (unsigned)(ib_slot * kBElems * 2) + t * 1024);
        if (has_tail && ib_k == nk - 1) {
            const int k0 = kbeg + ib_k * BK;
            const char* g = baseB + offB[t];
            if (k0 + b_row[t] >= kend) g -= (size_t)(k0 + b_row[t] - (kend - 1)) * p.ldb * 2;
            glds16(g, dB);
        } else {
            glds16_su<0>(offB[t], uniform_ptr(baseB), dB);
        }
        if (t == BPW - 1) {
            baseB += b_step;
            if (++ib_slot == NS) ib_slot = 0;
            if (++ib_k == nk) { ib_k = 0; if (++ib_ord < my_tiles) setup_b(ib_ord); }
        }
    };
    auto issue_stage = [&]() __attribute__((always_inline)) {       // a whole stage at once
        issue_a(std::integral_constant<int, 0>{}); issue_a(std::integral_constant<int, 1>{});
        issue_a(std::integral_constant<int, 2>{}); issue_a(std::integral_constant<int, 3>{});
        issue_b(std::integral_constant<int, 0>{}); issue_b(std::integral_constant<int, 1>{});
        issue_b(std::integral_constant<int, 2>{}); issue_b(std::integral_constant<int, 3>{});
    };

    // ---- per-lane fragment addresses inside ring slot 0 ---------------------------------------------------
    const int q = (lane & 15) >> 2, pp = lane & 3, hi = lane >> 4;
    const int g_lane = q | ((hi & 1) << 2);
    unsigned a_off[TM], b_off[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        if (A_KC) a_off[a] = lds_base + (wm * WTM + a * 16 + (lane & 15)) * 64 + ((hi ^ swz_f(q)) << 4);
        else a_off[a] = lds_base + (hi * 8 + q) * (BM * 2) + ((((wm * WTM + a * 16) / 8 + (pp >> 1)) ^ (g_lane << 1)) << 4) + (pp & 1) * 8;
    }
#pragma unroll
    for (int b = 0; b < TN; ++b)
        b_off[b] = lds_base + NS * kAElems * 2 + (hi * 8 + q) * (BN * 2) + ((((wn * WTN + b * 16) / 8 + (pp >> 1)) ^ (g_lane << 1)) << 4) + (pp & 1) * 8;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 amask;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        amask[j] = ((8 * hi + 2 * j < ktail) ? 0x0000FFFFu : 0u) | ((8 * hi + 2 * j + 1 < ktail) ? 0xFFFF0000u : 0u);

    f32x4 acc[TM][TN];                                             // AGPRs; written by the asm MFMAs only (first K-step: C = 0)

    // own pieces of stage s + 1 landed?  (called between the halves of step s, BEFORE the A pieces of stage s + 3 go out: the
    // youngest stage this wave has issued is s + 2 -- its A part in the second half of step s - 1, its B part just now)
    auto wait_next = [&](int s, bool fresh_epi) {
        if (s + 1 >= total) return;
        if (fresh_epi || s + 2 >= total) wait_vmcnt<0>();
        else wait_vmcnt<PW>();
    };
    // ---- fragments: ONE register set (64 VGPRs), refilled for stage s + 1 as the MFMAs of stage s retire their last use.
    // The 8 x 8 MFMA grid of a K-step runs as two halves: columns 0-3 (all rows), then columns 4-7 (all rows).
    //   fb[0..3] are free after the first half   -> refilled at the start of the second half
    //   fa[a]    is free after row a of the second half -> refilled there
    //   fb[4..7] are free after the last row's last four MFMAs -> refilled there; first used 32 MFMAs into the next K-step
    // so every read has >= ~450 cycles before its first use.  The barrier of a K-step (and its DMA issue) sits between the
    // halves: stage s + 1 is only read behind it, stage s + 3 only written behind it.
    bf16x8 fa[TM], fb[TN];
    unsigned a_cur = 0, b_cur = 0;                                 // byte offsets of the ring slot the next refills read
    auto refill_a = [&](auto a_c) __attribute__((always_inline)) {
        constexpr int a = decltype(a_c)::value;
        if (A_KC) lds_read_b128<0>(fa[a], a_off[a] + a_cur);
        else lds_read_tr_frag<4 * (BM * 2)>(fa[a], a_off[a] + a_cur);
    };
    auto refill_b = [&](auto b_c) __attribute__((always_inline)) {
        constexpr int b = decltype(b_c)::value;
        lds_read_tr_frag<4 * (BN * 2)>(fb[b], b_off[b] + b_cur);
    };
#define W4_EACH8(F) F(std::integral_constant<int, 0>{}); F(std::integral_constant<int, 1>{}); F(std::integral_constant<int, 2>{}); \
    F(std::integral_constant<int, 3>{}); F(std::integral_constant<int, 4>{}); F(std::integral_constant<int, 5>{}); \
    F(std::integral_constant<int, 6>{}); F(std::integral_constant<int, 7>{});

    // ---- prologue: D stages in flight, stage 0 landed for everyone, its fragments in registers -------------
    static_assert(APW == 4 && BPW == 4, "piece schedule below");
    setup_a(0); setup_b(0);
    for (int s = 0; s < D && s < total; ++s) issue_stage();
    if (total >= 3) wait_vmcnt<2 * PW>(); else if (total == 2) wait_vmcnt<PW>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    W4_EACH8(refill_a)
    W4_EACH8(refill_b)
    wait_lgkmcnt<0>();

    int rd_slot = 1, s = 0;
    bool fresh = false;
#ifdef ADN_GEMM_STAMPS
    // [0] top wait, [1] first half (32 MFMAs), [2] vmcnt wait, [3] barrier, [4] DMA issue + B refills, [5] second half, [6] epilogue
    const bool late = false;
    const bool stamping = blockIdx.x == 3 && blockIdx.y == 0 && wave == 0;
    unsigned long long gacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long gl_ = __builtin_amdgcn_s_memtime();
#endif
    // one K-step; FIRST: the first of its tile (accumulators defined by C = 0 MFMAs); last: the tile's last K-step
    auto step = [&](auto first_c, bool last) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value;
        auto mm = [&](auto a_c, auto b_c) __attribute__((always_inline)) {
            constexpr int a = decltype(a_c)::value, b = decltype(b_c)::value;
            if (FIRST) mfma_first(acc[a][b], fb[b], fa[a]);
            else mfma_acc(acc[a][b], fb[b], fa[a]);
        };
        auto row_left = [&](auto a_c) __attribute__((always_inline)) {
            mm(a_c, std::integral_constant<int, 0>{}); mm(a_c, std::integral_constant<int, 1>{});
            mm(a_c, std::integral_constant<int, 2>{}); mm(a_c, std::integral_constant<int, 3>{});
        };
        if (has_tail && last) {
            wait_lgkmcnt<0>();
            asm volatile("" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7]));
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, fa[a]) & amask);
            asm volatile("s_nop 3" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7]));
        } else {
            wait_lgkmcnt<15>();                                       // fb[0..3] and fa[0]: the oldest reads of the previous step
        }
        GSTAMP(0);
        // ---- first half: columns 0-3.  Row a needs fa[a]: the waits count what may still be in flight behind its refill (A_KC: 6 - a refills of later rows
        // + the 9 youngest reads; k-strided A: two reads per refill, capped by the 4-bit counter)
        row_left(std::integral_constant<int, 0>{});
        wait_lgkmcnt<A_KC ? 14 : 15>(); row_left(std::integral_constant<int, 1>{});
        wait_lgkmcnt<A_KC ? 13 : 15>(); row_left(std::integral_constant<int, 2>{});
        wait_lgkmcnt<A_KC ? 12 : 15>(); row_left(std::integral_constant<int, 3>{});
        wait_lgkmcnt<A_KC ? 11 : 14>(); row_left(std::integral_constant<int, 4>{});
        wait_lgkmcnt<A_KC ? 10 : 12>(); row_left(std::integral_constant<int, 5>{});
        wait_lgkmcnt<A_KC ? 9 : 10>(); row_left(std::integral_constant<int, 6>{});
        wait_lgkmcnt<0>();                                           // fa[7], fb[4..7] (no read has been issued since the step began)
        row_left(std::integral_constant<int, 7>{});
        GSTAMP(1);
        // ---- between the halves: stage s + 1 landed for everyone; the slot of stage s - 1 is free
        wait_next(s, fresh);
        fresh = false;
        GSTAMP(2);
        __builtin_amdgcn_s_barrier();
        GSTAMP(3);
        a_cur = rd_slot * (kAElems * 2); b_cur = rd_slot * (kBElems * 2);
        if (++rd_slot == NS) rd_slot = 0;
        if (s + D < total) issue_stage();                            // all eight pieces of stage s + 3 (see the DMA notes above)
        refill_b(std::integral_constant<int, 0>{}); refill_b(std::integral_constant<int, 1>{});
        refill_b(std::integral_constant<int, 2>{}); refill_b(std::integral_constant<int, 3>{});
        GSTAMP(4);
        // ---- second half: columns 4-7; fa[a] is refilled behind its row
#define W4_ROW_RIGHT(A) mm(std::integral_constant<int, A>{}, std::integral_constant<int, 4>{}); mm(std::integral_constant<int, A>{}, std::integral_constant<int, 5>{}); \
        mm(std::integral_constant<int, A>{}, std::integral_constant<int, 6>{}); mm(std::integral_constant<int, A>{}, std::integral_constant<int, 7>{}); \
        refill_a(std::integral_constant<int, A>{});
        W4_ROW_RIGHT(0) W4_ROW_RIGHT(1) W4_ROW_RIGHT(2) W4_ROW_RIGHT(3) W4_ROW_RIGHT(4) W4_ROW_RIGHT(5) W4_ROW_RIGHT(6)
        mm(std::integral_constant<int, 7>{}, std::integral_constant<int, 4>{}); refill_b(std::integral_constant<int, 4>{});
        mm(std::integral_constant<int, 7>{}, std::integral_constant<int, 5>{}); refill_b(std::integral_constant<int, 5>{});
        mm(std::integral_constant<int, 7>{}, std::integral_constant<int, 6>{}); refill_b(std::integral_constant<int, 6>{});
        mm(std::integral_constant<int, 7>{}, std::integral_constant<int, 7>{}); refill_b(std::integral_constant<int, 7>{});
        refill_a(std::integral_constant<int, 7>{});
#undef W4_ROW_RIGHT
        GSTAMP(5);
        ++s;
    };
    for (int ord = 0; ord < my_tiles; ++ord) {
        int grp, tile_m, tile_n;
        tile_of(ord, grp, tile_m, tile_n);
        step(std::true_type{}, nk == 1);
        for (int kt = 1; kt < nk; ++kt) step(std::false_type{}, kt == nk - 1);
        wait_lgkmcnt<0>();                                           // no fragment register is still being written when the
                                                                     // compiler's epilogue code may touch the register file
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");             // MFMA results -> accumulator reads of the epilogue (the hazard
                                                                     // recogniser does not look inside inline assembly)
        {
            // ---- epilogue of tile `ord` from the (transposed) accumulators: lane = row (lane & 15), columns 4 hi .. 4 hi + 3
            //      of every 16-column block (the eight-wave kernel's epilogue on an 8 x 8 grid of blocks)
            const GemmGroup gp = pick_group(p, grp);
            float* Cg = SPLIT ? p.partial + ((size_t)(grp * (int)gridDim.y + slice) * p.M) * p.ldc : gp.C;
            const int row0 = tile_m * BM + wm * WTM + (lane & 15), col0 = tile_n * BN + wn * WTN + 4 * hi;
            constexpr int TNH = 2;
#pragma clang loop unroll(full)
            for (int half = 0; half < TN / TNH; ++half) {
                bf16x4 yv[TM][TNH];
                if (!SPLIT && gp.Y16) {
#pragma clang loop unroll(full)
                    for (int bb = 0; bb < TNH; ++bb)
#pragma clang loop unroll(full)
                        for (int a = 0; a < TM; ++a) {
                            const int row = min(row0 + a * 16, p.M - 1), col = min(col0 + (half * TNH + bb) * 16, p.N - 4);
                            yv[a][bb] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(gp.Y16) + (size_t)row * p.ldy + col);
                        }
                }
                const int b0 = half * TNH;
                const int colA = col0 + b0 * 16, colB = colA + 16;
                float4 biasA = make_float4(0.f, 0.f, 0.f, 0.f), biasB = biasA, csumA = biasA, csumB = biasA;
                if (!SPLIT && gp.bias) {
                    biasA = *reinterpret_cast<const float4*>(gp.bias + min(colA, p.N - 4));
                    biasB = *reinterpret_cast<const float4*>(gp.bias + min(colB, p.N - 4));
                }
                const bool odd = hi & 1;
#pragma clang loop unroll(full)
                for (int a = 0; a < TM; ++a) {
                    __builtin_amdgcn_sched_barrier(0);             // (keeps the scheduler from pulling all 256 accumulators into VGPRs)
                    const int row = row0 + a * 16;
                    const bool rok = row < p.M;
                    const float4 vA = pp_epi4<SPLIT>(p, gp, Cg, acc[a][b0], biasA, yv[a][0], row, colA, rok && colA < p.N, csumA);
                    const float4 vB = pp_epi4<SPLIT>(p, gp, Cg, acc[a][b0 + 1], biasB, yv[a][1], row, colB, rok && colB < p.N, csumB);
                    if (!SPLIT && gp.C16) {
                        const uint2 pA = __builtin_bit_cast(uint2, cvt4(vA)), pB = __builtin_bit_cast(uint2, cvt4(vB));
                        const uint2 give = odd ? pA : pB;
                        uint2 take;
                        take.x = __shfl_xor(give.x, 16, 64); take.y = __shfl_xor(give.y, 16, 64);
                        const uint4 out = odd ? make_uint4(take.x, take.y, pB.x, pB.y) : make_uint4(pA.x, pA.y, take.x, take.y);
                        const int cs = odd ? colB - 4 : colA;
                        __bf16* dst = reinterpret_cast<__bf16*>(gp.C16) + (size_t)row * p.ldc + cs;
                        if (rok && cs + 8 <= p.N) *reinterpret_cast<uint4*>(dst) = out;
                        else if (rok && cs + 4 <= p.N) *reinterpret_cast<uint2*>(dst) = make_uint2(out.x, out.y);
                    }
                }
                if (!SPLIT && gp.colsum) {                         // column sums over this wave's WTM rows: [tiles_m * 2][colsum_ld]
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) {
                        csumA.x += __shfl_xor(csumA.x, o, 64); csumA.y += __shfl_xor(csumA.y, o, 64);
                        csumA.z += __shfl_xor(csumA.z, o, 64); csumA.w += __shfl_xor(csumA.w, o, 64);
                        csumB.x += __shfl_xor(csumB.x, o, 64); csumB.y += __shfl_xor(csumB.y, o, 64);
                        csumB.z += __shfl_xor(csumB.z, o, 64); csumB.w += __shfl_xor(csumB.w, o, 64);
                    }
                    if ((lane & 15) == 0) {
                        float* cs_row = gp.colsum + (size_t)(tile_m * 2 + wm) * p.colsum_ld;
                        if (colA < p.N) *reinterpret_cast<float4*>(cs_row + colA) = csumA;
                        if (colB < p.N) *reinterpret_cast<float4*>(cs_row + colB) = csumB;
                    }
                }
            }
        }
        fresh = true;
        GSTAMP(6);
    }
    GSTAMP_FLUSH;
#undef W4_EACH8
}

// C (+)= sum of the split-K partial slabs ([group][split][M][ldc] floats); float4 per lane
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p, int splits) {
    const size_t slab4 = (size_t)p.M * p.ldc / 4;                 // float4 per slab; pad columns (>= N) are never touched
    const int n4 = p.N / 4, ld4 = p.ldc / 4;
    const size_t work = (size_t)p.M * n4;
    for (int g = 0; g < p.ngroups; ++g) {
        const float4* part = reinterpret_cast<const float4*>(p.partial) + (size_t)g * splits * slab4;
        float4* C = reinterpret_cast<float4*>(pick_group(p, g).C);
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < work; e += (size_t)gridDim.x * 256) {
            const size_t i = (e / n4) * ld4 + (e % n4);
            float4 v = p.accumulate ? C[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            for (int s = 0; s < splits; ++s) {
                const float4 w = part[(size_t)s * slab4 + i];
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            C[i] = v;
        }
    }
}

// The same for MANY slabs over a small result (round 6): the weight gradients of the LSTMs -- 250 x 1000 and 150 x 1000 outputs cut
// into 21 - 31 K-slices so that their 8 - 12 tiles fill the device -- are 62 500 float4 per problem: one element per thread with
// `splits` loads one behind the other is a chain of round trips (27 us for 63 MB, 2.5 TB/s).  Here a workgroup takes 64 elements
// and its four thread rows take the slabs s = q, q + 4, ... each, four loads in flight per thread; the four partial sums meet in
// LDS and are added in row order on top of C: one fixed order (row q's slabs ascending, then rows 0..3), no atomics.
// blockIdx.y = problem.
__global__ __launch_bounds__(256) void splitk_reduce_wide_kernel(const GemmParams p, int splits) {
    __shared__ float4 part4[3][64];
    const size_t slab4 = (size_t)p.M * p.ldc / 4;
    const int n4 = p.N / 4, ld4 = p.ldc / 4;
    const size_t work = (size_t)p.M * n4;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const float4* part = reinterpret_cast<const float4*>(p.partial) + (size_t)blockIdx.y * splits * slab4;
    float4* C = reinterpret_cast<float4*>(pick_group(p, blockIdx.y).C);
    for (size_t e0 = (size_t)blockIdx.x * 64; e0 < work; e0 += (size_t)gridDim.x * 64) {
        const size_t e = e0 + lane;
        const bool ok = e < work;
        const size_t i = ok ? (e / n4) * ld4 + (e % n4) : 0;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            int s = q;
            for (; s + 12 < splits; s += 16) {                    // four slabs of this row in flight
                const float4 a = part[(size_t)s * slab4 + i], b = part[(size_t)(s + 4) * slab4 + i];
                const float4 c = part[(size_t)(s + 8) * slab4 + i], d = part[(size_t)(s + 12) * slab4 + i];
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w;
                v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
            }
            for (; s < splits; s += 4) {
                const float4 a = part[(size_t)s * slab4 + i];
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            }
        }
        if (q > 0) part4[q - 1][lane] = v;
        __syncthreads();
        if (q == 0 && ok) {
            float4 r = p.accumulate ? C[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
#pragma unroll
            for (int k = 0; k < 3; ++k) { const float4 w = part4[k][lane]; r.x += w.x; r.y += w.y; r.z += w.z; r.w += w.w; }
            C[i] = r;
        }
        __syncthreads();
    }
}

// The tail band of a persistent NN launch (gemm_f32.hip gemm_pp_try_impl: the row tiles whose tiles would open a nearly empty last
// round of the 256 workgroups run K-split over the whole device instead): sums the band's partial slabs ([group][split][Mt][ldc],
// Mt = p.M band rows, slice order: one fixed order) and applies the epilogue the persistent kernel would have -- bias, rectifier,
// rectify'(Y) mask from the bf16 copy of Y, fp32 store, bf16 copy or hi / lo planes, column sums (one partial row per kTailEpiRows
// band rows, behind the main launch's partial rows) -- at rows m_off + r of the problem's matrices.  blockIdx.y = problem.
__global__ __launch_bounds__(256) void splitk_tail_epilogue_kernel(const GemmParams p, int splits, int m_off, int cs_row0) {
    const GemmGroup gp = pick_group(p, blockIdx.y);
    const int r0 = blockIdx.x * kTailEpiRows, r1 = min(p.M, r0 + kTailEpiRows);
    const size_t slab = (size_t)p.M * p.ldc;
    const float* part = p.partial + (size_t)blockIdx.y * splits * slab;
    const int n4 = p.N / 4;
    for (int c4 = threadIdx.x; c4 < n4; c4 += 256) {
        const int col = 4 * c4;
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gp.bias) b4 = *reinterpret_cast<const float4*>(gp.bias + col);
        for (int r = r0; r < r1; ++r) {
            float4 v = b4;
            for (int s = 0; s < splits; ++s) {
                const float4 w = *reinterpret_cast<const float4*>(part + (size_t)s * slab + (size_t)r * p.ldc + col);
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            if (p.act == ADN_ACT_RECTIFY) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            const size_t row = (size_t)m_off + r;
            if (gp.Y16) {
                const bf16x4 y = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(gp.Y16) + row * p.ldy + col);
                v.x = (float)y[0] > 0.f ? v.x : 0.f; v.y = (float)y[1] > 0.f ? v.y : 0.f;
                v.z = (float)y[2] > 0.f ? v.z : 0.f; v.w = (float)y[3] > 0.f ? v.w : 0.f;
            }
            const size_t off = row * p.ldc + col;
            if (gp.C) *reinterpret_cast<float4*>(gp.C + off) = v;
            if (gp.C16) {
                const bf16x4 h = cvt4(v);
                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(gp.C16) + off) = h;
                if (gp.C16lo)
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(gp.C16lo) + off) =
                        cvt4(make_float4(v.x - (float)h[0], v.y - (float)h[1], v.z - (float)h[2], v.w - (float)h[3]));
            }
            cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
        }
        if (gp.colsum) *reinterpret_cast<float4*>(gp.colsum + (size_t)(cs_row0 + blockIdx.x) * p.colsum_ld + col) = cs;
    }
}

void launch_splitk_tail_epilogue(const GemmParams& p, int splits, int m_off, int cs_row0, hipStream_t s) {
    hipLaunchKernelGGL(splitk_tail_epilogue_kernel, dim3((unsigned)cdiv(p.M, kTailEpiRows), (unsigned)p.ngroups), dim3(256), 0, s, p, splits, m_off, cs_row0);
}

void launch_splitk_reduce(const GemmParams& p, int splits, hipStream_t s) {
    const size_t n4 = (size_t)p.M * (p.N / 4);
    static const bool no_wide = getenv("ADN_GEMM_NO_WIDE_REDUCE") != nullptr;      // (A/B)
    if (splits >= 8 && !no_wide) {                     // many slabs: the slabs of an element spread over four thread rows
        const int blocks = (int)std::min<size_t>(4096, (n4 + 63) / 64);
        hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3((unsigned)blocks, (unsigned)p.ngroups), dim3(256), 0, s, p, splits);
        return;
    }
    const int blocks = (int)std::min<size_t>(2048, (n4 + 255) / 256);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p, splits);
}

template <int BM, int BN, bool PLANES>
static void launch_pp_t(const GemmParams& p, int layout, bool split, dim3 grid, hipStream_t s) {
    if (layout == GEMM_NN) {
        if (split) hipLaunchKernelGGL((gemm_bf16_pp_kernel<BM, BN, true, true, PLANES>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((gemm_bf16_pp_kernel<BM, BN, true, false, PLANES>), grid, dim3(512), 0, s, p);
    } else {
        if (split) hipLaunchKernelGGL((gemm_bf16_pp_kernel<BM, BN, false, true, PLANES>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((gemm_bf16_pp_kernel<BM, BN, false, false, PLANES>), grid, dim3(512), 0, s, p);
    }
}

// tile_mode: 4 = 256 x 256, 5 = 256 x 128, 6 = 128 x 256 (eight waves); 7 = 256 x 256, four waves of 128 x 128
void launch_gemm_bf16_pp(const GemmParams& p, int layout, int tile_mode, int splits, dim3 grid, hipStream_t s, bool reduce) {
    const bool split = splits > 1;
    if (tile_mode == 7) {
        if (layout == GEMM_NN) {
            if (split) hipLaunchKernelGGL((gemm_bf16_w4_kernel<true, true>), grid, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((gemm_bf16_w4_kernel<true, false>), grid, dim3(256), 0, s, p);
        } else {
            if (split) hipLaunchKernelGGL((gemm_bf16_w4_kernel<false, true>), grid, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((gemm_bf16_w4_kernel<false, false>), grid, dim3(256), 0, s, p);
        }
    }
#ifndef ADN_W4_ONLY
    else if (tile_mode == 4) { if (p.kseg) launch_pp_t<256, 256, true>(p, layout, split, grid, s); else launch_pp_t<256, 256, false>(p, layout, split, grid, s); }
    else if (tile_mode == 5) launch_pp_t<256, 128, true>(p, layout, split, grid, s);      // (diagnostic tile shapes: the general form only)
    else launch_pp_t<128, 256, true>(p, layout, split, grid, s);
#endif
    if (split && reduce) launch_splitk_reduce(p, splits, s);
}

template <int BM, int BN, int WM, typename T>
static void launch_bf16_t(const GemmParams& p, int layout, dim3 grid, hipStream_t s) {
    const dim3 block(WM * 128);
    switch (layout) {
        case GEMM_NN: hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, true, false, T>), grid, block, 0, s, p); break;
        case GEMM_NT: hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, true, true, T>), grid, block, 0, s, p); break;
        default:      hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, false, false, T>), grid, block, 0, s, p); break;
    }
}

// ---------------------------------------------------------------------------------------------------------
// A-stationary NT product for a SHORT K under very many rows and a wide output:  C16 [M][N] (bf16 only) = A16 [M][K] . B16 [N][K]^T,
// K <= 256.  The conv auto-encoder's patch-gradient matrices (129024 x 2500 x 152 at batch 1024: dY W^T, 645 MB of output) are this
// shape.  With 64 x 64 tiles every workgroup reloads its A rows for each of the 40 column tiles and spends its life in the
// prologue / epilogue of a 5-step K loop (363 us, 1.8 TB/s of output).  Here a workgroup owns RB x 64 rows for ALL columns: its
// A fragments (RB row tiles x K / 32 steps) sit in registers, B streams through LDS in 64-column chunks (double-buffered,
// L2-resident: every workgroup reads the same 760 KB), and each chunk is multiplied as C^T = B A^T so that a lane ends up with
// 4 consecutive COLUMNS of one row: an 8-byte store, 32 contiguous bytes per row and tile, no LDS bounce.
// ---------------------------------------------------------------------------------------------------------
constexpr int kAsCols = 64;                      // columns per chunk
template <int KS, int RB>                        // k-steps (K <= 32 KS), 16-row tiles per wave
__global__ __launch_bounds__(256) void gemm_bf16_nt_astat_kernel(const __bf16* __restrict__ A, int lda, const __bf16* __restrict__ B, int ldb,
                                                                 __bf16* __restrict__ C, int ldc, int M, int N, int K) {
    constexpr int LS = 32 * KS + 8;               // LDS row stride of a B chunk (bf16): [64 columns][LS]
    static_assert(2 * kAsCols * LS * 2 <= 160 * 1024, "gemm_bf16_nt_astat_kernel: the two B chunks need gfx950's 160 KB of LDS (KS = 8: 66 KB)");
    __shared__ __attribute__((aligned(16))) __bf16 bs[2][kAsCols][LS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int row0 = (blockIdx.x * 4 + wave) * 16 * RB;
    // ---- this wave's A fragments (as the B operand of C^T = B A^T: lane (m = i, k = 32 s + 8 kq ..+7)); zero beyond M / K
    bf16x8 af[RB][KS];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            const int m = row0 + 16 * r + i, k = 32 * s_ + 8 * kq;
            bf16x8 v = bf16x8{};
            if (m < M && k < K) {
                v = *reinterpret_cast<const bf16x8*>(A + (size_t)m * lda + k);
                if (k + 8 > K) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (k + e >= K) v[e] = (__bf16)0.f;
                }
            }
            af[r][s_] = v;
        }
    // ---- B chunks: thread t loads 16-byte pieces (column t / CH, k 8 (t % CH)), CH pieces per column
    constexpr int CH = 4 * KS;
    constexpr int PIECES = kAsCols * CH, PER = (PIECES + 255) / 256;
    bf16x8 stage[PER];
    auto request = [&](int chunk) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = tid + 256 * q, col = chunk * kAsCols + e / CH, k = 8 * (e % CH);
            bf16x8 v = bf16x8{};
            if (e < PIECES && col < N && k < K) {
                v = *reinterpret_cast<const bf16x8*>(B + (size_t)col * ldb + k);
                if (k + 8 > K) {
#pragma unroll
                    for (int x = 0; x < 8; ++x) if (k + x >= K) v[x] = (__bf16)0.f;
                }
            }
            stage[q] = v;
        }
    };
    auto commit = [&](int par) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = tid + 256 * q;
            if (e < PIECES) *reinterpret_cast<bf16x8*>(&bs[par][e / CH][8 * (e % CH)]) = stage[q];
        }
    };
    const int chunks = (N + kAsCols - 1) / kAsCols;
    request(0);
    commit(0);
    __syncthreads();
    for (int c = 0; c < chunks; ++c) {
        const int par = c & 1;
        if (c + 1 < chunks) request(c + 1);          // the next chunk travels while this one is multiplied
        f32x4 acc[RB][4];
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            bf16x8 bf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) bf[t] = *reinterpret_cast<const bf16x8*>(&bs[par][16 * t + i][32 * s_ + 8 * kq]);
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[t], af[r][s_], acc[r][t], 0, 0, 0);
        }
        // C^T tile: lane (column of the tile = m = i, rows = n = 4 kq ..+3)  ->  C[m][16 t + 4 kq ..+3]
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int m = row0 + 16 * r + i;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int n0 = c * kAsCols + 16 * t + 4 * kq;
                if (m < M && n0 < N) {
                    bf16x4 o;
                    o[0] = (__bf16)acc[r][t][0]; o[1] = (__bf16)acc[r][t][1]; o[2] = (__bf16)acc[r][t][2]; o[3] = (__bf16)acc[r][t][3];
                    if (n0 + 4 <= N) *reinterpret_cast<bf16x4*>(C + (size_t)m * ldc + n0) = o;
                    else for (int e = 0; e < 4 && n0 + e < N; ++e) C[(size_t)m * ldc + n0 + e] = o[e];
                }
            }
        }
        if (c + 1 < chunks) commit(par ^ 1);         // (the chunk before this one was read before the barrier below, two iterations ago)
        __syncthreads();
    }
}

// may the A-stationary kernel take this product?  (bf16 operands, bf16-only output, no epilogue terms)
bool gemm_nt_astat_takes(int M, int N, int K, int lda, int ldb, int ldc, const void* A16, const void* B16, const void* C16) {
    static const bool off = getenv("ADN_GEMM_NO_ASTAT") != nullptr;      // (A/B switch)
    // (measured, conv auto-encoder at batch 1024: 129024 x 2500 x 152 363 -> 221 us, 35840 x 2500 x 152 101 -> 90; 15360 x 1368 x 200
    //  33 -> 42: with 240 workgroups each walks its 22 chunks alone on its CU, a barrier and a round trip per chunk)
    return !off && K >= 16 && K <= 256 && M >= 32768 && N >= 256 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 &&
           ((uintptr_t)A16 % 16) == 0 && ((uintptr_t)B16 % 16) == 0 && ((uintptr_t)C16 % 8) == 0;
}

template <int KS>
static void launch_astat(const void* A16, int lda, const void* B16, int ldb, void* C16, int ldc, int M, int N, int K, hipStream_t s) {
    const __bf16* A = reinterpret_cast<const __bf16*>(A16); const __bf16* B = reinterpret_cast<const __bf16*>(B16);
    __bf16* C = reinterpret_cast<__bf16*>(C16);
    // 128 rows per workgroup where that still gives every CU two workgroups, 64 otherwise
    if (cdiv(M, 128) >= 512) hipLaunchKernelGGL((gemm_bf16_nt_astat_kernel<KS, 2>), dim3(cdiv(M, 128)), dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, K);
    else hipLaunchKernelGGL((gemm_bf16_nt_astat_kernel<KS, 1>), dim3(cdiv(M, 64)), dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, K);
}

int gemm_nt_astat(const void* A16, int lda, const void* B16, int ldb, void* C16, int ldc, int M, int N, int K, hipStream_t s) {
    switch (cdiv(K, 32)) {
        case 1: launch_astat<1>(A16, lda, B16, ldb, C16, ldc, M, N, K, s); break;
        case 2: launch_astat<2>(A16, lda, B16, ldb, C16, ldc, M, N, K, s); break;
        case 3: launch_astat<3>(A16, lda, B16, ldb, C16, ldc, M, N, K, s); break;
        case 4: launch_astat<4>(A16, lda, B16, ldb, C16, ldc, M, N, K, s); break;
        case 5: launch_astat<5>(A16, lda, B16, ldb, C16, ldc, M, N, K, s); break;
        case 6: launch_astat<6>(A16, lda, B16, ldb, C16, ldc, M, N, K, s); break;
        case 7: launch_astat<7>(A16, lda, B16, ldb, C16, ldc, M, N, K, s); break;
        default: launch_astat<8>(A16, lda, B16, ldb, C16, ldc, M, N, K, s); break;
    }
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// tile_mode: 0 = 64 x 64, 1 = 128 x 128, 2 = 256 x 64 (tall: narrow outputs under many rows)
void launch_gemm_bf16(const GemmParams& p, int layout, int tile_mode, dim3 grid, hipStream_t s) {
#ifndef ADN_W4_ONLY
    const bool shadows = p.A16 && p.B16;      // operands already available as bf16 copies
    const bool big = tile_mode == 1;
    if (tile_mode == 2) {
        if (shadows) launch_bf16_t<256, 64, 4, __bf16>(p, layout, grid, s);
        else launch_bf16_t<256, 64, 4, float>(p, layout, grid, s);
        return;
    }
    if (shadows) {
        if (big) launch_bf16_t<128, 128, 2, __bf16>(p, layout, grid, s);
        else launch_bf16_t<64, 64, 2, __bf16>(p, layout, grid, s);
    } else {
        if (big) launch_bf16_t<128, 128, 2, float>(p, layout, grid, s);
        else launch_bf16_t<64, 64, 2, float>(p, layout, grid, s);
    }
#endif
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

__global__ __launch_bounds__(256) void transpose_bf16_batch_kernel(const TransposeItem* __restrict__ items, int n, int lo_part) {
    __shared__ float tile[32][33];
    int k = 0;
    while (k + 1 < n && (int)blockIdx.x >= items[k].block_end) ++k;
    const TransposeItem it = items[k];
    const int local = (int)blockIdx.x - (k ? items[k - 1].block_end : 0);
    const int ctiles = (it.cols + 31) / 32;
    const int c0 = (local % ctiles) * 32, r0 = (local / ctiles) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < it.rows && c < it.cols) ? it.W[(size_t)r * it.ld + c] : 0.f;
    }
    __syncthreads();
    __bf16* out = reinterpret_cast<__bf16*>(it.out);
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < it.cols && r < it.rows) {
            const float v = tile[tx][j];
            const __bf16 h = (__bf16)v;
            out[(size_t)c * it.ldT + r] = lo_part ? (__bf16)(v - (float)h) : h;      // (lo_part: the bf16x3 mode's second plane)
        }
    }
}

int transpose_to_bf16_batch(const TransposeItem* dev_items, int n, int total_blocks, hipStream_t s, int lo_part) {
    if (n <= 0 || total_blocks <= 0) return ADN_OK;
    hipLaunchKernelGGL(transpose_bf16_batch_kernel, dim3(total_blocks), dim3(256), 0, s, dev_items, n, lo_part);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// fp32 -> the two bf16 planes of the bf16x3 mode: hi = bf16(x), lo = bf16(x - hi); 8 elements per lane
__global__ __launch_bounds__(256) void split_hilo_kernel(const float* __restrict__ src, __bf16* __restrict__ hi, __bf16* __restrict__ lo,
                                                         size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
        bf16x4 ha, la, hb, lb;
        split4(a, ha, la); split4(b, hb, lb);
        reinterpret_cast<bf16x8*>(hi)[i] = join(ha, hb);
        reinterpret_cast<bf16x8*>(lo)[i] = join(la, lb);
    }
}

int split_hilo(const float* src, void* hi, void* lo, size_t n, hipStream_t s) {
    ADN_CHECK(n % 8 == 0, ADN_ERR_INVALID, "split_hilo: element count must be a multiple of 8");
    if (!n) return ADN_OK;
    const size_t n8 = n / 8;
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((n8 + 255) / 256, 4096));
    hipLaunchKernelGGL(split_hilo_kernel, dim3(grid), dim3(256), 0, s, src, reinterpret_cast<__bf16*>(hi), reinterpret_cast<__bf16*>(lo), n8);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// the same for up to kMaxSplitJobs tensors of ONE size in one launch (blockIdx.y = tensor): the state histories of the LSTMs of a
// launch, which become GEMM operands together -- three 7 us launches are latency, one is 9
struct SplitJobTable { const float* src[kMaxSplitJobs]; __bf16* hi[kMaxSplitJobs]; __bf16* lo[kMaxSplitJobs]; };
__global__ __launch_bounds__(256) void split_hilo_batch_kernel(const SplitJobTable t, size_t n8) {
    const float* __restrict__ src = t.src[0]; __bf16* __restrict__ hi = t.hi[0]; __bf16* __restrict__ lo = t.lo[0];
#pragma unroll
    for (int k = 1; k < kMaxSplitJobs; ++k) if ((int)blockIdx.y == k) { src = t.src[k]; hi = t.hi[k]; lo = t.lo[k]; }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
        bf16x4 ha, la, hb, lb;
        split4(a, ha, la); split4(b, hb, lb);
        reinterpret_cast<bf16x8*>(hi)[i] = join(ha, hb);
        reinterpret_cast<bf16x8*>(lo)[i] = join(la, lb);
    }
}
int split_hilo_batch(const float* const* src, void* const* hi, void* const* lo, int n, size_t count, hipStream_t s) {
    ADN_CHECK(n >= 1 && n <= kMaxSplitJobs && count % 8 == 0, ADN_ERR_INVALID, "split_hilo_batch: 1..8 tensors of a multiple of 8 elements");
    if (!count) return ADN_OK;
    if (n == 1) return split_hilo(src[0], hi[0], lo[0], count, s);
    SplitJobTable t;
    for (int k = 0; k < kMaxSplitJobs; ++k) {
        const int q = k < n ? k : 0;
        t.src[k] = src[q]; t.hi[k] = reinterpret_cast<__bf16*>(hi[q]); t.lo[k] = reinterpret_cast<__bf16*>(lo[q]);
    }
    const size_t n8 = count / 8;
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((n8 + 255) / 256, 2048));
    hipLaunchKernelGGL(split_hilo_batch_kernel, dim3(grid, n), dim3(256), 0, s, t, n8);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// ... and back: x' = hi + lo (exact in fp32; splitting x' again gives the same two planes)
__global__ __launch_bounds__(256) void join_hilo_kernel(const __bf16* __restrict__ hi, const __bf16* __restrict__ lo, float* __restrict__ dst,
                                                        size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const bf16x8 h = reinterpret_cast<const bf16x8*>(hi)[i];
        bf16x8 l;
        if (lo) l = reinterpret_cast<const bf16x8*>(lo)[i];
        else for (int j = 0; j < 8; ++j) l[j] = (__bf16)0.f;      // (a tensor that only ever had its hi plane written)
        // (hi + lo, except that hi = -0.0 stays -0.0 -- (-0) + (+0) is +0, and -0.0 is the rectifier's kink mark: adn_common.h)
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float lf = (float)l[j]; o[j] = lf == 0.f ? (float)h[j] : (float)h[j] + lf; }
        reinterpret_cast<float4*>(dst)[2 * i] = make_float4(o[0], o[1], o[2], o[3]);
        reinterpret_cast<float4*>(dst)[2 * i + 1] = make_float4(o[4], o[5], o[6], o[7]);
    }
}
int join_hilo(const void* hi, const void* lo, float* dst, size_t n, hipStream_t s) {
    ADN_CHECK(n % 8 == 0, ADN_ERR_INVALID, "join_hilo: element count must be a multiple of 8");
    if (!n) return ADN_OK;
    const size_t n8 = n / 8;
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((n8 + 255) / 256, 4096));
    hipLaunchKernelGGL(join_hilo_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const __bf16*>(hi), reinterpret_cast<const __bf16*>(lo), dst, n8);
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

// ---------------------------------------------------------------------------------------------------------
// bf16x3: fp32-grade products at the bf16 MFMA rate.  x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (|x - hi - lo| <=
// 2^-18 |x|); a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi (the dropped lo lo term is 2^-18 relative too), every product exact
// in the fp32 accumulator.  The three products are ONE GEMM over a three times deeper K: A' = [A_hi | A_hi | A_lo],
// B' = [B_hi ; B_lo ; B_hi] along k, each segment padded to Kp = round_up(K, 8) with zeros.  These kernels write the
// split images; the GEMM kernels above then run unchanged on them.
//   lo_mask: bit s set = segment s holds the lo part (A: 0b100, B: 0b010)
// ---------------------------------------------------------------------------------------------------------
// k along the COLUMNS of src [rows][ld_src]: dst [rows][3 Kp], dst[r][s Kp + k]
__global__ __launch_bounds__(256) void split3_cols_kernel(const float* __restrict__ src, int ld_src, int rows, int K, int Kp,
                                                          __bf16* __restrict__ dst, int lo_mask) {
    const int q = Kp / 4;
    const int64_t total = (int64_t)rows * q;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int k = (int)(e % q) * 4;
        const int64_t r = e / q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k + 4 <= K) v = *reinterpret_cast<const float4*>(src + (size_t)r * ld_src + k);
        else if (k < K) {
            const float* p = src + (size_t)r * ld_src + k;
            v.x = p[0]; if (k + 1 < K) v.y = p[1]; if (k + 2 < K) v.z = p[2];
        }
        bf16x4 hi, lo;
        split4(v, hi, lo);
        __bf16* d = dst + (size_t)r * 3 * Kp + k;
#pragma unroll
        for (int sgm = 0; sgm < 3; ++sgm) *reinterpret_cast<bf16x4*>(d + (size_t)sgm * Kp) = ((lo_mask >> sgm) & 1) ? lo : hi;
    }
}

// k along the ROWS of src [K][ld_src] (cols valid columns): dst [3 Kp][ld_dst], dst[s Kp + k][c]; pad rows / columns zero
__global__ __launch_bounds__(256) void split3_rows_kernel(const float* __restrict__ src, int ld_src, int K, int Kp, int cols,
                                                          __bf16* __restrict__ dst, int ld_dst, int lo_mask) {
    const int q = ld_dst / 4;
    const int64_t total = (int64_t)Kp * q;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % q) * 4;
        const int k = (int)(e / q);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K) {
            if (c + 4 <= cols) v = *reinterpret_cast<const float4*>(src + (size_t)k * ld_src + c);
            else if (c < cols) {
                const float* p = src + (size_t)k * ld_src + c;
                v.x = p[0]; if (c + 1 < cols) v.y = p[1]; if (c + 2 < cols) v.z = p[2];
            }
        }
        bf16x4 hi, lo;
        split4(v, hi, lo);
#pragma unroll
        for (int sgm = 0; sgm < 3; ++sgm)
            *reinterpret_cast<bf16x4*>(dst + ((size_t)sgm * Kp + k) * ld_dst + c) = ((lo_mask >> sgm) & 1) ? lo : hi;
    }
}

// src [R][ld_src] with k along its COLUMNS, written TRANSPOSED: dst [3 Kp][ld_dst], dst[s Kp + k][r] (an NT operand turned
// into the k-strided form the faster NN kernels read); 32 x 32 tiles through LDS, pad rows / columns zero
__global__ __launch_bounds__(256) void split3_transpose_kernel(const float* __restrict__ src, int ld_src, int R, int K, int Kp,
                                                               __bf16* __restrict__ dst, int ld_dst, int lo_mask) {
    __shared__ float tile[32][33];
    const int k0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, k = k0 + tx;
        tile[j][tx] = (r < R && k < K) ? src[(size_t)r * ld_src + k] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int k = k0 + j, r = r0 + tx;
        if (k >= Kp || r >= ld_dst) continue;
        const float v = tile[tx][j];
        const __bf16 hi = (__bf16)v, lo = (__bf16)(v - (float)hi);
#pragma unroll
        for (int sgm = 0; sgm < 3; ++sgm) dst[((size_t)sgm * Kp + k) * ld_dst + r] = ((lo_mask >> sgm) & 1) ? lo : hi;
    }
}

int split3_transpose(const float* src, int ld_src, int R, int K, int Kp, void* dst, int ld_dst, int lo_mask, hipStream_t s) {
    hipLaunchKernelGGL(split3_transpose_kernel, dim3(cdiv(Kp, 32), cdiv(ld_dst, 32)), dim3(256), 0, s, src, ld_src, R, K, Kp,
                       reinterpret_cast<__bf16*>(dst), ld_dst, lo_mask);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int split3_cols(const float* src, int ld_src, int rows, int K, int Kp, void* dst, int lo_mask, hipStream_t s) {
    const int64_t total = (int64_t)rows * (Kp / 4);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 16384));
    hipLaunchKernelGGL(split3_cols_kernel, dim3(grid), dim3(256), 0, s, src, ld_src, rows, K, Kp, reinterpret_cast<__bf16*>(dst), lo_mask);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int split3_rows(const float* src, int ld_src, int K, int Kp, int cols, void* dst, int ld_dst, int lo_mask, hipStream_t s) {
    const int64_t total = (int64_t)Kp * (ld_dst / 4);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 16384));
    hipLaunchKernelGGL(split3_rows_kernel, dim3(grid), dim3(256), 0, s, src, ld_src, K, Kp, cols, reinterpret_cast<__bf16*>(dst), ld_dst, lo_mask);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

}  // namespace adn

#ifdef ADN_GEMM_STAMPS
extern "C" int adn_debug_gemm_stamps(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(adn::g_gstamps), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(adn::g_gstamps), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif
