// HBM-bound kernels of the path: delta layer (+ layout change), fusion sums, bias/column reductions,
// classifier softmax + the reference's double-softmax temporal loss, Adam.
// Every kernel walks memory with the feature index on the lane (coalesced rows).
#include "adn_common.h"
#include <map>
#include <mutex>
#include <utility>
#include <algorithm>
#include <math.h>

namespace adn {

// =========================================================================================
// Delta layer: custom/layers.py:105-121 -> utils/signal.py:59-80.
//   d[t] = sum_{k=1..theta} (x[clamp(t+k)] - x[clamp(t-k)]) / (2k), clamp to [0, T-1] of the PADDED
//   tensor (mask-blind); out = [x | d(x) | d(d(x))].
// Input is batch-major (B,T,F) -- the order the encoder GEMMs produce -- and the output is
// time-major (T,B,3F), the order the recurrence consumes: the layout change is free here.
// One workgroup = one utterance x 32 features; the (T x 32) slab lives in LDS.
// =========================================================================================
constexpr int kDeltaFC = 32;

// The LDS columns carry theta extra rows at either end -- edge replicas in the forward kernel (the clamped reads become
// plain offsets), zeros in the backward kernel -- and TH > 0 fixes theta at compile time: the k-loop unrolls into 2 TH
// independent LDS reads with constant weights (theta = 9, the reference's window: 21 -> 10 us per launch at 520 x 40 x 50).
// 16-bit copies beside an fp32 store: the bf16 copy (bf16 mode) or the hi / lo planes (bf16x3 / mixed: lo != nullptr)
__device__ __forceinline__ void delta_put16(__bf16* __restrict__ hi, __bf16* __restrict__ lo, size_t idx, float v) {
    const __bf16 h = (__bf16)v;
    hi[idx] = h;
    if (lo) lo[idx] = (__bf16)(v - (float)h);
}

// Frame compaction (compact.hip) rides on the layout change: with j.row_map the batch-major side of the layer is the COMPACT
// matrix -- frame (b, t) lives in row row_map[b T + t] (every padding frame in the one zero-input row) -- so the forward kernel
// reads through the map (no expanded copy of the encoder output) and the backward kernel stores a valid frame's gradient at its
// compact row and adds the padding frames' gradients up: per utterance here (lane order, then the 8 row lanes in order), over the
// utterances in order by compact_pad_finish -- one fixed order, no atomics.
template <int TH>
__device__ __forceinline__ void delta_fwd_body(const DeltaJob& j, int B, int T, int theta_rt) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const float* __restrict__ in = j.src; const int ld_in = j.ld_src;
    float* __restrict__ out = j.dst; const int ld_out = j.ld_dst; const int F = j.F;
    __bf16* __restrict__ out16 = reinterpret_cast<__bf16*>(j.dst16);
    __bf16* __restrict__ out16lo = reinterpret_cast<__bf16*>(j.dst16lo);
    const int theta = TH ? TH : theta_rt;
    const int R = T + 2 * theta;
    float* xs = sm;                   // [R][32], row theta + t = frame t
    float* d1 = sm + R * kDeltaFC;    // [R][32]
    const int b = blockIdx.x, f0 = blockIdx.y * kDeltaFC;
    const int fl = threadIdx.x & 31, ts = threadIdx.x >> 5;
    const int f = f0 + fl;
    const bool fv = f < F;
    // time-major row of frame t: tm0 + t * tms for t < Tw (length buckets), the plain t B + b otherwise
    const size_t tm0 = j.tm_row0 ? (size_t)j.tm_row0[b] : (size_t)b, tms = j.tm_row0 ? (size_t)j.tm_stride : (size_t)B;
    const int Tw = j.tm_T ? j.tm_T[b] : T;
    for (int t = ts; t < T; t += 8) {
        const size_t row = j.row_map ? (size_t)j.row_map[(size_t)b * T + t] : (size_t)b * T + t;
        const float v = fv ? in[row * ld_in + f] : 0.f;
        xs[(theta + t) * kDeltaFC + fl] = v;
        if (t == 0) for (int k = 0; k < theta; ++k) xs[k * kDeltaFC + fl] = v;
        if (t == T - 1) for (int k = 1; k <= theta; ++k) xs[(theta + T - 1 + k) * kDeltaFC + fl] = v;
    }
    __syncthreads();
    if (!j.append) {
        for (int t = ts; t < Tw; t += 8)
            if (fv) {
                const float v = xs[(theta + t) * kDeltaFC + fl];
                const size_t o = (tm0 + (size_t)t * tms) * ld_out + f;
                out[o] = v;
                if (out16) delta_put16(out16, out16lo, o, v);
            }
        return;
    }
    for (int t = ts; t < T; t += 8) {
        const float* c = xs + (theta + t) * kDeltaFC + fl;
        float acc = 0.f;
#pragma unroll
        for (int k = 1; k <= (TH ? TH : theta); ++k) acc += (c[k * kDeltaFC] - c[-k * kDeltaFC]) * (0.5f / (float)k);
        d1[(theta + t) * kDeltaFC + fl] = acc;
        if (t == 0) for (int k = 0; k < theta; ++k) d1[k * kDeltaFC + fl] = acc;
        if (t == T - 1) for (int k = 1; k <= theta; ++k) d1[(theta + T - 1 + k) * kDeltaFC + fl] = acc;
    }
    __syncthreads();
    for (int t = ts; t < T; t += 8) {
        const float* c = d1 + (theta + t) * kDeltaFC + fl;
        float acc = 0.f;
#pragma unroll
        for (int k = 1; k <= (TH ? TH : theta); ++k) acc += (c[k * kDeltaFC] - c[-k * kDeltaFC]) * (0.5f / (float)k);
        if (fv && t < Tw) {
            const size_t o = (tm0 + (size_t)t * tms) * ld_out;
            const float x0 = xs[(theta + t) * kDeltaFC + fl], x1 = c[0];
            out[o + f] = x0;
            out[o + F + f] = x1;
            out[o + 2 * F + f] = acc;
            if (out16) {                                         // the 16-bit copies the projection GEMM / the folding LSTM kernel read
                delta_put16(out16, out16lo, o + f, x0);
                delta_put16(out16, out16lo, o + F + f, x1);
                delta_put16(out16, out16lo, o + 2 * F + f, acc);
            }
        }
    }
}

// (D^T g)[tau] for one feature column held in LDS (row stride kDeltaFC, zero rows beyond either end; c -> row tau).
// Interior rows: sum_k w_k (g[tau - k] - g[tau + k]); the clamped ends collect everything that was clamped onto them:
// tau = T-1 takes the suffix sums g[T-1-k .. T-1] as its plus term, tau = 0 the prefix sums g[0 .. k] as its minus term.
template <int TH>
__device__ __forceinline__ float delta_adjoint_at(const float* c, int tau, int T, int theta) {
    const bool first = tau == 0, last = tau == T - 1;
    float acc = 0.f, suf = c[0], pre = c[0];
#pragma unroll
    for (int k = 1; k <= (TH ? TH : theta); ++k) {
        const float p = c[-k * kDeltaFC], q = c[k * kDeltaFC];
        suf += p; pre += q;
        acc += ((last ? suf : p) - (first ? pre : q)) * (0.5f / (float)k);
    }
    return acc;
}

template <int TH>
__device__ __forceinline__ void delta_bwd_body(const DeltaJob& j, int B, int T, int theta_rt) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const float* __restrict__ dout = j.src; const int ld_out = j.ld_src;
    float* __restrict__ din = j.dst; const int ld_in = j.ld_dst; const int F = j.F;
    __bf16* __restrict__ din16 = reinterpret_cast<__bf16*>(j.dst16);
    __bf16* __restrict__ din16lo = reinterpret_cast<__bf16*>(j.dst16lo);
    const int theta = TH ? TH : theta_rt;
    const int R = T + 2 * theta;
    float* g2 = sm;                   // [R][32]  gradient wrt dd (row theta + t), zero rows at both ends
    float* r1 = sm + R * kDeltaFC;    // [R][32]  g1 + D^T g2, same layout
    const int b = blockIdx.x, f0 = blockIdx.y * kDeltaFC;
    const int fl = threadIdx.x & 31, ts = threadIdx.x >> 5;
    const int f = f0 + fl;
    const bool fv = f < F;
    const size_t tm0 = j.tm_row0 ? (size_t)j.tm_row0[b] : (size_t)b, tms = j.tm_row0 ? (size_t)j.tm_stride : (size_t)B;
    const int Tw = j.tm_T ? j.tm_T[b] : T;       // (frames t >= Tw have no time-major row: zero gradient)
    auto tmrow = [&](int t) { return (tm0 + (size_t)t * tms) * ld_out; };
    float pad = 0.f;                  // this lane's share of the padding frames' gradient (row_map: rows mapped to zrow)
    auto put = [&](int t, float v) {
        size_t row = (size_t)b * T + t;
        if (j.row_map) {
            const int c = j.row_map[row];
            if (c == j.zrow) { pad += v; return; }
            row = (size_t)c;
        }
        din[row * ld_in + f] = v;
        if (din16) delta_put16(din16, din16lo, row * ld_in + f, v);
    };
    if (!j.append) {
        for (int t = ts; t < T; t += 8)
            if (fv) put(t, t < Tw ? dout[tmrow(t) + f] : 0.f);
    } else {
        for (int k = ts; k < theta; k += 8) {
            g2[k * kDeltaFC + fl] = 0.f; g2[(theta + T + k) * kDeltaFC + fl] = 0.f;
            r1[k * kDeltaFC + fl] = 0.f; r1[(theta + T + k) * kDeltaFC + fl] = 0.f;
        }
        for (int t = ts; t < T; t += 8)
            g2[(theta + t) * kDeltaFC + fl] = (fv && t < Tw) ? dout[tmrow(t) + 2 * F + f] : 0.f;
        __syncthreads();
        for (int t = ts; t < T; t += 8) {
            const float g1 = (fv && t < Tw) ? dout[tmrow(t) + F + f] : 0.f;
            r1[(theta + t) * kDeltaFC + fl] = g1 + delta_adjoint_at<TH>(g2 + (theta + t) * kDeltaFC + fl, t, T, theta);
        }
        __syncthreads();
        for (int t = ts; t < T; t += 8) {
            if (fv) {
                const float g0 = t < Tw ? dout[tmrow(t) + f] : 0.f;
                put(t, g0 + delta_adjoint_at<TH>(r1 + (theta + t) * kDeltaFC + fl, t, T, theta));
            }
        }
    }
    if (j.pad_partial) {              // (uniform over the workgroup) the 8 row lanes' shares, added in lane order
        __syncthreads();              // g2 is free: its last readers are behind the barriers above (!append: never used)
        g2[ts * kDeltaFC + fl] = pad;
        __syncthreads();
        if (ts == 0 && fv) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += g2[k * kDeltaFC + fl];
            j.pad_partial[(size_t)b * F + f] = sum;
        }
    }
}

// the kernels: one tensor per launch, or up to kMaxDeltaJobs tensors of one (B, T, theta) -- the S input streams' delta layers --
// with blockIdx.z = tensor (at the reference's minibatch every one of these launches is a 6 us latency)
template <int TH>
__global__ __launch_bounds__(256) void delta_fwd_kernel(const DeltaJob j, int B, int T, int theta_rt) { delta_fwd_body<TH>(j, B, T, theta_rt); }
template <int TH>
__global__ __launch_bounds__(256) void delta_bwd_kernel(const DeltaJob j, int B, int T, int theta_rt) { delta_bwd_body<TH>(j, B, T, theta_rt); }
struct DeltaJobTable { DeltaJob j[kMaxDeltaJobs]; };
__device__ __forceinline__ DeltaJob pick_delta_job(const DeltaJobTable& t, int k) {      // (explicit selects: no scratch copy of the table)
    DeltaJob r = t.j[0];
    if (k == 1) r = t.j[1];
    if (k == 2) r = t.j[2];
    if (k == 3) r = t.j[3];
    return r;
}
template <int TH>
__global__ __launch_bounds__(256) void delta_fwd_batch_kernel(const DeltaJobTable tab, int B, int T, int theta_rt) {
    const DeltaJob j = pick_delta_job(tab, blockIdx.z);
    if ((int)blockIdx.y * kDeltaFC >= j.F) return;
    delta_fwd_body<TH>(j, B, T, theta_rt);
}
template <int TH>
__global__ __launch_bounds__(256) void delta_bwd_batch_kernel(const DeltaJobTable tab, int B, int T, int theta_rt) {
    const DeltaJob j = pick_delta_job(tab, blockIdx.z);
    if ((int)blockIdx.y * kDeltaFC >= j.F) return;
    delta_bwd_body<TH>(j, B, T, theta_rt);
}

// The column slab of one utterance lives in LDS: 2 (T + 2 theta) rows of 32 floats.  Up to 64 KiB that is a plain
// launch; beyond, the kernels' dynamic-LDS limit is raised to what the request needs (gfx950: 160 KiB per workgroup =
// 640 rows: T <= 622 at theta = 9; the reference's DeltaLayer has no limit, its longest utterances have ~ 40 frames).
constexpr size_t kDeltaMaxLds = 160 * 1024;
template <typename K>
static int delta_allow_lds(K kernel, size_t lds) {
    if (lds <= 64 * 1024) return ADN_OK;
    ADN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDeltaMaxLds));
    return ADN_OK;
}

// LDS of a delta kernel: the two column slabs; the backward kernel's padding sums reuse the first one (8 rows: there with T + 2 theta >= 8,
// asked for otherwise)
static size_t delta_lds_bytes(int T, int theta) { return (size_t)2 * std::max(T + 2 * theta, 8) * kDeltaFC * sizeof(float); }

template <bool FWD>
static int delta_one(const DeltaJob& j, int B, int T, int theta, hipStream_t s) {
    ADN_CHECK(T > 0 && B > 0 && j.F > 0 && theta >= 0, ADN_ERR_INVALID, "delta layer: empty tensor");
    const size_t lds = delta_lds_bytes(T, theta);
    ADN_CHECK(lds <= kDeltaMaxLds, ADN_ERR_INVALID, "delta layer: T + 2 theta too large (max 640 rows)");
    ProfScope prof(FWD ? PROF_DELTA_FWD : PROF_DELTA_BWD, 0.0, 4.0 * B * T * (double)j.F * (j.append ? 4.0 : 2.0), s);
    const dim3 grid(B, cdiv(j.F, kDeltaFC));
#define ADN_DELTA_ONE(K, TH) do { ADN_TRY(delta_allow_lds(&K<TH>, lds)); hipLaunchKernelGGL(K<TH>, grid, dim3(256), lds, s, j, B, T, theta); } while (0)
    // the reference's windows: 9 (video streams), 3 (OuluVS audio)
    if (FWD) { if (theta == 9) ADN_DELTA_ONE(delta_fwd_kernel, 9); else if (theta == 3) ADN_DELTA_ONE(delta_fwd_kernel, 3); else ADN_DELTA_ONE(delta_fwd_kernel, 0); }
    else { if (theta == 9) ADN_DELTA_ONE(delta_bwd_kernel, 9); else if (theta == 3) ADN_DELTA_ONE(delta_bwd_kernel, 3); else ADN_DELTA_ONE(delta_bwd_kernel, 0); }
#undef ADN_DELTA_ONE
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int delta_forward(const float* in, int ld_in, float* out, int ld_out, int B, int T, int F, int theta, int append,
                  hipStream_t s, void* out16) {
    return delta_one<true>(DeltaJob{in, ld_in, out, ld_out, F, append, out16}, B, T, theta, s);
}

int delta_backward(const float* dout, int ld_out, float* din, int ld_in, int B, int T, int F, int theta, int append,
                   hipStream_t s, void* din16) {
    return delta_one<false>(DeltaJob{dout, ld_out, din, ld_in, F, append, din16}, B, T, theta, s);
}

// the same for up to kMaxDeltaJobs tensors of one (B, T, theta) in ONE launch (jobs: src = layer input / output gradient)
template <bool FWD>
static int delta_batch(const DeltaJob* jobs, int n, int B, int T, int theta, hipStream_t s) {
    ADN_CHECK(n >= 1 && n <= kMaxDeltaJobs && T > 0 && B > 0 && theta >= 0, ADN_ERR_INVALID, "delta batch: bad arguments");
    const size_t lds = delta_lds_bytes(T, theta);
    ADN_CHECK(lds <= kDeltaMaxLds, ADN_ERR_INVALID, "delta layer: T + 2 theta too large (max 640 rows)");
    DeltaJobTable tab;
    int fmax = 0; double bytes = 0.0;
    for (int k = 0; k < n; ++k) {
        ADN_CHECK(jobs[k].F > 0, ADN_ERR_INVALID, "delta batch: empty tensor");
        tab.j[k] = jobs[k]; fmax = std::max(fmax, jobs[k].F);
        bytes += 4.0 * B * T * (double)jobs[k].F * (jobs[k].append ? 4.0 : 2.0);
    }
    for (int k = n; k < kMaxDeltaJobs; ++k) tab.j[k] = jobs[0];
    const dim3 grid(B, cdiv(fmax, kDeltaFC), n);
    ProfScope prof(FWD ? PROF_DELTA_FWD : PROF_DELTA_BWD, 0.0, bytes, s, n);
#define ADN_DELTA_LAUNCH(K, TH) do { ADN_TRY(delta_allow_lds(&K<TH>, lds)); hipLaunchKernelGGL(K<TH>, grid, dim3(256), lds, s, tab, B, T, theta); } while (0)
    if (FWD) { if (theta == 9) ADN_DELTA_LAUNCH(delta_fwd_batch_kernel, 9); else if (theta == 3) ADN_DELTA_LAUNCH(delta_fwd_batch_kernel, 3); else ADN_DELTA_LAUNCH(delta_fwd_batch_kernel, 0); }
    else { if (theta == 9) ADN_DELTA_LAUNCH(delta_bwd_batch_kernel, 9); else if (theta == 3) ADN_DELTA_LAUNCH(delta_bwd_batch_kernel, 3); else ADN_DELTA_LAUNCH(delta_bwd_batch_kernel, 0); }
#undef ADN_DELTA_LAUNCH
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}
int delta_forward_batch(const DeltaJob* jobs, int n, int B, int T, int theta, hipStream_t s) {
    if (n == 1) return delta_one<true>(jobs[0], B, T, theta, s);
    return delta_batch<true>(jobs, n, B, T, theta, s);
}
int delta_backward_batch(const DeltaJob* jobs, int n, int B, int T, int theta, hipStream_t s) {
    if (n == 1) return delta_one<false>(jobs[0], B, T, theta, s);
    return delta_batch<false>(jobs, n, B, T, theta, s);
}

// =========================================================================================
// small row-wise helpers
// =========================================================================================
struct SumKArgs {
    const float* in[ADN_MAX_STREAMS];
    const float* alpha[ADN_MAX_STREAMS];
    int n;
};

__global__ __launch_bounds__(256) void sum_k_kernel(SumKArgs a, int ld_in, float* __restrict__ out, int ld_out,
                                                    int rows, int cols4, __bf16* __restrict__ out16, __bf16* __restrict__ out16lo) {
    const int64_t total = (int64_t)rows * cols4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / cols4), c = (int)(e % cols4) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < a.n; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(a.in[k] + (size_t)r * ld_in + c);
            const float w = a.alpha[k] ? *a.alpha[k] : 1.f;
            acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
        }
        *reinterpret_cast<float4*>(out + (size_t)r * ld_out + c) = acc;
        if (out16) {                        // bf16 copy for the GEMM that reads the sum (same row stride)
            typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
            bf16x4_t h; h[0] = (__bf16)acc.x; h[1] = (__bf16)acc.y; h[2] = (__bf16)acc.z; h[3] = (__bf16)acc.w;
            *reinterpret_cast<bf16x4_t*>(out16 + (size_t)r * ld_out + c) = h;
            if (out16lo) {                  // bf16x3 / mixed: out16 is the hi plane, this the lo plane bf16(x - hi)
                bf16x4_t l; l[0] = (__bf16)(acc.x - (float)h[0]); l[1] = (__bf16)(acc.y - (float)h[1]);
                l[2] = (__bf16)(acc.z - (float)h[2]); l[3] = (__bf16)(acc.w - (float)h[3]);
                *reinterpret_cast<bf16x4_t*>(out16lo + (size_t)r * ld_out + c) = l;
            }
        }
    }
}

static inline int grid_for(int64_t work_items) {
    int64_t g = (work_items + 255) / 256;
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, 2048));
}

// NOTE: operates on whole float4 groups up to round_up(cols,4) <= ld; pad columns of the inputs are
// zero by construction, so the pad columns of the output stay zero.
int sum_k(int n_in, const float* const* in, const float* const* alpha, int ld_in, float* out, int ld_out, int rows,
          int cols, hipStream_t s, void* out16, void* out16lo) {
    ADN_CHECK(n_in >= 1 && n_in <= ADN_MAX_STREAMS && (!out16lo || out16), ADN_ERR_INVALID, "sum_k: bad operand count");
    SumKArgs a;
    a.n = n_in;
    for (int k = 0; k < n_in; ++k) { a.in[k] = in[k]; a.alpha[k] = alpha ? alpha[k] : nullptr; }
    const int cols4 = cdiv(cols, 4);
    hipLaunchKernelGGL(sum_k_kernel, dim3(grid_for((int64_t)rows * cols4)), dim3(256), 0, s, a, ld_in, out, ld_out,
                       rows, cols4, reinterpret_cast<__bf16*>(out16), reinterpret_cast<__bf16*>(out16lo));
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int scale_by(const float* in, int ld_in, const float* alpha, float* out, int ld_out, int rows, int cols,
             hipStream_t s) {
    const float* ins[1] = {in};
    const float* al[1] = {alpha};
    return sum_k(1, ins, al, ld_in, out, ld_out, rows, cols, s);
}

// column sums: grid (col tiles of 64, row splits); 4 row-lanes per column, LDS combine, one atomic per
// (column, split)
// Deterministic mode keeps the row splits (one block per column tile walked 20800 rows alone: 0.4 - 0.6 ms per call at the bench
// geometry) and replaces the atomics: a split leaves its partial sums in ws[split][cols], col_sum_finish_kernel adds them in
// split order -- ONE add into `out` per column and call, as before.  The scratch belongs to the (device, stream) of the call.
namespace {
struct DetWs { float* ptr = nullptr; size_t floats = 0; };
std::mutex g_detws_mutex;
std::map<std::pair<int, hipStream_t>, DetWs> g_detws;
int det_workspace(hipStream_t s, size_t floats, float** out) {
    int dev = 0;
    ADN_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_detws_mutex);
    DetWs& w = g_detws[std::make_pair(dev, s)];
    if (w.floats < floats) {
        if (w.ptr) { ADN_HIP_CHECK(hipStreamSynchronize(s)); ADN_HIP_CHECK(hipFree(w.ptr)); w.ptr = nullptr; w.floats = 0; }
        const size_t want = std::max(floats, (size_t)1 << 20);
        ADN_HIP_CHECK(hipMalloc((void**)&w.ptr, want * sizeof(float)));
        w.floats = want;
    }
    *out = w.ptr;
    return ADN_OK;
}
}  // namespace
__global__ __launch_bounds__(256) void col_sum_finish_kernel(const float* __restrict__ ws, int splits, int cols, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += ws[(size_t)s * cols + c];
    out[c] += v;
}
__global__ __launch_bounds__(256) void col_sum_finish_batch_kernel(const ColSumBatch b) {
    const ColSumItem& it = b.it[blockIdx.y];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if ((int)blockIdx.y >= b.n || c >= it.cols) return;
    float v = 0.f;
    for (int s = 0; s < it.splits; ++s) v += b.ws[(size_t)it.ws_off + (size_t)s * it.cols + c];
    it.out[c] += v;
}

__global__ __launch_bounds__(256) void col_sum_kernel(const float* __restrict__ in, int ld, int rows, int cols,
                                                      float* __restrict__ out, int rows_per_split, float* __restrict__ ws) {
    __shared__ float part[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * rows_per_split, r1 = min(rows, r0 + rows_per_split);
    float acc = 0.f;
    if (c < cols) {                                  // four independent loads in flight per lane (one per iteration: latency-bound)
        float a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int r = r0 + rl;
        for (; r + 12 < r1; r += 16) {
            acc += in[(size_t)r * ld + c]; a1 += in[(size_t)(r + 4) * ld + c];
            a2 += in[(size_t)(r + 8) * ld + c]; a3 += in[(size_t)(r + 12) * ld + c];
        }
        for (; r < r1; r += 4) acc += in[(size_t)r * ld + c];
        acc += a1 + a2 + a3;
    }
    part[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c < cols) {
        const float v = part[0][cl] + part[1][cl] + part[2][cl] + part[3][cl];
        if (ws) ws[(size_t)blockIdx.y * cols + c] = v;
        else atomicAdd(out + c, v);
    }
}

int col_sum(const float* in, int ld, int rows, int cols, float* out, int accumulate, hipStream_t s) {
    if (!accumulate) ADN_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)cols * sizeof(float), s));
    if (rows <= 0) return ADN_OK;
    const int ctiles = cdiv(cols, 64);
    int splits = std::max(1, std::min(cdiv(rows, 64), cdiv(rows >= (1 << 18) ? 4096 : 1024, ctiles)));
    const int rps = cdiv(rows, splits);
    splits = cdiv(rows, rps);
    float* ws = nullptr;
    if (deterministic() && splits > 1) ADN_TRY(det_workspace(s, (size_t)splits * cols, &ws));      // (one split: its single add is ordered already)
    hipLaunchKernelGGL(col_sum_kernel, dim3(ctiles, splits), dim3(256), 0, s, in, ld, rows, cols, out, rps, ws);
    if (ws) hipLaunchKernelGGL(col_sum_finish_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, s, ws, splits, cols, out);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// the same for up to 8 matrices in one launch: block -> item through the running block_end
__global__ __launch_bounds__(256) void col_sum_batch_kernel(const ColSumBatch b) {
    __shared__ float part[4][64];
    int k = 0;
    while (k + 1 < b.n && (int)blockIdx.x >= b.it[k].block_end) ++k;
    const ColSumItem& it = b.it[k];
    const int local = (int)blockIdx.x - (k ? b.it[k - 1].block_end : 0);
    const int ct = local % it.ctiles, sp = local / it.ctiles;
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = ct * 64 + cl;
    const int r0 = sp * it.rps, r1 = min(it.rows, r0 + it.rps);
    float acc = 0.f;
    if (c < it.cols) {
        float a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int r = r0 + rl;
        for (; r + 12 < r1; r += 16) {
            acc += it.in[(size_t)r * it.ld + c]; a1 += it.in[(size_t)(r + 4) * it.ld + c];
            a2 += it.in[(size_t)(r + 8) * it.ld + c]; a3 += it.in[(size_t)(r + 12) * it.ld + c];
        }
        for (; r < r1; r += 4) acc += it.in[(size_t)r * it.ld + c];
        acc += a1 + a2 + a3;
    }
    part[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c < it.cols) {
        const float v = part[0][cl] + part[1][cl] + part[2][cl] + part[3][cl];
        if (b.ws) b.ws[(size_t)it.ws_off + (size_t)sp * it.cols + c] = v;
        else atomicAdd(it.out + c, v);
    }
}

void col_sum_batch_add(ColSumBatch& b, const float* in, int ld, int rows, int cols, float* out) {
    if (rows <= 0 || cols <= 0 || b.n >= 8) return;
    ColSumItem& it = b.it[b.n];
    it.in = in; it.out = out; it.ld = ld; it.rows = rows; it.cols = cols;
    it.ctiles = cdiv(cols, 64);
    int splits = std::max(1, std::min(cdiv(rows, 64), cdiv(1024, it.ctiles)));
    it.rps = cdiv(rows, splits);
    it.splits = cdiv(rows, it.rps);
    it.block_end = (b.n ? b.it[b.n - 1].block_end : 0) + it.ctiles * it.splits;
    it.ws_off = b.n ? b.it[b.n - 1].ws_off + b.it[b.n - 1].splits * b.it[b.n - 1].cols : 0;
    ++b.n;
}

int col_sum_batch(ColSumBatch& b, hipStream_t s) {
    if (b.n <= 0) return ADN_OK;
    b.ws = nullptr;
    int max_cols = 0;
    if (deterministic()) {                           // partial sums per (item, split), then one ordered add per column
        const ColSumItem& last = b.it[b.n - 1];
        ADN_TRY(det_workspace(s, (size_t)last.ws_off + (size_t)last.splits * last.cols, &b.ws));
        for (int k = 0; k < b.n; ++k) max_cols = std::max(max_cols, b.it[k].cols);
    }
    hipLaunchKernelGGL(col_sum_batch_kernel, dim3(b.it[b.n - 1].block_end), dim3(256), 0, s, b);
    if (b.ws) hipLaunchKernelGGL(col_sum_finish_batch_kernel, dim3(cdiv(max_cols, 256), b.n), dim3(256), 0, s, b);
    b.ws = nullptr;
    b.n = 0;
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// bf16 copies of up to 4 weight matrices [nblk * rows][ld] with every block of `rows` rows re-spaced to `rows_pad` rows
// (the K-padding of the concatenated-input GEMM); one launch, pointers by value
struct RepackArgs { const float* in[4]; void* out[4]; };
__global__ __launch_bounds__(256) void repack_rows_bf16_kernel(RepackArgs a, int nblk, int rows, int rows_pad, int ld, int lo_part) {
    const float* in = a.in[blockIdx.y];
    __bf16* out = reinterpret_cast<__bf16*>(a.out[blockIdx.y]);
    const int ld4 = ld / 4;
    const int64_t total = (int64_t)nblk * rows * ld4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % ld4), r = (int)(e / ld4), j = r / rows, k = r % rows;
        const float4 v = reinterpret_cast<const float4*>(in)[(size_t)r * ld4 + c];
        __bf16 o[4] = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        if (lo_part) {                       // the bf16x3 mode's second plane: bf16(v - bf16(v))
            o[0] = (__bf16)(v.x - (float)o[0]); o[1] = (__bf16)(v.y - (float)o[1]);
            o[2] = (__bf16)(v.z - (float)o[2]); o[3] = (__bf16)(v.w - (float)o[3]);
        }
        *reinterpret_cast<uint2*>(out + ((size_t)(j * rows_pad + k) * ld + 4 * c)) = *reinterpret_cast<const uint2*>(o);
    }
}

static int repack_rows_impl(int n, const float* const* in, void* const* out, int nblk, int rows, int rows_pad, int ld, hipStream_t s,
                            int lo_part) {
    ADN_CHECK(n >= 1 && n <= 4 && ld % 4 == 0 && rows_pad >= rows, ADN_ERR_INVALID, "repack_rows_bf16: bad shape");
    RepackArgs a{};
    for (int j = 0; j < n; ++j) { a.in[j] = in[j]; a.out[j] = out[j]; }
    const int64_t total = (int64_t)nblk * rows * (ld / 4);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 2048));
    hipLaunchKernelGGL(repack_rows_bf16_kernel, dim3(grid, n), dim3(256), 0, s, a, nblk, rows, rows_pad, ld, lo_part);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}
int repack_rows_bf16(int n, const float* const* in, void* const* out, int nblk, int rows, int rows_pad, int ld, hipStream_t s) {
    return repack_rows_impl(n, in, out, nblk, rows, rows_pad, ld, s, 0);
}
int repack_rows_bf16_lo(int n, const float* const* in, void* const* out, int nblk, int rows, int rows_pad, int ld, hipStream_t s) {
    return repack_rows_impl(n, in, out, nblk, rows, rows_pad, ld, s, 1);
}

// out[r][j * cols + c] = in_j[r][c]  (bf16; cols a multiple of 8): the materialised concat of up to 4 matrices
// (blockIdx.y = plane: the hi and the lo concat of the bf16x3 / mixed arithmetics leave in one launch -- in[4 + j] and out2)
struct ConcatArgs { const void* in[8]; };
__global__ __launch_bounds__(256) void concat_cols_bf16_kernel(ConcatArgs a, int n, int ld_in, uint4* __restrict__ out, int ld_out,
                                                               int rows, int cols, uint4* __restrict__ out2) {
    const int cpr = cols / 8, per_row = n * cpr;                       // 16-byte chunks
    const int64_t total = (int64_t)rows * per_row;
    const int pl = (int)blockIdx.y;
    uint4* __restrict__ dst = pl ? out2 : out;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / per_row), q = (int)(e % per_row), j = q / cpr, c = q % cpr;
        const void* src = a.in[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) if (4 * pl + j == k) src = a.in[k];        // (explicit selects: no scratch copy of the table)
        dst[(size_t)r * (ld_out / 8) + q] = reinterpret_cast<const uint4*>(src)[(size_t)r * (ld_in / 8) + c];
    }
}

int concat_cols_bf16(int n, const void* const* in, int ld_in, void* out, int ld_out, int rows, int cols, hipStream_t s,
                     const void* const* in_lo, void* out_lo) {
    ADN_CHECK(n >= 1 && n <= 4 && cols % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && (in_lo == nullptr) == (out_lo == nullptr), ADN_ERR_INVALID,
              "concat_cols_bf16: bad shape");
    if (rows <= 0) return ADN_OK;
    ConcatArgs a{};
    for (int j = 0; j < n; ++j) { a.in[j] = in[j]; a.in[4 + j] = in_lo ? in_lo[j] : in[j]; }
    const int64_t total = (int64_t)rows * n * (cols / 8);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, in_lo ? 4096 : 8192));
    hipLaunchKernelGGL(concat_cols_bf16_kernel, dim3(grid, in_lo ? 2 : 1), dim3(256), 0, s, a, n, ld_in, reinterpret_cast<uint4*>(out), ld_out,
                       rows, cols, reinterpret_cast<uint4*>(out_lo));
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// dst[(j * rows_valid + k) * ld + c] += src[(j * rows_pad + k) * ld + c]   (j < nblk, k < rows_valid, c < cols; ld % 4 == 0)
__global__ __launch_bounds__(256) void add_row_blocks_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int ld4,
                                                             int nblk, int rows_valid, int rows_pad, int cols4) {
    const int64_t total = (int64_t)nblk * rows_valid * cols4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % cols4), rk = (int)(e / cols4), j = rk / rows_valid, k = rk % rows_valid;
        const float4 a = src[(size_t)(j * rows_pad + k) * ld4 + c];
        float4 d = dst[(size_t)(j * rows_valid + k) * ld4 + c];
        d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
        dst[(size_t)(j * rows_valid + k) * ld4 + c] = d;
    }
}

// ... for up to 4 (src, dst) pairs of one geometry in one launch (blockIdx.y = pair): the aggregation LSTMs' scratch matrices
struct RowBlocksPairs { const float4* src[4]; float4* dst[4]; };
__global__ __launch_bounds__(256) void add_row_blocks_batch_kernel(const RowBlocksPairs t, int ld4, int nblk, int rows_valid, int rows_pad, int cols4) {
    const float4* __restrict__ src = t.src[0]; float4* __restrict__ dst = t.dst[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) if ((int)blockIdx.y == k) { src = t.src[k]; dst = t.dst[k]; }
    const int64_t total = (int64_t)nblk * rows_valid * cols4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % cols4), rk = (int)(e / cols4), j = rk / rows_valid, k = rk % rows_valid;
        const float4 a = src[(size_t)(j * rows_pad + k) * ld4 + c];
        float4 d = dst[(size_t)(j * rows_valid + k) * ld4 + c];
        d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
        dst[(size_t)(j * rows_valid + k) * ld4 + c] = d;
    }
}
int add_row_blocks_batch(const float* const* src, float* const* dst, int n, int ld, int nblk, int rows_valid, int rows_pad, int cols, hipStream_t s) {
    ADN_CHECK(n >= 1 && n <= 4 && ld % 4 == 0, ADN_ERR_INVALID, "add_row_blocks_batch: 1..4 pairs, ld a multiple of 4");
    if (n == 1) return add_row_blocks(src[0], dst[0], ld, nblk, rows_valid, rows_pad, cols, s);
    const int cols4 = (cols + 3) / 4;
    const int64_t total = (int64_t)nblk * rows_valid * cols4;
    if (total <= 0) return ADN_OK;
    RowBlocksPairs t;
    for (int k = 0; k < 4; ++k) { const int q = k < n ? k : 0; t.src[k] = reinterpret_cast<const float4*>(src[q]); t.dst[k] = reinterpret_cast<float4*>(dst[q]); }
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 2048));
    hipLaunchKernelGGL(add_row_blocks_batch_kernel, dim3(grid, n), dim3(256), 0, s, t, ld / 4, nblk, rows_valid, rows_pad, cols4);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int add_row_blocks(const float* src, float* dst, int ld, int nblk, int rows_valid, int rows_pad, int cols, hipStream_t s) {
    ADN_CHECK(ld % 4 == 0, ADN_ERR_INVALID, "add_row_blocks: ld must be a multiple of 4");
    const int cols4 = (cols + 3) / 4;
    const int64_t total = (int64_t)nblk * rows_valid * cols4;
    if (total <= 0) return ADN_OK;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 4096));
    hipLaunchKernelGGL(add_row_blocks_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const float4*>(src),
                       reinterpret_cast<float4*>(dst), ld / 4, nblk, rows_valid, rows_pad, cols4);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// DropoutLayer (lasagne, rescale=True; modelzoo/adenet_v3.py:112,123,134,154): out = x * keep / (1 - p).  The mask is
// a counter-based hash of (seed, call counter, layer id, element index in (B,T,width) C order) -- the same function
// as oracle/adenet_oracle.py::dropout_uniform, so both draw identical masks; backward re-derives it instead of
// storing it.  The matrix is time-major (row t*B + b) and may be a column block [off, off + cols) of a `width`-wide
// logical tensor (the concat of the stream outputs).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool dropout_keep(uint32_t key, uint32_t idx, float p) {
    uint32_t x = idx * 0x9E3779B1u + key;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return (float)(x >> 8) * (1.0f / 16777216.0f) >= p;
}

__global__ __launch_bounds__(256) void dropout_kernel(const float* in, int ld_in, float* out, int ld_out,     // in == out allowed
                                                      int B, int T, int cols, int width, int off, float p, float scale, uint32_t key) {
    const int64_t total = (int64_t)B * T * cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % cols), r = (int)(e / cols), t = r / B, b = r % B;
        const uint32_t idx = (uint32_t)(((int64_t)b * T + t) * width + off + c);
        const float v = in[(size_t)r * ld_in + c];
        out[(size_t)r * ld_out + c] = dropout_keep(key, idx, p) ? v * scale : 0.f;
    }
}

int dropout_apply(const float* in, int ld_in, float* out, int ld_out, int B, int T, int cols, int width, int off, float p,
                  uint32_t seed, uint32_t counter, uint32_t layer, hipStream_t s) {
    ADN_CHECK(p >= 0.f && p < 1.f, ADN_ERR_INVALID, "dropout probability must be in [0, 1)");
    const int64_t total = (int64_t)B * T * cols;
    if (total <= 0) return ADN_OK;
    const uint32_t key = seed ^ (layer * 0x85EBCA77u) ^ (counter * 0xC2B2AE3Du);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 16384));
    hipLaunchKernelGGL(dropout_kernel, dim3(grid), dim3(256), 0, s, in, ld_in, out, ld_out, B, T, cols, width, off, p,
                       1.f / (1.f - p), key);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// last-timestep head: softmax over C classes of B rows + categorical cross-entropy
// (lasagne.objectives.categorical_crossentropy(...).mean(), avletters/trimodal.py:327-328):
//   probs[b] = softmax(z[b]);  row_loss[b] = -log probs[b][y_b];  dz[b] = (probs[b] - onehot(y_b)) / total
// one wave per row, class on the lane (C <= 64)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ z, int ldz, int B, int T, int C,
                                                         const int32_t* __restrict__ y_bt, const float* __restrict__ total,
                                                         float* __restrict__ probs, float* __restrict__ row_loss,
                                                         float* __restrict__ dz, int lddz) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float v = lane < C ? z[(size_t)b * ldz + lane] : -INFINITY;
    float mx = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float e = lane < C ? expf(v - mx) : 0.f;
    float sum = e;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float pr = e / sum;
    if (lane < C && probs) probs[(size_t)b * C + lane] = pr;
    if (y_bt) {
        const int y = y_bt[(size_t)b * T];                      // the label is repeated over T (runners/3stream.py:360-361)
        if (row_loss && lane == y) row_loss[b] = -logf(pr);
        if (dz && lane < C) dz[(size_t)b * lddz + lane] = (pr - (lane == y ? 1.f : 0.f)) / total[0];
    }
}

int softmax_ce(const float* z, int ldz, int B, int T, int C, const int32_t* y_bt, const float* total, float* probs,
               float* row_loss, float* dz, int lddz, hipStream_t s) {
    ADN_CHECK(C >= 1 && C <= 64, ADN_ERR_INVALID, "softmax_ce: 1..64 classes");
    if (B <= 0) return ADN_OK;
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, z, ldz, B, T, C, y_bt, total, probs, row_loss, dz, lddz);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// lasagne.updates.sgd / momentum / nesterov_momentum / adadelta on the flat buffers
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ vel,
                                                  int64_t n4, float lr, float mu, int nesterov, const float* poison, int* sticky) {
    if (poison && *poison != 0.f) {          // a rank's LSTM exchange timed out: the gradients are invalid on EVERY rank
        if (sticky && blockIdx.x == 0 && threadIdx.x == 0) *sticky = 1;
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 P = reinterpret_cast<float4*>(p)[i];
        const float4 G = reinterpret_cast<const float4*>(g)[i];
        float* pp = reinterpret_cast<float*>(&P);
        const float* gg = reinterpret_cast<const float*>(&G);
        if (mu == 0.f) {
#pragma unroll
            for (int e = 0; e < 4; ++e) pp[e] = pp[e] - lr * gg[e];
        } else {
            float4 V = reinterpret_cast<float4*>(vel)[i];
            float* vv = reinterpret_cast<float*>(&V);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                vv[e] = mu * vv[e] - lr * gg[e];
                pp[e] = pp[e] + (nesterov ? (mu * vv[e] - lr * gg[e]) : vv[e]);
            }
            reinterpret_cast<float4*>(vel)[i] = V;
        }
        reinterpret_cast<float4*>(p)[i] = P;
    }
}

int sgd_update(float* p, const float* g, float* vel, int64_t n, float lr, float momentum, int nesterov, hipStream_t s,
               const float* poison, int* sticky) {
    ADN_CHECK(n % 4 == 0, ADN_ERR_INVALID, "sgd_update: element count must be a multiple of 4");
    if (!n) return ADN_OK;
    const int64_t n4 = n / 4;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n4 + 255) / 256, 8192));
    hipLaunchKernelGGL(sgd_kernel, dim3(grid), dim3(256), 0, s, p, g, vel, n4, lr, momentum, nesterov, poison, sticky);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

__global__ __launch_bounds__(256) void adadelta_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ accu,
                                                       float* __restrict__ delta, int64_t n4, float lr, float rho, float eps,
                                                       const float* poison, int* sticky) {
    if (poison && *poison != 0.f) {          // a rank's LSTM exchange timed out: the gradients are invalid on EVERY rank
        if (sticky && blockIdx.x == 0 && threadIdx.x == 0) *sticky = 1;
        return;
    }
    const float omr = 1.f - rho;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 P = reinterpret_cast<float4*>(p)[i], A = reinterpret_cast<float4*>(accu)[i], D = reinterpret_cast<float4*>(delta)[i];
        const float4 G = reinterpret_cast<const float4*>(g)[i];
        float *pp = reinterpret_cast<float*>(&P), *aa = reinterpret_cast<float*>(&A), *dd = reinterpret_cast<float*>(&D);
        const float* gg = reinterpret_cast<const float*>(&G);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            aa[e] = rho * aa[e] + omr * gg[e] * gg[e];
            const float upd = gg[e] * sqrtf(dd[e] + eps) / sqrtf(aa[e] + eps);
            pp[e] = pp[e] - lr * upd;
            dd[e] = rho * dd[e] + omr * upd * upd;
        }
        reinterpret_cast<float4*>(p)[i] = P; reinterpret_cast<float4*>(accu)[i] = A; reinterpret_cast<float4*>(delta)[i] = D;
    }
}

int adadelta_update(float* p, const float* g, float* accu, float* delta, int64_t n, float lr, float rho, float eps, hipStream_t s,
                    const float* poison, int* sticky) {
    ADN_CHECK(n % 4 == 0, ADN_ERR_INVALID, "adadelta_update: element count must be a multiple of 4");
    if (!n) return ADN_OK;
    const int64_t n4 = n / 4;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n4 + 255) / 256, 8192));
    hipLaunchKernelGGL(adadelta_kernel, dim3(grid), dim3(256), 0, s, p, g, accu, delta, n4, lr, rho, eps, poison, sticky);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void dot_all_kernel(const float* __restrict__ a, int lda,
                                                      const float* __restrict__ b, int ldb, int rows, int cols,
                                                      float* __restrict__ out) {
    __shared__ float part[4];
    const int64_t total = (int64_t)rows * cols;
    float acc = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / cols), c = (int)(e % cols);
        acc += a[(size_t)r * lda + c] * b[(size_t)r * ldb + c];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

int dot_all(const float* a, int lda, const float* b, int ldb, int rows, int cols, float* out, float* /*scratch*/,
            hipStream_t s) {
    hipLaunchKernelGGL(dot_all_kernel, dim3(deterministic() ? 1 : grid_for((int64_t)rows * cols / 4 + 1)), dim3(256), 0, s, a, lda, b, ldb,
                       rows, cols, out);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

__device__ __forceinline__ float act_grad_y(int act, float y) {
    switch (act) {
        case ADN_ACT_RECTIFY: return y > 0.f ? 1.f : 0.f;
        case ADN_ACT_SIGMOID: return y * (1.f - y);
        case ADN_ACT_TANH: return 1.f - y * y;
        case ADN_ACT_LEAKY_RECTIFY: return y > 0.f ? 1.f : 0.01f;
        case ADN_ACT_VERY_LEAKY_RECTIFY: return y > 0.f ? 1.f : (1.f / 3.f);
        case ADN_ACT_SCALED_TANH: { const float t = y * (1.f / 2.4f); return 1.2f * (1.f - t * t); }
        case ADN_ACT_SCALED_TANH_LECUN: { const float t = y * (1.f / 1.7159f); return (2.f / 3.f) * 1.7159f * (1.f - t * t); }
        case kActRectifyHalf: return y > 0.f ? 1.f : (__float_as_uint(y) == 0x80000000u ? 0.5f : 0.f);      // (adn_common.h)
        default: return 1.f;
    }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(float* __restrict__ dy, int ld_dy, const float* __restrict__ y,
                                                      int ld_y, int rows, int cols, int act) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / cols), c = (int)(e % cols);
        dy[(size_t)r * ld_dy + c] *= act_grad_y(act, y[(size_t)r * ld_y + c]);
    }
}

int act_backward(float* dy, int ld_dy, const float* y, int ld_y, int rows, int cols, int act, hipStream_t s) {
    if (act == ADN_ACT_LINEAR) return ADN_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, s, dy, ld_dy, y, ld_y,
                       rows, cols, act);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// mask (B,T) -> (T,B); total = number of valid frames.  One workgroup of 1024 threads (B*T is tiny: the count needs no second pass),
// 16 mask bytes per thread and pass -- as one block of 256 threads reading byte by byte this was 11 us at 520 x 40 (27 with the
// lengths compared), a visible slice of a 3 ms step.
// (lens: the lengths the caller announced for this batch -- frame compaction, compact.hip -- which the mask must agree with)
__global__ __launch_bounds__(1024) void mask_prepare_kernel(const uint8_t* __restrict__ m_bt, uint8_t* __restrict__ m_tb,
                                                            int B, int T, float* __restrict__ total, const int32_t* __restrict__ lens,
                                                            int* __restrict__ flag, int bit, const int32_t* __restrict__ tm_row0,
                                                            const int32_t* __restrict__ tm_T, int tm_stride) {
    __shared__ int part[16];
    const int n = B * T;
    const bool vec = (reinterpret_cast<uintptr_t>(m_bt) & 15) == 0;
    int cnt = 0, bad = 0;
    for (int e0 = (int)threadIdx.x * 16; e0 < n; e0 += 1024 * 16) {
        uint8_t v[16];
        if (vec && e0 + 16 <= n) *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(m_bt + e0);
        else for (int k = 0; k < 16; ++k) v[k] = e0 + k < n ? m_bt[e0 + k] : 0;
        int b = e0 / T, t = e0 - b * T;
        int len = lens ? lens[min(b, B - 1)] : 0;
        // (length buckets, model.hip TmPlan: frame (b, t) lives in row tm_row0[b] + t tm_stride while t < tm_T[b]; rows no frame
        //  lives in are zeroed by whoever installs the tables)
        int r0 = tm_row0 ? tm_row0[min(b, B - 1)] : 0, tw = tm_T ? tm_T[min(b, B - 1)] : T;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (e0 + k < n) {
                const int u = v[k] ? 1 : 0;
                if (!tm_row0) m_tb[(size_t)t * B + b] = (uint8_t)u;
                else if (t < tw) m_tb[(size_t)r0 + (size_t)t * tm_stride] = (uint8_t)u;
                cnt += u;
                if (lens) bad |= u ^ (t < len ? 1 : 0);
            }
            if (++t == T) {
                t = 0; ++b;
                if (lens) len = lens[min(b, B - 1)];
                if (tm_row0) { r0 = tm_row0[min(b, B - 1)]; tw = tm_T[min(b, B - 1)]; }
            }
        }
    }
    if (bad) atomicOr(flag, bit);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0 && total) {
        int sum = 0;
        for (int k = 0; k < 16; ++k) sum += part[k];
        *total = (float)sum;
    }
}

int mask_prepare(const uint8_t* mask_bt, uint8_t* mask_tb, int B, int T, float* total, hipStream_t s, const int32_t* lens, int* flag, int bit,
                 const int32_t* tm_row0, const int32_t* tm_T, int tm_stride) {
    hipLaunchKernelGGL(mask_prepare_kernel, dim3(1), dim3(1024), 0, s, mask_bt, mask_tb, B, T, total, flag ? lens : nullptr, flag, bit,
                       tm_row0, tm_T, tm_stride);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

__global__ __launch_bounds__(256) void broadcast_rows_kernel(const float* __restrict__ vec, float* __restrict__ dst,
                                                             int ld, int rows, int cols) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / cols), c = (int)(e % cols);
        dst[(size_t)r * ld + c] = vec[c];
    }
}

// initial LSTM state of every utterance: h[r][:] = hid_init, c[r][:] = cell_init, plus the bf16 copy of h (one launch
// instead of two broadcasts and a conversion per LSTM)
__global__ __launch_bounds__(256) void lstm_init_state_kernel(const float* __restrict__ hid, const float* __restrict__ cell,
                                                              float* __restrict__ h, float* __restrict__ c,
                                                              __bf16* __restrict__ h16, int ld, int rows, int cols) {
    const int64_t total = (int64_t)rows * ld;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int col = (int)(e % ld);
        const float hv = col < cols ? hid[col] : 0.f, cv = col < cols ? cell[col] : 0.f;
        h[e] = hv; c[e] = cv;
        if (h16) h16[e] = (__bf16)hv;
    }
}

// ... for up to kMaxInitJobs LSTMs of one shape in ONE launch (blockIdx.y = LSTM)
struct InitJobTable { LstmInitJob j[kMaxInitJobs]; };
__global__ __launch_bounds__(256) void lstm_init_state_batch_kernel(const InitJobTable tab, int ld, int rows, int cols) {
    LstmInitJob j = tab.j[0];
#pragma unroll
    for (int k = 1; k < kMaxInitJobs; ++k) if ((int)blockIdx.y == k) j = tab.j[k];
    const int64_t total = (int64_t)rows * ld;
    __bf16* h16 = reinterpret_cast<__bf16*>(j.h16);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int col = (int)(e % ld);
        const float hv = col < cols ? j.hid[col] : 0.f, cv = col < cols ? j.cell[col] : 0.f;
        j.h[e] = hv; j.c[e] = cv;
        if (h16) h16[e] = (__bf16)hv;
    }
}
int lstm_init_state_batch(const LstmInitJob* jobs, int n, int ld, int rows, int cols, hipStream_t s) {
    for (int k0 = 0; k0 < n; k0 += kMaxInitJobs) {
        const int nn = std::min(kMaxInitJobs, n - k0);
        if (nn == 1) { ADN_TRY(lstm_init_state_rows(jobs[k0].hid, jobs[k0].cell, jobs[k0].h, jobs[k0].c, jobs[k0].h16, ld, rows, cols, s)); continue; }
        InitJobTable tab;
        for (int k = 0; k < kMaxInitJobs; ++k) tab.j[k] = jobs[k0 + (k < nn ? k : 0)];
        hipLaunchKernelGGL(lstm_init_state_batch_kernel, dim3(std::min(grid_for((int64_t)rows * ld), 1024), nn), dim3(256), 0, s, tab, ld, rows, cols);
        ADN_HIP_CHECK(hipGetLastError());
    }
    return ADN_OK;
}

int lstm_init_state_rows(const float* hid, const float* cell, float* h, float* c, void* h16, int ld, int rows, int cols, hipStream_t s);
int lstm_init_state_rows(const float* hid, const float* cell, float* h, float* c, void* h16, int ld, int rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(lstm_init_state_kernel, dim3(grid_for((int64_t)rows * ld)), dim3(256), 0, s, hid, cell, h, c,
                       reinterpret_cast<__bf16*>(h16), ld, rows, cols);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int broadcast_rows(const float* vec, float* dst, int ld, int rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(broadcast_rows_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, s, vec, dst, ld,
                       rows, cols);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// =========================================================================================
// classifier softmax (modelzoo/adenet_v2.py:89-92) + temporal_softmax_loss (custom/objectives.py:4-39)
// The loss soft-maxes the network's probabilities a second time; kept.
// One group of G lanes (32 or 64) per frame, class index on the lane.
// =========================================================================================
template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, G));
    return v;
}
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
    return v;
}

template <int G>
__global__ __launch_bounds__(256) void softmax_loss_kernel(const float* __restrict__ z, int ldz, int B, int T, int C,
                                                           const uint8_t* __restrict__ mask_tb,
                                                           const int32_t* __restrict__ y_bt,
                                                           const float* __restrict__ total,
                                                           float* __restrict__ probs_bt, float* __restrict__ row_loss,
                                                           float* __restrict__ dz, int lddz, __bf16* __restrict__ dz16,
                                                           const int32_t* __restrict__ bt_of_row, int rows, __bf16* __restrict__ dz16lo) {
    const int r = (blockIdx.x * 256 + threadIdx.x) / G;
    const int c = threadIdx.x % G;
    if (r >= rows) return;                 // whole groups exit together (256 % G == 0)
    // the frame b T + t this row holds: t B + b -> (b, t), or what the table says (length buckets, model.hip TmPlan; -1 = a row no
    // frame lives in: its gradient row must read as zero, it has no probabilities and no loss)
    const int bt = bt_of_row ? bt_of_row[r] : (r % B) * T + r / B;
    const bool cv = c < C;
    if (bt < 0) {
        if (row_loss && c == 0) row_loss[r] = 0.f;
        if (dz && cv) {
            dz[(size_t)r * lddz + c] = 0.f;
            if (dz16) dz16[(size_t)r * lddz + c] = (__bf16)0.f;
            if (dz16lo) dz16lo[(size_t)r * lddz + c] = (__bf16)0.f;
        }
        return;
    }
    const float zc = cv ? z[(size_t)r * ldz + c] : -INFINITY;
    const float m1 = group_max<G>(zc);
    const float e1 = cv ? expf(zc - m1) : 0.f;
    const float p = e1 / group_sum<G>(e1);
    if (probs_bt && cv) probs_bt[(size_t)bt * C + c] = p;
    if (!y_bt) return;
    const float m2 = group_max<G>(cv ? p : -INFINITY);
    const float e2 = cv ? expf(p - m2) : 0.f;
    const float s2 = group_sum<G>(e2);
    const float q = e2 / s2;
    const int y = y_bt[bt];
    const float msk = mask_tb[r] ? 1.f : 0.f;
    if (row_loss && c == y) row_loss[r] = msk * -(p - m2 - logf(s2));
    if (dz) {
        const float dp = cv ? (msk / total[0]) * (q - (c == y ? 1.f : 0.f)) : 0.f;
        const float dot = group_sum<G>(dp * p);
        const float g = p * (dp - dot);
        if (cv) dz[(size_t)r * lddz + c] = g;
        if (cv && dz16) {
            const __bf16 h = (__bf16)g;
            dz16[(size_t)r * lddz + c] = h;
            if (dz16lo) dz16lo[(size_t)r * lddz + c] = (__bf16)(g - (float)h);       // (planes: hi beside it)
        }
    }
}

int softmax_loss(const float* z, int ldz, int B, int T, int C, const uint8_t* mask_tb, const int32_t* y_bt,
                 const float* total, float* probs_bt, float* row_loss, float* dz, int lddz, hipStream_t s, void* dz16,
                 const int32_t* bt_of_row, int table_rows, void* dz16lo) {
    ADN_CHECK(C >= 1 && C <= ADN_MAX_CLASSES, ADN_ERR_INVALID, "softmax: unsupported number of classes");
    const int rows = bt_of_row ? table_rows : B * T;
    ProfScope prof(PROF_SOFTMAX_LOSS, 0.0, 4.0 * rows * (double)C * (dz ? 3.0 : 2.0), s);
    if (C <= 32) {
        hipLaunchKernelGGL(softmax_loss_kernel<32>, dim3(cdiv(rows, 8)), dim3(256), 0, s, z, ldz, B, T, C, mask_tb,
                           y_bt, total, probs_bt, row_loss, dz, lddz, reinterpret_cast<__bf16*>(dz16), bt_of_row, rows,
                           reinterpret_cast<__bf16*>(dz16lo));
    } else {
        hipLaunchKernelGGL(softmax_loss_kernel<64>, dim3(cdiv(rows, 4)), dim3(256), 0, s, z, ldz, B, T, C, mask_tb,
                           y_bt, total, probs_bt, row_loss, dz, lddz, reinterpret_cast<__bf16*>(dz16), bt_of_row, rows,
                           reinterpret_cast<__bf16*>(dz16lo));
    }
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// single block, fixed summation order -> reproducible loss value
__global__ __launch_bounds__(256) void reduce_loss_kernel(const float* __restrict__ v, int n,
                                                          const float* __restrict__ total, float* __restrict__ out) {
    __shared__ float part[256];
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // 8 independent loads in flight per lane; fixed order
    for (int base = threadIdx.x; base < n; base += 256 * 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {                            // clamped address + select: the loads stay unconditional
            const int i = base + 256 * k;
            const float x = v[min(i, n - 1)];
            a[k] += i < n ? x : 0.f;
        }
    }
    const float acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0] / total[0];
}

int reduce_loss(const float* v, int n, const float* total, float* out, hipStream_t s) {
    hipLaunchKernelGGL(reduce_loss_kernel, dim3(1), dim3(256), 0, s, v, n, total, out);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// =========================================================================================
// Adam: lasagne.updates.adam == custom/updates.py:73-99.  a_t = lr*sqrt(1-b2^t)/(1-b1^t) from the host.
// 7 floats of traffic per parameter (read p,g,m,v; write p,m,v): pure HBM stream, float4 per lane.
// =========================================================================================
__device__ __forceinline__ void adam_body(float* __restrict__ p, const float* __restrict__ g,
                                          float* __restrict__ m, float* __restrict__ v, int64_t n4,
                                          int64_t n, float a_t, float b1, float b2, float eps,
                                          __bf16* __restrict__ p16, __bf16* __restrict__ p16lo) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
#define ADN_ADAM1(x)                                         \
        mm.x = b1 * mm.x + (1.f - b1) * gg.x;                \
        vv.x = b2 * vv.x + (1.f - b2) * gg.x * gg.x;         \
        pp.x = pp.x - a_t * mm.x / (sqrtf(vv.x) + eps);
        ADN_ADAM1(x) ADN_ADAM1(y) ADN_ADAM1(z) ADN_ADAM1(w)
#undef ADN_ADAM1
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (p16) {                                               // the bf16 shadow the next step's GEMMs read
            __bf16 o[4] = {(__bf16)pp.x, (__bf16)pp.y, (__bf16)pp.z, (__bf16)pp.w};
            reinterpret_cast<uint2*>(p16)[i] = *reinterpret_cast<const uint2*>(o);
            if (p16lo) {                                         // bf16x3: the lo plane bf16(p - bf16(p)) beside it
                __bf16 l[4] = {(__bf16)(pp.x - (float)o[0]), (__bf16)(pp.y - (float)o[1]), (__bf16)(pp.z - (float)o[2]), (__bf16)(pp.w - (float)o[3])};
                reinterpret_cast<uint2*>(p16lo)[i] = *reinterpret_cast<const uint2*>(l);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) {      // tail (n not a multiple of 4)
        const int64_t i = n4 * 4 + threadIdx.x;
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float pi = p[i] - a_t * mi / (sqrtf(vi) + eps);
        p[i] = pi;
        if (p16) { p16[i] = (__bf16)pi; if (p16lo) p16lo[i] = (__bf16)(pi - (float)p16[i]); }
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n4,
                                                   int64_t n, float a_t, float b1, float b2, float eps,
                                                   __bf16* __restrict__ p16, const float* poison, int* sticky,
                                                   __bf16* __restrict__ p16lo) {
    if (poison && *poison != 0.f) {          // a rank's LSTM exchange timed out: the gradients are invalid on EVERY rank
        if (sticky && blockIdx.x == 0 && threadIdx.x == 0) *sticky = 1;
        return;
    }
    adam_body(p, g, m, v, n4, n, a_t, b1, b2, eps, p16, p16lo);
}
// the same update on up to kMaxAdamRanges ranges of the flat buffers in ONE launch (blockIdx.y = range): the data-parallel
// wrapper updates every gradient bucket whose reduction has landed with one launch
__global__ __launch_bounds__(256) void adam_ranges_kernel(const AdamRanges R, float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, float a_t, float b1, float b2,
                                                          float eps, __bf16* __restrict__ p16, const float* poison, int* sticky,
                                                          __bf16* __restrict__ p16lo) {
    if (poison && *poison != 0.f) {
        if (sticky && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *sticky = 1;
        return;
    }
    int64_t b = R.begin[0], e = R.end[0];
#pragma unroll
    for (int k = 1; k < kMaxAdamRanges; ++k) if ((int)blockIdx.y == k) { b = R.begin[k]; e = R.end[k]; }
    const int64_t n = e - b;
    if (n <= 0) return;
    adam_body(p + b, g + b, m + b, v + b, n / 4, n, a_t, b1, b2, eps, p16 ? p16 + b : nullptr, p16lo ? p16lo + b : nullptr);
}

// float4 copy (bench.py's measured-HBM yardstick)
__global__ __launch_bounds__(256) void copy4_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}
// U float4 per thread and iteration, all U loads issued ahead of the stores (U x 16 B x 256 threads x resident blocks in flight per CU)
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy4u_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
            if constexpr (NT) __builtin_nontemporal_store(nt_f32x4{v[u].x, v[u].y, v[u].z, v[u].w}, reinterpret_cast<nt_f32x4*>(dst + i + u * stride));
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}
constexpr int kCopyDefaultVariant = 0;
int copy_bench(const float* src, float* dst, int64_t n, int repeats, hipStream_t s, float* ms) {
    ADN_CHECK(src && dst && ms && n >= 4 && repeats >= 1, ADN_ERR_INVALID, "copy_bench: bad argument");
    ADN_CHECK(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, ADN_ERR_INVALID, "copy_bench: 16-byte alignment");
    hipEvent_t a, b;
    ADN_HIP_CHECK(hipEventCreate(&a)); ADN_HIP_CHECK(hipEventCreate(&b));
    const int64_t n4 = n / 4;
    // ADN_COPY_VARIANT (experiments): 0 = one float4 per thread and iteration, grid-stride (round 1-4: 4.96 - 5.06 TB/s);
    // U >= 2 = U float4 loads in flight per thread ahead of their stores; +16 = non-temporal stores
    static const int variant = getenv("ADN_COPY_VARIANT") ? atoi(getenv("ADN_COPY_VARIANT")) : kCopyDefaultVariant;
    // (profiles/r05/copy_variants.txt: 7 forms x 4 grid sizes on one box -- 4.1 - 5.3 TB/s, none near the guide's 6.29; the plain
    //  grid-stride form with 64 blocks per CU is the fastest: 5.31 against 5.17 at 32)
    static const int blocks_per_cu = getenv("ADN_COPY_BLOCKS") ? atoi(getenv("ADN_COPY_BLOCKS")) : 64;
    const int unroll = variant & 15;
    const bool nt = (variant & 16) != 0;
    const int grid = (int)std::min<int64_t>((n4 / std::max(unroll, 1) + 255) / 256, 256 * blocks_per_cu);
    auto launch = [&]() {
        const float4* s4 = reinterpret_cast<const float4*>(src); float4* d4 = reinterpret_cast<float4*>(dst);
        if (unroll == 4) { if (nt) hipLaunchKernelGGL((copy4u_kernel<4, true>), dim3(grid), dim3(256), 0, s, s4, d4, n4); else hipLaunchKernelGGL((copy4u_kernel<4, false>), dim3(grid), dim3(256), 0, s, s4, d4, n4); }
        else if (unroll == 8) { if (nt) hipLaunchKernelGGL((copy4u_kernel<8, true>), dim3(grid), dim3(256), 0, s, s4, d4, n4); else hipLaunchKernelGGL((copy4u_kernel<8, false>), dim3(grid), dim3(256), 0, s, s4, d4, n4); }
        else if (unroll == 2) { if (nt) hipLaunchKernelGGL((copy4u_kernel<2, true>), dim3(grid), dim3(256), 0, s, s4, d4, n4); else hipLaunchKernelGGL((copy4u_kernel<2, false>), dim3(grid), dim3(256), 0, s, s4, d4, n4); }
        else hipLaunchKernelGGL(copy4_kernel, dim3(grid), dim3(256), 0, s, s4, d4, n4);
    };
    launch();
    ADN_HIP_CHECK(hipEventRecord(a, s));
    for (int r = 0; r < repeats; ++r) launch();
    ADN_HIP_CHECK(hipEventRecord(b, s));
    ADN_HIP_CHECK(hipEventSynchronize(b));
    ADN_HIP_CHECK(hipEventElapsedTime(ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ADN_OK;
}

// tail[1] of the gradient buffer = 1 when this device's LSTM-exchange error word is raised (it rides through the data-parallel
// all-reduce, so that the optimiser kernels of EVERY rank see it and skip the update together)
// (loss / tail0, optional: the rank's cost share goes into tail[0] with the same launch -- it has the same deadline, bucket 0's release)
__global__ void poison_tail_kernel(const int* __restrict__ err_word, float* __restrict__ tail1, const float* __restrict__ loss,
                                   float* __restrict__ tail0) {
    *tail1 = (*err_word != 0) ? 1.f : 0.f;
    if (loss) *tail0 = *loss;
}
int poison_tail(const int* err_word, float* tail1, hipStream_t s, const float* loss, float* tail0) {
    hipLaunchKernelGGL(poison_tail_kernel, dim3(1), dim3(1), 0, s, err_word, tail1, loss, tail0);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int adam_update(float* p, const float* g, float* m, float* v, int64_t n, float a_t, float beta1, float beta2,
                float eps, hipStream_t s, void* p16, const float* poison, int* sticky, void* p16lo) {
    if (n <= 0) return ADN_OK;
    const int64_t n4 = n / 4;
    ProfScope prof(PROF_ADAM, 0.0, 7.0 * 4.0 * (double)n, s);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(std::max<int64_t>(n4, 1))), dim3(256), 0, s, p, g, m, v, n4, n, a_t,
                       beta1, beta2, eps, reinterpret_cast<__bf16*>(p16), poison, sticky, reinterpret_cast<__bf16*>(p16lo));
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int adam_update_ranges(float* p, const float* g, float* m, float* v, const int64_t* begin, const int64_t* end, int n_ranges, float a_t,
                       float beta1, float beta2, float eps, hipStream_t s, void* p16, const float* poison, int* sticky, void* p16lo) {
    for (int k0 = 0; k0 < n_ranges; k0 += kMaxAdamRanges) {
        const int nn = std::min(kMaxAdamRanges, n_ranges - k0);
        AdamRanges R;
        int64_t longest = 0; double total = 0.0;
        for (int k = 0; k < kMaxAdamRanges; ++k) {
            R.begin[k] = k < nn ? begin[k0 + k] : 0; R.end[k] = k < nn ? end[k0 + k] : 0;
            longest = std::max(longest, R.end[k] - R.begin[k]); total += (double)(R.end[k] - R.begin[k]);
        }
        if (longest <= 0) continue;
        ProfScope prof(PROF_ADAM, 0.0, 7.0 * 4.0 * total, s);
        hipLaunchKernelGGL(adam_ranges_kernel, dim3(grid_for(std::max<int64_t>(longest / 4, 1)), nn), dim3(256), 0, s, R, p, g, m, v, a_t,
                           beta1, beta2, eps, reinterpret_cast<__bf16*>(p16), poison, sticky, reinterpret_cast<__bf16*>(p16lo));
        ADN_HIP_CHECK(hipGetLastError());
    }
    return ADN_OK;
}

}  // namespace adn
