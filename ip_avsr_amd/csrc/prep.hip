// Feature front-end on the GPU: the per-frame / per-utterance array transforms the reference runs in
// NumPy before batching (reference utils/preprocessing.py).  All of them are HBM-bound streaming kernels over the
// (sum of lengths) x D frame matrix, row-major, fp32; utterance structure comes in as per-frame or per-utterance
// int32 index vectors built once on the host from the length vector.  Reference quirks are reproduced (cited inline).
#include "adn_common.h"
#include <algorithm>
#include <cmath>

namespace adn {

// ---------------------------------------------------------------------------------------------------------
// deltas along time, one level (utils/preprocessing.py:17-51 applied per utterance as :465-489 does):
//   out[t][f] = sum_{m = h .. -h, m != 0} m * x[clampq(t + m)][f]
// where positions before the utterance read its SECOND frame (the reference pads with column 1, App. E-4) and
// positions after it read its last frame; taps are accumulated in the order m = h, h-1, ..., -h (lfilter's order).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seq_deltas_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out,
                                                         int ld_out, const int* __restrict__ first, const int* __restrict__ last,
                                                         int n_frames, int F, int h) {
    const int64_t total = (int64_t)n_frames * F;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int t = (int)(e / F), f = (int)(e % F);
        const int s0 = first[t], s1 = last[t];
        const int second = min(s0 + 1, s1);
        float acc = 0.f;
        for (int m = h; m >= -h; --m) {
            if (!m) continue;
            const int p = t + m;
            const int src = p < s0 ? second : (p > s1 ? s1 : p);
            acc += (float)m * in[(size_t)src * ld_in + f];
        }
        out[(size_t)t * ld_out + f] = acc;
    }
}

int prep_seq_deltas(const float* in, int ld_in, float* out, int ld_out, const int* first, const int* last, int n_frames, int F,
                    int w, hipStream_t s) {
    ADN_CHECK(in && out && first && last, ADN_ERR_INVALID, "prep_seq_deltas: null argument");
    ADN_CHECK(w >= 1 && w < 4096, ADN_ERR_INVALID, "prep_seq_deltas: window out of range");
    if (n_frames <= 0 || F <= 0) return ADN_OK;
    const int64_t total = (int64_t)n_frames * F;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 16384));
    hipLaunchKernelGGL(seq_deltas_kernel, dim3(grid), dim3(256), 0, s, in, ld_in, out, ld_out, first, last, n_frames, F, w / 2);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// frame differences per utterance; frame 0 receives a copy of the first difference (utils/preprocessing.py:506-517)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void diff_images_kernel(const float* __restrict__ in, float* __restrict__ out, int ld,
                                                          const int* __restrict__ first, const int* __restrict__ last,
                                                          int n_frames, int D) {
    const int64_t total = (int64_t)n_frames * D;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int t = (int)(e / D), c = (int)(e % D);
        const int s0 = first[t], s1 = last[t];
        const int hi = (t == s0) ? min(s0 + 1, s1) : t;             // frame 0: x[1] - x[0]
        const int lo = (t == s0) ? s0 : t - 1;
        out[(size_t)t * ld + c] = in[(size_t)hi * ld + c] - in[(size_t)lo * ld + c];
    }
}

int prep_diff_images(const float* in, float* out, int ld, const int* first, const int* last, int n_frames, int D, hipStream_t s) {
    ADN_CHECK(in && out && first && last && in != out, ADN_ERR_INVALID, "prep_diff_images: null or aliased argument");
    if (n_frames <= 0 || D <= 0) return ADN_OK;
    const int64_t total = (int64_t)n_frames * D;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 16384));
    hipLaunchKernelGGL(diff_images_kernel, dim3(grid), dim3(256), 0, s, in, out, ld, first, last, n_frames, D);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// subtract each utterance's mean frame (utils/preprocessing.py:260-277): grid (utterance, 256-column tiles)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mean_image_kernel(const float* __restrict__ in, float* __restrict__ out, int ld,
                                                         const int* __restrict__ starts, const int* __restrict__ lens, int D) {
    const int u = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (c >= D) return;
    const int s0 = starts[u], L = lens[u];
    float acc = 0.f;
    for (int t = 0; t < L; ++t) acc += in[(size_t)(s0 + t) * ld + c];
    const float mean = acc / (float)L;
    for (int t = 0; t < L; ++t) out[(size_t)(s0 + t) * ld + c] = in[(size_t)(s0 + t) * ld + c] - mean;
}

int prep_mean_image_subtraction(const float* in, float* out, int ld, const int* starts, const int* lens, int n_utt, int D,
                                hipStream_t s) {
    ADN_CHECK(in && out && starts && lens, ADN_ERR_INVALID, "prep_mean_image_subtraction: null argument");
    if (n_utt <= 0 || D <= 0) return ADN_OK;
    hipLaunchKernelGGL(mean_image_kernel, dim3(n_utt, cdiv(D, 256)), dim3(256), 0, s, in, out, ld, starts, lens, D);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// per-frame z-normalisation, population std, in place (utils/preprocessing.py:218-242): one wave per row
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void normalize_rows_kernel(float* __restrict__ x, int ld, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += gridDim.x * 4) {
        float* row = x + (size_t)r * ld;
        float s = 0.f;
        for (int c = lane; c < cols; c += 64) s += row[c];
        const float mean = wave_sum_f(s) / (float)cols;
        float q = 0.f;
        for (int c = lane; c < cols; c += 64) { const float d = row[c] - mean; q += d * d; }
        const float inv = 1.f / sqrtf(wave_sum_f(q) / (float)cols);
        for (int c = lane; c < cols; c += 64) row[c] = (row[c] - mean) * inv;
    }
}

int prep_normalize_rows(float* x, int ld, int rows, int cols, hipStream_t s) {
    ADN_CHECK(x, ADN_ERR_INVALID, "prep_normalize_rows: null argument");
    if (rows <= 0 || cols <= 0) return ADN_OK;
    const int grid = std::max(1, std::min(cdiv(rows, 4), 8192));
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(grid), dim3(256), 0, s, x, ld, rows, cols);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// column statistics over all frames (utils/preprocessing.py:245-257): mean, then the population std of the centred
// data; fp64 accumulators (the frame count is in the tens of thousands).  ws: 2*cols doubles, zeroed here.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void col_accum_kernel(const float* __restrict__ x, int ld, int rows, int cols,
                                                        const float* __restrict__ mean, double* __restrict__ acc, int rows_per_split) {
    __shared__ double part[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * rows_per_split, r1 = min(rows, r0 + rows_per_split);
    double a = 0.0;
    if (c < cols) {
        const float mu = mean ? mean[c] : 0.f;
        for (int r = r0 + rl; r < r1; r += 4) {
            const float v = x[(size_t)r * ld + c] - mu;          // second pass: centred in fp32 exactly as the reference
            a += mean ? (double)v * (double)v : (double)v;
        }
    }
    part[rl][cl] = a;
    __syncthreads();
    if (rl == 0 && c < cols) atomicAdd(acc + c, part[0][cl] + part[1][cl] + part[2][cl] + part[3][cl]);
}

__global__ void col_finish_kernel(const double* __restrict__ acc, int cols, int rows, int is_std, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < cols) out[c] = is_std ? (float)sqrt(acc[c] / (double)rows) : (float)(acc[c] / (double)rows);
}

int prep_column_stats(const float* x, int ld, int rows, int cols, double* ws, float* mean, float* std, hipStream_t s) {
    ADN_CHECK(x && ws && mean && std, ADN_ERR_INVALID, "prep_column_stats: null argument");
    if (rows <= 0 || cols <= 0) return ADN_OK;
    ADN_HIP_CHECK(hipMemsetAsync(ws, 0, (size_t)2 * cols * sizeof(double), s));
    const int ctiles = cdiv(cols, 64);
    int splits = std::max(1, std::min(cdiv(rows, 64), cdiv(2048, ctiles)));
    const int rps = cdiv(rows, splits);
    splits = cdiv(rows, rps);
    hipLaunchKernelGGL(col_accum_kernel, dim3(ctiles, splits), dim3(256), 0, s, x, ld, rows, cols, (const float*)nullptr, ws, rps);
    hipLaunchKernelGGL(col_finish_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, s, ws, cols, rows, 0, mean);
    hipLaunchKernelGGL(col_accum_kernel, dim3(ctiles, splits), dim3(256), 0, s, x, ld, rows, cols, (const float*)mean, ws + cols, rps);
    hipLaunchKernelGGL(col_finish_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, s, ws + cols, cols, rows, 1, std);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// out = (x - mean) / std per column (train statistics applied to val / test, runners/3stream.py:102-108)
__global__ __launch_bounds__(256) void apply_col_norm_kernel(const float* __restrict__ x, float* __restrict__ out, int ld, int rows,
                                                             int cols, const float* __restrict__ mean, const float* __restrict__ std) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / cols), c = (int)(e % cols);
        out[(size_t)r * ld + c] = (x[(size_t)r * ld + c] - mean[c]) / std[c];
    }
}

int prep_apply_column_norm(const float* x, float* out, int ld, int rows, int cols, const float* mean, const float* std,
                           hipStream_t s) {
    ADN_CHECK(x && out && mean && std, ADN_ERR_INVALID, "prep_apply_column_norm: null argument");
    if (rows <= 0 || cols <= 0) return ADN_OK;
    const int64_t total = (int64_t)rows * cols;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 16384));
    hipLaunchKernelGGL(apply_col_norm_kernel, dim3(grid), dim3(256), 0, s, x, out, ld, rows, cols, mean, std);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// out[r][j] = in[r][perm[j]]: pixel re-ordering (utils/preprocessing.py:492-503) and coefficient selection
__global__ __launch_bounds__(256) void gather_cols_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out,
                                                          int ld_out, const int* __restrict__ perm, int rows, int cols) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / cols), c = (int)(e % cols);
        out[(size_t)r * ld_out + c] = in[(size_t)r * ld_in + perm[c]];
    }
}

int prep_gather_columns(const float* in, int ld_in, float* out, int ld_out, const int* perm, int rows, int cols, hipStream_t s) {
    ADN_CHECK(in && out && perm && in != out, ADN_ERR_INVALID, "prep_gather_columns: null or aliased argument");
    if (rows <= 0 || cols <= 0) return ADN_OK;
    const int64_t total = (int64_t)rows * cols;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 16384));
    hipLaunchKernelGGL(gather_cols_kernel, dim3(grid), dim3(256), 0, s, in, ld_in, out, ld_out, perm, rows, cols);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// LeCun local contrast normalisation of single-channel images (reference utils/lcn.py:24-61, 64-104):
//   blur(Z)[y][x] = sum_{i,j} f[i][j] Z[y + mid - i][x + mid - j]     ('full' convolution cropped by mid = k / 2: zero
//                                                                      outside the image; f = normalised k x k Gaussian)
//   cen   = X - blur(X);   den = sqrt(blur(cen^2))
//   div   = max(max(colmean[x], den), threshold)  with colmean[x] = mean over the ROWS of den[:, x]  (the reference's
//           `denom.mean(axis=[1, 2])` of a (B, 1, H, W) tensor averages channel and rows, not the whole image)
//   out   = cen / div
// One workgroup per image: the image (later den), cen and a row-filtered scratch live in LDS (3 H W floats <= 96 KB), every
// global byte is read and written once, coalesced.  A separable filter runs as 1-d row and column passes (2 k taps
// per pixel and blur instead of k^2).
// ---------------------------------------------------------------------------------------------------------
struct LcnFilter { float w[kLcnMaxK * kLcnMaxK]; float row[kLcnMaxK], col[kLcnMaxK]; };

// blur of the LDS image `in` at pixel (py, px): SEP = false the k x k taps; SEP = true `in` already holds the horizontal
// pass and only the k column taps remain
template <bool SEP>
__device__ __forceinline__ float lcn_blur(const float* in, const LcnFilter& f, int H, int W, int k, int mid, int py, int px) {
    float b = 0.f;
    for (int i = 0; i < k; ++i) {
        const int yy = py + mid - i;
        if (yy < 0 || yy >= H) continue;
        if (SEP) { b += f.col[i] * in[yy * W + px]; continue; }
        for (int j = 0; j < k; ++j) {
            const int xx = px + mid - j;
            if (xx >= 0 && xx < W) b += f.w[i * k + j] * in[yy * W + xx];
        }
    }
    return b;
}

// horizontal pass of a separable filter: out[y][x] = sum_j row[j] (SQ ? in^2 : in)[y][x + mid - j]
template <bool SQ>
__device__ __forceinline__ void lcn_rows(const float* in, float* out, const LcnFilter& f, int HW, int W, int k, int mid) {
    for (int e = threadIdx.x; e < HW; e += 256) {
        const int px = e % W, base = e - px;
        float b = 0.f;
        for (int j = 0; j < k; ++j) {
            const int xx = px + mid - j;
            if (xx >= 0 && xx < W) { const float v = in[base + xx]; b += f.row[j] * (SQ ? v * v : v); }
        }
        out[e] = b;
    }
}

template <bool SEP>
__global__ __launch_bounds__(256) void lcn_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int k,
                                                  float threshold, const LcnFilter f) {
    extern __shared__ float lcn_lds[];
    float* img = lcn_lds;                 // X, later den
    float* cen = lcn_lds + H * W;
    float* tmp = cen + H * W;             // SEP: row-filtered image; generic: unused
    float* colmean = tmp + (SEP ? H * W : 0);
    const int HW = H * W, mid = k / 2;
    const float* src = x + (size_t)blockIdx.x * HW;
    for (int e = threadIdx.x; e < HW; e += 256) img[e] = src[e];
    __syncthreads();
    if (SEP) { lcn_rows<false>(img, tmp, f, HW, W, k, mid); __syncthreads(); }
    for (int e = threadIdx.x; e < HW; e += 256)
        cen[e] = img[e] - lcn_blur<SEP>(SEP ? tmp : img, f, H, W, k, mid, e / W, e % W);
    __syncthreads();
    if (SEP) {
        lcn_rows<true>(cen, tmp, f, HW, W, k, mid);
        __syncthreads();
        for (int e = threadIdx.x; e < HW; e += 256) img[e] = sqrtf(fmaxf(lcn_blur<true>(tmp, f, H, W, k, mid, e / W, e % W), 0.f));
    } else {
        float den_r[kLcnMaxPerThread];
        int n = 0;
        for (int e = threadIdx.x; e < HW; e += 256, ++n) {
            const int py = e / W, px = e % W;
            float b = 0.f;
            for (int i = 0; i < k; ++i) {
                const int yy = py + mid - i;
                if (yy < 0 || yy >= H) continue;
                for (int j = 0; j < k; ++j) {
                    const int xx = px + mid - j;
                    if (xx >= 0 && xx < W) { const float c = cen[yy * W + xx]; b += f.w[i * k + j] * (c * c); }
                }
            }
            den_r[n] = sqrtf(fmaxf(b, 0.f));
        }
        n = 0;
        for (int e = threadIdx.x; e < HW; e += 256, ++n) img[e] = den_r[n];     // X is no longer needed
    }
    __syncthreads();
    for (int c = threadIdx.x; c < W; c += 256) {
        float a = 0.f;
        for (int r = 0; r < H; ++r) a += img[r * W + c];
        colmean[c] = a / (float)H;
    }
    __syncthreads();
    float* dst = y + (size_t)blockIdx.x * HW;
    for (int e = threadIdx.x; e < HW; e += 256)
        dst[e] = cen[e] / fmaxf(fmaxf(colmean[e % W], img[e]), threshold);
}

int prep_lcn(const float* x, float* y, int n_images, int H, int W, const float* filter_host, int k, float threshold, hipStream_t s) {
    ADN_CHECK(x && y && filter_host, ADN_ERR_INVALID, "prep_lcn: null argument");
    ADN_CHECK(k >= 1 && k <= kLcnMaxK && (k & 1), ADN_ERR_INVALID, "prep_lcn: the kernel size must be odd and <= 15");
    ADN_CHECK(H >= 1 && W >= 1 && (int64_t)H * W <= 256 * kLcnMaxPerThread, ADN_ERR_INVALID, "prep_lcn: image larger than 8192 pixels");
    if (n_images <= 0) return ADN_OK;
    LcnFilter f;
    for (int e = 0; e < k * k; ++e) f.w[e] = filter_host[e];
    // a rank-1 filter (the reference's Gaussian is one: exp(-(a^2 + b^2) / 2 sigma^2) = g(a) g(b)) runs as two 1-d passes
    const int mid = k / 2;
    const float centre = filter_host[mid * k + mid];
    bool sep = centre > 0.f;
    if (sep) {
        const float r = sqrtf(centre);
        for (int i = 0; i < k; ++i) { f.col[i] = filter_host[i * k + mid] / r; f.row[i] = filter_host[mid * k + i] / r; }
        float fmax = 0.f;
        for (int e = 0; e < k * k; ++e) fmax = std::max(fmax, std::fabs(filter_host[e]));
        for (int i = 0; i < k && sep; ++i)
            for (int j = 0; j < k; ++j)
                if (std::fabs(f.col[i] * f.row[j] - filter_host[i * k + j]) > 1e-6f * fmax) { sep = false; break; }
    }
    const size_t lds = ((size_t)(sep ? 3 : 2) * H * W + W) * sizeof(float);
    if (sep) {
        ADN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lcn_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(lcn_kernel<true>, dim3(n_images), dim3(256), lds, s, x, y, H, W, k, threshold, f);
    } else {
        ADN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lcn_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(lcn_kernel<false>, dim3(n_images), dim3(256), lds, s, x, y, H, W, k, threshold, f);
    }
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

}  // namespace adn
