// lasagne.layers.BatchNormLayer over the rows of a [rows][C] matrix (row stride ld): the dense form the reference uses
// behind the encoder bottleneck (modelzoo/adenet_v1.py:82: axes = (0,), one statistic per feature over all B*T frames,
// padded ones included) and, with NHWC activations, the per-channel form of the convolutional auto-encoder variants
// (modelzoo/avletters_convae_bn.py:49-59: axes = (0, 2, 3) of a (B, C, H, W) tensor = the columns of [B*H*W][C]).
//
//   training pass (get_output(..., deterministic=False)):
//       mean = E[x], var = E[(x - mean)^2] (biased), inv_std = 1 / sqrt(var + eps)          eps = 1e-4
//       y = (x - mean) * inv_std * gamma + beta
//       running mean    <- (1 - alpha) running mean    + alpha mean                         alpha = 0.1
//       running inv_std <- (1 - alpha) running inv_std + alpha inv_std      (Lasagne averages inv_std itself)
//   deterministic pass: y = (x - running mean) * running inv_std * gamma + beta
//   backward of the training pass (statistics are functions of x):
//       dbeta = sum dy,  dgamma = sum dy xhat,  dx = gamma inv_std (dy - dbeta / rows - xhat dgamma / rows)
//
// HBM-bound streaming kernels: statistics through the fp64 column accumulators of prep.hip (two passes: mean, then the
// centred second moment), everything else one float per lane and pass.
#include "adn_common.h"
#include <algorithm>

namespace adn {

namespace {

__global__ __launch_bounds__(256) void bn_finish_stats_kernel(const float* __restrict__ std_in, float* __restrict__ inv_std,
                                                              const float* __restrict__ mean, float* __restrict__ run_mean,
                                                              float* __restrict__ run_inv_std, int C, float eps, float alpha) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float s = std_in[c];
    const float is = 1.f / sqrtf(s * s + eps);
    inv_std[c] = is;
    if (run_mean) {
        run_mean[c] = (1.f - alpha) * run_mean[c] + alpha * mean[c];
        run_inv_std[c] = (1.f - alpha) * run_inv_std[c] + alpha * is;
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int rows,
                                                       int C, const float* __restrict__ mean, const float* __restrict__ inv_std,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       __bf16* __restrict__ y16) {
    const int64_t total = (int64_t)rows * C;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / C), c = (int)(e % C);
        const float v = (x[(size_t)r * ldx + c] - mean[c]) * inv_std[c] * gamma[c] + beta[c];
        y[(size_t)r * ldy + c] = v;
        if (y16) y16[(size_t)r * ldy + c] = (__bf16)v;
    }
}

// per-column sums of dy and dy * xhat: 32 columns x 8 row lanes per workgroup, rows split over gridDim.y, float atomics
// into a zeroed [2][C] workspace (C is small: 50 ... 3000; rows up to a few 10^4)
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy,
                                                          int rows, int C, const float* __restrict__ mean,
                                                          const float* __restrict__ inv_std, float* __restrict__ sums, int rps) {
    __shared__ float red[2][8][33];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const int r0 = blockIdx.y * rps, r1 = min(rows, r0 + rps);
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        const float mu = mean[c], is = inv_std[c];
        for (int r = r0 + rl; r < r1; r += 8) {
            const float g = dy[(size_t)r * lddy + c];
            s0 += g;
            s1 += g * (x[(size_t)r * ldx + c] - mu) * is;
        }
    }
    red[0][rl][cl] = s0; red[1][rl][cl] = s1;
    __syncthreads();
    if (rl == 0 && c < C) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { a += red[0][k][cl]; b += red[1][k][cl]; }
        atomicAdd(&sums[c], a);
        atomicAdd(&sums[C + c], b);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy,
                                                           float* __restrict__ dx, int lddx, int rows, int C,
                                                           const float* __restrict__ mean, const float* __restrict__ inv_std,
                                                           const float* __restrict__ gamma, const float* __restrict__ sums,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, __bf16* __restrict__ dx16,
                                                           int batch_stats) {
    const int64_t total = (int64_t)rows * C;
    const float inv_n = batch_stats ? 1.f / (float)rows : 0.f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / C), c = (int)(e % C);
        const float is = inv_std[c];
        const float xhat = (x[(size_t)r * ldx + c] - mean[c]) * is;
        const float v = gamma[c] * is * (dy[(size_t)r * lddy + c] - sums[c] * inv_n - xhat * sums[C + c] * inv_n);
        dx[(size_t)r * lddx + c] = v;
        if (dx16) dx16[(size_t)r * lddx + c] = (__bf16)v;
    }
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += 256) {
            if (dbeta) dbeta[c] += sums[c];
            if (dgamma) dgamma[c] += sums[C + c];
        }
}

int grid_of(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 16384)); }

}  // namespace

size_t batchnorm_ws_bytes(int C) { return (size_t)2 * C * sizeof(double) + (size_t)4 * C * sizeof(float); }

// ws: batchnorm_ws_bytes(C).  save_mean / save_inv_std: the batch statistics (kept for the backward pass).
// run_mean / run_inv_std: the layer's running averages, updated in place (null: not updated).
int batchnorm_forward_train(const float* x, int ldx, float* y, int ldy, int rows, int C, const float* gamma, const float* beta,
                            float eps, float alpha, float* save_mean, float* save_inv_std, float* run_mean, float* run_inv_std,
                            void* ws, hipStream_t s, void* y16) {
    ADN_CHECK(x && y && gamma && beta && save_mean && save_inv_std && ws && rows > 0 && C > 0, ADN_ERR_INVALID, "batchnorm: bad argument");
    double* acc = static_cast<double*>(ws);
    float* std_tmp = reinterpret_cast<float*>(acc + 2 * (size_t)C);
    ADN_TRY(prep_column_stats(x, ldx, rows, C, acc, save_mean, std_tmp, s));
    hipLaunchKernelGGL(bn_finish_stats_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, std_tmp, save_inv_std, save_mean, run_mean,
                       run_inv_std, C, eps, alpha);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_of((int64_t)rows * C)), dim3(256), 0, s, x, ldx, y, ldy, rows, C, save_mean,
                       save_inv_std, gamma, beta, reinterpret_cast<__bf16*>(y16));
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int batchnorm_forward_eval(const float* x, int ldx, float* y, int ldy, int rows, int C, const float* gamma, const float* beta,
                           const float* run_mean, const float* run_inv_std, hipStream_t s, void* y16) {
    ADN_CHECK(x && y && gamma && beta && run_mean && run_inv_std && rows > 0 && C > 0, ADN_ERR_INVALID, "batchnorm: bad argument");
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_of((int64_t)rows * C)), dim3(256), 0, s, x, ldx, y, ldy, rows, C, run_mean,
                       run_inv_std, gamma, beta, reinterpret_cast<__bf16*>(y16));
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// dx may alias dy.  dgamma / dbeta are accumulated into (+=).
int batchnorm_backward(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, int rows, int C, const float* gamma,
                       const float* save_mean, const float* save_inv_std, int batch_stats, float* dgamma, float* dbeta, void* ws,
                       hipStream_t s, void* dx16) {
    ADN_CHECK(x && dy && dx && gamma && save_mean && save_inv_std && ws && rows > 0 && C > 0, ADN_ERR_INVALID, "batchnorm: bad argument");
    float* sums = reinterpret_cast<float*>(static_cast<double*>(ws) + 2 * (size_t)C) + C;     // behind std_tmp
    ADN_HIP_CHECK(hipMemsetAsync(sums, 0, (size_t)2 * C * sizeof(float), s));
    const int ctiles = cdiv(C, 32);
    int splits = std::max(1, std::min(cdiv(rows, 64), cdiv(1024, ctiles)));
    if (deterministic()) splits = 1;                 // one add per channel: no arrival order
    const int rps = cdiv(rows, splits);
    splits = cdiv(rows, rps);
    hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(ctiles, splits), dim3(256), 0, s, x, ldx, dy, lddy, rows, C, save_mean, save_inv_std,
                       sums, rps);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_of((int64_t)rows * C)), dim3(256), 0, s, x, ldx, dy, lddy, dx, lddx, rows, C,
                       save_mean, save_inv_std, gamma, sums, dgamma, dbeta, reinterpret_cast<__bf16*>(dx16), batch_stats);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

}  // namespace adn
