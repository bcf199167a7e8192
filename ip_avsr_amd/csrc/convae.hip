// Convolutional auto-encoder (reference modelzoo/avletters_convae.py:33-69, its BatchNorm / dropout
// variants avletters_convae_{bn,drop,bndrop}.py and the training step avletters/avletters_convae.py:254-262): im2col + MFMA
// GEMM for every convolution and its adjoint; BatchNorm through csrc/batchnorm.hip (NHWC: a channel is a column), dropout
// masks from the model's counter-based hash.
//
//   (B,1,30,40) -> conv 5x5 (100) -> maxpool 2 -> conv 5x5 (150) -> maxpool 2 pad (1,0) -> conv 3x3 (200) -> 3000
//               -> dense 500 -> bottleneck 50 (linear) -> dense8 (W_b^T) -> dense9 (W_7^T) -> (200,3,5)
//               -> deconv (conv5.W) -> upscale 2 -> deconv (conv3.W) -> upscale 2 -> deconv crop (1,0) (conv1.W) -> 1200
//
// Layout: activations are NHWC fp32 ([B][H][W][C]), so a convolution's output IS the GEMM result
// [B*OH*OW][C_out] and its patches matrix is [B*OH*OW][kh*kw*C_in] with (i, j, c) column order.  Weights are kept in
// the matching GEMM form Wm[(i*kw + j)*C_in + c][o] = W[o][c][kh-1-i][kw-1-j] (Lasagne's W, filters flipped: true
// convolution); dense7 rows are in (h, w, c) order.  The C ABI converts to / from Lasagne's layouts on the host, so
// checkpoints and the oracle see (out, in, kh, kw) and (c*h*w, units).
//   conv forward    cols = im2col(x);  y = act(cols Wm + b)                                GEMM NN
//   conv backward   dWm += cols^T dy;  db += colsum(dy);  dx = col2im(dy Wm^T)              GEMM TN, NT
//   deconv forward  z = act(col2im(x Wm^T) + b)       (exact adjoint of the tied convolution)  GEMM NT
//   deconv backward cols' = im2col(dz padded by crop);  dx = cols' Wm;  dWm += cols'^T x     GEMM NN, TN
// Every GEMM goes through gemm() (fp32 MFMA, or bf16 MFMA converting in flight with precision = bf16).
#include "adn_common.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

using namespace adn;

namespace {

// Filters per convolution: 100 / 150 / 200 (modelzoo/avletters_convae.py:34-36), 125 / 300 / 400 in the dropout variant
// (avletters_convae_drop.py:36-38).  The GEMM loaders need row strides that are multiples of 4 floats (8 for the bf16
// operands): channel counts that are not multiples of 4 are padded to the next multiple of 8 with zero channels (zero
// filters / zero bias / zero BatchNorm beta and gamma give 0, and nothing ever flows into or out of them).
int pad_channels(int f, bool first) { return (first && f % 4 == 0) ? f : (int)round_up(f, 8); }   // (the first layer's C_in is 1: f32 path)

__device__ __forceinline__ float cae_act(int act, float v) {
    if (act == ADN_ACT_SCALED_TANH) return 2.4f * tanhf(0.5f * v);
    if (act == ADN_ACT_SCALED_TANH_LECUN) return 1.7159f * tanhf((2.f / 3.f) * v);
    return v;
}
__device__ __forceinline__ bool cae_keep(uint32_t key, uint32_t idx, float p) {       // = elementwise.hip::dropout_keep
    uint32_t x = idx * 0x9E3779B1u + key;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return (float)(x >> 8) * (1.0f / 16777216.0f) >= p;
}

// ---------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------
// The first convolution has ONE input channel: its patches are 25 pixels of the frame, a GEMM over them (K = 25) spends its time
// writing and re-reading a patch matrix 25 times the size of its input and runs at 20 TFLOP/s.  Direct form, no patch matrix: a
// workgroup takes frames into LDS (4.8 KB each); thread (channel quad cq, pixel lane pl) keeps its 25 x 4 filter taps in
// registers and walks the output pixels pl, pl + lanes, ...: 25 LDS reads (broadcast) + 100 fp32 FMAs + one 16-byte store per
// pixel; a pixel's O channels leave as one contiguous row.  fp32 arithmetic throughout (the GEMM path rounds the operands to
// bfloat16 in bf16 mode).   y[b][oy][ox][o] = act(bias[o] + sum_{i,j} x[b][oy + i][ox + j] Wm[i k + j][o])
// (tanh through exp2 + rcp, 2 ulp-ish: with tanhf the kernel is bound by the activation's ~40 instructions per value -- 156 us,
//  twice its store time)
__device__ __forceinline__ float cae_act_fast(int act, float v) {
    if (act == ADN_ACT_LINEAR) return v;
    const float s_in = act == ADN_ACT_SCALED_TANH ? 0.5f : (2.f / 3.f), s_out = act == ADN_ACT_SCALED_TANH ? 2.4f : 1.7159f;
    const float t = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.88539008177792681f * s_in * v));
    return s_out * t;
}

template <int KK>
__global__ __launch_bounds__(256) void conv1_direct_fwd_kernel(const float* __restrict__ x, const float* __restrict__ Wm,
                                                               const float* __restrict__ bias, float* __restrict__ y, int B, int H,
                                                               int W, int k, int O, int OH, int OW, int act) {
    extern __shared__ float img[];                       // [H][W]
    const int tid = threadIdx.x, Q = O / 4, lanes = 256 / Q;
    const int cq = tid % Q, pl = tid / Q;
    float4 w[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) w[t] = pl < lanes ? *reinterpret_cast<const float4*>(Wm + (size_t)t * O + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 bq = pl < lanes ? *reinterpret_cast<const float4*>(bias + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();                                 // (the previous frame has been consumed)
        for (int e = tid; e < H * W; e += 256) img[e] = x[(size_t)b * H * W + e];
        __syncthreads();
        if (pl >= lanes) continue;
        float* yb = y + (size_t)b * OH * OW * O + 4 * cq;
        for (int p = pl; p < OH * OW; p += lanes) {
            const int oy = p / OW, ox = p - oy * OW;
            const float* ip = img + oy * W + ox;
            float4 a = bq;
#pragma unroll
            for (int t = 0; t < KK; ++t) {
                const float xv = ip[(t / 5) * W + (t % 5)];      // (KK = 25: a 5 x 5 window)
                a.x += xv * w[t].x; a.y += xv * w[t].y; a.z += xv * w[t].z; a.w += xv * w[t].w;
            }
            a.x = cae_act_fast(act, a.x); a.y = cae_act_fast(act, a.y); a.z = cae_act_fast(act, a.z); a.w = cae_act_fast(act, a.w);
            *reinterpret_cast<float4*>(yb + (size_t)p * O) = a;
        }
    }
}

// ... and its weight gradient: dW[i k + j][o] = sum_{b, oy, ox} x[b][oy + i][ox + j] g[b][oy][ox][o].  Same thread map; a thread
// accumulates its 25 x 4 taps over the frames of its workgroup, the pixel lanes are then summed in lane order through LDS and the
// workgroup's 25 x O block goes to its own slot (conv1_direct_dw_reduce_kernel adds the slots in order: no atomics anywhere).
template <int KK>
__global__ __launch_bounds__(256) void conv1_direct_dw_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                              float* __restrict__ slots, int B, int H, int W, int k, int O, int OH, int OW) {
    extern __shared__ float sm[];                        // [H][W] frame, then [KK][O] block sums
    float* img = sm;
    float* red = sm + H * W;
    const int tid = threadIdx.x, Q = O / 4, lanes = 256 / Q;
    const int cq = tid % Q, pl = tid / Q;
    float4 acc[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        for (int e = tid; e < H * W; e += 256) img[e] = x[(size_t)b * H * W + e];
        __syncthreads();
        if (pl >= lanes) continue;
        const float* gb = g + (size_t)b * OH * OW * O + 4 * cq;
        for (int p0 = pl; p0 < OH * OW; p0 += 4 * lanes) {       // four gradient rows in flight per thread (one alone: a round trip per pixel)
            float4 gv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = p0 + u * lanes;
                gv[u] = p < OH * OW ? *reinterpret_cast<const float4*>(gb + (size_t)p * O) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = min(p0 + u * lanes, OH * OW - 1);     // (a pixel past the end multiplies a zero gradient)
                const int oy = p / OW, ox = p - oy * OW;
                const float* ip = img + oy * W + ox;
#pragma unroll
                for (int t = 0; t < KK; ++t) {
                    const float xv = ip[(t / 5) * W + (t % 5)];
                    acc[t].x += xv * gv[u].x; acc[t].y += xv * gv[u].y; acc[t].z += xv * gv[u].z; acc[t].w += xv * gv[u].w;
                }
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < KK * O; e += 256) red[e] = 0.f;
    for (int l = 0; l < lanes; ++l) {                    // pixel lanes in order: a fixed summation order
        __syncthreads();
        if (pl == l) {
#pragma unroll
            for (int t = 0; t < KK; ++t) {
                float4* r4 = reinterpret_cast<float4*>(red + t * O + 4 * cq);
                float4 v = *r4;
                v.x += acc[t].x; v.y += acc[t].y; v.z += acc[t].z; v.w += acc[t].w;
                *r4 = v;
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < KK * O; e += 256) slots[(size_t)blockIdx.x * KK * O + e] = red[e];
}

// 64 outputs per workgroup x 4 slot quarters: a thread adds its quarter's slots in order, the quarters are added in order
// (elements [0, nW) of a slot belong to dW, [nW, nW + nb) -- the fused forms' bias sums -- to db, the rest -- the column sums of an
//  input gradient, i.e. the NEXT layer's bias gradient -- to dc)
__global__ __launch_bounds__(256) void conv1_direct_dw_reduce_kernel(const float* __restrict__ slots, int nslots, int n, float* __restrict__ dW,
                                                                     int nW, float* __restrict__ db, int nb = 1 << 30,
                                                                     float* __restrict__ dc = nullptr) {
    __shared__ float part[4][64];
    const int o = threadIdx.x & 63, q = threadIdx.x >> 6, e = blockIdx.x * 64 + o;
    const int per = (nslots + 3) / 4;
    float v = 0.f;
    if (e < n)
        for (int s_ = q * per; s_ < min(nslots, (q + 1) * per); ++s_) v += slots[(size_t)s_ * n + e];
    part[q][o] = v;
    __syncthreads();
    if (q == 0 && e < n) {
        const float sum = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
        if (e < nW) dW[e] += sum;                        // (the tied deconvolution adds to the same gradient)
        else if (e - nW < nb) db[e - nW] += sum;
        else dc[e - nW - nb] += sum;
    }
}

// cols[(b, oy, ox)][(i*kw + j)*C + c] = x[b][oy + i - ph][ox + j - pw][c]  (0 outside); row stride ldc
// `up` = 1: the patch rows belong to a 2x upscaled grid and are wanted SUMMED over every 2 x 2 block of it (OH, OW = the
// compact grid): row (b, oy, ox) = sum over dy, dx in {0, 1} of the patch at (2 oy + dy, 2 ox + dx) -- see deconv_bwd
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, float* __restrict__ cols, int B, int H, int W,
                                                     int C, int kh, int kw, int ph, int pw, int OH, int OW, int ldc, int up) {
    const int K = kh * kw * C;
    const int64_t total = (int64_t)B * OH * OW * K;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int k = (int)(e % K);
        const int64_t r = e / K;
        const int c = k % C, ij = k / C, j = ij % kw, i = ij / kw;
        const int ox = (int)(r % OW), oy = (int)((r / OW) % OH), b = (int)(r / ((int64_t)OW * OH));
        float v = 0.f;
        for (int dy = 0; dy <= up; ++dy)
            for (int dx = 0; dx <= up; ++dx) {
                const int y = (up ? 2 * oy + dy : oy) + i - ph, xx = (up ? 2 * ox + dx : ox) + j - pw;
                if (y >= 0 && y < H && xx >= 0 && xx < W) v += x[(((size_t)b * H + y) * W + xx) * C + c];
            }
        cols[(size_t)r * ldc + k] = v;
    }
}

// out[b][y][x][c] = sum_{i,j} dcols[(b, y + ph - i, x + pw - j)][(i*kw + j)*C + c]  over valid patch positions
// (gather form of col2im: no atomics); optional bias[c] and activation
// `up` = 1: the patch rows exist for the compact (OH / 2) x (OW / 2) grid only and every row stands for the 2 x 2 block of the
// upscaled grid it was repeated into (Upscale2DLayer folded into the gather: x_up W^T = repeat(x W^T))
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcols, int ldc, float* __restrict__ out, int B, int H,
                                                     int W, int C, int kh, int kw, int ph, int pw, int OH, int OW,
                                                     const float* __restrict__ bias, int act, int up) {
    const int64_t total = (int64_t)B * H * W * C;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const int64_t p = e / C;
        const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((int64_t)W * H));
        float acc = bias ? bias[c] : 0.f;
        for (int i = 0; i < kh; ++i) {
            const int oy = y + ph - i;
            if (oy < 0 || oy >= OH) continue;
            for (int j = 0; j < kw; ++j) {
                const int ox = x + pw - j;
                if (ox < 0 || ox >= OW) continue;
                acc += dcols[(up ? ((size_t)b * (OH / 2) + oy / 2) * (OW / 2) + ox / 2 : ((size_t)b * OH + oy) * OW + ox) * ldc + (i * kw + j) * C + c];
            }
        }
        out[e] = cae_act(act, acc);
    }
}

// the same two kernels for C % 4 == 0, four channels per thread (float4 both ways, a quarter of the index arithmetic)
__global__ __launch_bounds__(256) void im2col4_kernel(const float4* __restrict__ x, float4* __restrict__ cols, int B, int H, int W,
                                                      int C4, int kh, int kw, int ph, int pw, int OH, int OW, int ldc4, int up) {
    const int K4 = kh * kw * C4;
    const int64_t total = (int64_t)B * OH * OW * K4;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int k = (int)(e % K4);
        const int64_t r = e / K4;
        const int c = k % C4, ij = k / C4, j = ij % kw, i = ij / kw;
        const int ox = (int)(r % OW), oy = (int)((r / OW) % OH), b = (int)(r / ((int64_t)OW * OH));
        float4 v = z;
        for (int dy = 0; dy <= up; ++dy)
            for (int dx = 0; dx <= up; ++dx) {
                const int y = (up ? 2 * oy + dy : oy) + i - ph, xx = (up ? 2 * ox + dx : ox) + j - pw;
                if (y >= 0 && y < H && xx >= 0 && xx < W) {
                    const float4 t = x[(((size_t)b * H + y) * W + xx) * C4 + c];
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
            }
        cols[(size_t)r * ldc4 + k] = v;
    }
}

__global__ __launch_bounds__(256) void col2im4_kernel(const float4* __restrict__ dcols, int ldc4, float4* __restrict__ out, int B,
                                                      int H, int W, int C4, int kh, int kw, int ph, int pw, int OH, int OW,
                                                      const float4* __restrict__ bias, int act, int up) {
    const int64_t total = (int64_t)B * H * W * C4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % C4);
        const int64_t p = e / C4;
        const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((int64_t)W * H));
        float4 acc = bias ? bias[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < kh; ++i) {
            const int oy = y + ph - i;
            if (oy < 0 || oy >= OH) continue;
            for (int j = 0; j < kw; ++j) {
                const int ox = x + pw - j;
                if (ox < 0 || ox >= OW) continue;
                const float4 v = dcols[(up ? ((size_t)b * (OH / 2) + oy / 2) * (OW / 2) + ox / 2 : ((size_t)b * OH + oy) * OW + ox) * ldc4 + (i * kw + j) * C4 + c];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        acc.x = cae_act(act, acc.x); acc.y = cae_act(act, acc.y); acc.z = cae_act(act, acc.z); acc.w = cae_act(act, acc.w);
        out[e] = acc;
    }
}

// im2col straight to bf16 (C % 4 == 0): the patches matrix of the bf16 GEMMs, half the bytes of the fp32 one
typedef __bf16 cae_bf16x4 __attribute__((ext_vector_type(4)));

// col2im4_kernel reading a bf16 matrix (round 4): the product dy Wm^T leaves its GEMM as bf16 only -- the largest one, the
// input gradient of the second convolution at batch 1024, is 1.29 GB as fp32 -- and is summed here in fp32
__global__ __launch_bounds__(256) void col2im4_bf16_kernel(const cae_bf16x4* __restrict__ dcols, int ldc4, float4* __restrict__ out, int B,
                                                           int H, int W, int C4, int kh, int kw, int ph, int pw, int OH, int OW,
                                                           const float4* __restrict__ bias, int act, int up) {
    const int64_t total = (int64_t)B * H * W * C4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % C4);
        const int64_t p = e / C4;
        const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((int64_t)W * H));
        float4 acc = bias ? bias[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (kw == 5) {                                  // a filter row's five taps requested together (one at a time behind its own
            for (int i = 0; i < kh; ++i) {              // bounds test: a round trip per tap, 2.6 TB/s)
                const int oy = y + ph - i;
                if (oy < 0 || oy >= OH) continue;
                cae_bf16x4 v[5];
                bool ok[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int ox = x + pw - j;
                    ok[j] = ox >= 0 && ox < OW;
                    const int oxc = ok[j] ? ox : 0;
                    v[j] = dcols[(up ? ((size_t)b * (OH / 2) + oy / 2) * (OW / 2) + oxc / 2 : ((size_t)b * OH + oy) * OW + oxc) * ldc4 + (i * 5 + j) * C4 + c];
                }
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    if (ok[j]) { acc.x += (float)v[j][0]; acc.y += (float)v[j][1]; acc.z += (float)v[j][2]; acc.w += (float)v[j][3]; }
            }
        } else {
        for (int i = 0; i < kh; ++i) {
            const int oy = y + ph - i;
            if (oy < 0 || oy >= OH) continue;
            for (int j = 0; j < kw; ++j) {
                const int ox = x + pw - j;
                if (ox < 0 || ox >= OW) continue;
                const cae_bf16x4 v = dcols[(up ? ((size_t)b * (OH / 2) + oy / 2) * (OW / 2) + ox / 2 : ((size_t)b * OH + oy) * OW + ox) * ldc4 + (i * kw + j) * C4 + c];
                acc.x += (float)v[0]; acc.y += (float)v[1]; acc.z += (float)v[2]; acc.w += (float)v[3];
            }
        }
        }
        acc.x = cae_act(act, acc.x); acc.y = cae_act(act, acc.y); acc.z = cae_act(act, acc.z); acc.w = cae_act(act, acc.w);
        out[e] = acc;
    }
}
__global__ __launch_bounds__(256) void im2col4_bf16_kernel(const float4* __restrict__ x, cae_bf16x4* __restrict__ cols, int B, int H,
                                                           int W, int C4, int kh, int kw, int ph, int pw, int OH, int OW, int ldc4, int up) {
    const int K4 = kh * kw * C4;
    const int64_t total = (int64_t)B * OH * OW * K4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int k = (int)(e % K4);
        const int64_t r = e / K4;
        const int c = k % C4, ij = k / C4, j = ij % kw, i = ij / kw;
        const int ox = (int)(r % OW), oy = (int)((r / OW) % OH), b = (int)(r / ((int64_t)OW * OH));
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int dy = 0; dy <= up; ++dy)
            for (int dx = 0; dx <= up; ++dx) {
                const int y = (up ? 2 * oy + dy : oy) + i - ph, xx = (up ? 2 * ox + dx : ox) + j - pw;
                if (y >= 0 && y < H && xx >= 0 && xx < W) {
                    const float4 t = x[(((size_t)b * H + y) * W + xx) * C4 + c];
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
            }
        cae_bf16x4 o; o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
        cols[(size_t)r * ldc4 + k] = o;
    }
}

// The same patches matrix for an unpadded, un-upscaled convolution, by RUNS: in NHWC the patch of output pixel r is kh runs of
// kw * C contiguous values (one per filter row), so a workgroup of 128 threads copies the runs of 4 output pixels with no index
// arithmetic per element (the generic kernel spends 6 integer divisions per 8-byte store and runs at half the HBM rate).
__global__ __launch_bounds__(128) void im2col_runs_bf16_kernel(const float4* __restrict__ x, cae_bf16x4* __restrict__ cols, int R, int H,
                                                               int W, int C4, int kh, int kw, int OH, int OW, int ldc4) {
    const int run4 = kw * C4;
    for (int rr = 0; rr < 4; ++rr) {
        const int r = blockIdx.x * 4 + rr;
        if (r >= R) return;
        const int ox = r % OW, oy = (r / OW) % OH, b = r / (OW * OH);
        const float4* src = x + (((size_t)b * H + oy) * W + ox) * C4;
        cae_bf16x4* dst = cols + (size_t)r * ldc4;
        for (int i = 0; i < kh; ++i)
            for (int q = threadIdx.x; q < run4; q += 128) {
                const float4 t = src[(size_t)i * W * C4 + q];
                cae_bf16x4 o; o[0] = (__bf16)t.x; o[1] = (__bf16)t.y; o[2] = (__bf16)t.z; o[3] = (__bf16)t.w;
                dst[i * run4 + q] = o;
            }
    }
}

// ... for the patches of a 2x upscaled grid summed over its 2 x 2 blocks (`up`, see im2col_kernel): the four source runs of a patch
// row are contiguous too (the generic kernel built this matrix at 1.2 TB/s)
__global__ __launch_bounds__(128) void im2col_runs_up_bf16_kernel(const float4* __restrict__ x, cae_bf16x4* __restrict__ cols, int R, int H,
                                                                  int W, int C4, int kh, int kw, int oh, int ow, int ldc4) {
    const int run4 = kw * C4;
    for (int rr = 0; rr < 4; ++rr) {
        const int r = blockIdx.x * 4 + rr;
        if (r >= R) return;
        const int ox = r % ow, oy = (r / ow) % oh, b = r / (ow * oh);
        const float4* src = x + (((size_t)b * H + 2 * oy) * W + 2 * ox) * C4;
        cae_bf16x4* dst = cols + (size_t)r * ldc4;
        for (int i = 0; i < kh; ++i)
            for (int q = threadIdx.x; q < run4; q += 128) {
                const float4* s0 = src + (size_t)i * W * C4 + q;
                const float4 a = s0[0], b1 = s0[C4], c = s0[(size_t)W * C4], d = s0[(size_t)W * C4 + C4];
                cae_bf16x4 o;
                o[0] = (__bf16)(((a.x + b1.x) + c.x) + d.x); o[1] = (__bf16)(((a.y + b1.y) + c.y) + d.y);
                o[2] = (__bf16)(((a.z + b1.z) + c.z) + d.z); o[3] = (__bf16)(((a.w + b1.w) + c.w) + d.w);
                dst[i * run4 + q] = o;
            }
    }
}

// ... from a bf16 copy of the input (the fused first layer leaves one beside its fp32 output): a plain copy of runs, half the bytes read
__global__ __launch_bounds__(128) void im2col_runs_from_bf16_kernel(const cae_bf16x4* __restrict__ x, cae_bf16x4* __restrict__ cols, int R,
                                                                    int H, int W, int C4, int kh, int kw, int OH, int OW, int ldc4) {
    const int run4 = kw * C4;
    for (int rr = 0; rr < 4; ++rr) {
        const int r = blockIdx.x * 4 + rr;
        if (r >= R) return;
        const int ox = r % OW, oy = (r / OW) % OH, b = r / (OW * OH);
        const cae_bf16x4* src = x + (((size_t)b * H + oy) * W + ox) * C4;
        cae_bf16x4* dst = cols + (size_t)r * ldc4;
        for (int i = 0; i < kh; ++i)
            for (int q = threadIdx.x; q < run4; q += 128) dst[i * run4 + q] = src[(size_t)i * W * C4 + q];
    }
}

// 2x2 / stride 2 max pooling, ignore_border, `ph` rows of padding above and below that never win; code = dy*2 + dx
// post_act != LINEAR: x holds PRE-activations and the (monotone) activation is applied to the maximum -- pool(act(x)) = act(pool(x)),
// on a quarter of the elements -- so that the convolution in front runs with a plain bias epilogue
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ arg,
                                                          int B, int H, int W, int C, int ph, int OH, int OW, int post_act) {
    const int64_t total = (int64_t)B * OH * OW * C;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const int64_t p = e / C;
        const int ox = (int)(p % OW), oy = (int)((p / OW) % OH), b = (int)(p / ((int64_t)OW * OH));
        float best = -INFINITY; int code = 0;
        for (int dy = 0; dy < 2; ++dy) {
            const int yy = 2 * oy + dy - ph;
            if (yy < 0 || yy >= H) continue;
            for (int dx = 0; dx < 2; ++dx) {
                const int xx = 2 * ox + dx;
                if (xx >= W) continue;
                const float v = x[(((size_t)b * H + yy) * W + xx) * C + c];
                if (v > best) { best = v; code = dy * 2 + dx; }
            }
        }
        y[e] = cae_act(post_act, best); arg[e] = (uint8_t)code;
    }
}

// (act_y != null: the pooled tensor was act(z); the gradient is multiplied by act'(z) from the activation's output in the same
//  pass -- one sweep over the full-resolution tensor instead of two)
__device__ __forceinline__ float cae_act_grad(int act, float y) {
    if (act == ADN_ACT_SCALED_TANH) { const float t = y * (1.f / 2.4f); return 1.2f * (1.f - t * t); }
    if (act == ADN_ACT_SCALED_TANH_LECUN) { const float t = y * (1.f / 1.7159f); return (2.f / 3.f) * 1.7159f * (1.f - t * t); }
    return 1.f;
}
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ arg,
                                                          float* __restrict__ dx, int B, int H, int W, int C, int ph, int OH, int OW,
                                                          const float* __restrict__ act_y, int act) {
    const int64_t total = (int64_t)B * H * W * C;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const int64_t p = e / C;
        const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((int64_t)W * H));
        const int oy = (y + ph) / 2, ox = x / 2;
        float g = 0.f;
        if (oy < OH && ox < OW) {
            const size_t o = (((size_t)b * OH + oy) * OW + ox) * C + c;
            if ((int)arg[o] == ((y + ph) & 1) * 2 + (x & 1)) g = dy[o];
        }
        if (act_y && g != 0.f) g *= cae_act_grad(act, act_y[e]);
        dx[e] = g;
    }
}

// The first convolution FUSED with the 2 x 2 pooling behind it (no BatchNorm between them, no dropout): the full-resolution
// activation (383 MB at batch 1024) is never written or read.  Forward: a thread computes the four pre-activations of a pooled
// position from its 6 x 6 window (36 LDS reads), keeps the maximum (first one on ties, like maxpool_fwd_kernel) and applies the
// -- monotone -- activation to it: pool(act(x)) = act(pool(x)), a quarter of the activations.
template <int KK>
__global__ __launch_bounds__(256) void conv1_pool_direct_fwd_kernel(const float* __restrict__ x, const float* __restrict__ Wm,
                                                                    const float* __restrict__ bias, float* __restrict__ pooled,
                                                                    cae_bf16x4* __restrict__ pooled16, uint8_t* __restrict__ arg, int B, int H,
                                                                    int W, int O, int PH, int PW, int act) {
    extern __shared__ float img[];
    const int tid = threadIdx.x, Q = O / 4, lanes = 256 / Q;
    const int cq = tid % Q, pl = tid / Q;
    float4 w[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) w[t] = pl < lanes ? *reinterpret_cast<const float4*>(Wm + (size_t)t * O + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 bq = pl < lanes ? *reinterpret_cast<const float4*>(bias + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        for (int e = tid; e < H * W; e += 256) img[e] = x[(size_t)b * H * W + e];
        __syncthreads();
        if (pl >= lanes) continue;
        for (int q = pl; q < PH * PW; q += lanes) {
            const int py = q / PW, px = q - py * PW;
            const float* ip = img + 2 * py * W + 2 * px;
            float xw[36];
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c) xw[r * 6 + c] = ip[r * W + c];
            float4 best = bq;
            uchar4 code = make_uchar4(0, 0, 0, 0);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                float4 a = bq;
#pragma unroll
                for (int t = 0; t < KK; ++t) {
                    const float xv = xw[(t / 5 + (d >> 1)) * 6 + (t % 5) + (d & 1)];
                    a.x += xv * w[t].x; a.y += xv * w[t].y; a.z += xv * w[t].z; a.w += xv * w[t].w;
                }
                if (d == 0) best = a;
                else {
                    if (a.x > best.x) { best.x = a.x; code.x = d; }
                    if (a.y > best.y) { best.y = a.y; code.y = d; }
                    if (a.z > best.z) { best.z = a.z; code.z = d; }
                    if (a.w > best.w) { best.w = a.w; code.w = d; }
                }
            }
            best.x = cae_act_fast(act, best.x); best.y = cae_act_fast(act, best.y); best.z = cae_act_fast(act, best.z); best.w = cae_act_fast(act, best.w);
            const size_t o = ((size_t)b * PH * PW + q) * O + 4 * cq;
            *reinterpret_cast<float4*>(pooled + o) = best;
            *reinterpret_cast<uchar4*>(arg + o) = code;
            if (pooled16) {                              // the next convolution's patch matrix is built from this copy
                cae_bf16x4 h; h[0] = (__bf16)best.x; h[1] = (__bf16)best.y; h[2] = (__bf16)best.z; h[3] = (__bf16)best.w;
                pooled16[o / 4] = h;
            }
        }
    }
}

// Backward of the fused pair: the pooled position's gradient (times act' of the pooled value) belongs to its winner's conv
// position, per channel -- dW[i k + j][o] = sum x[b][2 py + dy_o + i][2 px + dx_o + j] g[b][py][px][o], db[o] = sum g: a quarter of the
// multiply-adds of the unfused weight gradient, and neither the 383 MB full-resolution gradient nor the pooling's adjoint pass.
template <int KK>
__global__ __launch_bounds__(256) void conv1_pool_direct_dw_kernel(const float* __restrict__ x, const float* __restrict__ gp,
                                                                   const float* __restrict__ pooled, const uint8_t* __restrict__ arg,
                                                                   float* __restrict__ slots, int B, int H, int W, int O, int PH, int PW, int act) {
    extern __shared__ float sm[];                        // [H][W] frame, then [KK][O] block sums + [O] bias sums
    float* img = sm;
    float* red = sm + H * W;
    const int tid = threadIdx.x, Q = O / 4, lanes = 256 / Q;
    const int cq = tid % Q, pl = tid / Q;
    float4 acc[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        for (int e = tid; e < H * W; e += 256) img[e] = x[(size_t)b * H * W + e];
        __syncthreads();
        if (pl >= lanes) continue;
        for (int q0 = pl; q0 < PH * PW; q0 += 2 * lanes) {       // two positions in flight per thread
            float4 g[2], y[2];
            uchar4 a[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = q0 + u * lanes;
                const size_t o = ((size_t)b * PH * PW + min(q, PH * PW - 1)) * O + 4 * cq;
                g[u] = q < PH * PW ? *reinterpret_cast<const float4*>(gp + o) : make_float4(0.f, 0.f, 0.f, 0.f);
                y[u] = *reinterpret_cast<const float4*>(pooled + o);
                a[u] = *reinterpret_cast<const uchar4*>(arg + o);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = min(q0 + u * lanes, PH * PW - 1);      // (a position past the end carries a zero gradient)
                const int py = q / PW, px = q - py * PW;
                float4 gv = g[u];
                gv.x *= cae_act_grad(act, y[u].x); gv.y *= cae_act_grad(act, y[u].y); gv.z *= cae_act_grad(act, y[u].z); gv.w *= cae_act_grad(act, y[u].w);
                bsum.x += gv.x; bsum.y += gv.y; bsum.z += gv.z; bsum.w += gv.w;
                const float* base = img + 2 * py * W + 2 * px;
                const float* ix = base + (a[u].x >> 1) * W + (a[u].x & 1);
                const float* iy = base + (a[u].y >> 1) * W + (a[u].y & 1);
                const float* iz = base + (a[u].z >> 1) * W + (a[u].z & 1);
                const float* iw = base + (a[u].w >> 1) * W + (a[u].w & 1);
#pragma unroll
                for (int t = 0; t < KK; ++t) {
                    const int off = (t / 5) * W + (t % 5);
                    acc[t].x += ix[off] * gv.x; acc[t].y += iy[off] * gv.y; acc[t].z += iz[off] * gv.z; acc[t].w += iw[off] * gv.w;
                }
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < KK * O + O; e += 256) red[e] = 0.f;
    for (int l = 0; l < lanes; ++l) {                    // pixel lanes in order: a fixed summation order
        __syncthreads();
        if (pl == l) {
#pragma unroll
            for (int t = 0; t < KK; ++t) {
                float4* r4 = reinterpret_cast<float4*>(red + t * O + 4 * cq);
                float4 v = *r4;
                v.x += acc[t].x; v.y += acc[t].y; v.z += acc[t].z; v.w += acc[t].w;
                *r4 = v;
            }
            float4* b4 = reinterpret_cast<float4*>(red + KK * O + 4 * cq);
            float4 v = *b4;
            v.x += bsum.x; v.y += bsum.y; v.z += bsum.z; v.w += bsum.w;
            *b4 = v;
        }
    }
    __syncthreads();
    for (int e = tid; e < KK * O + O; e += 256) slots[(size_t)blockIdx.x * (KK * O + O) + e] = red[e];
}

// The LAST deconvolution (tied to the first convolution: ONE output channel, 5 x 5, its input upscaled 2 x) without patch matrices.
// Forward, per frame: P[r][t] = x[r][:] . Wm[t][:] for the CH x CW compact input pixels r and the 25 taps t (thread = pixel, its
// 25 sums in registers, Wm in LDS), then z[y][x] = act(b + sum over the taps whose source (y + ph - i, x + pw - j) lies inside the
// upscaled grid of P[((y + ph - i) / 2, (x + pw - j) / 2)][i k + j]) out of LDS.  (GEMM + col2im: 46 + 61 us at batch 1024.)
template <int KK>
__global__ __launch_bounds__(256) void deconv1_direct_fwd_kernel(const float* __restrict__ x, const float* __restrict__ Wm,
                                                                 const float* __restrict__ bias, float* __restrict__ z, int B, int H, int W,
                                                                 int O, int CH, int CW, int ph, int pw, int act) {
    extern __shared__ float sm[];                        // Wl [KK][O], then P [CH CW][KK + 1]
    float* Wl = sm;
    float* P = sm + KK * O;
    const int tid = threadIdx.x, OH = 2 * CH, OW = 2 * CW;
    for (int e = tid; e < KK * O; e += 256) Wl[e] = Wm[e];
    const float b0 = bias ? bias[0] : 0.f;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();                                 // (Wl is in place; the previous frame's P has been consumed)
        for (int r = tid; r < CH * CW; r += 256) {
            float acc[KK];
#pragma unroll
            for (int t = 0; t < KK; ++t) acc[t] = 0.f;
            const float4* xr = reinterpret_cast<const float4*>(x + ((size_t)b * CH * CW + r) * O);
            for (int o4 = 0; o4 < O / 4; ++o4) {
                const float4 a = xr[o4];
#pragma unroll
                for (int t = 0; t < KK; ++t) {
                    const float4 w = *reinterpret_cast<const float4*>(Wl + t * O + 4 * o4);
                    acc[t] += ((a.x * w.x + a.y * w.y) + a.z * w.z) + a.w * w.w;
                }
            }
#pragma unroll
            for (int t = 0; t < KK; ++t) P[r * (KK + 1) + t] = acc[t];
        }
        __syncthreads();
        for (int e = tid; e < H * W; e += 256) {
            const int y = e / W, xx = e - y * W;
            float acc = b0;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int oy = y + ph - i;
                if (oy < 0 || oy >= OH) continue;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int ox = xx + pw - j;
                    if (ox < 0 || ox >= OW) continue;
                    acc += P[((oy >> 1) * CW + (ox >> 1)) * (KK + 1) + i * 5 + j];
                }
            }
            z[(size_t)b * H * W + e] = cae_act(act, acc);
        }
    }
}

// ... and its backward: patches[r][t] = sum over the 2 x 2 block of the upscaled grid of dz[2 cy + dy + i - ph][2 cx + dx + j - pw]
// (zero outside the frame) into LDS, then thread (channel quad, pixel lane): dx[r][o] = sum_t patches[r][t] Wm[t][o] stored, and
// dW[t][o] += patches[r][t] x[r][o] accumulated in registers over the workgroup's frames -> slots (conv1_direct_dw_reduce_kernel);
// the bias gradient (the sum of dz) rides in the slot's last element.  (im2col + two K = 25 GEMMs + a column sum before.)
template <int KK>
__global__ __launch_bounds__(256) void deconv1_direct_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                                 const float* __restrict__ Wm, float* __restrict__ dx, float* __restrict__ slots,
                                                                 int B, int H, int W, int O, int CH, int CW, int ph, int pw, int act_dx,
                                                                 int want_colsum) {
    extern __shared__ float sm[];                        // Wl [KK][O] (later the block sums), img [H][W], patches [CH CW][KK + 1], bred [256]
    float* Wl = sm;
    float* img = sm + KK * O;
    float* pt = img + H * W;
    float* bred = pt + CH * CW * (KK + 1);
    const int tid = threadIdx.x, Q = O / 4, lanes = 256 / Q;
    const int cq = tid % Q, pl = tid / Q;
    for (int e = tid; e < KK * O; e += 256) Wl[e] = Wm[e];
    float4 acc[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    float bsum = 0.f;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);       // column sums of the stored input gradient (want_colsum)
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        for (int e = tid; e < H * W; e += 256) { const float v = dz[(size_t)b * H * W + e]; img[e] = v; bsum += v; }
        __syncthreads();
        for (int idx = tid; idx < CH * CW * KK; idx += 256) {
            const int r = idx / KK, t = idx - r * KK;
            const int cy = r / CW, cx = r - cy * CW, i = t / 5, j = t - 5 * i;
            float v = 0.f;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dxx = 0; dxx < 2; ++dxx) {
                    const int y = 2 * cy + dy + i - ph, xx = 2 * cx + dxx + j - pw;
                    if (y >= 0 && y < H && xx >= 0 && xx < W) v += img[y * W + xx];
                }
            pt[r * (KK + 1) + t] = v;
        }
        __syncthreads();
        if (pl < lanes) {
            for (int r = pl; r < CH * CW; r += lanes) {
                const size_t o = ((size_t)b * CH * CW + r) * O + 4 * cq;
                const float4 a = *reinterpret_cast<const float4*>(x + o);
                float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t < KK; ++t) {
                    const float pv = pt[r * (KK + 1) + t];
                    const float4 w = *reinterpret_cast<const float4*>(Wl + t * O + 4 * cq);
                    d.x += pv * w.x; d.y += pv * w.y; d.z += pv * w.z; d.w += pv * w.w;
                    acc[t].x += pv * a.x; acc[t].y += pv * a.y; acc[t].z += pv * a.z; acc[t].w += pv * a.w;
                }
                // x is the OUTPUT of the layer below (its activation applied): the gradient leaves multiplied by act'(x), and its
                // column sums -- that layer's bias gradient -- are taken on the way (a separate act' pass and a column-sum pass before)
                d.x *= cae_act_grad(act_dx, a.x); d.y *= cae_act_grad(act_dx, a.y); d.z *= cae_act_grad(act_dx, a.z); d.w *= cae_act_grad(act_dx, a.w);
                csum.x += d.x; csum.y += d.y; csum.z += d.z; csum.w += d.w;
                *reinterpret_cast<float4*>(dx + o) = d;
            }
        }
    }
    __syncthreads();                                     // Wl has been read for the last time: it becomes the block sums
    for (int e = tid; e < KK * O; e += 256) Wl[e] = 0.f;
    for (int l = 0; l < lanes; ++l) {                    // pixel lanes in order: a fixed summation order
        __syncthreads();
        if (pl == l) {
#pragma unroll
            for (int t = 0; t < KK; ++t) {
                float4* r4 = reinterpret_cast<float4*>(Wl + t * O + 4 * cq);
                float4 v = *r4;
                v.x += acc[t].x; v.y += acc[t].y; v.z += acc[t].z; v.w += acc[t].w;
                *r4 = v;
            }
        }
    }
    bred[tid] = bsum;
    __syncthreads();
    const size_t slot = (size_t)blockIdx.x * (KK * O + 1 + (want_colsum ? O : 0));
    for (int e = tid; e < KK * O; e += 256) slots[slot + e] = Wl[e];
    if (tid == 0) {
        float v = 0.f;
        for (int k = 0; k < 256; ++k) v += bred[k];      // (thread order: fixed)
        slots[slot + KK * O] = v;
    }
    if (want_colsum) {                                   // pixel lanes in order again, into the first O words of Wl
        __syncthreads();
        for (int e = tid; e < O; e += 256) Wl[e] = 0.f;
        for (int l = 0; l < lanes; ++l) {
            __syncthreads();
            if (pl == l) {
                float4* r4 = reinterpret_cast<float4*>(Wl + 4 * cq);
                float4 v = *r4;
                v.x += csum.x; v.y += csum.y; v.z += csum.z; v.w += csum.w;
                *r4 = v;
            }
        }
        __syncthreads();
        for (int e = tid; e < O; e += 256) slots[slot + KK * O + 1 + e] = Wl[e];
    }
}

// The same adjoint from the POOLED side (round 4; 2 x 2 windows that tile the padded input exactly, C % 4 == 0): one thread per
// (pooled position, 4 channels) writes its window -- the winner gets dy * act'(.), the other three positions zero.  act' comes
// from the pooled VALUE (the activation is applied before the pooling, so the pooled value is the winner's activation): the
// full-resolution activation tensor is not read at all (383 MB at batch 1024 for the first convolution).  Also leaves, in the
// same pass, the bf16 copy of the gradient that the convolution's backward GEMMs read (dx16) and the bias gradient: the column
// sums of the full-resolution gradient are the column sums of the winners, accumulated per workgroup in LDS and added with one
// float atomic per channel and workgroup (dbias; null in deterministic mode: the caller sums the columns instead).
__global__ __launch_bounds__(256) void maxpool_bwd_pooled_kernel(const float4* __restrict__ dy, const uchar4* __restrict__ arg,
                                                                 const float4* __restrict__ pooled, float4* __restrict__ dx,
                                                                 cae_bf16x4* __restrict__ dx16, float* __restrict__ dbias, int B, int H,
                                                                 int W, int C4, int ph, int OH, int OW, int act) {
    extern __shared__ float csum[];                  // [4 C4]
    for (int c = threadIdx.x; c < 4 * C4; c += 256) csum[c] = 0.f;
    __syncthreads();
    const int64_t total = (int64_t)B * OH * OW * C4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % C4);
        const int64_t p = e / C4;
        const int ox = (int)(p % OW), oy = (int)((p / OW) % OH), b = (int)(p / ((int64_t)OW * OH));
        float4 g = dy[e];
        const uchar4 a = arg[e];
        if (pooled) {
            const float4 y = pooled[e];
            g.x *= cae_act_grad(act, y.x); g.y *= cae_act_grad(act, y.y); g.z *= cae_act_grad(act, y.z); g.w *= cae_act_grad(act, y.w);
        }
        if (dbias) {
            atomicAdd(&csum[4 * c], g.x); atomicAdd(&csum[4 * c + 1], g.y); atomicAdd(&csum[4 * c + 2], g.z); atomicAdd(&csum[4 * c + 3], g.w);
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int yy = 2 * oy + (d >> 1) - ph, xx = 2 * ox + (d & 1);
            if (yy < 0 || yy >= H || xx >= W) continue;
            const float4 v = make_float4(a.x == d ? g.x : 0.f, a.y == d ? g.y : 0.f, a.z == d ? g.z : 0.f, a.w == d ? g.w : 0.f);
            const size_t o = (((size_t)b * H + yy) * W + xx) * C4 + c;
            dx[o] = v;
            if (dx16) { cae_bf16x4 h; h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w; dx16[o] = h; }
        }
    }
    if (dbias) {
        __syncthreads();
        for (int c = threadIdx.x; c < 4 * C4; c += 256) if (csum[c] != 0.f) atomicAdd(dbias + c, csum[c]);
    }
}

// d = scale * (recon - target);  sq[i] = (recon - target)^2 summed into *loss_acc by the caller's dot product
__global__ __launch_bounds__(256) void mse_grad_kernel(const float* __restrict__ r, const float* __restrict__ t, float* __restrict__ d,
                                                       int64_t n, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = scale * (r[i] - t[i]);
}

__global__ void scale_scalar_kernel(float* p, float s) { *p *= s; }

// DropoutLayer (rescale) in place on an NHWC tensor [rows = B*HW][Cp] with C logical channels: the mask is indexed by the
// element's position in the reference's (B, C, H, W) tensor (for the flattened tensor that IS the (c, h, w) feature index),
// so that it does not depend on this file's layout; forward and backward apply the same kernel.
__global__ __launch_bounds__(256) void cae_dropout_kernel(float* __restrict__ x, int64_t rows, int HW, int C, int Cp, float p,
                                                          float scale, uint32_t key) {
    const int64_t total = rows * Cp;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % Cp);
        if (c >= C) continue;
        const int64_t r = e / Cp;
        const uint32_t idx = (uint32_t)(((r / HW) * C + c) * HW + r % HW);
        x[e] = cae_keep(key, idx, p) ? x[e] * scale : 0.f;
    }
}

int grid_of(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 16384)); }

// one convolution (or the convolution a deconv layer is the adjoint of): input [B][H][W][C] padded by (ph, pw),
// k x k filters, O output channels, output grid OH x OW; patches matrix [B*OH*OW][ldk]
struct ConvGeom { int H, W, C, k, O, ph, pw, OH, OW, K, ldk; };

ConvGeom conv_geom(int H, int W, int C, int k, int O, int ph = 0, int pw = 0) {
    ConvGeom g{H, W, C, k, O, ph, pw, H + 2 * ph - k + 1, W + 2 * pw - k + 1, k * k * C, 0};
    g.ldk = (int)round_up(g.K, 8);      // 8: the bf16 patches matrix must keep 16-byte rows too
    return g;
}

// kind: 0 vector, 1 conv W (Op x Cp physical filters / channels), 2 dense7 W, 3 bottleneck W, 4 dense9 b
struct Tensor { std::string name; int ndim; int64_t dims[4]; size_t off; size_t floats; int kind; int Op, Cp; size_t phys; };

}  // namespace

struct BnSlot {              // one BatchNormLayer: parameter offsets, width (physical / logical), what it normalises
    bool on = false;
    size_t beta = 0, gamma = 0, mean = 0, inv_std = 0;
    int C = 0;
    float *out = nullptr, *save_mean = nullptr, *save_inv = nullptr;
    void* ws = nullptr;
};

struct adn_cae {
    adn_cae_config cfg;
    hipStream_t stream = nullptr;
    int H = 30, W = 40, D7 = 500, NB = 50, ldb = 52, precision = ADN_PRECISION_F32;
    int variant = ADN_CAE_NORMAL, S = ADN_ACT_SCALED_TANH;
    int F1 = 100, F2 = 152, F3 = 200, F1L = 100, F2L = 150, F3L = 200;       // physical (padded) and logical filter counts
    int bn_mode = 0;                      // 0 none, 1 behind the poolings / flatten / dense, 2 behind the convolutions / dense
    bool drop = false;
    bool training = false;                // this pass: batch statistics + running-average update + dropout masks
    uint32_t drop_seed = 0x5EED1234u, drop_counter = 0;
    BnSlot bn[4];
    ConvGeom c1, c3, c5, d11, d13, d15;
    int p2h = 0, p2w = 0, p4h = 0, p4w = 0, flat = 0;
    std::vector<Tensor> params;
    size_t flat_floats = 0;
    float* buf[4] = {nullptr, nullptr, nullptr, nullptr};       // parameters, gradients, optimiser state 0 / 1
    size_t W1 = 0, b1 = 0, W3 = 0, b3 = 0, W5 = 0, b5 = 0, W7 = 0, b7 = 0, Wb = 0, bb = 0, b8 = 0, b9 = 0, b11 = 0, b13 = 0, b15 = 0;
    char* slab = nullptr; size_t slab_bytes = 0; int wsB = 0;
    float *x0 = nullptr, *cols1 = nullptr, *a1 = nullptr, *p2 = nullptr, *cols3 = nullptr, *a3 = nullptr, *p4 = nullptr,
          *cols5 = nullptr, *a5 = nullptr, *a7 = nullptr, *code = nullptr, *a8 = nullptr, *a9 = nullptr, *a11 = nullptr,
          *a13 = nullptr, *a15 = nullptr, *target = nullptr, *scratch = nullptr,
          *gA = nullptr, *gB = nullptr, *loss_dev = nullptr, *f6d = nullptr, *a7d = nullptr, *splitk = nullptr;
    size_t splitk_floats = 0;               // split-K slabs of the ping-pong kernel (mm16 with pp)
    // what each layer actually read in the last forward pass (the BatchNorm output where one sits in front of it)
    const float *in3 = nullptr, *in5 = nullptr, *in7 = nullptr, *inb = nullptr;
    uint8_t *arg2 = nullptr, *arg4 = nullptr;
    // bf16 mode: bf16 copy of the parameters (same layout), of the deconv inputs, and a scratch for gradients
    char *p16 = nullptr, *a9_16 = nullptr, *u12_16 = nullptr, *t16 = nullptr;
    bool p16_dirty = true;
    bool preact3 = false;                 // last forward pass: a3 holds conv3's pre-activations (see forward())
    bool fused1 = false;                  // last forward pass: conv1 + pool2 ran fused (no a1; conv1_pool_direct_*_kernel)
    float* cs_ws = nullptr; size_t cs_ws_floats = 0;      // per-tile column sums of a GEMM's fused bias-gradient epilogue
    char* p2_16 = nullptr;                // ... and left this bf16 copy of p2 for conv3's patch matrix
    bool p2_16_valid = false;
    bool grads_valid = false;
    int adam_t = 0;
    float* P(size_t off) const { return buf[0] + off; }
    float* G(size_t off) const { return buf[1] + off; }
};

namespace {

size_t add_tensor(adn_cae* m, const char* name, int kind, std::initializer_list<int64_t> dims, int Op = 0, int Cp = 0) {
    Tensor t; t.name = name; t.kind = kind; t.off = m->flat_floats; t.ndim = (int)dims.size(); t.Op = Op; t.Cp = Cp;
    size_t n = 1; int k = 0;
    for (auto d : dims) { t.dims[k++] = d; n *= (size_t)d; }
    t.floats = n;                                    // logical (host) element count
    if (kind == 1) t.phys = (size_t)t.dims[2] * t.dims[3] * Cp * Op;
    else if (kind == 3) t.phys = (size_t)t.dims[0] * m->ldb;
    else t.phys = (size_t)round_up((int64_t)std::max<size_t>(n, (size_t)Op), 4);      // vectors: Op = physical length
    t.phys = (size_t)round_up((int64_t)t.phys, 4);
    m->params.push_back(t);
    m->flat_floats += t.phys;
    return t.off;
}

struct Carve {
    char* base; size_t cur = 0;
    template <typename T> T* take(size_t n) {
        T* p = base ? reinterpret_cast<T*>(base + cur) : nullptr;
        cur += (size_t)round_up((int64_t)(n * sizeof(T)), 256);
        return p;
    }
};

size_t rows_of(const ConvGeom& g, int B) { return (size_t)B * g.OH * g.OW; }

size_t carve(adn_cae* m, char* base, int B) {
    Carve c{base};
    const size_t N = B;
    m->x0 = c.take<float>(N * m->H * m->W);
    m->cols1 = c.take<float>(rows_of(m->c1, B) * m->c1.ldk);
    m->a1 = c.take<float>(rows_of(m->c1, B) * m->F1);
    m->p2 = c.take<float>(N * m->p2h * m->p2w * m->F1); m->arg2 = c.take<uint8_t>(N * m->p2h * m->p2w * m->F1);
    m->cols3 = c.take<float>(rows_of(m->c3, B) * m->c3.ldk);
    m->a3 = c.take<float>(rows_of(m->c3, B) * m->F2);
    m->p4 = c.take<float>(N * m->p4h * m->p4w * m->F2); m->arg4 = c.take<uint8_t>(N * m->p4h * m->p4w * m->F2);
    m->cols5 = c.take<float>(rows_of(m->c5, B) * m->c5.ldk);
    m->a5 = c.take<float>(N * m->flat);
    m->a7 = c.take<float>(N * m->D7); m->code = c.take<float>(N * m->ldb);
    m->a8 = c.take<float>(N * m->D7); m->a9 = c.take<float>(N * m->flat);
    m->a11 = c.take<float>(N * m->d11.H * m->d11.W * m->F2);
    m->a13 = c.take<float>(N * m->d13.H * m->d13.W * m->F1);
    m->a15 = c.take<float>(N * m->H * m->W);
    m->target = c.take<float>(N * m->H * m->W);
    size_t sc = 0, act = N * m->H * m->W;
    for (const ConvGeom* g : {&m->c1, &m->c3, &m->c5, &m->d11, &m->d13, &m->d15}) {
        sc = std::max(sc, rows_of(*g, B) * g->ldk);
        act = std::max(act, std::max(rows_of(*g, B) * (size_t)g->O, N * g->H * g->W * (size_t)g->C));
    }
    m->scratch = c.take<float>(sc);
    m->gA = c.take<float>(act); m->gB = c.take<float>(act);
    m->loss_dev = c.take<float>(8);
    m->splitk_floats = std::max<size_t>((size_t)16 << 20, N * 16384);     // (25 slabs of 2504 x 152; 2 of 32256 x 152 at batch 1024)
    m->splitk = c.take<float>(m->splitk_floats);
    m->a9_16 = c.take<char>(N * m->flat * 2);
    m->u12_16 = c.take<char>(N * m->d11.H * m->d11.W * m->F2 * 2);           // bf16 copy of a11 (the compact input of deconv2d13)
    m->t16 = c.take<char>(act * 2);
    m->p2_16 = c.take<char>(N * m->p2h * m->p2w * m->F1 * 2);
    m->cs_ws_floats = (size_t)1 << 20; m->cs_ws = c.take<float>(m->cs_ws_floats);
    if (m->drop && m->bn_mode == 0) { m->f6d = c.take<float>(N * m->flat); m->a7d = c.take<float>(N * m->D7); }   // dropped copies of a5 / a7
    if (m->bn_mode) {                                 // BatchNorm outputs, batch statistics, workspaces
        const size_t rows[4] = {m->bn_mode == 1 ? N * m->p2h * m->p2w : rows_of(m->c1, B), m->bn_mode == 1 ? N * m->p4h * m->p4w : rows_of(m->c3, B),
                                m->bn_mode == 1 ? N : rows_of(m->c5, B), N};
        for (int k = 0; k < 4; ++k) {
            BnSlot& q = m->bn[k];
            q.out = c.take<float>(rows[k] * q.C);
            q.save_mean = c.take<float>(q.C); q.save_inv = c.take<float>(q.C);
            q.ws = c.take<char>(batchnorm_ws_bytes(q.C));
        }
    }
    return c.cur;
}

int ensure_ws(adn_cae* m, int B) {
    if (B == m->wsB && m->slab) return ADN_OK;
    const size_t need = carve(m, nullptr, B);
    if (need > m->slab_bytes) {
        if (m->slab) { ADN_HIP_CHECK(hipStreamSynchronize(m->stream)); ADN_HIP_CHECK(hipFree(m->slab)); m->slab = nullptr; }
        ADN_HIP_CHECK(hipMalloc((void**)&m->slab, need));
        m->slab_bytes = need;
    }
    carve(m, m->slab, B);
    ADN_HIP_CHECK(hipMemsetAsync(m->slab, 0, need, m->stream));       // pad columns of the patches matrices read as zero
    m->wsB = B;
    return ADN_OK;
}

int mm(adn_cae* m, int layout, int M, int N, int K, const float* A, int lda, const float* Bm, int ldb, float* C, int ldc,
       const float* bias = nullptr, int act = ADN_ACT_LINEAR, int accumulate = 0) {
    GemmArgs g;
    g.layout = layout; g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = Bm; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.bias = bias; g.act = act; g.accumulate = accumulate; g.precision = m->precision;
    return gemm(g, m->stream);
}

// the heavy layers (150 / 200 filters, K = 2500 / 1368) run on the bf16-operand GEMM kernels in bf16 mode
bool fast16(const adn_cae* m, const ConvGeom& g) {
    return m->precision == ADN_PRECISION_BF16 && g.O % 8 == 0 && g.C % 4 == 0;      // (row strides: ldk and O)
}
const void* W16(const adn_cae* m, size_t off) { return m->p16 + off * 2; }

// pp: the persistent ping-pong kernel wherever it can run (GemmArgs::pp_force) -- the auto-encoder's big products are narrow
// (100 - 200 output channels) under very many rows, a shape the AdeNet-tuned selection leaves to the register-staged kernels
int mm16(adn_cae* m, int layout, int M, int N, int K, const void* A16, int lda, const void* B16, int ldb, float* C, int ldc,
         const float* bias = nullptr, int act = ADN_ACT_LINEAR, int accumulate = 0, bool pp = false) {
    static const bool no_pp = getenv("ADN_CAE_NO_PP") != nullptr;       // (A/B switch)
    GemmArgs g;
    g.layout = layout; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.A = reinterpret_cast<const float*>(A16); g.B = reinterpret_cast<const float*>(B16);      // (only the bf16 views are read)
    g.A16 = A16; g.B16 = B16;
    g.bias = bias; g.act = act; g.accumulate = accumulate; g.precision = ADN_PRECISION_BF16;
    if (pp && !no_pp) { g.pp_force = 1; g.splitk_ws = m->splitk; g.splitk_ws_floats = m->splitk_floats; }
    return gemm(g, m->stream);
}

// `up`: the rows are wanted for the compact (OH / 2) x (OW / 2) grid, each the sum of the 2 x 2 block of the upscaled grid
int im2col16(adn_cae* m, const float* x, const ConvGeom& g, int B, void* cols16, int up = 0) {
    const int oh = up ? g.OH / 2 : g.OH, ow = up ? g.OW / 2 : g.OW;
    const int64_t total = (int64_t)B * oh * ow * g.K;
    static const bool no_runs = getenv("ADN_CAE_IM2COL_GENERIC") != nullptr;     // (A/B switch)
    if (!up && g.ph == 0 && g.pw == 0 && !no_runs) {
        const int R = B * g.OH * g.OW;
        if (x == m->p2 && m->p2_16_valid) {
            hipLaunchKernelGGL(im2col_runs_from_bf16_kernel, dim3((R + 3) / 4), dim3(128), 0, m->stream, reinterpret_cast<const cae_bf16x4*>(m->p2_16),
                               reinterpret_cast<cae_bf16x4*>(cols16), R, g.H, g.W, g.C / 4, g.k, g.k, g.OH, g.OW, g.ldk / 4);
            ADN_HIP_CHECK(hipGetLastError());
            return ADN_OK;
        }
        hipLaunchKernelGGL(im2col_runs_bf16_kernel, dim3((R + 3) / 4), dim3(128), 0, m->stream, reinterpret_cast<const float4*>(x),
                           reinterpret_cast<cae_bf16x4*>(cols16), R, g.H, g.W, g.C / 4, g.k, g.k, g.OH, g.OW, g.ldk / 4);
        ADN_HIP_CHECK(hipGetLastError());
        return ADN_OK;
    }
    if (up && g.ph == 0 && g.pw == 0 && !no_runs) {      // (2 (oh - 1) + 1 + k - 1 = OH + k - 2 < H: every source run lies inside the frame)
        const int R = B * oh * ow;
        hipLaunchKernelGGL(im2col_runs_up_bf16_kernel, dim3((R + 3) / 4), dim3(128), 0, m->stream, reinterpret_cast<const float4*>(x),
                           reinterpret_cast<cae_bf16x4*>(cols16), R, g.H, g.W, g.C / 4, g.k, g.k, oh, ow, g.ldk / 4);
        ADN_HIP_CHECK(hipGetLastError());
        return ADN_OK;
    }
    hipLaunchKernelGGL(im2col4_bf16_kernel, dim3(grid_of(total / 4)), dim3(256), 0, m->stream, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<cae_bf16x4*>(cols16), B, g.H, g.W, g.C / 4, g.k, g.k, g.ph, g.pw, oh, ow, g.ldk / 4, up);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int im2col(adn_cae* m, const float* x, const ConvGeom& g, int B, float* cols, int up = 0) {
    const int oh = up ? g.OH / 2 : g.OH, ow = up ? g.OW / 2 : g.OW;
    const int64_t total = (int64_t)B * oh * ow * g.K;
    if (g.C % 4 == 0)
        hipLaunchKernelGGL(im2col4_kernel, dim3(grid_of(total / 4)), dim3(256), 0, m->stream, reinterpret_cast<const float4*>(x),
                           reinterpret_cast<float4*>(cols), B, g.H, g.W, g.C / 4, g.k, g.k, g.ph, g.pw, oh, ow, g.ldk / 4, up);
    else
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_of(total)), dim3(256), 0, m->stream, x, cols, B, g.H, g.W, g.C, g.k, g.k, g.ph, g.pw,
                       oh, ow, g.ldk, up);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// `up`: dcols holds the rows of the compact grid; every row stands for a 2 x 2 block of the (OH x OW) grid
int col2im(adn_cae* m, const float* dcols, const ConvGeom& g, int B, float* out, const float* bias, int act, int up = 0) {
    const int64_t total = (int64_t)B * g.H * g.W * g.C;
    if (g.C % 4 == 0)
        hipLaunchKernelGGL(col2im4_kernel, dim3(grid_of(total / 4)), dim3(256), 0, m->stream, reinterpret_cast<const float4*>(dcols),
                           g.ldk / 4, reinterpret_cast<float4*>(out), B, g.H, g.W, g.C / 4, g.k, g.k, g.ph, g.pw, g.OH, g.OW,
                           reinterpret_cast<const float4*>(bias), act, up);
    else
    hipLaunchKernelGGL(col2im_kernel, dim3(grid_of(total)), dim3(256), 0, m->stream, dcols, g.ldk, out, B, g.H, g.W, g.C, g.k, g.k,
                       g.ph, g.pw, g.OH, g.OW, bias, act, up);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int col2im16(adn_cae* m, const void* dcols16, const ConvGeom& g, int B, float* out, const float* bias, int act, int up = 0) {
    const int64_t total = (int64_t)B * g.H * g.W * g.C;
    hipLaunchKernelGGL(col2im4_bf16_kernel, dim3(grid_of(total / 4)), dim3(256), 0, m->stream, reinterpret_cast<const cae_bf16x4*>(dcols16),
                       g.ldk / 4, reinterpret_cast<float4*>(out), B, g.H, g.W, g.C / 4, g.k, g.k, g.ph, g.pw, g.OH, g.OW,
                       reinterpret_cast<const float4*>(bias), act, up);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// a product whose only reader takes bf16: the GEMM writes the bf16 matrix alone (no fp32 copy)
int mm16_lean(adn_cae* m, int layout, int M, int N, int K, const void* A16, int lda, const void* B16, int ldb, void* C16, int ldc) {
    GemmArgs g;
    g.layout = layout; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.C = nullptr; g.C16 = C16; g.ldc = ldc;
    g.A = reinterpret_cast<const float*>(A16); g.B = reinterpret_cast<const float*>(B16);
    g.A16 = A16; g.B16 = B16; g.precision = ADN_PRECISION_BF16;
    return gemm(g, m->stream);
}
static bool lean_scratch() { static const bool off = getenv("ADN_CAE_FP32_SCRATCH") != nullptr; return !off; }   // (A/B switch)

// y = act(conv(x) + b): patches kept in `cols` for the backward pass
// preact: y = conv(x) + b WITHOUT the activation (the pooling behind it applies it to its maxima): a plain bias epilogue, which the
// ping-pong kernel has -- 129024 x 152 x 2500 at batch 1024
// the one-input-channel first convolution without a patch matrix (conv1_direct_*_kernel): bf16 mode only -- the fp32 mode keeps
// its GEMM path, the oracle-parity reference
bool direct1(const adn_cae* m, const ConvGeom& g) {
    static const bool off = getenv("ADN_CAE_NO_DIRECT1") != nullptr;      // (A/B switch)
    const size_t lds = ((size_t)g.H * g.W + (size_t)26 * g.O) * sizeof(float);      // (frame + block sums: within a plain launch's 64 KB)
    return !off && m->precision == ADN_PRECISION_BF16 && g.C == 1 && g.k == 5 && g.ph == 0 && g.pw == 0 && g.O % 4 == 0 && g.O <= 256 &&
           lds <= 65536 && (m->S == ADN_ACT_SCALED_TANH || m->S == ADN_ACT_SCALED_TANH_LECUN || m->S == ADN_ACT_LINEAR);
}

// ... and the last deconvolution, tied to it (deconv1_direct_*_kernel; its input arrives 2 x upscaled)
bool direct15(const adn_cae* m, const ConvGeom& g) {
    static const bool off = getenv("ADN_CAE_NO_DIRECT15") != nullptr;     // (A/B switch)
    // (the kernels' LDS -- filter, frame, per-frame tap sums / patches -- must stay within the 64 KB a plain launch may ask for)
    const size_t lds = ((size_t)25 * g.O + (size_t)g.H * g.W + (size_t)(g.OH / 2) * (g.OW / 2) * 26 + 256) * sizeof(float);
    return !off && m->precision == ADN_PRECISION_BF16 && g.C == 1 && g.k == 5 && g.O % 4 == 0 && g.O <= 256 && g.OH % 2 == 0 && g.OW % 2 == 0 &&
           lds <= 65536;
}

int conv_fwd(adn_cae* m, const float* x, const ConvGeom& g, int B, float* cols, size_t W, size_t b, float* y, bool preact = false) {
    if (!preact && direct1(m, g)) {
        hipLaunchKernelGGL(conv1_direct_fwd_kernel<25>, dim3(std::min(B, 2048)), dim3(256), (size_t)g.H * g.W * sizeof(float), m->stream,
                           x, m->P(W), m->P(b), y, B, g.H, g.W, g.k, g.O, g.OH, g.OW, m->S);
        ADN_HIP_CHECK(hipGetLastError());
        return ADN_OK;
    }
    if (fast16(m, g)) {                              // `cols` holds the bf16 patches matrix in this mode
        ADN_TRY(im2col16(m, x, g, B, cols));
        const int R = (int)rows_of(g, B);
        if (preact) return mm16(m, GEMM_NN, R, g.O, g.K, cols, g.ldk, W16(m, W), g.O, y, g.O, m->P(b), ADN_ACT_LINEAR, 0, R >= 65536);
        return mm16(m, GEMM_NN, R, g.O, g.K, cols, g.ldk, W16(m, W), g.O, y, g.O, m->P(b), m->S);
    }
    ADN_TRY(im2col(m, x, g, B, cols));
    return mm(m, GEMM_NN, (int)rows_of(g, B), g.O, g.K, cols, g.ldk, m->P(W), g.O, y, g.O, m->P(b), m->S);
}

// dy (already multiplied by act') -> dW, db, and (optionally) dx.  `ready`: the pass that produced dy already left its bf16 copy
// in t16 and added the bias gradient (maxpool_bwd's pooled-grid form)
// x_direct: the layer's input when conv_fwd took the direct form (no patches in `cols`)
int conv_bwd(adn_cae* m, const ConvGeom& g, int B, const float* cols, const float* dy, size_t W, size_t b, float* dx, bool ready = false,
             const float* x_direct = nullptr) {
    const int R = (int)rows_of(g, B);
    if (x_direct) {
        ADN_CHECK(direct1(m, g) && !dx, ADN_ERR_STATE, "conv AE: direct first-layer backward without its forward");
        const int nslots = std::min(B, 512), n = g.K * g.O;
        ADN_CHECK((size_t)nslots * n <= m->splitk_floats, ADN_ERR_STATE, "conv AE: slot workspace too small");
        hipLaunchKernelGGL(conv1_direct_dw_kernel<25>, dim3(nslots), dim3(256), (size_t)(g.H * g.W + n) * sizeof(float), m->stream,
                           x_direct, dy, m->splitk, B, g.H, g.W, g.k, g.O, g.OH, g.OW);
        hipLaunchKernelGGL(conv1_direct_dw_reduce_kernel, dim3(cdiv(n, 64)), dim3(256), 0, m->stream, m->splitk, nslots, n, m->G(W), n, (float*)nullptr);
        ADN_HIP_CHECK(hipGetLastError());
        if (!ready) ADN_TRY(col_sum(dy, g.O, R, g.O, m->G(b), 1, m->stream));
        return ADN_OK;
    }
    if (!fast16(m, g) && ready) {                    // (fp32-operand layer: only the bias gradient came with dy)
        ADN_TRY(mm(m, GEMM_TN, g.K, g.O, R, cols, g.ldk, dy, g.O, m->G(W), g.O, nullptr, ADN_ACT_LINEAR, 1));
        if (dx) {
            ADN_TRY(mm(m, GEMM_NT, R, g.K, g.O, dy, g.O, m->P(W), g.O, m->scratch, g.ldk));
            ADN_TRY(col2im(m, m->scratch, g, B, dx, nullptr, ADN_ACT_LINEAR));
        }
        return ADN_OK;
    }
    if (fast16(m, g)) {
        if (!ready) ADN_TRY(to_bf16(dy, m->t16, (size_t)R * g.O, m->stream));
        ADN_TRY(mm16(m, GEMM_TN, g.K, g.O, R, cols, g.ldk, m->t16, g.O, m->G(W), g.O, nullptr, ADN_ACT_LINEAR, 1, true));
        if (!ready) ADN_TRY(col_sum(dy, g.O, R, g.O, m->G(b), 1, m->stream));
        if (dx) {
            if (lean_scratch()) {                    // dy Wm^T as bf16 only: its one reader sums it in fp32
                ADN_TRY(mm16_lean(m, GEMM_NT, R, g.K, g.O, m->t16, g.O, W16(m, W), g.O, m->scratch, g.ldk));
                ADN_TRY(col2im16(m, m->scratch, g, B, dx, nullptr, ADN_ACT_LINEAR));
            } else {
                ADN_TRY(mm16(m, GEMM_NT, R, g.K, g.O, m->t16, g.O, W16(m, W), g.O, m->scratch, g.ldk));
                ADN_TRY(col2im(m, m->scratch, g, B, dx, nullptr, ADN_ACT_LINEAR));
            }
        }
        return ADN_OK;
    }
    ADN_TRY(mm(m, GEMM_TN, g.K, g.O, R, cols, g.ldk, dy, g.O, m->G(W), g.O, nullptr, ADN_ACT_LINEAR, 1));
    ADN_TRY(col_sum(dy, g.O, R, g.O, m->G(b), 1, m->stream));
    if (dx) {
        ADN_TRY(mm(m, GEMM_NT, R, g.K, g.O, dy, g.O, m->P(W), g.O, m->scratch, g.ldk));
        ADN_TRY(col2im(m, m->scratch, g, B, dx, nullptr, ADN_ACT_LINEAR));
    }
    return ADN_OK;
}

// z = act(adjoint_conv(x) + b): x [B*OH*OW][O] -> z [B][H][W][C]   (g = the tied convolution, crop = its padding)
// x16: bf16 copy of x (kept for the backward pass), null = fp32 path.
// up = 1: the layer's input is Upscale2DLayer(x), x on the compact (OH / 2) x (OW / 2) grid.  The upscaled tensor is never
// built: repeat(x) W^T = repeat(x W^T), so the GEMM runs on a quarter of the rows and the gather reads row (oy / 2, ox / 2)
// (modelzoo/avletters_convae.py:62-66: upscale2d12 / upscale2d14 in front of deconv2d13 / deconv2d14; deconv2d13 was 106 of
// the forward pass's 231 MFLOP per frame)
int deconv_fwd(adn_cae* m, const float* x, void* x16, const ConvGeom& g, int B, size_t W, size_t b, float* z, int up = 0) {
    const int R = (int)rows_of(g, B) / (up ? 4 : 1);
    if (x16 && fast16(m, g)) {
        ADN_TRY(to_bf16(x, x16, (size_t)R * g.O, m->stream));
        if (lean_scratch()) {
            ADN_TRY(mm16_lean(m, GEMM_NT, R, g.K, g.O, x16, g.O, W16(m, W), g.O, m->scratch, g.ldk));
            return col2im16(m, m->scratch, g, B, z, m->P(b), m->S, up);
        }
        ADN_TRY(mm16(m, GEMM_NT, R, g.K, g.O, x16, g.O, W16(m, W), g.O, m->scratch, g.ldk));
        return col2im(m, m->scratch, g, B, z, m->P(b), m->S, up);
    }
    if (up && direct15(m, g)) {
        const int CH = g.OH / 2, CW = g.OW / 2;
        hipLaunchKernelGGL(deconv1_direct_fwd_kernel<25>, dim3(std::min(B, 2048)), dim3(256), (size_t)(25 * g.O + CH * CW * 26) * sizeof(float),
                           m->stream, x, m->P(W), m->P(b), z, B, g.H, g.W, g.O, CH, CW, g.ph, g.pw, m->S);
        ADN_HIP_CHECK(hipGetLastError());
        return ADN_OK;
    }
    ADN_TRY(mm(m, GEMM_NT, R, g.K, g.O, x, g.O, m->P(W), g.O, m->scratch, g.ldk));
    return col2im(m, m->scratch, g, B, z, m->P(b), m->S, up);
}

// dz (already multiplied by act') -> db, dW (tied), dx.  up = 1: x and dx live on the compact grid; the adjoint of the 2 x 2
// repetition is a sum, and it commutes with the products: dx = (sum4 cols') Wm, dW += (sum4 cols')^T x -- the patches are
// summed while they are gathered (im2col `up`), both GEMMs run on a quarter of the rows
// x_act / x_bias_grad / x_done: x is the activated output of the layer below; where this layer runs as the direct kernel, dx leaves
// multiplied by act'(x) and its column sums go to that layer's bias gradient (*x_done = true: the caller skips both passes).
// bias_ready: this layer's own bias gradient was added by the layer above in that way
int deconv_bwd(adn_cae* m, const ConvGeom& g, int B, const float* x, const void* x16, const float* dz, size_t W, size_t b,
               float* dx, int up = 0, int x_act = ADN_ACT_LINEAR, float* x_bias_grad = nullptr, bool* x_done = nullptr, bool bias_ready = false) {
    const int R = (int)rows_of(g, B) / (up ? 4 : 1);
    if (x_done) *x_done = false;
    if (!x16 && up && direct15(m, g)) {
        static const bool no_fuse = getenv("ADN_CAE_NO_FUSE15") != nullptr;      // (A/B switch)
        const bool fuse = x_done && x_bias_grad && !no_fuse;
        const int CH = g.OH / 2, CW = g.OW / 2, nslots = std::min(B, 512), nW = 25 * g.O, n = nW + 1 + (fuse ? g.O : 0);
        ADN_CHECK((size_t)nslots * n <= m->splitk_floats, ADN_ERR_STATE, "conv AE: slot workspace too small");
        hipLaunchKernelGGL(deconv1_direct_bwd_kernel<25>, dim3(nslots), dim3(256),
                           (size_t)(25 * g.O + g.H * g.W + CH * CW * 26 + 256) * sizeof(float), m->stream, dz, x, m->P(W), dx, m->splitk, B, g.H,
                           g.W, g.O, CH, CW, g.ph, g.pw, fuse ? x_act : (int)ADN_ACT_LINEAR, fuse ? 1 : 0);
        hipLaunchKernelGGL(conv1_direct_dw_reduce_kernel, dim3(cdiv(n, 64)), dim3(256), 0, m->stream, m->splitk, nslots, n, m->G(W), nW, m->G(b),
                           1, fuse ? x_bias_grad : nullptr);
        ADN_HIP_CHECK(hipGetLastError());
        if (fuse) *x_done = true;
        return ADN_OK;
    }
    if (!bias_ready) ADN_TRY(col_sum(dz, g.C, B * g.H * g.W, g.C, m->G(b), 1, m->stream));
    if (x16 && fast16(m, g)) {
        ADN_TRY(im2col16(m, dz, g, B, m->scratch, up));
        // (NN over the ping-pong kernel only where its 256-row tiles fill the device: 35840 x 152 x 2500 took 157 us on 140
        //  tiles against ~110 on the register-staged kernel, 15360 x 200 x 1368 68 against 37)
        // dx = d (this layer's input) = d (the activated output of the layer below): where the register-staged kernel runs it,
        // its epilogue multiplies by act'(x) and sums the columns -- that layer's bias gradient -- in the same pass
        static const bool no_epi = getenv("ADN_CAE_NO_DX_EPILOGUE") != nullptr;      // (A/B switch)
        if (x_done && x_bias_grad && !no_epi && R < 65536 && x_act != ADN_ACT_LINEAR) {
            GemmArgs q;
            q.layout = GEMM_NN; q.M = R; q.N = g.O; q.K = g.K; q.lda = g.ldk; q.ldb = g.O; q.C = dx; q.ldc = g.O;
            q.A = reinterpret_cast<const float*>(m->scratch); q.B = reinterpret_cast<const float*>(W16(m, W));
            q.A16 = m->scratch; q.B16 = W16(m, W); q.precision = ADN_PRECISION_BF16;
            q.Y = x; q.ldy = g.O; q.act_grad = x_act;
            int fused = 0;
            q.colsum = x_bias_grad; q.colsum_done = &fused; q.colsum_ws = m->cs_ws; q.colsum_ws_floats = m->cs_ws_floats;
            ADN_TRY(gemm(q, m->stream));
            if (!fused) ADN_TRY(col_sum(dx, g.O, R, g.O, x_bias_grad, 1, m->stream));
            *x_done = true;
        } else
        ADN_TRY(mm16(m, GEMM_NN, R, g.O, g.K, m->scratch, g.ldk, W16(m, W), g.O, dx, g.O, nullptr, ADN_ACT_LINEAR, 0, R >= 65536));
        return mm16(m, GEMM_TN, g.K, g.O, R, m->scratch, g.ldk, x16, g.O, m->G(W), g.O, nullptr, ADN_ACT_LINEAR, 1, true);
    }
    ADN_TRY(im2col(m, dz, g, B, m->scratch, up));
    ADN_TRY(mm(m, GEMM_NN, R, g.O, g.K, m->scratch, g.ldk, m->P(W), g.O, dx, g.O));
    return mm(m, GEMM_TN, g.K, g.O, R, m->scratch, g.ldk, x, g.O, m->G(W), g.O, nullptr, ADN_ACT_LINEAR, 1);
}

int maxpool_fwd(adn_cae* m, const float* x, int B, int H, int W, int C, int ph, int OH, int OW, float* y, uint8_t* arg,
                int post_act = ADN_ACT_LINEAR) {
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_of((int64_t)B * OH * OW * C)), dim3(256), 0, m->stream, x, y, arg, B, H, W, C, ph, OH, OW,
                       post_act);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}
// may the pooling's adjoint run from the pooled grid (maxpool_bwd_pooled_kernel)?  2 x 2 windows that cover every input position
bool pool_bwd_from_pooled(int H, int W, int C, int ph, int OH, int OW) {
    static const bool old_form = getenv("ADN_CAE_POOL_BWD_FULL") != nullptr;      // (A/B switch)
    return !old_form && 2 * OH - ph >= H && 2 * OW >= W && C % 4 == 0;
}

// `pooled` (optional): the pooling's OUTPUT, still as the forward pass wrote it -- act' is then taken from it (the pooled value is
// the winner's activation) instead of from the full-resolution activation act_y.  `dbias` / `*ready`: bf16 mode, the pooled-grid
// kernel also leaves the gradient's bf16 copy in t16 and adds the bias gradient; conv_bwd then skips both passes
int maxpool_bwd(adn_cae* m, const float* dy, const uint8_t* arg, int B, int H, int W, int C, int ph, int OH, int OW, float* dx,
                const float* act_y = nullptr, const float* pooled = nullptr, float* dbias = nullptr, bool* ready = nullptr, bool want16 = true) {
    if (ready) *ready = false;
    if (pool_bwd_from_pooled(H, W, C, ph, OH, OW) && (pooled || !act_y)) {
        const bool fuse = dbias && ready && !deterministic();
        const bool copy16 = fuse && m->precision == ADN_PRECISION_BF16 && want16;
        const int64_t total = (int64_t)B * OH * OW * (C / 4);
        // (with the fused bias sums: at most 2048 workgroups -- every workgroup ends with one float atomic per channel into the
        //  same C addresses, and 16384 of them queueing there cost more than the column-sum pass they replace)
        hipLaunchKernelGGL(maxpool_bwd_pooled_kernel, dim3(fuse ? std::min(grid_of(total), 2048) : grid_of(total)), dim3(256), (size_t)C * sizeof(float), m->stream,
                           reinterpret_cast<const float4*>(dy), reinterpret_cast<const uchar4*>(arg), reinterpret_cast<const float4*>(pooled),
                           reinterpret_cast<float4*>(dx), copy16 ? reinterpret_cast<cae_bf16x4*>(m->t16) : nullptr, fuse ? dbias : nullptr,
                           B, H, W, C / 4, ph, OH, OW, m->S);
        ADN_HIP_CHECK(hipGetLastError());
        if (fuse) *ready = true;
        return ADN_OK;
    }
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_of((int64_t)B * H * W * C)), dim3(256), 0, m->stream, dy, arg, dx, B, H, W, C, ph, OH, OW,
                       act_y, m->S);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}
int stage(adn_cae* m, const float* x, const float* target, int B, int flags) {
    const hipMemcpyKind kind = (flags & ADN_FLAG_DEVICE_INPUTS) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const size_t bytes = (size_t)B * m->H * m->W * sizeof(float);
    ADN_HIP_CHECK(hipMemcpyAsync(m->x0, x, bytes, kind, m->stream));
    if (target) ADN_HIP_CHECK(hipMemcpyAsync(m->target, target, bytes, kind, m->stream));
    return ADN_OK;
}

// DropoutLayer k (0..4) in place on [rows][Cp] (no-op in deterministic passes and in the variants without dropout)
int dropout_inplace(adn_cae* m, int layer, float* x, int64_t rows, int HW, int C, int Cp) {
    if (!m->drop || !m->training) return ADN_OK;
    const float p = layer == 0 ? 0.2f : 0.5f;                    // avletters_convae_drop.py:46 (p=0.2), DropoutLayer's default
    const uint32_t key = m->drop_seed ^ ((uint32_t)layer * 0x85EBCA77u) ^ (m->drop_counter * 0xC2B2AE3Du);
    hipLaunchKernelGGL(cae_dropout_kernel, dim3(grid_of(rows * Cp)), dim3(256), 0, m->stream, x, rows, HW, C, Cp, p, 1.f / (1.f - p), key);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// BatchNormLayer k on x [rows][C]: -> q.out (training: batch statistics kept for backward, running averages updated)
int bn_fwd(adn_cae* m, int k, const float* x, int rows) {
    BnSlot& q = m->bn[k];
    if (m->training)
        return batchnorm_forward_train(x, q.C, q.out, q.C, rows, q.C, m->P(q.gamma), m->P(q.beta), kBnEps, kBnAlpha, q.save_mean,
                                       q.save_inv, m->P(q.mean), m->P(q.inv_std), q.ws, m->stream);
    return batchnorm_forward_eval(x, q.C, q.out, q.C, rows, q.C, m->P(q.gamma), m->P(q.beta), m->P(q.mean), m->P(q.inv_std), m->stream);
}
// d(out) -> d(x) in place, dgamma / dbeta accumulated
int bn_bwd(adn_cae* m, int k, const float* x, float* d, int rows) {
    BnSlot& q = m->bn[k];
    return batchnorm_backward(x, q.C, d, q.C, d, q.C, rows, q.C, m->P(q.gamma), m->training ? q.save_mean : m->P(q.mean),
                              m->training ? q.save_inv : m->P(q.inv_std), m->training ? 1 : 0, m->G(q.gamma), m->G(q.beta), q.ws, m->stream);
}

int forward(adn_cae* m, int B, bool decode) {
    const int S = m->S;
    if (m->precision == ADN_PRECISION_BF16 && m->p16_dirty) {
        if (!m->p16) ADN_HIP_CHECK(hipMalloc((void**)&m->p16, (size_t)round_up((int64_t)m->flat_floats, 8) * 2));
        ADN_TRY(to_bf16(m->buf[0], m->p16, (size_t)(m->flat_floats / 8 * 8), m->stream));
        m->p16_dirty = false;
    }
    const int R1 = (int)rows_of(m->c1, B), R3 = (int)rows_of(m->c3, B), R5 = (int)rows_of(m->c5, B);
    const int P2 = B * m->p2h * m->p2w, P4 = B * m->p4h * m->p4w;
    // (a dropout layer works in place on the tensor the next layer reads: a BatchNorm output, a pooling output -- whose
    //  backward needs only the argmax codes -- or the staged input)
    ADN_TRY(dropout_inplace(m, 0, m->x0, (int64_t)B * m->H * m->W, m->H * m->W, 1, 1));
    // the one-channel first convolution and the pooling behind it in one kernel where nothing else needs the full-resolution
    // activation: no BatchNorm between them, no dropout (whose in-place rescaling of the pooled tensor sends the backward pass
    // to it for act')
    static const bool no_fuse1 = getenv("ADN_CAE_NO_FUSE1") != nullptr;       // (A/B switch)
    m->fused1 = direct1(m, m->c1) && m->bn_mode != 2 && !m->drop && !no_fuse1 && 2 * m->p2h == m->c1.OH && 2 * m->p2w == m->c1.OW;
    const float* t = m->a1;
    if (m->fused1) {
        hipLaunchKernelGGL(conv1_pool_direct_fwd_kernel<25>, dim3(std::min(B, 2048)), dim3(256), (size_t)m->H * m->W * sizeof(float), m->stream,
                           m->x0, m->P(m->W1), m->P(m->b1), m->p2, reinterpret_cast<cae_bf16x4*>(m->p2_16), m->arg2, B, m->H, m->W, m->F1,
                           m->p2h, m->p2w, S);
        ADN_HIP_CHECK(hipGetLastError());
    } else {
    ADN_TRY(conv_fwd(m, m->x0, m->c1, B, m->cols1, m->W1, m->b1, m->a1));
    if (m->bn_mode == 2) { ADN_TRY(bn_fwd(m, 0, t, R1)); t = m->bn[0].out; }
    ADN_TRY(maxpool_fwd(m, t, B, m->c1.OH, m->c1.OW, m->F1, 0, m->p2h, m->p2w, m->p2, m->arg2));
    }
    float* u = m->p2;
    if (m->bn_mode == 1) { ADN_TRY(bn_fwd(m, 0, u, P2)); u = m->bn[0].out; }
    ADN_TRY(dropout_inplace(m, 1, u, P2, m->p2h * m->p2w, m->F1L, m->F1));
    m->in3 = u;
    // bf16 mode, activation directly in front of the pooling and nobody else reading it (a dropout layer that rescales the pooled
    // tensor in place sends the backward pass to the full-resolution activation for act'): a3 then holds PRE-activations
    static const bool no_preact = getenv("ADN_CAE_NO_PREACT") != nullptr;       // (A/B switch)
    m->preact3 = fast16(m, m->c3) && m->bn_mode != 2 && !m->drop && !no_preact &&
                 pool_bwd_from_pooled(m->c3.OH, m->c3.OW, m->F2, 1, m->p4h, m->p4w);
    m->p2_16_valid = m->fused1 && u == m->p2;         // (a BatchNorm behind the pooling puts another tensor in front of conv3)
    ADN_TRY(conv_fwd(m, u, m->c3, B, m->cols3, m->W3, m->b3, m->a3, m->preact3));
    m->p2_16_valid = false;
    t = m->a3;
    if (m->bn_mode == 2) { ADN_TRY(bn_fwd(m, 1, t, R3)); t = m->bn[1].out; }
    ADN_TRY(maxpool_fwd(m, t, B, m->c3.OH, m->c3.OW, m->F2, 1, m->p4h, m->p4w, m->p4, m->arg4, m->preact3 ? S : ADN_ACT_LINEAR));
    u = m->p4;
    if (m->bn_mode == 1) { ADN_TRY(bn_fwd(m, 1, u, P4)); u = m->bn[1].out; }
    ADN_TRY(dropout_inplace(m, 2, u, P4, m->p4h * m->p4w, m->F2L, m->F2));
    m->in5 = u;
    ADN_TRY(conv_fwd(m, u, m->c5, B, m->cols5, m->W5, m->b5, m->a5));
    u = m->a5;
    if (m->bn_mode == 2) { ADN_TRY(bn_fwd(m, 2, u, R5)); u = m->bn[2].out; }             // per channel
    if (m->bn_mode == 1) { ADN_TRY(bn_fwd(m, 2, u, B)); u = m->bn[2].out; }              // per flattened feature
    if (m->drop && m->training && u == m->a5) {       // (dropout3 must not overwrite a5: its nonlinearity's backward reads it)
        ADN_HIP_CHECK(hipMemcpyAsync(m->f6d, m->a5, (size_t)B * m->flat * 4, hipMemcpyDeviceToDevice, m->stream));
        u = m->f6d;
    }
    ADN_TRY(dropout_inplace(m, 3, u, (int64_t)B * m->c5.OH * m->c5.OW, m->c5.OH * m->c5.OW, m->F3L, m->F3));
    m->in7 = u;
    ADN_TRY(mm(m, GEMM_NN, B, m->D7, m->flat, u, m->flat, m->P(m->W7), m->D7, m->a7, m->D7, m->P(m->b7), S));
    u = m->a7;
    if (m->bn_mode) { ADN_TRY(bn_fwd(m, 3, u, B)); u = m->bn[3].out; }
    if (m->drop && m->training && u == m->a7) {
        ADN_HIP_CHECK(hipMemcpyAsync(m->a7d, m->a7, (size_t)B * m->D7 * 4, hipMemcpyDeviceToDevice, m->stream));
        u = m->a7d;
    }
    ADN_TRY(dropout_inplace(m, 4, u, B, 1, m->D7, m->D7));
    m->inb = u;
    ADN_TRY(mm(m, GEMM_NN, B, m->NB, m->D7, u, m->D7, m->P(m->Wb), m->ldb, m->code, m->ldb, m->P(m->bb)));
    if (!decode) return ADN_OK;
    ADN_TRY(mm(m, GEMM_NT, B, m->D7, m->NB, m->code, m->ldb, m->P(m->Wb), m->ldb, m->a8, m->D7, m->P(m->b8)));
    ADN_TRY(mm(m, GEMM_NT, B, m->flat, m->D7, m->a8, m->D7, m->P(m->W7), m->D7, m->a9, m->flat, m->P(m->b9), S));
    ADN_TRY(deconv_fwd(m, m->a9, m->a9_16, m->d11, B, m->W5, m->b11, m->a11));
    // (upscale2d12 / upscale2d14 are folded into the deconvolutions behind them)
    ADN_TRY(deconv_fwd(m, m->a11, m->u12_16, m->d13, B, m->W3, m->b13, m->a13, 1));
    return deconv_fwd(m, m->a13, nullptr, m->d15, B, m->W1, m->b15, m->a15, 1);
}

// loss_dev[0] = mean((recon - target)^2); gA = d loss / d recon when want_grad
int mse(adn_cae* m, int B, bool want_grad) {
    const int64_t n = (int64_t)B * m->H * m->W;
    hipLaunchKernelGGL(mse_grad_kernel, dim3(grid_of(n)), dim3(256), 0, m->stream, m->a15, m->target, m->gA, n, 1.f);
    ADN_HIP_CHECK(hipMemsetAsync(m->loss_dev, 0, sizeof(float), m->stream));
    ADN_TRY(dot_all(m->gA, m->W, m->gA, m->W, B * m->H, m->W, m->loss_dev, nullptr, m->stream));
    hipLaunchKernelGGL(scale_scalar_kernel, dim3(1), dim3(1), 0, m->stream, m->loss_dev, 1.f / (float)n);
    if (want_grad)
        hipLaunchKernelGGL(mse_grad_kernel, dim3(grid_of(n)), dim3(256), 0, m->stream, m->a15, m->target, m->gA, n, 2.f / (float)n);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int backward(adn_cae* m, int B) {
    const int S = m->S;
    hipStream_t s = m->stream;
    ADN_HIP_CHECK(hipMemsetAsync(m->buf[1], 0, m->flat_floats * sizeof(float), s));
    float *gA = m->gA, *gB = m->gB;
    const int F1 = m->F1, F2 = m->F2;
    const int R1 = (int)rows_of(m->c1, B), R3 = (int)rows_of(m->c3, B), R5 = (int)rows_of(m->c5, B);
    const int P2 = B * m->p2h * m->p2w, P4 = B * m->p4h * m->p4w;
    // decoder
    ADN_TRY(act_backward(gA, 1, m->a15, 1, B * m->H * m->W, 1, S, s));
    bool a13_done = false;                             // (the direct kernel applies act'(a13) and takes b13's gradient on the way)
    ADN_TRY(deconv_bwd(m, m->d15, B, m->a13, nullptr, gA, m->W1, m->b15, gB, 1, S, m->G(m->b13), &a13_done));   // gB = d a13 (through upscale2d14)
    if (!a13_done) ADN_TRY(act_backward(gB, F1, m->a13, F1, B * m->d13.H * m->d13.W, F1, S, s));
    bool a11_done = false;
    ADN_TRY(deconv_bwd(m, m->d13, B, m->a11, m->u12_16, gB, m->W3, m->b13, gA, 1, S, m->G(m->b11), &a11_done, a13_done));   // gA = d a11 (through upscale2d12)
    if (!a11_done) ADN_TRY(act_backward(gA, F2, m->a11, F2, B * m->d11.H * m->d11.W, F2, S, s));
    ADN_TRY(deconv_bwd(m, m->d11, B, m->a9, m->a9_16, gA, m->W5, m->b11, gB, 0, ADN_ACT_LINEAR, nullptr, nullptr, a11_done));   // gB = d a9 (as [B][flat])
    ADN_TRY(act_backward(gB, m->flat, m->a9, m->flat, B, m->flat, S, s));
    ADN_TRY(col_sum(gB, m->flat, B, m->flat, m->G(m->b9), 1, s));
    ADN_TRY(mm(m, GEMM_TN, m->flat, m->D7, B, gB, m->flat, m->a8, m->D7, m->G(m->W7), m->D7, nullptr, ADN_ACT_LINEAR, 1));
    ADN_TRY(mm(m, GEMM_NN, B, m->D7, m->flat, gB, m->flat, m->P(m->W7), m->D7, gA, m->D7));          // gA = d a8
    ADN_TRY(col_sum(gA, m->D7, B, m->D7, m->G(m->b8), 1, s));
    ADN_TRY(mm(m, GEMM_TN, m->D7, m->NB, B, gA, m->D7, m->code, m->ldb, m->G(m->Wb), m->ldb, nullptr, ADN_ACT_LINEAR, 1));
    ADN_TRY(mm(m, GEMM_NN, B, m->NB, m->D7, gA, m->D7, m->P(m->Wb), m->ldb, gB, m->ldb));             // gB = d code
    // encoder: every optional layer is undone where the forward pass applied it (dropout: the same mask on the gradient)
    ADN_TRY(col_sum(gB, m->ldb, B, m->NB, m->G(m->bb), 1, s));
    ADN_TRY(mm(m, GEMM_TN, m->D7, m->NB, B, m->inb, m->D7, gB, m->ldb, m->G(m->Wb), m->ldb, nullptr, ADN_ACT_LINEAR, 1));
    ADN_TRY(mm(m, GEMM_NT, B, m->D7, m->NB, gB, m->ldb, m->P(m->Wb), m->ldb, gA, m->D7));             // gA = d (bottleneck input)
    ADN_TRY(dropout_inplace(m, 4, gA, B, 1, m->D7, m->D7));
    if (m->bn_mode) ADN_TRY(bn_bwd(m, 3, m->a7, gA, B));                                   // gA = d a7
    ADN_TRY(act_backward(gA, m->D7, m->a7, m->D7, B, m->D7, S, s));
    ADN_TRY(col_sum(gA, m->D7, B, m->D7, m->G(m->b7), 1, s));
    ADN_TRY(mm(m, GEMM_TN, m->flat, m->D7, B, m->in7, m->flat, gA, m->D7, m->G(m->W7), m->D7, nullptr, ADN_ACT_LINEAR, 1));
    ADN_TRY(mm(m, GEMM_NT, B, m->flat, m->D7, gA, m->D7, m->P(m->W7), m->D7, gB, m->flat));           // gB = d (dense input)
    ADN_TRY(dropout_inplace(m, 3, gB, (int64_t)B * m->c5.OH * m->c5.OW, m->c5.OH * m->c5.OW, m->F3L, m->F3));
    if (m->bn_mode == 1) ADN_TRY(bn_bwd(m, 2, m->a5, gB, B));
    if (m->bn_mode == 2) ADN_TRY(bn_bwd(m, 2, m->a5, gB, R5));                             // gB = d a5
    ADN_TRY(act_backward(gB, m->flat, m->a5, m->flat, B, m->flat, S, s));
    ADN_TRY(conv_bwd(m, m->c5, B, m->cols5, gB, m->W5, m->b5, gA));                       // gA = d (conv5 input)
    ADN_TRY(dropout_inplace(m, 2, gA, P4, m->p4h * m->p4w, m->F2L, F2));
    if (m->bn_mode == 1) ADN_TRY(bn_bwd(m, 1, m->p4, gA, P4));                             // gA = d p4
    // (the pooling's input is the activation itself unless a BatchNorm sits between them: act' rides on the pooling's backward)
    // (the pooled tensor still holds the winners' activations unless a dropout layer rescaled it in place)
    const bool pooled_intact = !(m->drop && m->training && m->bn_mode == 0);
    bool ready3 = false, ready1 = false;
    ADN_CHECK(!m->preact3 || (m->bn_mode != 2 && pooled_intact), ADN_ERR_STATE, "conv AE: pre-activation forward without intact pooled values");
    ADN_TRY(maxpool_bwd(m, gA, m->arg4, B, m->c3.OH, m->c3.OW, F2, 1, m->p4h, m->p4w, gB, m->bn_mode == 2 ? nullptr : m->a3,
                        (m->bn_mode != 2 && pooled_intact) ? m->p4 : nullptr, m->bn_mode == 2 ? nullptr : m->G(m->b3), &ready3));   // gB = d (pool input)
    if (m->bn_mode == 2) {
        ADN_TRY(bn_bwd(m, 1, m->a3, gB, R3));                                             // gB = d a3
        ADN_TRY(act_backward(gB, F2, m->a3, F2, R3, F2, S, s));
    }
    ADN_TRY(conv_bwd(m, m->c3, B, m->cols3, gB, m->W3, m->b3, gA, ready3));   // gA = d (conv3 input)
    ADN_TRY(dropout_inplace(m, 1, gA, P2, m->p2h * m->p2w, m->F1L, F1));
    if (m->bn_mode == 1) ADN_TRY(bn_bwd(m, 0, m->p2, gA, P2));                             // gA = d p2
    if (m->fused1) {                                  // pooling adjoint + weight and bias gradient of conv1 in one kernel
        ADN_CHECK(m->bn_mode != 2 && pooled_intact, ADN_ERR_STATE, "conv AE: fused first layer without intact pooled values");
        const int nslots = std::min(B, 512), nW = m->c1.K * F1, n = nW + F1;
        ADN_CHECK((size_t)nslots * n <= m->splitk_floats, ADN_ERR_STATE, "conv AE: slot workspace too small");
        hipLaunchKernelGGL(conv1_pool_direct_dw_kernel<25>, dim3(nslots), dim3(256), (size_t)(m->H * m->W + n) * sizeof(float), s,
                           m->x0, gA, m->p2, m->arg2, m->splitk, B, m->H, m->W, F1, m->p2h, m->p2w, S);
        hipLaunchKernelGGL(conv1_direct_dw_reduce_kernel, dim3(cdiv(n, 64)), dim3(256), 0, s, m->splitk, nslots, n, m->G(m->W1), nW, m->G(m->b1));
        ADN_HIP_CHECK(hipGetLastError());
        m->grads_valid = true;
        return ADN_OK;
    }
    // (conv1 is not a bf16-operand layer (C_in = 1): no bf16 copy of its gradient, the bias sums ride along all the same)
    ADN_TRY(maxpool_bwd(m, gA, m->arg2, B, m->c1.OH, m->c1.OW, F1, 0, m->p2h, m->p2w, gB, m->bn_mode == 2 ? nullptr : m->a1,
                        (m->bn_mode != 2 && pooled_intact) ? m->p2 : nullptr, m->bn_mode == 2 ? nullptr : m->G(m->b1), &ready1,
                        fast16(m, m->c1)));           // gB = d (pool input)
    if (m->bn_mode == 2) {
        ADN_TRY(bn_bwd(m, 0, m->a1, gB, R1));                                             // gB = d a1
        ADN_TRY(act_backward(gB, F1, m->a1, F1, R1, F1, S, s));
    }
    ADN_TRY(conv_bwd(m, m->c1, B, m->cols1, gB, m->W1, m->b1, nullptr, ready1, direct1(m, m->c1) ? m->x0 : nullptr));
    m->grads_valid = true;
    return ADN_OK;
}

int fetch(adn_cae* m, void* dst, const void* src, size_t bytes, int flags) {
    if (flags & ADN_FLAG_DEVICE_OUTPUTS) {
        ADN_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, m->stream));
    } else {
        ADN_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, m->stream));
        ADN_HIP_CHECK(hipStreamSynchronize(m->stream));
    }
    return ADN_OK;
}

// host <-> device layout conversion of one tensor (Lasagne layout on the host side)
void to_internal(const adn_cae* m, const Tensor& t, const float* host, std::vector<float>& dev) {
    dev.assign(t.phys, 0.f);
    if (t.kind == 1) {                               // (O, C, kh, kw) -> [(i*kw + j)*Cp + c][o] (row stride Op), filters flipped
        const int O = (int)t.dims[0], C = (int)t.dims[1], kh = (int)t.dims[2], kw = (int)t.dims[3];
        for (int o = 0; o < O; ++o) for (int c = 0; c < C; ++c) for (int i = 0; i < kh; ++i) for (int j = 0; j < kw; ++j)
            dev[((size_t)(i * kw + j) * t.Cp + c) * t.Op + o] = host[(((size_t)o * C + c) * kh + (kh - 1 - i)) * kw + (kw - 1 - j)];
    } else if (t.kind == 2) {                        // rows (c, h, w) -> rows (h, w, c)
        const int hh = m->c5.OH, ww = m->c5.OW, U = (int)t.dims[1];
        for (int c = 0; c < m->F3; ++c) for (int y = 0; y < hh; ++y) for (int x = 0; x < ww; ++x)
            memcpy(&dev[((size_t)(y * ww + x) * m->F3 + c) * U], &host[((size_t)(c * hh + y) * ww + x) * U], (size_t)U * 4);
    } else if (t.kind == 3) {                        // bottleneck W: rows padded to ldb
        const int U = (int)t.dims[1];
        for (int r = 0; r < (int)t.dims[0]; ++r) memcpy(&dev[(size_t)r * m->ldb], &host[(size_t)r * U], (size_t)U * 4);
    } else if (t.kind == 4) {                        // dense9.b: (c, h, w) -> (h, w, c)
        const int hh = m->c5.OH, ww = m->c5.OW;
        for (int c = 0; c < m->F3; ++c) for (int y = 0; y < hh; ++y) for (int x = 0; x < ww; ++x)
            dev[(size_t)(y * ww + x) * m->F3 + c] = host[(size_t)(c * hh + y) * ww + x];
    } else {
        memcpy(dev.data(), host, t.floats * 4);
    }
}

void to_host(const adn_cae* m, const Tensor& t, const std::vector<float>& dev, float* host) {
    if (t.kind == 1) {
        const int O = (int)t.dims[0], C = (int)t.dims[1], kh = (int)t.dims[2], kw = (int)t.dims[3];
        for (int o = 0; o < O; ++o) for (int c = 0; c < C; ++c) for (int i = 0; i < kh; ++i) for (int j = 0; j < kw; ++j)
            host[(((size_t)o * C + c) * kh + (kh - 1 - i)) * kw + (kw - 1 - j)] = dev[((size_t)(i * kw + j) * t.Cp + c) * t.Op + o];
    } else if (t.kind == 2) {
        const int hh = m->c5.OH, ww = m->c5.OW, U = (int)t.dims[1];
        for (int c = 0; c < m->F3; ++c) for (int y = 0; y < hh; ++y) for (int x = 0; x < ww; ++x)
            memcpy(&host[((size_t)(c * hh + y) * ww + x) * U], &dev[((size_t)(y * ww + x) * m->F3 + c) * U], (size_t)U * 4);
    } else if (t.kind == 3) {
        const int U = (int)t.dims[1];
        for (int r = 0; r < (int)t.dims[0]; ++r) memcpy(&host[(size_t)r * U], &dev[(size_t)r * m->ldb], (size_t)U * 4);
    } else if (t.kind == 4) {
        const int hh = m->c5.OH, ww = m->c5.OW;
        for (int c = 0; c < m->F3; ++c) for (int y = 0; y < hh; ++y) for (int x = 0; x < ww; ++x)
            host[(size_t)(c * hh + y) * ww + x] = dev[(size_t)(y * ww + x) * m->F3 + c];
    } else {
        memcpy(host, dev.data(), t.floats * 4);
    }
}

// get_output(..., deterministic=...): training passes are adn_cae_compute_grads unless ADN_FLAG_DETERMINISTIC, and
// adn_cae_loss / adn_cae_forward only with ADN_FLAG_STOCHASTIC (avletters/avletters_convae.py:254-268: train and
// train_cost_fn are non-deterministic, eval_cost_fn and recon_fn deterministic)
void begin_pass(adn_cae* m, int flags, bool training_default) {
    m->training = (flags & ADN_FLAG_STOCHASTIC) ? true : (flags & ADN_FLAG_DETERMINISTIC) ? false : training_default;
    if (m->training && m->bn_mode) m->p16_dirty = true;          // the running averages live in the parameter buffer
}
void end_pass(adn_cae* m) {
    if (m->training && m->drop) m->drop_counter += 1;
    m->training = false;
}

}  // namespace

extern "C" {

int adn_cae_set_dropout_state(adn_cae* m, uint32_t seed, uint32_t counter) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    m->drop_seed = seed; m->drop_counter = counter;
    return ADN_OK;
}

int adn_cae_create(const adn_cae_config* cfg, adn_cae** out) {
    ADN_CHECK(cfg && out, ADN_ERR_INVALID, "null argument");
    ADN_CHECK(cfg->image_h >= 22 && cfg->image_w >= 22 && cfg->image_h <= 512 && cfg->image_w <= 512, ADN_ERR_INVALID,
              "image size out of range (the three valid convolutions and two poolings need at least 22 x 22)");
    ADN_CHECK(cfg->dense >= 1 && cfg->bottleneck >= 1, ADN_ERR_INVALID, "layer widths must be positive");
    ADN_CHECK(cfg->precision == ADN_PRECISION_F32 || cfg->precision == ADN_PRECISION_BF16, ADN_ERR_INVALID, "unknown precision");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device visible"); return ADN_ERR_NO_DEVICE; }
    ADN_CHECK(cfg->variant >= ADN_CAE_NORMAL && cfg->variant <= ADN_CAE_BNDROP, ADN_ERR_INVALID, "unknown conv auto-encoder variant");
    adn_cae* m = new adn_cae();
    m->cfg = *cfg; m->H = cfg->image_h; m->W = cfg->image_w; m->D7 = cfg->dense; m->NB = cfg->bottleneck;
    m->ldb = (int)round_up(m->NB, 4); m->precision = cfg->precision;
    m->variant = cfg->variant;
    m->drop = m->variant == ADN_CAE_DROPOUT || m->variant == ADN_CAE_BNDROP;
    m->bn_mode = m->variant == ADN_CAE_BATCHNORM ? 1 : m->variant == ADN_CAE_BNDROP ? 2 : 0;
    m->S = m->variant == ADN_CAE_BNDROP ? ADN_ACT_SCALED_TANH_LECUN : ADN_ACT_SCALED_TANH;     // avletters_convae_bndrop.py:8
    if (m->variant == ADN_CAE_DROPOUT) { m->F1L = 125; m->F2L = 300; m->F3L = 400; }           // avletters_convae_drop.py:36-38
    m->F1 = pad_channels(m->F1L, true); m->F2 = pad_channels(m->F2L, false); m->F3 = pad_channels(m->F3L, false);
    if (m->F3 != m->F3L || (m->bn_mode && m->D7 % 4)) {
        delete m; set_error("layer widths: the last filter count and (with BatchNorm) the dense width must be multiples of 4");
        return ADN_ERR_INVALID;
    }
    const int F1 = m->F1, F2 = m->F2, F3 = m->F3, F1L = m->F1L, F2L = m->F2L, F3L = m->F3L;
    m->c1 = conv_geom(m->H, m->W, 1, 5, F1);
    m->p2h = (m->c1.OH - 2) / 2 + 1; m->p2w = (m->c1.OW - 2) / 2 + 1;
    m->c3 = conv_geom(m->p2h, m->p2w, F1, 5, F2);
    m->p4h = (m->c3.OH + 2 - 2) / 2 + 1; m->p4w = (m->c3.OW - 2) / 2 + 1;
    m->c5 = conv_geom(m->p4h, m->p4w, F2, 3, F3);
    if (m->c5.OH < 1 || m->c5.OW < 1) { delete m; set_error("image too small for the encoder"); return ADN_ERR_INVALID; }
    m->flat = F3 * m->c5.OH * m->c5.OW;
    // deconv layers = adjoints of convolutions on their OUTPUT images (Deconv2DLayer full padding = valid conv's adjoint)
    m->d11 = conv_geom(m->c5.OH + 2, m->c5.OW + 2, F2, 3, F3);
    m->d13 = conv_geom(2 * m->d11.H + 4, 2 * m->d11.W + 4, F1, 5, F2);
    m->d15 = conv_geom(2 * m->d13.H + 4 - 2, 2 * m->d13.W + 4, 1, 5, F1, 1, 0);
    if (m->d15.H != m->H || m->d15.W != m->W) {
        delete m; set_error("the decoder does not reproduce this image size (use e.g. 30 x 40)"); return ADN_ERR_INVALID;
    }
    // layer names of the variant's model-zoo file (the BatchNorm file numbers its layers differently); parameters in
    // lasagne.layers.get_all_params order
    const bool bnfile = m->variant == ADN_CAE_BATCHNORM;
    const char* n_c1 = "conv2d1"; const char* n_c3 = bnfile ? "conv2d4" : "conv2d3"; const char* n_c5 = bnfile ? "conv2d7" : "conv2d5";
    const char* n_d7 = bnfile ? "dense10" : "dense7"; const char* n_d8 = bnfile ? "dense12" : "dense8"; const char* n_d9 = bnfile ? "dense13" : "dense9";
    const char* n_dc11 = bnfile ? "deconv2d19" : "deconv2d11"; const char* n_dc13 = bnfile ? "deconv2d17" : "deconv2d13";
    const char* bn_names[4] = {bnfile ? "batchnorm2" : "batchnorm1", bnfile ? "batchnorm3" : "batchnorm2",
                               bnfile ? "batchnorm8" : "batchnorm3", bnfile ? "batchnorm11" : "batchnorm4"};
    auto nm = [](const char* layer, const char* p) { return std::string(layer) + "." + p; };
    auto add_bn = [&](int k, int logical, int physical, bool flat_order) {
        if (!m->bn_mode) return;
        BnSlot& q = m->bn[k];
        q.on = true; q.C = physical;
        const int kind = flat_order ? 4 : 0;
        q.beta = add_tensor(m, nm(bn_names[k], "beta").c_str(), kind, {logical}, physical);
        q.gamma = add_tensor(m, nm(bn_names[k], "gamma").c_str(), kind, {logical}, physical);
        q.mean = add_tensor(m, nm(bn_names[k], "mean").c_str(), kind, {logical}, physical);
        q.inv_std = add_tensor(m, nm(bn_names[k], "inv_std").c_str(), kind, {logical}, physical);
    };
    m->W1 = add_tensor(m, nm(n_c1, "W").c_str(), 1, {F1L, 1, 5, 5}, F1, 1); m->b1 = add_tensor(m, nm(n_c1, "b").c_str(), 0, {F1L}, F1);
    add_bn(0, F1L, F1, false);
    m->W3 = add_tensor(m, nm(n_c3, "W").c_str(), 1, {F2L, F1L, 5, 5}, F2, F1); m->b3 = add_tensor(m, nm(n_c3, "b").c_str(), 0, {F2L}, F2);
    add_bn(1, F2L, F2, false);
    m->W5 = add_tensor(m, nm(n_c5, "W").c_str(), 1, {F3L, F2L, 3, 3}, F3, F2); m->b5 = add_tensor(m, nm(n_c5, "b").c_str(), 0, {F3L}, F3);
    if (m->bn_mode == 1) add_bn(2, m->flat, m->flat, true);      // per flattened feature, (c, h, w) order on the host
    else add_bn(2, F3L, F3, false);                              // per channel
    m->W7 = add_tensor(m, nm(n_d7, "W").c_str(), 2, {m->flat, m->D7}); m->b7 = add_tensor(m, nm(n_d7, "b").c_str(), 0, {m->D7});
    add_bn(3, m->D7, m->D7, false);
    m->Wb = add_tensor(m, "bottleneck.W", 3, {m->D7, m->NB});
    m->bb = add_tensor(m, "bottleneck.b", 0, {m->NB});
    m->b8 = add_tensor(m, nm(n_d8, "b").c_str(), 0, {m->D7}); m->b9 = add_tensor(m, nm(n_d9, "b").c_str(), 4, {m->flat});
    m->b11 = add_tensor(m, nm(n_dc11, "b").c_str(), 0, {F2L}, F2); m->b13 = add_tensor(m, nm(n_dc13, "b").c_str(), 0, {F1L}, F1);
    m->b15 = add_tensor(m, "deconv2d14.b", 0, {1});
    for (int k = 0; k < 4; ++k) {
        if (hipMalloc((void**)&m->buf[k], m->flat_floats * sizeof(float)) != hipSuccess ||
            hipMemset(m->buf[k], 0, m->flat_floats * sizeof(float)) != hipSuccess) {
            set_error("allocating the parameter buffers failed"); adn_cae_destroy(m); return ADN_ERR_HIP;
        }
    }
    *out = m;
    return ADN_OK;
}

void adn_cae_destroy(adn_cae* m) {
    if (!m) return;
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    for (int k = 0; k < 4; ++k) if (m->buf[k]) (void)hipFree(m->buf[k]);
    if (m->slab) (void)hipFree(m->slab);
    if (m->p16) (void)hipFree(m->p16);
    delete m;
}

int adn_cae_set_stream(adn_cae* m, void* hip_stream) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    m->stream = static_cast<hipStream_t>(hip_stream);
    return ADN_OK;
}

int adn_cae_num_params(const adn_cae* m) { return m ? (int)m->params.size() : 0; }

int adn_cae_param_info(const adn_cae* m, int index, adn_param_info_t* info) {
    ADN_CHECK(m && info, ADN_ERR_INVALID, "null argument");
    ADN_CHECK(index >= 0 && index < (int)m->params.size(), ADN_ERR_INVALID, "parameter index out of range");
    const Tensor& t = m->params[index];
    memset(info, 0, sizeof(*info));
    strncpy(info->name, t.name.c_str(), sizeof(info->name) - 1);
    info->ndim = t.ndim <= 2 ? t.ndim : 2;           // 4-d filters are reported as (out, in*kh*kw)
    info->dims[0] = t.dims[0];
    info->dims[1] = t.ndim == 4 ? t.dims[1] * t.dims[2] * t.dims[3] : (t.ndim >= 2 ? t.dims[1] : 1);
    info->numel = (int64_t)t.floats;
    return ADN_OK;
}

int adn_cae_read_tensor(adn_cae* m, int buffer, int index, float* host_dst) {
    ADN_CHECK(m && host_dst, ADN_ERR_INVALID, "null argument");
    ADN_CHECK(buffer >= 0 && buffer < 4 && index >= 0 && index < (int)m->params.size(), ADN_ERR_INVALID, "bad buffer / index");
    const Tensor& t = m->params[index];
    const size_t phys = t.phys;
    std::vector<float> dev(phys);
    ADN_HIP_CHECK(hipStreamSynchronize(m->stream));
    ADN_HIP_CHECK(hipMemcpy(dev.data(), m->buf[buffer] + t.off, phys * 4, hipMemcpyDeviceToHost));
    to_host(m, t, dev, host_dst);
    return ADN_OK;
}

int adn_cae_write_tensor(adn_cae* m, int buffer, int index, const float* host_src) {
    ADN_CHECK(m && host_src, ADN_ERR_INVALID, "null argument");
    ADN_CHECK(buffer >= 0 && buffer < 4 && index >= 0 && index < (int)m->params.size(), ADN_ERR_INVALID, "bad buffer / index");
    const Tensor& t = m->params[index];
    std::vector<float> dev;
    to_internal(m, t, host_src, dev);
    ADN_HIP_CHECK(hipStreamSynchronize(m->stream));
    ADN_HIP_CHECK(hipMemcpy(m->buf[buffer] + t.off, dev.data(), dev.size() * 4, hipMemcpyHostToDevice));
    if (buffer == ADN_BUF_PARAM) m->p16_dirty = true;
    return ADN_OK;
}

int adn_cae_flat_buffer(adn_cae* m, int buffer, float** ptr, int64_t* floats) {
    ADN_CHECK(m && ptr && floats, ADN_ERR_INVALID, "null argument");
    ADN_CHECK(buffer >= 0 && buffer < 4, ADN_ERR_INVALID, "bad buffer id");
    *ptr = m->buf[buffer]; *floats = (int64_t)m->flat_floats;
    return ADN_OK;
}

int adn_cae_forward(adn_cae* m, const float* x, int B, int flags, float* recon, float* code) {
    ADN_CHECK(m && x, ADN_ERR_INVALID, "null argument");
    ADN_CHECK(B >= 1 && B <= (1 << 16), ADN_ERR_INVALID, "batch size out of range");
    ADN_TRY(ensure_ws(m, B));
    ADN_TRY(stage(m, x, nullptr, B, flags));
    begin_pass(m, flags, false);
    ADN_TRY(forward(m, B, recon != nullptr));
    end_pass(m);
    if (code) {
        if (flags & ADN_FLAG_DEVICE_OUTPUTS) {
            ADN_HIP_CHECK(hipMemcpy2DAsync(code, (size_t)m->NB * 4, m->code, (size_t)m->ldb * 4, (size_t)m->NB * 4, B,
                                           hipMemcpyDeviceToDevice, m->stream));
        } else {
            ADN_HIP_CHECK(hipMemcpy2DAsync(code, (size_t)m->NB * 4, m->code, (size_t)m->ldb * 4, (size_t)m->NB * 4, B,
                                           hipMemcpyDeviceToHost, m->stream));
            ADN_HIP_CHECK(hipStreamSynchronize(m->stream));
        }
    }
    if (recon) return fetch(m, recon, m->a15, (size_t)B * m->H * m->W * 4, flags);
    return ADN_OK;
}

int adn_cae_loss(adn_cae* m, const float* x, const float* target, int B, int flags, float* loss) {
    ADN_CHECK(m && x && loss, ADN_ERR_INVALID, "null argument");
    ADN_CHECK(B >= 1 && B <= (1 << 16), ADN_ERR_INVALID, "batch size out of range");
    ADN_TRY(ensure_ws(m, B));
    ADN_TRY(stage(m, x, target ? target : x, B, flags));
    begin_pass(m, flags, false);
    ADN_TRY(forward(m, B, true));
    end_pass(m);
    ADN_TRY(mse(m, B, false));
    return fetch(m, loss, m->loss_dev, sizeof(float), flags);
}

int adn_cae_compute_grads(adn_cae* m, const float* x, const float* target, int B, int flags, float* loss) {
    ADN_CHECK(m && x, ADN_ERR_INVALID, "null argument");
    ADN_CHECK(B >= 1 && B <= (1 << 16), ADN_ERR_INVALID, "batch size out of range");
    ADN_TRY(ensure_ws(m, B));
    ADN_TRY(stage(m, x, target ? target : x, B, flags));
    begin_pass(m, flags, true);
    ADN_TRY(forward(m, B, true));
    ADN_TRY(mse(m, B, true));
    ADN_TRY(backward(m, B));
    end_pass(m);
    if (loss) return fetch(m, loss, m->loss_dev, sizeof(float), flags);
    return ADN_OK;
}

int adn_cae_apply_adadelta(adn_cae* m, float learning_rate, float rho, float epsilon) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    ADN_CHECK(m->grads_valid, ADN_ERR_STATE, "adn_cae_apply_adadelta called without gradients");
    ADN_TRY(adadelta_update(m->buf[0], m->buf[1], m->buf[2], m->buf[3], (int64_t)m->flat_floats, learning_rate, rho, epsilon, m->stream));
    m->grads_valid = false;
    m->p16_dirty = true;
    return ADN_OK;
}

int adn_cae_apply_adam(adn_cae* m, float learning_rate) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    ADN_CHECK(m->grads_valid, ADN_ERR_STATE, "adn_cae_apply_adam called without gradients");
    m->adam_t += 1;
    const float t = (float)m->adam_t;
    const float a_t = learning_rate * sqrtf(1.f - powf(0.999f, t)) / (1.f - powf(0.9f, t));
    ADN_TRY(adam_update(m->buf[0], m->buf[1], m->buf[2], m->buf[3], (int64_t)m->flat_floats, a_t, 0.9f, 0.999f, 1e-8f, m->stream));
    m->grads_valid = false;
    m->p16_dirty = true;
    return ADN_OK;
}

int adn_cae_synchronize(adn_cae* m) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    ADN_HIP_CHECK(hipStreamSynchronize(m->stream));
    return ADN_OK;
}

}  // extern "C"
