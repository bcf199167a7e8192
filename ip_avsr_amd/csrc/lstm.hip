// LSTM recurrence, forward and BPTT (Lasagne LSTMLayer semantics, SURVEY.md App. A-3, as configured by
// custom/layers.py:10-25,55-80 and modelzoo/adenet_v2.py:45-63: mask holds (c,h) on padded steps,
// learn_init, optional peepholes, grad_clipping=5 on the gate pre-activations, backwards=True for b_*).
//
// Data layout (all time-major, row r = t*B + b):
//   xproj / gates / dG : [T*B][ldg], gate columns INTERLEAVED per hidden unit: column 4*u+g holds gate
//                        g in (i,f,g,o) of unit u, so one float4 per (row, unit) carries all four gates.
//                        W_in, W_hid, b are stored with the same column order, which makes every GEMM
//                        on the LSTM side (projection, dX, dW_in, dW_hid) layout-agnostic.
//   hbuf / cbuf        : [(T+1)*B][ldh] = T+1 blocks of B rows.  Forward LSTM: block 0 = initial state,
//                        block t+1 = state after frame t.  backwards=True: block T = initial state,
//                        block t = state after frame t.  Either way the T output blocks and the T
//                        "previous state" blocks are each one contiguous [T*B][ldh] matrix, so
//                        dW_hid = H_prev^T dG is ONE GEMM after the recurrence instead of T small ones.
//
// v1 structure: one launch per time step covering every LSTM that is independent at that depth
// (the S stream LSTMs, then the two aggregation LSTMs), blockIdx.z = LSTM.  Each workgroup owns a
// (rows x hidden-units) tile, splits the K dimension of  h_prev * W_hid  over its 4 waves (one per
// SIMD = one MFMA pipe each), reduces the four partial tiles through LDS and applies the gate math to
// its own (row, unit) elements -- the recurrent GEMM and the gate fusion never leave the CU.
// Operands go straight from L2 to VGPRs: with K split over the waves nothing is shared between them.
#include "adn_common.h"
#include <algorithm>
#include <cstdlib>

namespace adn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct LstmLaunch {
    LstmStep l[kMaxLstmPerLaunch];
};

// Gate nonlinearities on the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1 ulp each): the gate
// math of 3*B*H (row, unit) pairs per step is otherwise ~300 VALU instructions per pair with libm's expf /
// tanhf / IEEE division and becomes the longest part of a step launch.  Absolute error ~1e-7.
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.88539008177792681f * x));
}
__device__ __forceinline__ float clip5(float x) { return fminf(fmaxf(x, -5.f), 5.f); }

// -----------------------------------------------------------------------------------------
// forward step: tile = 32 batch rows x 8 hidden units (x 4 gates = 32 gate columns), MFMA 32x32x2 f32
// -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(const LstmLaunch L, const uint8_t* __restrict__ mask_tb,
                                                            int B, int T, int H, int ldh, int ldg, int step,
                                                            int kc) {
    __shared__ __attribute__((aligned(16))) float part[4][32][36];
    const LstmStep& P = L.l[blockIdx.z];
    const int t = P.backwards ? (T - 1 - step) : step;
    const int prev_blk = t + (P.backwards ? 1 : 0), out_blk = t + (P.backwards ? 0 : 1);
    const int r0 = blockIdx.x * 32, u0 = blockIdx.y * 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;

    const float* hprev = P.hbuf + (size_t)prev_blk * B * ldh;
    const int arow = min(r0 + i, B - 1);
    const int bcol = u0 * 4 + i;                       // gate column owned by this lane (B operand)
    const bool bcol_ok = (u0 + (i >> 2)) < H;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int kbeg = wave * kc;
    for (int kk = 0; kk < kc; kk += 8) {
        const int k = kbeg + kk + 4 * hh;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        if (k < H) {
            a = *reinterpret_cast<const float4*>(hprev + (size_t)arow * ldh + k);
            if (bcol_ok) {
                const float* w = P.W_hid + (size_t)k * ldg + bcol;
                b0 = w[0];
                if (k + 1 < H) b1 = w[ldg];
                if (k + 2 < H) b2 = w[2 * (size_t)ldg];
                if (k + 3 < H) b3 = w[3 * (size_t)ldg];
            }
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b3, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * hh][i] = acc[r];
    __syncthreads();

    const int row = tid >> 3, ul = tid & 7;
    const int r = r0 + row, u = u0 + ul;
    if (r >= B || u >= H) return;
    float4 g = *reinterpret_cast<const float4*>(&part[0][row][ul * 4]);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float4 x = *reinterpret_cast<const float4*>(&part[w][row][ul * 4]);
        g.x += x.x; g.y += x.y; g.z += x.z; g.w += x.w;
    }
    const float4 xp = *reinterpret_cast<const float4*>(P.xproj + ((size_t)t * B + r) * ldg + u * 4);
    float a_i = xp.x + g.x, a_f = xp.y + g.y, a_g = xp.z + g.z, a_o = xp.w + g.w;
    const size_t prev_idx = ((size_t)prev_blk * B + r) * ldh + u;
    const size_t out_idx = ((size_t)out_blk * B + r) * ldh + u;
    const float c_prev = P.cbuf[prev_idx], h_prev = P.hbuf[prev_idx];
    if (P.peep) { a_i += c_prev * P.peep[u]; a_f += c_prev * P.peep[ldh + u]; }
    const float gi = sigmoidf_(a_i), gf = sigmoidf_(a_f), gg = tanhf_(a_g);
    const float c_new = gf * c_prev + gi * gg;
    if (P.peep) a_o += c_new * P.peep[2 * ldh + u];
    const float go = sigmoidf_(a_o);
    const float h_new = go * tanhf_(c_new);
    const bool m = mask_tb[(size_t)t * B + r] != 0;
    P.cbuf[out_idx] = m ? c_new : c_prev;
    P.hbuf[out_idx] = m ? h_new : h_prev;
    if (P.gates) *reinterpret_cast<float4*>(P.gates + ((size_t)t * B + r) * ldg + u * 4) = make_float4(gi, gf, gg, go);
}

// =========================================================================================
// bf16 mode: the recurrent products run on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; c, the gate
// math, xproj, the saved gates and every gradient stay fp32.  MFMA fragments are loaded STRAIGHT from
// L2 (16 bytes = 8 consecutive k per lane): h_prev from the bf16 shadow of hbuf that the previous step's
// launch wrote, W_hid from a transposed bf16 copy refreshed once per optimiser step.  No LDS staging, no
// K-split: one wave owns 32 rows x 16 gate columns (4 hidden units) for the full K.
// =========================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 ld8(const __bf16* p, bool ok) {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
    return ok ? *reinterpret_cast<const bf16x8*>(p) : z;
}

__global__ __launch_bounds__(256) void lstm_fwd_step_bf16_kernel(const LstmLaunch L, const uint8_t* __restrict__ mask_tb,
                                                                 int B, int T, int H, int ldh, int ldg, int ldk,
                                                                 int step) {
    __shared__ __attribute__((aligned(16))) float part[4][32][20];
    const LstmStep& P = L.l[blockIdx.z];
    const int t = P.backwards ? (T - 1 - step) : step;
    const int prev_blk = t + (P.backwards ? 1 : 0), out_blk = t + (P.backwards ? 0 : 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int r0 = blockIdx.x * 32;
    const int c0 = blockIdx.y * 64 + wave * 16;          // first gate column of this wave (4 units x 4 gates)
    const __bf16* h16 = reinterpret_cast<const __bf16*>(P.h16) + (size_t)prev_blk * B * ldh;
    const __bf16* wt = reinterpret_cast<const __bf16*>(P.W_hid16T);
    const int rowA0 = min(r0 + i, B - 1), rowA1 = min(r0 + 16 + i, B - 1);
    const int colB = min(c0 + i, 4 * H - 1);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < ldk; k0 += 32) {
        const int k = k0 + kq * 8;
        const bool kin = k + 8 <= ldh;                   // beyond the padded row: W^T is zero there anyway
        const bf16x8 a0 = ld8(h16 + (size_t)rowA0 * ldh + k, kin);
        const bf16x8 a1 = ld8(h16 + (size_t)rowA1 * ldh + k, kin);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(wt + (size_t)colB * ldk + k);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, acc1, 0, 0, 0);
    }
    // 16x16 C/D map: col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        part[wave][4 * kq + r][i] = acc0[r];
        part[wave][16 + 4 * kq + r][i] = acc1[r];
    }
    __syncthreads();
    __bf16* h16out = reinterpret_cast<__bf16*>(P.h16);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pr = lane + 64 * j;                    // (row, unit) pair inside this wave's 32 x 4 tile
        const int row = pr >> 2, ul = pr & 3;
        const int r = r0 + row, u = (c0 >> 2) + ul;
        if (r >= B || u >= H) continue;
        const float4 g = *reinterpret_cast<const float4*>(&part[wave][row][ul * 4]);
        const float4 xp = *reinterpret_cast<const float4*>(P.xproj + ((size_t)t * B + r) * ldg + u * 4);
        float a_i = xp.x + g.x, a_f = xp.y + g.y, a_g = xp.z + g.z, a_o = xp.w + g.w;
        const size_t prev_idx = ((size_t)prev_blk * B + r) * ldh + u;
        const size_t out_idx = ((size_t)out_blk * B + r) * ldh + u;
        const float c_prev = P.cbuf[prev_idx], h_prev = P.hbuf[prev_idx];
        if (P.peep) { a_i += c_prev * P.peep[u]; a_f += c_prev * P.peep[ldh + u]; }
        const float gi = sigmoidf_(a_i), gf = sigmoidf_(a_f), gg = tanhf_(a_g);
        const float c_new = gf * c_prev + gi * gg;
        if (P.peep) a_o += c_new * P.peep[2 * ldh + u];
        const float go = sigmoidf_(a_o);
        const float h_new = go * tanhf_(c_new);
        const bool m = mask_tb[(size_t)t * B + r] != 0;
        const float h_out = m ? h_new : h_prev;
        P.cbuf[out_idx] = m ? c_new : c_prev;
        P.hbuf[out_idx] = h_out;
        h16out[out_idx] = (__bf16)h_out;
        if (P.gates)
            *reinterpret_cast<float4*>(P.gates + ((size_t)t * B + r) * ldg + u * 4) = make_float4(gi, gf, gg, go);
    }
}

// BPTT step, bf16 recurrent product: tile = 32 rows x 32 hidden units, K = 4H gate columns split over the
// 4 waves (fragments straight from L2: dG shadow rows, and W_hid rows, both k-contiguous as stored).
__global__ __launch_bounds__(256) void lstm_bwd_step_bf16_kernel(const LstmLaunch L, const uint8_t* __restrict__ mask_tb,
                                                                 int B, int T, int H, int ldh, int ldg, int step,
                                                                 int kc) {
    __shared__ __attribute__((aligned(16))) float part[4][32][33];
    __shared__ float pred[3][32][33];
    const LstmStep& P = L.l[blockIdx.z];
    const int t = P.backwards ? step : (T - 1 - step);
    const int t_done = P.backwards ? (t - 1) : (t + 1);
    const int r0 = blockIdx.x * 32, u0 = blockIdx.y * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;

    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (step > 0) {
        const __bf16* dg = reinterpret_cast<const __bf16*>(P.dG16) + (size_t)t_done * B * ldg;
        const __bf16* w = reinterpret_cast<const __bf16*>(P.W_hid16);
        const int ra0 = min(r0 + i, B - 1), ra1 = min(r0 + 16 + i, B - 1);
        const int ub0 = min(u0 + i, H - 1), ub1 = min(u0 + 16 + i, H - 1);
        const int kbeg = wave * kc;
        for (int kk = 0; kk < kc; kk += 32) {
            const int k = kbeg + kk + kq * 8;
            const bool kin = k + 8 <= ldg;
            const bf16x8 a0 = ld8(dg + (size_t)ra0 * ldg + k, kin), a1 = ld8(dg + (size_t)ra1 * ldg + k, kin);
            const bf16x8 b0 = ld8(w + (size_t)ub0 * ldg + k, kin), b1 = ld8(w + (size_t)ub1 * ldg + k, kin);
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[wave][a * 16 + 4 * kq + r][b * 16 + i] = acc[a][b][r];
    __syncthreads();

    __bf16* dg16 = reinterpret_cast<__bf16*>(P.dG16);
    float pw[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pr = tid + 256 * j;
        const int row = pr >> 5, ul = pr & 31;
        const int r = r0 + row, u = u0 + ul;
        pw[0][j] = pw[1][j] = pw[2][j] = 0.f;
        if (r >= B || u >= H) continue;
        const float rec = part[0][row][ul] + part[1][row][ul] + part[2][row][ul] + part[3][row][ul];
        const size_t sidx = (size_t)r * ldh + u;
        if (step == T) { P.dh_carry[sidx] += rec; continue; }
        const size_t ridx = (size_t)t * B + r;
        const float dh = P.dhs[ridx * (P.ld_dhs ? P.ld_dhs : ldh) + u] + P.dh_carry[sidx] + rec;
        const float dc = P.dc_state[sidx];
        float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mask_tb[ridx]) {
            const float4 gt = *reinterpret_cast<const float4*>(P.gates + ridx * ldg + u * 4);
            const int prev_blk = t + (P.backwards ? 1 : 0), out_blk = t + (P.backwards ? 0 : 1);
            const float c_t = P.cbuf[((size_t)out_blk * B + r) * ldh + u];
            const float c_prev = P.cbuf[((size_t)prev_blk * B + r) * ldh + u];
            const float tc = tanhf_(c_t);
            const float da_o = dh * tc * gt.w * (1.f - gt.w);
            float dcn = dc + dh * gt.w * (1.f - tc * tc);
            if (P.peep) { dcn += da_o * P.peep[2 * ldh + u]; pw[2][j] = da_o * c_t; }
            const float da_i = dcn * gt.z * gt.x * (1.f - gt.x);
            const float da_f = dcn * c_prev * gt.y * (1.f - gt.y);
            const float da_g = dcn * gt.x * (1.f - gt.z * gt.z);
            float dcp = dcn * gt.y;
            if (P.peep) {
                dcp += da_i * P.peep[u] + da_f * P.peep[ldh + u];
                pw[0][j] = da_i * c_prev; pw[1][j] = da_f * c_prev;
            }
            dg = make_float4(clip5(da_i), clip5(da_f), clip5(da_g), clip5(da_o));
            P.dh_carry[sidx] = 0.f;
            P.dc_state[sidx] = dcp;
        } else {
            P.dh_carry[sidx] = dh;
        }
        *reinterpret_cast<float4*>(P.dG + ridx * ldg + u * 4) = dg;
        __bf16* o16 = dg16 + ridx * ldg + u * 4;
        o16[0] = (__bf16)dg.x; o16[1] = (__bf16)dg.y; o16[2] = (__bf16)dg.z; o16[3] = (__bf16)dg.w;
    }
    if (P.dpeep_part && step < T) {                    // block-uniform branch
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pr = tid + 256 * j;
            pred[0][pr >> 5][pr & 31] = pw[0][j]; pred[1][pr >> 5][pr & 31] = pw[1][j]; pred[2][pr >> 5][pr & 31] = pw[2][j];
        }
        __syncthreads();
        if (tid < 96) {
            const int which = tid >> 5, c = tid & 31;
            float sum = 0.f;
#pragma unroll
            for (int rr = 0; rr < 32; ++rr) sum += pred[which][rr][c];
            if (u0 + c < H) {
                if (P.det_ws) P.det_ws[(size_t)blockIdx.x * P.det_stride + ldg + (2 + which) * (size_t)ldh + u0 + c] += sum;   // deterministic mode
                else atomicAdd(P.dpeep_part + (size_t)which * ldh + u0 + c, sum);
            }
        }
    }
}

// W [H][ldg] fp32 -> out [ldg][ldk] bf16 (transposed, zero padded in k)
__global__ __launch_bounds__(256) void pack_whid_t_kernel(const float* __restrict__ W, __bf16* __restrict__ out, int H,
                                                          int ldg, int ldk) {
    const int total = ldg * ldk;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int col = e / ldk, k = e % ldk;
        out[e] = (__bf16)(k < H ? W[(size_t)k * ldg + col] : 0.f);
    }
}

int lstm_pack_whid_t(const float* W, void* out, int H, hipStream_t s) {
    const int ldg = ld_of(4 * H), ldk = lstm_ldk(H);
    hipLaunchKernelGGL(pack_whid_t_kernel, dim3(std::min(1024, cdiv((int64_t)ldg * ldk, 256))), dim3(256), 0, s, W,
                       reinterpret_cast<__bf16*>(out), H, ldg, ldk);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

long long g_lstm_family_forwards[4] = {0, 0, 0, 0};
long long g_lstm_family_backwards[4] = {0, 0, 0, 0};

// Length buckets (LstmStep::T_own / mask_own): only the four weight-stationary kernels of H <= 256 read them; the dispatchers
// below refuse such entries on any other route rather than run them over the launch's T and mask.
static bool has_bucket_entries(const LstmStep* l, int n) {
    for (int k = 0; k < n; ++k) if (l[k].T_own || l[k].mask_own) return true;
    return false;
}
bool lstm_takes_length_buckets(const LstmStep* l, int n, int B, int T, int H, int precision, bool backward) {
    if (n < 1 || n > kMaxLstmPerLaunch || H > 256 || deterministic()) return false;
    if (precision == ADN_PRECISION_BF16X3) return backward ? lstm_cluster_x3_bwd_supported(l, n, B, T, H) : lstm_cluster_x3_supported(l, n, B, T, H);
    if (precision != ADN_PRECISION_BF16) return false;
    for (int k = 0; k < n; ++k)
        if (backward ? !(l[k].W_hid16 && l[k].dG16) : !(l[k].W_hid16T && l[k].h16)) return false;
    if (!lstm_persistent_supported(H) || !(backward ? l[0].W_frag_bwd : l[0].W_frag_fwd)) return false;
    if (backward && getenv("ADN_LSTM_NO_CLUSTER_BWD")) return false;
    return lstm_cluster_supported(l, n, B, T, H);
}

int lstm_forward(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, int precision, hipStream_t s) {
    ADN_CHECK(n >= 1 && n <= kMaxLstmPerLaunch, ADN_ERR_INVALID, "lstm_forward: bad LSTM count");
    ADN_CHECK(!has_bucket_entries(l, n) || lstm_takes_length_buckets(l, n, B, T, H, precision, false), ADN_ERR_STATE,
              "lstm_forward: length-bucket entries on a kernel family that does not take them");
    // bf16x3 mode: the weight-stationary forward kernel with fp32-grade products where it applies, the fp32 step kernels else
    if (precision == ADN_PRECISION_BF16X3) {
        if (lstm_cluster_x3_supported(l, n, B, T, H)) { g_lstm_family_forwards[3] += n; return lstm_forward_cluster_x3(l, n, mask_tb, B, T, H, s); }
        if (lstm_cluster_x3w_supported(l, n, B, T, H)) { g_lstm_family_forwards[3] += n; return lstm_forward_cluster_x3w(l, n, mask_tb, B, T, H, s); }
        precision = ADN_PRECISION_F32;
    }
    LstmLaunch L;
    bool have16 = precision == ADN_PRECISION_BF16;
    for (int k = 0; k < n; ++k) { L.l[k] = l[k]; have16 = have16 && l[k].W_hid16T && l[k].h16; }
    const int ldh = ld_of(H), ldg = ld_of(4 * H);
    // resident-weight kernels: the cluster kernels (H <= 512) or, where those cannot run, the one-workgroup persistent
    // kernel (H <= 256 only: at H > 256 streaming 2 MB of W_hid per step into one CU is slower than the per-step launches)
    if (have16 && lstm_persistent_supported(H) && l[0].W_frag_fwd && (H <= 256 || lstm_cluster_supported(l, n, B, T, H) || getenv("ADN_LSTM_WIDE_PERSISTENT")))
        return lstm_forward_persistent(l, n, mask_tb, B, T, H, s);
    g_lstm_family_forwards[0] += n;
    if (have16) {
        const double bytes = n * (4.0 * (12.0 * B * H + 4.0 * H * H) + B), flops = n * 8.0 * B * H * H;
        const dim3 grid16(cdiv(B, 32), cdiv(4 * H, 64), n);
        ProfScope prof(PROF_LSTM_FWD, flops * T, bytes * T, s, T);
        for (int step = 0; step < T; ++step)
            hipLaunchKernelGGL(lstm_fwd_step_bf16_kernel, grid16, dim3(256), 0, s, L, mask_tb, B, T, H, ldh, ldg,
                               lstm_ldk(H), step);
        ADN_HIP_CHECK(hipGetLastError());
        return ADN_OK;
    }
    const int kc = (int)round_up(cdiv(H, 4), 8);
    const dim3 grid(cdiv(B, 32), cdiv(H, 8), n);
    // algorithmic work per LSTM per step: bytes = 4*(12BH + 4H^2) + B, flops = 8BH^2
    const double bytes = n * (4.0 * (12.0 * B * H + 4.0 * H * H) + B), flops = n * 8.0 * B * H * H;
    ProfScope prof(PROF_LSTM_FWD, flops * T, bytes * T, s, T);
    for (int step = 0; step < T; ++step)
        hipLaunchKernelGGL(lstm_fwd_step_kernel, grid, dim3(256), 0, s, L, mask_tb, B, T, H, ldh, ldg, step, kc);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// -----------------------------------------------------------------------------------------
// BPTT step: tile = 16 batch rows x 16 hidden units, MFMA 16x16x4 f32, K = 4H gate columns.
//   rec   = dG[previous BPTT step] * W_hid^T                (recurrent gradient, skipped at step 0)
//   dh    = dhs[t] + dh_carry + rec ;  dc = dc_state
//   mask ? (gate gradients, clipped to +-5, -> dG[t];  dh_carry = 0;        dc_state = dcn*f [+peep])
//        : (dG[t] = 0;                                   dh_carry = dh;      dc_state = dc)
// step == T is the epilogue: dh_carry += rec only (gradient wrt the initial hidden state).
// -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(const LstmLaunch L, const uint8_t* __restrict__ mask_tb,
                                                            int B, int T, int H, int ldh, int ldg, int step,
                                                            int kc) {
    __shared__ __attribute__((aligned(16))) float part[4][16][17];
    __shared__ float pred[3][16][17];
    const LstmStep& P = L.l[blockIdx.z];
    // BPTT visits frames in the reverse of the forward order
    const int t = P.backwards ? step : (T - 1 - step);
    const int t_done = P.backwards ? (t - 1) : (t + 1);      // frame handled by the previous BPTT step
    const int r0 = blockIdx.x * 16, u0 = blockIdx.y * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int G4 = 4 * H;

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (step > 0) {
        const float* dgp = P.dG + (size_t)t_done * B * ldg;
        const int arow = min(r0 + i, B - 1);
        const int bunit = u0 + i;
        const bool bok = bunit < H;
        const int kbeg = wave * kc;
        for (int kk = 0; kk < kc; kk += 16) {
            const int k = kbeg + kk + 4 * kq;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < G4) {
                a = *reinterpret_cast<const float4*>(dgp + (size_t)arow * ldg + k);
                if (bok) b = *reinterpret_cast<const float4*>(P.W_hid + (size_t)bunit * ldg + k);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
        }
    }
    // 16x16 C/D map: col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][4 * kq + r][i] = acc[r];
    __syncthreads();

    const int row = tid >> 4, ul = tid & 15;
    const int r = r0 + row, u = u0 + ul;
    const bool valid = r < B && u < H;
    const float rec = part[0][row][ul] + part[1][row][ul] + part[2][row][ul] + part[3][row][ul];
    float pw_i = 0.f, pw_f = 0.f, pw_o = 0.f;           // peephole-weight gradient contributions
    if (valid) {
        const size_t sidx = (size_t)r * ldh + u;
        if (step == T) {
            P.dh_carry[sidx] += rec;
        } else {
            const size_t ridx = (size_t)t * B + r;
            const float dh = P.dhs[ridx * (P.ld_dhs ? P.ld_dhs : ldh) + u] + P.dh_carry[sidx] + rec;
            const float dc = P.dc_state[sidx];
            float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
            if (mask_tb[ridx]) {
                const float4 gt = *reinterpret_cast<const float4*>(P.gates + ridx * ldg + u * 4);
                const int prev_blk = t + (P.backwards ? 1 : 0), out_blk = t + (P.backwards ? 0 : 1);
                const float c_t = P.cbuf[((size_t)out_blk * B + r) * ldh + u];
                const float c_prev = P.cbuf[((size_t)prev_blk * B + r) * ldh + u];
                const float tc = tanhf_(c_t);
                const float da_o = dh * tc * gt.w * (1.f - gt.w);
                float dcn = dc + dh * gt.w * (1.f - tc * tc);
                if (P.peep) { dcn += da_o * P.peep[2 * ldh + u]; pw_o = da_o * c_t; }
                const float da_i = dcn * gt.z * gt.x * (1.f - gt.x);
                const float da_f = dcn * c_prev * gt.y * (1.f - gt.y);
                const float da_g = dcn * gt.x * (1.f - gt.z * gt.z);
                float dcp = dcn * gt.y;
                if (P.peep) {
                    dcp += da_i * P.peep[u] + da_f * P.peep[ldh + u];
                    pw_i = da_i * c_prev; pw_f = da_f * c_prev;
                }
                dg = make_float4(clip5(da_i), clip5(da_f), clip5(da_g), clip5(da_o));
                P.dh_carry[sidx] = 0.f;
                P.dc_state[sidx] = dcp;
            } else {
                P.dh_carry[sidx] = dh;                 // state was held: gradient passes straight through
            }
            *reinterpret_cast<float4*>(P.dG + ridx * ldg + u * 4) = dg;
        }
    }
    if (P.dpeep_part && step < T) {                    // block-uniform branch
        pred[0][row][ul] = pw_i; pred[1][row][ul] = pw_f; pred[2][row][ul] = pw_o;
        __syncthreads();
        if (tid < 48) {
            const int which = tid >> 4, c = tid & 15;
            float sum = 0.f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) sum += pred[which][rr][c];
            if (u0 + c < H) {
                if (P.det_ws) P.det_ws[(size_t)blockIdx.x * P.det_stride + ldg + (2 + which) * (size_t)ldh + u0 + c] += sum;   // deterministic mode
                else atomicAdd(P.dpeep_part + (size_t)which * ldh + u0 + c, sum);
            }
        }
    }
}

// ---- deterministic mode: the kernels' group sums go through slots (LstmStep::det_ws), then this fixed-order pass
struct DetReduceArgs { const float* ws[kMaxLstmPerLaunch]; float* dbias[kMaxLstmPerLaunch]; float* dhid[kMaxLstmPerLaunch];
                       float* dcell[kMaxLstmPerLaunch]; float* dpeep[kMaxLstmPerLaunch]; };
__global__ __launch_bounds__(256) void lstm_det_reduce_kernel(const DetReduceArgs a, int slots, int stride, int ldg, int ldh) {
    const float* __restrict__ ws = a.ws[blockIdx.y];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < stride; e += gridDim.x * 256) {
        float sum = 0.f;
        for (int sl = 0; sl < slots; ++sl) sum += ws[(size_t)sl * stride + e];          // slot order: the same every run
        float* dst = e < ldg ? (a.dbias[blockIdx.y] ? a.dbias[blockIdx.y] + e : nullptr)
                   : e < ldg + ldh ? (a.dhid[blockIdx.y] ? a.dhid[blockIdx.y] + (e - ldg) : nullptr)
                   : e < ldg + 2 * ldh ? (a.dcell[blockIdx.y] ? a.dcell[blockIdx.y] + (e - ldg - ldh) : nullptr)
                   : (a.dpeep[blockIdx.y] ? a.dpeep[blockIdx.y] + (e - ldg - 2 * ldh) : nullptr);
        if (dst) *dst += sum;
    }
}

static int lstm_backward_impl(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, int precision, hipStream_t s,
                              bool* sums_done);

int lstm_backward(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, int precision, hipStream_t s,
                  bool* sums_done) {
    ADN_CHECK(n >= 1 && n <= kMaxLstmPerLaunch, ADN_ERR_INVALID, "lstm_backward: bad LSTM count");
    if (!deterministic()) return lstm_backward_impl(l, n, mask_tb, B, T, H, precision, s, sums_done);
    // slots: >= 2 per 32-utterance group (weight-stationary kernels) and >= 1 per 16-row slice (the other families)
    const int ldh = ld_of(H), ldg = ld_of(4 * H), stride = ldg + 5 * ldh, slots = cdiv(B, 16) + 2;
    // (per device: one process may drive several; within a device one stream drives the models in this mode)
    static float* ws_dev[64] = {}; static size_t ws_floats_dev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    float*& ws = ws_dev[dev]; size_t& ws_floats = ws_floats_dev[dev];
    const size_t need = (size_t)n * slots * stride;
    if (need > ws_floats) {
        if (ws) { ADN_HIP_CHECK(hipDeviceSynchronize()); (void)hipFree(ws); ws = nullptr; ws_floats = 0; }
        ADN_HIP_CHECK(hipMalloc((void**)&ws, need * sizeof(float)));
        ws_floats = need;
    }
    ADN_HIP_CHECK(hipMemsetAsync(ws, 0, need * sizeof(float), s));
    LstmStep q[kMaxLstmPerLaunch];
    DetReduceArgs a{};
    for (int k = 0; k < n; ++k) {
        q[k] = l[k];
        q[k].det_ws = ws + (size_t)k * slots * stride; q[k].det_stride = stride;
        a.ws[k] = q[k].det_ws; a.dbias[k] = l[k].dbias; a.dhid[k] = l[k].dhid_init; a.dcell[k] = l[k].dcell_init; a.dpeep[k] = l[k].dpeep_part;
    }
    bool done = false;
    ADN_TRY(lstm_backward_impl(q, n, mask_tb, B, T, H, precision, s, &done));
    if (sums_done) *sums_done = done;
    if (!done)                       // (the kernel left bias / initial state to the caller's column sums: only the peephole slots hold anything)
        for (int k = 0; k < n; ++k) a.dbias[k] = a.dhid[k] = a.dcell[k] = nullptr;
    hipLaunchKernelGGL(lstm_det_reduce_kernel, dim3(cdiv(stride, 256), n), dim3(256), 0, s, a, slots, stride, ldg, ldh);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

static int lstm_backward_impl(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, int precision, hipStream_t s,
                              bool* sums_done) {
    ADN_CHECK(n >= 1 && n <= kMaxLstmPerLaunch, ADN_ERR_INVALID, "lstm_backward: bad LSTM count");
    ADN_CHECK(!has_bucket_entries(l, n) || lstm_takes_length_buckets(l, n, B, T, H, precision, true), ADN_ERR_STATE,
              "lstm_backward: length-bucket entries on a kernel family that does not take them");
    if (sums_done) *sums_done = false;
    if (precision == ADN_PRECISION_BF16X3) {          // fp32-grade products on the bf16 matrix pipe, or the fp32 step kernels
        if (lstm_cluster_x3_bwd_supported(l, n, B, T, H)) {
            if (sums_done) *sums_done = true;         // bias / initial-state gradients are added inside the kernel
            g_lstm_family_backwards[3] += n;
            return lstm_backward_cluster_x3(l, n, mask_tb, B, T, H, s);
        }
        if (lstm_cluster_x3w_bwd_supported(l, n, B, T, H)) {
            if (sums_done) *sums_done = true;
            g_lstm_family_backwards[3] += n;
            return lstm_backward_cluster_x3w(l, n, mask_tb, B, T, H, s);
        }
        precision = ADN_PRECISION_F32;
    }
    LstmLaunch L;
    bool have16 = precision == ADN_PRECISION_BF16;
    for (int k = 0; k < n; ++k) { L.l[k] = l[k]; have16 = have16 && l[k].W_hid16 && l[k].dG16; }
    const int ldh = ld_of(H), ldg = ld_of(4 * H);
    if (have16 && lstm_persistent_supported(H) && l[0].W_frag_bwd &&
        (H <= 256 || (lstm_cluster_supported(l, n, B, T, H) && !getenv("ADN_LSTM_NO_CLUSTER_BWD")) ||
         getenv("ADN_LSTM_WIDE_PERSISTENT")))
        return lstm_backward_persistent(l, n, mask_tb, B, T, H, s, sums_done);
    g_lstm_family_backwards[0] += n;
    for (int k = 0; k < n; ++k) {
        ADN_HIP_CHECK(hipMemsetAsync(l[k].dh_carry, 0, (size_t)B * ldh * sizeof(float), s));
        ADN_HIP_CHECK(hipMemsetAsync(l[k].dc_state, 0, (size_t)B * ldh * sizeof(float), s));
    }
    if (have16) {
        const double bytes = n * 4.0 * (15.0 * B * H + 4.0 * H * H), flops = n * 8.0 * B * H * H;
        const int kc16 = (int)round_up(cdiv(4 * H, 4), 32);
        const dim3 grid16(cdiv(B, 32), cdiv(H, 32), n);
        ProfScope prof(PROF_LSTM_BWD, flops * T, bytes * T, s, T + 1);
        for (int step = 0; step <= T; ++step)
            hipLaunchKernelGGL(lstm_bwd_step_bf16_kernel, grid16, dim3(256), 0, s, L, mask_tb, B, T, H, ldh, ldg, step,
                               kc16);
        ADN_HIP_CHECK(hipGetLastError());
        return ADN_OK;
    }
    const int kc = (int)round_up(cdiv(4 * H, 4), 16);
    const dim3 grid(cdiv(B, 16), cdiv(H, 16), n);
    // algorithmic work per LSTM per step: bytes = 4*(15BH + 4H^2), flops = 8BH^2
    const double bytes = n * 4.0 * (15.0 * B * H + 4.0 * H * H), flops = n * 8.0 * B * H * H;
    ProfScope prof(PROF_LSTM_BWD, flops * T, bytes * T, s, T + 1);
    for (int step = 0; step <= T; ++step)
        hipLaunchKernelGGL(lstm_bwd_step_kernel, grid, dim3(256), 0, s, L, mask_tb, B, T, H, ldh, ldg, step, kc);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

}  // namespace adn
