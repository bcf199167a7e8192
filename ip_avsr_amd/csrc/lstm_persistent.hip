// Persistent LSTM kernels (bf16 mode): ONE launch runs all T time steps of every LSTM that is independent
// at that depth.  The recurrence couples nothing across utterances, so a workgroup owns a 16-utterance slice
// of one LSTM for the whole sequence and never talks to another workgroup -- no grid barrier, no flags, no
// kernel boundary per step (a step launch costs ~16 us on MI355X, dominated by the launch boundary and the
// write-back of the step's dirty lines; T = 40 steps x 4 phases made that 2.7 ms of a 9.7 ms train step).
//
// Per workgroup (512 threads = 8 waves; wave w owns hidden units [UW*w, UW*w + UW)):
//   * h_{t-1} of its 16 rows lives in LDS as bf16 (double buffered), c and the fp32 h in registers;
//   * W_hid is NOT resident (H x 4H bf16 = 512 KB > LDS): every wave streams its own 1/8 of the transposed
//     bf16 copy from L2 each step, 16 bytes per lane straight into MFMA B fragments;
//   * one v_mfma_f32_16x16x32_bf16 tile = 16 rows x 4 units x 4 gates (gate-interleaved columns), so a
//     4x4 register transpose inside each lane quad (two DPP rounds) hands every lane the four gates of ONE
//     (row, unit) pair: the gate math needs no LDS round trip.
// The backward kernel keeps dG_{t+1} of its rows in LDS (bf16), streams W_hid rows (k-contiguous as stored),
// and its accumulator layout (unit on the lane, 4 rows per lane) already matches the per-(row, unit) BPTT math.
#include "adn_common.h"
#include <algorithm>

namespace adn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct LstmLaunchP {
    LstmStep l[kMaxLstmPerLaunch];
};

__device__ __forceinline__ float p_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float p_tanh(float x) {
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.88539008177792681f * x));
}
__device__ __forceinline__ float p_clip5(float x) { return fminf(fmaxf(x, -5.f), 5.f); }

template <int CTRL>
__device__ __forceinline__ float quad_perm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// 4x4 transpose across (register index, lane-in-quad): out[lane g].reg[j] = in[lane j].reg[g]
__device__ __forceinline__ void quad_transpose(f32x4& a, int lane) {
    const bool odd = lane & 1, hi = lane & 2;
    // round 1: exchange with lane^1 (quad_perm [1,0,3,2])
    const float p0 = quad_perm<0xB1>(a[0]), p1 = quad_perm<0xB1>(a[1]), p2 = quad_perm<0xB1>(a[2]), p3 = quad_perm<0xB1>(a[3]);
    const float b0 = odd ? p1 : a[0], b1 = odd ? a[1] : p0, b2 = odd ? p3 : a[2], b3 = odd ? a[3] : p2;
    // round 2: exchange with lane^2 (quad_perm [2,3,0,1])
    const float q0 = quad_perm<0x4E>(b0), q1 = quad_perm<0x4E>(b1), q2 = quad_perm<0x4E>(b2), q3 = quad_perm<0x4E>(b3);
    a[0] = hi ? q2 : b0; a[1] = hi ? q3 : b1; a[2] = hi ? b2 : q0; a[3] = hi ? b3 : q1;
}

// workgroup barrier that orders LDS traffic only (__syncthreads() would also wait for every outstanding global store
// and load of the step: see lstm_cluster.hip)
__device__ __forceinline__ void p_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

constexpr int kPRows = 16;      // utterances per workgroup
constexpr int kPWaves = 8;

// =========================================================================================
// forward
// =========================================================================================
template <int UW>                // hidden units per wave: 32 (H <= 256) or 64 (H <= 512)
__global__ __launch_bounds__(512) void lstm_fwd_persistent_kernel(const LstmLaunchP L, const uint8_t* __restrict__ mask_tb,
                                                                  int B, int T, int H, int ldh, int ldg) {
    constexpr int UT = UW / 16;                      // 16-unit tiles per wave; each has 4 MFMA tiles (one per gate)
    constexpr int KS = UW * kPWaves / 32;            // k-steps of 32 over the padded hidden size
    constexpr int HS = UW * kPWaves + 8;             // LDS row stride (bf16)
    __shared__ __attribute__((aligned(16))) __bf16 hs[2][kPRows][HS];
    const LstmStep& P = L.l[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int r0 = blockIdx.x * kPRows;
    const int ubase = wave * UW;
    // W_hid in MFMA-fragment order: [unit tile][gate][k-step][lane][8 bf16] -- one wave load = 1 KiB contiguous.
    // A tile holds ONE gate of 16 consecutive units, so after the four MFMA chains of a unit tile every lane
    // owns all four gates of its 4 (row, unit) pairs (rows 4*kq..+3, unit = lane&15): no cross-lane exchange,
    // and all global traffic of the gate math is row-contiguous across the 16 lanes of a group.
    const __bf16* wt = reinterpret_cast<const __bf16*>(P.W_frag_fwd) + ((size_t)wave * UT * 4 * KS * 64 + lane) * 8;
    __bf16* h16g = reinterpret_cast<__bf16*>(P.h16);

    const int blk0 = P.backwards ? T : 0;
    for (int e = tid; e < 2 * kPRows * HS / 8; e += 512) reinterpret_cast<bf16x8*>(&hs[0][0][0])[e] = bf16x8{};
    __syncthreads();
    for (int e = tid; e < kPRows * (ldh / 8); e += 512) {
        const int rr = e / (ldh / 8), cc = (e % (ldh / 8)) * 8;
        const int gr = min(r0 + rr, B - 1);
        *reinterpret_cast<bf16x8*>(&hs[0][rr][cc]) =
            *reinterpret_cast<const bf16x8*>(h16g + ((size_t)blk0 * B + gr) * ldh + cc);
    }
    float c_st[UT][4], h_st[UT][4];
#pragma unroll
    for (int ut = 0; ut < UT; ++ut) {
        const int uc = min(ubase + 16 * ut + i, H - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t idx = ((size_t)blk0 * B + min(r0 + 4 * kq + r, B - 1)) * ldh + uc;
            c_st[ut][r] = P.cbuf[idx];
            h_st[ut][r] = P.hbuf[idx];
        }
    }
    __syncthreads();

    int cur = 0;
    for (int step = 0; step < T; ++step) {
        const int t = P.backwards ? (T - 1 - step) : step;
        const int out_blk = t + (P.backwards ? 0 : 1);
        // ---- phase 1: masks and input projections of the step
        uint8_t m[4];             // raw bytes, compared where used: the request must not wait for its own data
        float4 xp[UT][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t ridx = (size_t)t * B + min(r0 + 4 * kq + r, B - 1);
            m[r] = mask_tb[ridx];
#pragma unroll
            for (int ut = 0; ut < UT; ++ut)
                xp[ut][r] = *reinterpret_cast<const float4*>(P.xproj + ridx * ldg + min(ubase + 16 * ut + i, H - 1) * 4);
        }
        // ---- phase 2: recurrent product; k outermost so that the 4*UT accumulator chains are independent
        f32x4 acc[UT][4];
#pragma unroll
        for (int ut = 0; ut < UT; ++ut)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[ut][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int s = 0; s < KS; ++s) {                // 2 k-steps = 16 fragment loads in flight per wave
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(&hs[cur][i][s * 32 + kq * 8]);
#pragma unroll
            for (int ut = 0; ut < UT; ++ut)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[ut][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        a, *reinterpret_cast<const bf16x8*>(wt + ((size_t)(ut * 4 + g) * KS + s) * 512), acc[ut][g], 0, 0, 0);
        }
        // ---- phase 3: gate math and stores; lane = (unit, 4 rows)
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
            const int u = ubase + 16 * ut + i;
            const int uc = min(u, H - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * kq + r, grow = r0 + row;
                float a_i = xp[ut][r].x + acc[ut][0][r], a_f = xp[ut][r].y + acc[ut][1][r];
                float a_g = xp[ut][r].z + acc[ut][2][r], a_o = xp[ut][r].w + acc[ut][3][r];
                const float c_prev = c_st[ut][r], h_prev = h_st[ut][r];
                if (P.peep) { a_i += c_prev * P.peep[uc]; a_f += c_prev * P.peep[ldh + uc]; }
                const float gi = p_sigmoid(a_i), gf = p_sigmoid(a_f), gg = p_tanh(a_g);
                const float c_new = gf * c_prev + gi * gg;
                if (P.peep) a_o += c_new * P.peep[2 * ldh + uc];
                const float go = p_sigmoid(a_o);
                const float h_new = go * p_tanh(c_new);
                const float c_out = m[r] ? c_new : c_prev, h_out = m[r] ? h_new : h_prev;
                c_st[ut][r] = c_out; h_st[ut][r] = h_out;
                if (u < H) {
                    hs[cur ^ 1][row][u] = (__bf16)h_out;
                    if (grow < B) {
                        const size_t ridx = (size_t)t * B + grow;
                        const size_t oidx = ((size_t)out_blk * B + grow) * ldh + u;
                        P.cbuf[oidx] = c_out;
                        P.hbuf[oidx] = h_out;
                        h16g[oidx] = (__bf16)h_out;
                        if (P.gates) *reinterpret_cast<float4*>(P.gates + ridx * ldg + u * 4) = make_float4(gi, gf, gg, go);
                    }
                }
            }
        }
        p_lds_barrier();
        cur ^= 1;
    }
}

// =========================================================================================
// backward (BPTT): see lstm.hip for the per-step math; dh_carry / dc_state live in registers here
// =========================================================================================
template <int UW>
__global__ __launch_bounds__(512) void lstm_bwd_persistent_kernel(const LstmLaunchP L, const uint8_t* __restrict__ mask_tb,
                                                                  int B, int T, int H, int ldh, int ldg) {
    constexpr int CT = UW / 16;                      // 16-unit MFMA column tiles per wave
    constexpr int GK = UW * kPWaves * 4;             // padded number of gate columns (K of the recurrent product)
    constexpr int KS = GK / 32;
    constexpr int GS = GK + 8;                       // LDS row stride (bf16)
    extern __shared__ __attribute__((aligned(16))) __bf16 dgs_raw[];
    __bf16 (*dgs)[kPRows][GS] = reinterpret_cast<__bf16 (*)[kPRows][GS]>(dgs_raw);   // [2][16][GS]
    const LstmStep& P = L.l[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int r0 = blockIdx.x * kPRows;
    const int ubase = wave * UW;
    const __bf16* w16 = reinterpret_cast<const __bf16*>(P.W_frag_bwd) + ((size_t)wave * CT * KS * 64 + lane) * 8;
    __bf16* dg16g = reinterpret_cast<__bf16*>(P.dG16);

    for (int e = tid; e < 2 * kPRows * GS / 8; e += 512) reinterpret_cast<bf16x8*>(dgs_raw)[e] = bf16x8{};
    float dh_c[CT][4], dc_s[CT][4];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) { dh_c[ct][r] = 0.f; dc_s[ct][r] = 0.f; }
    __syncthreads();

    int cur = 0;
    for (int step = 0; step <= T; ++step) {
        const int t = P.backwards ? step : (T - 1 - step);
        f32x4 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (step > 0) {
#pragma unroll 8
            for (int s = 0; s < KS; ++s) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&dgs[cur][i][s * 32 + kq * 8]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        a, *reinterpret_cast<const bf16x8*>(w16 + ((size_t)ct * KS + s) * 512), acc[ct], 0, 0, 0);
            }
        }
        // accumulator map: unit = ubase + 16*ct + (lane&15), row = 4*kq + r
        if (step == T) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) dh_c[ct][r] += acc[ct][r];
            break;
        }
        // every load of this step's gate math is issued before its first store (one exposed round trip)
        float l_dhs[CT][4], l_ct[CT][4], l_cp[CT][4];
        float4 l_gt[CT][4];
        uint8_t l_m[4];
        const int prev_blk = t + (P.backwards ? 1 : 0), out_blk = t + (P.backwards ? 0 : 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) l_m[r] = mask_tb[(size_t)t * B + min(r0 + 4 * kq + r, B - 1)];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int uc = min(ubase + 16 * ct + i, H - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int growc = min(r0 + 4 * kq + r, B - 1);
                const size_t ridx = (size_t)t * B + growc;
                l_dhs[ct][r] = P.dhs[ridx * (P.ld_dhs ? P.ld_dhs : ldh) + uc];
                l_gt[ct][r] = *reinterpret_cast<const float4*>(P.gates + ridx * ldg + uc * 4);
                l_ct[ct][r] = P.cbuf[((size_t)out_blk * B + growc) * ldh + uc];
                l_cp[ct][r] = P.cbuf[((size_t)prev_blk * B + growc) * ldh + uc];
            }
        }
        float pw_i = 0.f, pw_f = 0.f, pw_o = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int u = ubase + 16 * ct + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * kq + r, grow = r0 + row;
                const bool ok = grow < B && u < H;
                float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) {
                    const size_t ridx = (size_t)t * B + grow;
                    const float dh = l_dhs[ct][r] + dh_c[ct][r] + acc[ct][r];
                    const float dc = dc_s[ct][r];
                    if (l_m[r]) {
                        const float4 gt = l_gt[ct][r];
                        const float c_t = l_ct[ct][r], c_prev = l_cp[ct][r];
                        const float tc = p_tanh(c_t);
                        const float da_o = dh * tc * gt.w * (1.f - gt.w);
                        float dcn = dc + dh * gt.w * (1.f - tc * tc);
                        if (P.peep) { dcn += da_o * P.peep[2 * ldh + u]; pw_o += da_o * c_t; }
                        const float da_i = dcn * gt.z * gt.x * (1.f - gt.x);
                        const float da_f = dcn * c_prev * gt.y * (1.f - gt.y);
                        const float da_g = dcn * gt.x * (1.f - gt.z * gt.z);
                        float dcp = dcn * gt.y;
                        if (P.peep) {
                            dcp += da_i * P.peep[u] + da_f * P.peep[ldh + u];
                            pw_i += da_i * c_prev; pw_f += da_f * c_prev;
                        }
                        dg = make_float4(p_clip5(da_i), p_clip5(da_f), p_clip5(da_g), p_clip5(da_o));
                        dh_c[ct][r] = 0.f;
                        dc_s[ct][r] = dcp;
                    } else {
                        dh_c[ct][r] = dh;
                    }
                    *reinterpret_cast<float4*>(P.dG + ridx * ldg + u * 4) = dg;
                    bf16x4 d16;
                    d16[0] = (__bf16)dg.x; d16[1] = (__bf16)dg.y; d16[2] = (__bf16)dg.z; d16[3] = (__bf16)dg.w;
                    *reinterpret_cast<bf16x4*>(dg16g + ridx * ldg + u * 4) = d16;
                    *reinterpret_cast<bf16x4*>(&dgs[cur ^ 1][row][u * 4]) = d16;
                } else if (u < H) {
                    *reinterpret_cast<bf16x4*>(&dgs[cur ^ 1][row][u * 4]) = bf16x4{};
                }
            }
            if (P.dpeep_part) {                      // sum this lane's 4 rows, then the 4 row-groups (kq) of the wave
                float si = pw_i, sf = pw_f, so = pw_o;
                si += __shfl_xor(si, 16, 64); si += __shfl_xor(si, 32, 64);
                sf += __shfl_xor(sf, 16, 64); sf += __shfl_xor(sf, 32, 64);
                so += __shfl_xor(so, 16, 64); so += __shfl_xor(so, 32, 64);
                if (kq == 0 && u < H) {
                    if (P.det_ws) {                  // deterministic mode: this row slice's own slot, plain adds (one writer per address)
                        float* ds = P.det_ws + (size_t)blockIdx.x * P.det_stride + ldg + 2 * ldh;
                        ds[u] += si; ds[ldh + u] += sf; ds[2 * ldh + u] += so;
                    } else {
                        atomicAdd(P.dpeep_part + u, si);
                        atomicAdd(P.dpeep_part + ldh + u, sf);
                        atomicAdd(P.dpeep_part + 2 * (size_t)ldh + u, so);
                    }
                }
                pw_i = pw_f = pw_o = 0.f;
            }
        }
        p_lds_barrier();
        cur ^= 1;
    }
    // gradient wrt the initial state of every row of this slice (summed over rows by the caller)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int u = ubase + 16 * ct + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int grow = r0 + 4 * kq + r;
            if (grow < B && u < H) {
                P.dh_carry[(size_t)grow * ldh + u] = dh_c[ct][r];
                P.dc_state[(size_t)grow * ldh + u] = dc_s[ct][r];
            }
        }
    }
}

// W [H][ldg] fp32 (gate-interleaved columns) -> the two fragment-ordered bf16 copies the persistent kernels stream
//   fwd: [UW*8/16 unit tiles][4 gates][UW*8/32 k-steps][64 lanes][8]   value = W[k][4*unit + gate], unit = 16*tile + (lane&15)
//   bwd: [UW*8/16 unit tiles][4*UW*8/32 k-steps][64 lanes][8]     value = W[unit][k], unit = 16*tile + (lane&15)
// with k = 32*step + 8*(lane>>4) + j; zero outside the matrix.
struct PackFragArgs { const float* W[8]; void* fwd[8]; void* bwd[8]; };
// lo_part: the images of W - bf16(W) instead of W (bf16x3 mode: hi and lo fragment images); null targets are skipped
__global__ __launch_bounds__(256) void pack_frags_kernel(const PackFragArgs a, int H, int ldg, int UW, int lo_part) {
    const float* __restrict__ W = a.W[blockIdx.y];
    __bf16* __restrict__ fwd = reinterpret_cast<__bf16*>(a.fwd[blockIdx.y]);
    __bf16* __restrict__ bwd = reinterpret_cast<__bf16*>(a.bwd[blockIdx.y]);
    const int HP = UW * kPWaves, GP = 4 * HP;        // padded H and 4H
    const int total = GP * HP;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        {   // forward image
            const int j = e & 7, lane = (e >> 3) & 63, rest = e >> 9;
            const int ks = HP / 32, s = rest % ks, tile = rest / ks;      // tile = 4 * (16-unit tile) + gate
            const int unit = 16 * (tile >> 2) + (lane & 15), col = 4 * unit + (tile & 3);
            const int k = 32 * s + 8 * (lane >> 4) + j;
            float v = (k < H && unit < H) ? W[(size_t)k * ldg + col] : 0.f;
            if (lo_part) v -= (float)(__bf16)v;
            if (fwd) fwd[e] = (__bf16)v;
        }
        {   // backward image
            const int j = e & 7, lane = (e >> 3) & 63, rest = e >> 9;
            const int ks = GP / 32, s = rest % ks, tile = rest / ks;
            const int unit = 16 * tile + (lane & 15), k = 32 * s + 8 * (lane >> 4) + j;
            float v = (unit < H && k < 4 * H) ? W[(size_t)unit * ldg + k] : 0.f;
            if (lo_part) v -= (float)(__bf16)v;
            if (bwd) bwd[e] = (__bf16)v;
        }
    }
}

size_t lstm_frag_elems(int H) { const int UW = H <= 256 ? 32 : 64; return (size_t)4 * UW * kPWaves * UW * kPWaves; }

int lstm_pack_frags_batch(int n, const float* const* W, void* const* fwd, void* const* bwd, int H, hipStream_t s, int lo_part) {
    ADN_CHECK(n >= 1 && n <= 8, ADN_ERR_INVALID, "lstm_pack_frags_batch: 1..8 matrices per launch");
    const int UW = H <= 256 ? 32 : 64;
    PackFragArgs a{};
    for (int k = 0; k < n; ++k) { a.W[k] = W[k]; a.fwd[k] = fwd ? fwd[k] : nullptr; a.bwd[k] = bwd ? bwd[k] : nullptr; }
    hipLaunchKernelGGL(pack_frags_kernel, dim3(1024, n), dim3(256), 0, s, a, H, ld_of(4 * H), UW, lo_part);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int lstm_pack_frags(const float* W, void* fwd, void* bwd, int H, hipStream_t s) {
    return lstm_pack_frags_batch(1, &W, &fwd, &bwd, H, s);
}

bool lstm_persistent_supported(int H) { return H <= 512 && !getenv("ADN_LSTM_STEPWISE"); }

int lstm_forward_persistent(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, hipStream_t s) {
    if (lstm_cluster_supported(l, n, B, T, H)) { g_lstm_family_forwards[2] += n; return lstm_forward_cluster(l, n, mask_tb, B, T, H, s); }
    g_lstm_family_forwards[1] += n;
    LstmLaunchP L;
    for (int k = 0; k < n; ++k) L.l[k] = l[k];
    const int ldh = ld_of(H), ldg = ld_of(4 * H);
    const dim3 grid(cdiv(B, kPRows), n);
    const double bytes = (double)n * T * (4.0 * (12.0 * B * H + 4.0 * H * H) + B), flops = (double)n * T * 8.0 * B * H * H;
    ProfScope prof(PROF_LSTM_FWD, flops, bytes, s, T);
    if (H <= 256) {
        hipLaunchKernelGGL(lstm_fwd_persistent_kernel<32>, grid, dim3(512), 0, s, L, mask_tb, B, T, H, ldh, ldg);
    } else {
        hipLaunchKernelGGL(lstm_fwd_persistent_kernel<64>, grid, dim3(512), 0, s, L, mask_tb, B, T, H, ldh, ldg);
    }
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int lstm_backward_persistent(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, hipStream_t s,
                             bool* sums_done) {
    if (lstm_cluster_supported(l, n, B, T, H) && !getenv("ADN_LSTM_NO_CLUSTER_BWD")) {
        if (sums_done) *sums_done = true;             // bias / initial-state gradients are added inside the kernel
        g_lstm_family_backwards[2] += n;
        return lstm_backward_cluster(l, n, mask_tb, B, T, H, s);
    }
    g_lstm_family_backwards[1] += n;
    LstmLaunchP L;
    for (int k = 0; k < n; ++k) L.l[k] = l[k];
    const int ldh = ld_of(H), ldg = ld_of(4 * H);
    const dim3 grid(cdiv(B, kPRows), n);
    const double bytes = (double)n * T * 4.0 * (15.0 * B * H + 4.0 * H * H), flops = (double)n * T * 8.0 * B * H * H;
    ProfScope prof(PROF_LSTM_BWD, flops, bytes, s, T + 1);
    if (H <= 256) {
        const size_t lds = (size_t)2 * kPRows * (32 * kPWaves * 4 + 8) * 2;
        ADN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_persistent_kernel<32>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(lstm_bwd_persistent_kernel<32>, grid, dim3(512), lds, s, L, mask_tb, B, T, H, ldh, ldg);
    } else {
        const size_t lds = (size_t)2 * kPRows * (64 * kPWaves * 4 + 8) * 2;
        ADN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_persistent_kernel<64>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(lstm_bwd_persistent_kernel<64>, grid, dim3(512), lds, s, L, mask_tb, B, T, H, ldh, ldg);
    }
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

}  // namespace adn
