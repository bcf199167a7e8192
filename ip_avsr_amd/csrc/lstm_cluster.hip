// Weight-stationary LSTM kernels (bf16 mode, H <= 512; bf16x3 mode: the *_x3 kernels (H <= 256) and *_x3w kernels (H <= 512)
// further down): W_hid never
// leaves the CU.
//
// The persistent kernels of lstm_persistent.hip give a 16-utterance slice to ONE workgroup, which must then stream
// the whole W_hid (512 KB as bf16 at H = 250, 2 MB at H = 500) from L2 every time step: 12-14 us per step (34-70 us at
// H = 500), bound by what one CU can take in (~40 GB/s), with only 33 workgroups per LSTM busy.  Here a group of CWG
// workgroups shares a 32-utterance slice (CWG = 4 for H <= 256, 8 for H <= 512); workgroup j owns hidden units
// [64 j, 64 j + 64): its 256 gate columns of W_hid (forward) or its 256 rows of W_hid^T (backward), in MFMA-fragment
// order, stay on the CU for all T steps, split between LDS and the registers of the waves that multiply them (see
// ClusterGeom: at CWG = 4 the 128 KB slice sits in registers in the forward kernel and half / half in the backward one; at
// CWG = 8 the slice is 256 KB).  What the workgroups
// must exchange each step is tiny and goes through L2 / the memory side as 8-byte tagged granules
// {payload32, tag32}: the consumer polls the payload's own address until the tag of the step shows up (one hop,
// no separate flag; relaxed agent-scope 64-bit atomics = sc1 stores / loads, single-copy atomic):
//   forward   all-gather of h_t  (bf16):   granule = h of 2 rows of one unit       32 x 64 (CWG - 1) values in per workgroup
//   backward  reduce-scatter of the partial dh_t (fp32, tag in the low mantissa bits): 2 rows of one unit per
//             granule                                                                 (CWG - 1) x 32 x 64 values in
// Two parities of the exchange buffer suffice: a workgroup can only be one step ahead of its partners.
// Every workgroup of a launch must be resident at once (they wait for each other): the host launches at most one
// workgroup per CU (registers and LDS allow no second one) and splits larger sets of LSTMs over several launches; an LSTM whose groups
// do not fit the device at all (B > 32 * CUs / CWG) runs on the other kernels.  Polls are bounded (10 s of wall clock); a
// poll that gives up raises the launch's error word and the host reports ADN_ERR_STATE -- never a hang.
#include "adn_common.h"
#include <algorithm>
#include <vector>
#include <cstdlib>

namespace adn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// A launch covers a range of (LSTM, utterance group) PAIRS of one call -- pair = LSTM index * groups + group, CWG consecutive
// workgroups each.  Pairs are independent of each other (the exchange stays inside a group), so a call's pairs are dealt over
// as few launches as the device holds resident at once, in equal shares: four 512-unit LSTMs of 17 groups x 8 workgroups are
// three launches of 23 / 23 / 22 pairs, not four of 17 (+ 120 idle CUs each).
struct LstmClusterP {
    LstmStep l[kMaxLstmPerLaunch];
    int pair0, groups;
};
static_assert(sizeof(LstmClusterP) + 64 <= 4096, "kernel arguments: the launch descriptor + the scalars must fit 4 KB");

__device__ __forceinline__ float c_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float c_tanh(float x) {
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.88539008177792681f * x));
}
__device__ __forceinline__ float c_clip5(float x) { return fminf(fmaxf(x, -5.f), 5.f); }

constexpr int kCRows = 32;       // utterances per group
constexpr int kCUnits = 64;      // hidden units per workgroup
// Geometry of a group of CWG workgroups (4: H <= 256, everything LDS-resident;  8: H <= 512, where a workgroup's W slice
// is 256 KB and its k-steps are split between LDS and the waves' registers).
template <int CWG, int KXS = 0> struct ClusterGeom {     // (KXS > 0: the forward kernel that also multiplies x_t W_in)
    static constexpr int HP = CWG * kCUnits;                // padded hidden size (256 | 512)
    static constexpr int KS = HP / 32;                      // k-steps of the forward product (8 | 16)
    static constexpr int KSL = CWG == 4 ? (KXS > 0 ? 1 : 0) : 7;   // ... of which LDS-resident; the rest sit in registers
                                                            // (CWG = 4: all 8 in registers, 128 VGPRs -- the product phase
                                                            // is LDS-bandwidth bound otherwise: -0.2 us per step; the folding
                                                            // kernel gives one k-step's 16 registers to its x pipeline: with all
                                                            // 8 held the loop reloads spilled addresses from scratch every step,
                                                            // +1.1 us per step)
    static constexpr int KSR = KS - KSL;
    static constexpr int WElems = kCUnits * 4 * HP;         // bf16 elements of one workgroup's W slice (128 | 256 KB)
    static constexpr int WLdsFwd = 4 * 4 * KSL * 512;       // ... of which in LDS, forward  [4 unit tiles][4 gates][KSL][64][8]
    static constexpr int HS = HP + 8;                       // LDS row stride of the h image (bf16)
    static constexpr int NB = 2 * (CWG - 1);                // backward: foreign granules per thread and step (6 | 14)
    // forward: with 8 workgroups a thread polls ALL 8 slots of its (row pair, unit) column, its own workgroup's included
    // (already there): the 16 addresses are then one base plus constants, which keeps 28 registers free
    static constexpr bool PollOwn = CWG > 4;
    static constexpr int NF = PollOwn ? 2 * CWG : 2 * (CWG - 1);
    static constexpr int KSLB = 4;                          // backward: LDS-resident k-steps of the 8 a workgroup owns
    static constexpr int KSRB = 8 - KSLB;
    static constexpr int WLdsBwd = (HP / 16) * KSLB * 512;  // [HP/16 unit tiles][KSLB][64][8]
    static constexpr int NRT = CWG / 4;                     // backward product: row tiles per wave (4: wave = (row tile, destination);
                                                            //   8: wave = destination, both row tiles -- no W fragment is held twice)
};
constexpr int kCDS = 4 * kCUnits + 8;                       // LDS row stride of the dG image: a workgroup's own 256 gate columns
// A poll gives up after kPollTimeoutTicks of the 100 MHz wall clock (s_memrealtime), checked every 1024 spins.  Generous
// on purpose: partners can be late for reasons that are not a deadlock -- e.g. a collective kernel of the data-parallel
// path holding CUs until a slower rank arrives, so that not every workgroup of a launch is resident yet.
constexpr unsigned long long kPollTimeoutTicks = 10ull * 100000000ull;   // 10 s

__device__ __forceinline__ unsigned long long wall_ticks() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

// optional phase timing of one forward workgroup (build with -DADN_LSTM_STAMPS; read with adn_debug_lstm_stamps):
// 100 MHz wall-clock ticks spent in [product, gate math + publish, outputs, poll, fill + barrier], summed over steps
#ifdef ADN_LSTM_STAMPS
__device__ unsigned long long g_stamps[8];
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
#define STAMP(k) do { if (blockIdx.x == 5 && blockIdx.y == 0 && tid == 0) { const unsigned long long now_ = stamp_now(); \
    atomicAdd(&g_stamps[k], now_ - last_); last_ = now_; } } while (0)
#define STAMP_INIT unsigned long long last_ = stamp_now();
#else
#define STAMP(k) do {} while (0)
#define STAMP_INIT
#endif

// one term of a gradient that sums over utterance groups: a float atomic into the gradient (arrival order), or -- deterministic
// mode -- a plain add into the caller's own slot, which no other lane writes
__device__ __forceinline__ void group_sum_add(float* grad, float* slot, float v) {
    if (slot) *slot += v; else atomicAdd(grad, v);
}

__device__ __forceinline__ unsigned long long granule_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void granule_store(unsigned long long* p, unsigned payload, unsigned tag) {
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for the HBM
// round trips of the step's operand requests and for the published granules' acknowledgements -- measured 1.5-2 us
// per backward step; the compiler still waits for each load's data where it is used.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// Polls N granules (addresses ptr[k]) until all carry `tag`; returns false when it gave up.
template <int N>
__device__ __forceinline__ bool granule_wait(const unsigned long long* const (&ptr)[N], unsigned tag, unsigned (&payload)[N],
                                             int* err) {
    unsigned pending = (1u << N) - 1u;
    unsigned long long t_start = 0;
    for (int spin = 0; pending; ++spin) {
        unsigned long long g[N];
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (pending & (1u << k)) g[k] = granule_load(ptr[k]);
#pragma unroll
        for (int k = 0; k < N; ++k)
            if ((pending & (1u << k)) && (unsigned)(g[k] >> 32) == tag) { payload[k] = (unsigned)g[k]; pending &= ~(1u << k); }
        if (pending && (spin & 1023) == 1023) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
            const unsigned long long now = wall_ticks();
            if (!t_start) t_start = now;
            else if (now - t_start > kPollTimeoutTicks) {
                atomicCAS(err, 0, 1 | ((int)(tag & 1023u) << 4) | ((int)blockIdx.x << 16));
                return false;
            }
        }
    }
    return true;
}

// =========================================================================================
// forward
// =========================================================================================
// grid (CWG * groups, n LSTMs); 512 threads: wave w -> row tile w >> 2 (16 rows), local unit tile w & 3 (16 units);
// lane -> unit (lane & 15), rows 4 (lane >> 4) .. +3 of the row tile.
// KXS > 0: the kernel also computes the LSTM's INPUT projection x_t W_in + b itself (k-steps of 32 input features, KXS = 3: up to
// 96, KXS = 5: up to 160 -- the 150 = 3 x 50 delta features of an encoder stream): the workgroup's 256 gate columns of W_in stay in
// LDS beside the h image ([16 tiles][KXS][64 lanes][8] bf16: 80 KB at KXS = 5; the W_hid slice of a 4-workgroup group sits in
// registers), x_{t+1} arrives as MFMA A fragments straight from the bf16 feature matrix (requested at the top of step t), and its
// 4 x KXS MFMAs per wave run between the step's output stores and the poll for the partners' h_t, i.e. while the own granules
// travel.  Against the projection as a GEMM ahead of the launch this drops a [T B][4H] fp32 matrix from HBM twice (written by
// the GEMM: 250 MB for the three stream LSTMs of the bench model, 91 us; read back here, where the step waited ~0.7 us for it)
// and a launch.  Same products: bf16 operands, fp32 accumulation over the k-steps in order, bias added last.
template <int CWG, int KXS = 0>
__global__ __launch_bounds__(512) void lstm_fwd_cluster_kernel(const LstmClusterP L, const uint8_t* __restrict__ mask_arg,
                                                               int B, int T_arg, int H, int ldh, int ldg, unsigned tag0, int* err) {
    using G = ClusterGeom<CWG, KXS>;
    constexpr int HP = G::HP, KS = G::KS, KSL = G::KSL, KSR = G::KSR, HS = G::HS, NF = G::NF;
    constexpr bool FOLD = KXS > 0;
    static_assert(!FOLD || CWG == 4, "the folded input projection needs the LDS the 8-workgroup geometry gives to W_hid");
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    __bf16* wl = lds;                                 // [4 unit tiles][4 gates][KSL k-steps][64 lanes][8]
    __bf16 (*hs)[HS] = reinterpret_cast<__bf16 (*)[HS]>(lds + G::WLdsFwd);     // [32][HS]
    bf16x8* win = reinterpret_cast<bf16x8*>(lds + G::WLdsFwd + kCRows * HS);   // FOLD: [16 (unit tile, gate)][KXS][64 lanes] fragments
    constexpr int XS = 32 * KXS + 8;                  // FOLD: row stride of the x tiles [2 parities][32 rows][XS] (bf16)
    __bf16* xs = lds + G::WLdsFwd + kCRows * HS + 16 * KXS * 64 * 8;
    const int pair_ = L.pair0 + (int)blockIdx.x / CWG, lstm_ = pair_ / L.groups;
    const LstmStep& P = L.l[lstm_];
    // (a launch entry may be one LENGTH BUCKET of an LSTM -- model.hip, TmPlan: its own step count and its own rows of the mask)
    const int T = P.T_own ? P.T_own : T_arg;
    const uint8_t* __restrict__ mask_tb = P.mask_own ? P.mask_own : mask_arg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int group = pair_ - lstm_ * L.groups, j = blockIdx.x % CWG;
    const int r0 = group * kCRows;
    const int rt = wave >> 2, ut = wave & 3;
    const int u = kCUnits * j + 16 * ut + i;          // this lane's hidden unit
    const int uc = min(u, H - 1);
    __bf16* h16g = reinterpret_cast<__bf16*>(P.h16);
    // exchange buffer of the group: [2 parities][16 row pairs][HP units] granules
    unsigned long long* xb = reinterpret_cast<unsigned long long*>(P.xchg) + (size_t)group * 2 * 16 * HP;

    // ---- resident W slice: unit tiles 4j .. 4j+3 of the fragment image ([tile][gate][KS][64][8]); k-steps < KSL go to
    //      LDS, the others into this wave's registers (its own unit tile only)
    const bf16x8* wsrc = reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(P.W_frag_fwd) + (size_t)j * G::WElems);
    if constexpr (KSL > 0) {
        for (int e = tid; e < G::WLdsFwd / 8; e += 512) {
            const int l64 = e & 63, s_ = (e >> 6) % KSL, tg = (e >> 6) / KSL;     // tg = 4 * unit tile + gate
            reinterpret_cast<bf16x8*>(wl)[e] = wsrc[((size_t)tg * KS + s_) * 64 + l64];
        }
    }
    bf16x8 wreg[4][KSR > 0 ? KSR : 1];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int s_ = 0; s_ < KSR; ++s_) wreg[g][s_] = wsrc[((size_t)(4 * ut + g) * KS + KSL + s_) * 64 + lane];
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (FOLD) {
        const bf16x8* wi = reinterpret_cast<const bf16x8*>(P.W_in_frag) + (size_t)j * 16 * KXS * 64;
        for (int e = tid; e < 16 * KXS * 64; e += 512) win[e] = wi[e];
        bias4 = *reinterpret_cast<const float4*>(P.b_in + uc * 4);
    }
    // ---- initial state
    const int blk0 = P.backwards ? T : 0;
    for (int e = tid; e < kCRows * (HS / 8); e += 512) {
        const int rr = e / (HS / 8), cc = (e % (HS / 8)) * 8;
        bf16x8 v = bf16x8{};
        if (cc < ldh && cc < HP) v = *reinterpret_cast<const bf16x8*>(h16g + ((size_t)blk0 * B + min(r0 + rr, B - 1)) * ldh + cc);
        *reinterpret_cast<bf16x8*>(&hs[rr][cc]) = v;
    }
    float c_st[4], h_st[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const size_t idx = ((size_t)blk0 * B + min(r0 + 16 * rt + 4 * kq + r, B - 1)) * ldh + uc;
        c_st[r] = P.cbuf[idx];
        h_st[r] = P.hbuf[idx];
    }
    __syncthreads();

    // FOLD: the group's x_t tile (32 rows x 32 KXS features) travels HBM -> registers -> LDS two steps ahead of its use: thread
    // e (and e + 512) owns the 16-byte chunk (row e / (4 KXS), features 8 (e % (4 KXS))) -- one or two chunks per thread, a
    // whole step in registers so that no wave ever waits for the load; features beyond Kx are cleared on the way (the pad
    // columns of the matrix may hold anything; W_in's image is zero there, but 0 x NaN is not)
    const __bf16* x16 = reinterpret_cast<const __bf16*>(P.x16);
    constexpr int XCH = FOLD ? 4 * KXS : 1;           // chunks per row
    constexpr int XN = FOLD ? (kCRows * XCH + 511) / 512 : 1;
    bf16x8 xreg[XN];
    f32x4 xacc[4];                                    // x_t W_in of the step whose gate math comes next (4 gates x 4 rows of this lane)
    auto request_x = [&](int step_) {
        const int t_ = P.backwards ? (T - 1 - step_) : step_;
#pragma unroll
        for (int q = 0; q < XN; ++q) {
            const int e = tid + 512 * q;
            if (e < kCRows * XCH)
                xreg[q] = *reinterpret_cast<const bf16x8*>(x16 + ((size_t)t_ * B + min(r0 + e / XCH, B - 1)) * P.ld_x + 8 * (e % XCH));
        }
    };
    auto stage_x = [&](int par) {
#pragma unroll
        for (int q = 0; q < XN; ++q) {
            const int e = tid + 512 * q;
            if (e < kCRows * XCH) {
                const int keep = P.Kx - 8 * (e % XCH);
                uint4 d = __builtin_bit_cast(uint4, xreg[q]);
                d.x &= (keep > 0 ? 0xffffu : 0u) | (keep > 1 ? 0xffff0000u : 0u);
                d.y &= (keep > 2 ? 0xffffu : 0u) | (keep > 3 ? 0xffff0000u : 0u);
                d.z &= (keep > 4 ? 0xffffu : 0u) | (keep > 5 ? 0xffff0000u : 0u);
                d.w &= (keep > 6 ? 0xffffu : 0u) | (keep > 7 ? 0xffff0000u : 0u);
                *reinterpret_cast<uint4*>(xs + ((size_t)par * kCRows + e / XCH) * XS + 8 * (e % XCH)) = d;
            }
        }
    };
    auto project_x = [&](int par) {                   // xacc = x W_in for the tile in parity `par`
#pragma unroll
        for (int g = 0; g < 4; ++g) xacc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bf16x8* wf = win + (size_t)ut * 4 * KXS * 64 + lane;
        const __bf16* xr = xs + ((size_t)par * kCRows + 16 * rt + i) * XS + 8 * kq;
#pragma unroll
        for (int s_ = 0; s_ < KXS; ++s_) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(xr + 32 * s_);
#pragma unroll
            for (int g = 0; g < 4; ++g) xacc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wf[(g * KXS + s_) * 64], xacc[g], 0, 0, 0);
        }
    };
    if constexpr (FOLD) {
        // pipeline fill: x_0 -> its projection; x_1 staged for step 0's project_x; x_2 on its way
        request_x(0); stage_x(0);
        if (T > 1) { request_x(1); stage_x(1); }
        __syncthreads();
        project_x(0);
        if (T > 2) request_x(2);
    }

    // foreign granules this thread fetches every step: (CWG - 1) partners x 16 row pairs x 64 units = NF per thread;
    // granule k of the thread: id = tid + 512 k -> partner (j + 1 + id / 1024) % CWG, row pair (id % 1024) / 64, unit id % 64
    const int f_rp0 = tid >> 6, f_ul = tid & 63;      // id % 1024 = tid + 512 (k & 1): row pair f_rp0 + 8 (k & 1)
    const bf16x8* wfrag = reinterpret_cast<const bf16x8*>(wl) + (size_t)ut * 4 * KSL * 64 + lane;
    uint8_t m[4];           // raw mask bytes: compared where they are used, so that the request does not wait for its own data
    float4 xp[4];
    auto request_inputs = [&](int step_, uint8_t (&mm)[4], float4 (&xx)[4]) {
        const int t_ = P.backwards ? (T - 1 - step_) : step_;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t ridx = (size_t)t_ * B + min(r0 + 16 * rt + 4 * kq + r, B - 1);
            mm[r] = mask_tb[ridx];
            xx[r] = *reinterpret_cast<const float4*>(P.xproj + ridx * ldg + uc * 4);
        }
    };

    STAMP_INIT
    for (int step = 0; step < T; ++step) {
        const int t = P.backwards ? (T - 1 - step) : step;
        const int out_blk = t + (P.backwards ? 0 : 1);
        const unsigned tag = tag0 + (unsigned)step;
        unsigned long long* xpar = xb + (size_t)(step & 1) * 16 * HP;
        // masks and input projections of the step (the round trip hides under the recurrent product; requesting them a
        // step ahead, before or after the polls, measured no faster)
        if constexpr (FOLD) {
#pragma unroll
            for (int r = 0; r < 4; ++r) m[r] = mask_tb[(size_t)t * B + min(r0 + 16 * rt + 4 * kq + r, B - 1)];
        } else {
            request_inputs(step, m, xp);
        }
        // ---- recurrent product: 4 gate tiles x KS k-steps, W fragments out of LDS (s < KSL) or registers
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(&hs[16 * rt + i][s * 32 + kq * 8]);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, s < KSL ? wfrag[(g * KSL + s) * 64] : wreg[g][s < KSL ? 0 : s - KSL],
                                                                 acc[g], 0, 0, 0);
        }
        lds_barrier();                                // every wave has read h_{t-1}: the image may be overwritten
        STAMP(0);
        if constexpr (FOLD) {
            // x_{t+2} (requested a step ago) goes into the tile project_x read during the previous step; x_{t+3} sets out
            if (step + 2 < T) stage_x(step & 1);
            if (step + 3 < T) request_x(step + 3);
        }
        // ---- gate math; lane = (unit, 4 rows)
        float h_out[4];
        float4 gts[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // (FOLD: the projection of this step, (x_t W_in) + b, the bias added last like the GEMM's epilogue adds it)
            if constexpr (FOLD) xp[r] = make_float4(xacc[0][r] + bias4.x, xacc[1][r] + bias4.y, xacc[2][r] + bias4.z, xacc[3][r] + bias4.w);
            float a_i = xp[r].x + acc[0][r], a_f = xp[r].y + acc[1][r];
            float a_g = xp[r].z + acc[2][r], a_o = xp[r].w + acc[3][r];
            const float c_prev = c_st[r], h_prev = h_st[r];
            if (P.peep) { a_i += c_prev * P.peep[uc]; a_f += c_prev * P.peep[ldh + uc]; }
            const float gi = c_sigmoid(a_i), gf = c_sigmoid(a_f), gg = c_tanh(a_g);
            const float c_new = gf * c_prev + gi * gg;
            if (P.peep) a_o += c_new * P.peep[2 * ldh + uc];
            const float go = c_sigmoid(a_o);
            const float h_new = go * c_tanh(c_new);
            c_st[r] = m[r] ? c_new : c_prev;
            h_out[r] = m[r] ? h_new : h_prev;
            h_st[r] = h_out[r];
            gts[r] = make_float4(gi, gf, gg, go);
        }
        if (u >= H) {
#pragma unroll
            for (int r = 0; r < 4; ++r) h_out[r] = 0.f;
        }
        // ---- publish h_t of this lane's unit first (the partners are waiting for it), 2 rows per granule
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            bf16x2 pr; pr[0] = (__bf16)h_out[2 * rp]; pr[1] = (__bf16)h_out[2 * rp + 1];
            granule_store(xpar + (size_t)(8 * rt + 2 * kq + rp) * HP + u, __builtin_bit_cast(unsigned, pr), tag);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) hs[16 * rt + 4 * kq + r][u] = (__bf16)h_out[r];
        STAMP(1);
        // ---- the step's outputs
        if (u < H) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = r0 + 16 * rt + 4 * kq + r;
                if (grow < B) {
                    const size_t ridx = (size_t)t * B + grow;
                    const size_t oidx = ((size_t)out_blk * B + grow) * ldh + u;
                    P.cbuf[oidx] = c_st[r];
                    P.hbuf[oidx] = h_out[r];
                    h16g[oidx] = (__bf16)h_out[r];
                    if (P.gates) *reinterpret_cast<float4*>(P.gates + ridx * ldg + u * 4) = gts[r];
                }
            }
        }
        if constexpr (FOLD) {
            if (step + 1 < T) project_x((step + 1) & 1);   // the own granules are on their way: the matrix pipe is idle until the poll
        }
        STAMP(2);
        // ---- gather the partners' h_t
        if (step + 1 < T) {
            const unsigned long long* ptr[NF];
            unsigned pay[NF];
#pragma unroll
            for (int k = 0; k < NF; ++k) {
                pay[k] = 0u;
                ptr[k] = xpar + (size_t)(f_rp0 + 8 * (k & 1)) * HP + kCUnits * (G::PollOwn ? (k >> 1) : (j + 1 + (k >> 1)) % CWG) + f_ul;
            }
            granule_wait<NF>(ptr, tag, pay, err);
            STAMP(3);
#pragma unroll
            for (int k = 0; k < NF; ++k) {
                const bf16x2 pr = __builtin_bit_cast(bf16x2, pay[k]);
                const int fu = kCUnits * (G::PollOwn ? (k >> 1) : (j + 1 + (k >> 1)) % CWG) + f_ul, frp = f_rp0 + 8 * (k & 1);
                hs[2 * frp][fu] = pr[0];
                hs[2 * frp + 1][fu] = pr[1];
            }
        }
        lds_barrier();
        STAMP(4);
    }
}

// ---------------------------------------------------------------------------------------------------------
// bf16x3 forward (ADN_PRECISION_BF16X3, H <= 256): the same schedule with fp32-grade recurrent products.
//   h = hi + lo and W_hid = W_hi + W_lo in bf16;  h W ~ h_hi W_hi + h_lo W_hi + h_hi W_lo  (fp32 accumulate, 2^-17 per product)
// Resources of a workgroup (4 per 32-utterance group): the hi AND lo fragments of its 128 K-element W slice sit in
// registers with no copy held twice -- for the product, wave w takes unit tile w & 3 and the gate PAIR w >> 2 for both
// 16-row tiles (2 gates x 8 k-steps x (hi + lo) = 128 VGPRs; the bf16 kernel's (row tile, unit tile) split keeps every
// fragment in two waves, which leaves no room for the lo part).  The accumulators then cross to the gate-math lanes
// ((row tile, unit tile) per wave: all 4 gates of a (row, unit) in one lane) through 32 KB of LDS, on the barrier the
// step needs anyway; both h images live in LDS (2 x 16.5 KB).  A step costs 96 MFMAs per wave (1.3 us of the matrix pipe
// at 2 waves per SIMD) against 32 in the bf16 kernel.
// Exchange: h_t travels as q = h rounded to a 16-bit significand -- exactly what hi + lo can hold (hi = bf16(q), lo = q - hi
// is exact in bf16) -- in the top 24 bits of its fp32 pattern; the freed low byte carries half of a 16-bit step tag, so one
// 8-byte granule holds 2 rows of one unit as in the bf16 kernel (6 foreign granules per thread and step: 3 partners x
// 2 row pairs).  The owner builds its own images from the same q: all four workgroups multiply identical operands.  A 16-bit tag cannot be made
// unique over all launches of a process the way the bf16 kernel's 32-bit tag is; the host numbers the launches PER
// EXCHANGE BUFFER instead (x3_launch_seq): a slot can only hold values of the previous launch on the same buffer or of
// this one, and those differ in the sequence bits or, within a launch, in the step.  State, gates and outputs stay
// fp32; no bf16 shadows are read or written.
// ---------------------------------------------------------------------------------------------------------
struct LstmClusterX3P {
    LstmStep l[kMaxLstmPerLaunch];
    unsigned tag0[kMaxLstmPerLaunch];                                  // 1024 seq + 1, seq in [0, 64): tag0 + step < 65536
    int pair0, groups;
};
static_assert(sizeof(LstmClusterX3P) + 64 <= 4096, "kernel arguments: the launch descriptor + the scalars must fit 4 KB");
__device__ __forceinline__ unsigned x3_quant(float h) { return (__builtin_bit_cast(unsigned, h) + 0x80u) & ~0xffu; }
__device__ __forceinline__ void x3_split(unsigned qbits, __bf16& hi, __bf16& lo) {
    const float q = __builtin_bit_cast(float, qbits);
    hi = (__bf16)q;
    lo = (__bf16)(q - (float)hi);
}
constexpr int kX3AccLds = 4 * 2 * 4 * 64 * 4;                          // fp32 words: [4 gates][2 row tiles][4 unit tiles][64 lanes] x 4 rows
constexpr int kX3FwdLds = 3;                                           // lo k-steps of a wave's W fragments that live in LDS
__global__ __launch_bounds__(512) void lstm_fwd_cluster_x3_kernel(const LstmClusterX3P L, const uint8_t* __restrict__ mask_arg,
                                                                  int B, int T_arg, int H, int ldh, int ldg, int* err) {
    using G = ClusterGeom<4>;
    constexpr int CWG = 4, HP = G::HP, KS = G::KS, HS = G::HS, NF = 8;
    const unsigned tag0 = L.tag0[(L.pair0 + (int)blockIdx.x / CWG) / L.groups];
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    __bf16 (*hs_hi)[HS] = reinterpret_cast<__bf16 (*)[HS]>(lds);
    __bf16 (*hs_lo)[HS] = reinterpret_cast<__bf16 (*)[HS]>(lds + kCRows * HS);
    f32x4* xacc = reinterpret_cast<f32x4*>(lds + 2 * kCRows * HS);       // (2 * 32 * 264 * 2 bytes: 16-byte aligned)
    bf16x8* wl = reinterpret_cast<bf16x8*>(lds + 2 * kCRows * HS + kX3AccLds * 2);   // [8 waves][2 gates][kX3FwdLds][64 lanes]
    const int pair_ = L.pair0 + (int)blockIdx.x / CWG, lstm_ = pair_ / L.groups;
    const LstmStep& P = L.l[lstm_];
    // (a launch entry may be one LENGTH BUCKET of an LSTM -- model.hip, TmPlan: its own step count and its own rows of the mask)
    const int T = P.T_own ? P.T_own : T_arg;
    const uint8_t* __restrict__ mask_tb = P.mask_own ? P.mask_own : mask_arg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int group = pair_ - lstm_ * L.groups, j = blockIdx.x % CWG;
    const int r0 = group * kCRows;
    const int rt = wave >> 2, ut = wave & 3;          // gate math: row tile, unit tile;  product: gate pair rt, unit tile ut
    const int u = kCUnits * j + 16 * ut + i;
    const int uc = min(u, H - 1);
    // exchange buffer of the group: [2 parities][16 row pairs][HP units] granules
    unsigned long long* xb = reinterpret_cast<unsigned long long*>(P.xchg) + (size_t)group * 2 * 16 * HP;

    const bf16x8* wsrc_hi = reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(P.W_frag_fwd) + (size_t)j * G::WElems);
    const bf16x8* wsrc_lo = reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(P.W_frag_fwd_lo) + (size_t)j * G::WElems);
    // (the last kX3FwdLds lo k-steps of a wave sit in LDS: with all 128 fragment registers taken the step's other values spill)
    constexpr int KR = KS - kX3FwdLds;
    bf16x8 whi[2][KS], wlo[2][KR];
    bf16x8* wmine = wl + (size_t)wave * 2 * kX3FwdLds * 64 + lane;
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            const int ks = (2 * j + s_) & (KS - 1);          // register slot s_ holds k-step ks: the own units' two k-steps first
            whi[gi][s_] = wsrc_hi[((size_t)(4 * ut + 2 * rt + gi) * KS + ks) * 64 + lane];
            const bf16x8 lo = wsrc_lo[((size_t)(4 * ut + 2 * rt + gi) * KS + ks) * 64 + lane];
            if (s_ < KR) wlo[gi][s_ < KR ? s_ : 0] = lo;
            else wmine[(gi * kX3FwdLds + (s_ - KR)) * 64] = lo;
        }
    // ---- initial state: both images from the fp32 block
    const int blk0 = P.backwards ? T : 0;
    for (int e = tid; e < kCRows * HP; e += 512) {
        const int rr = e / HP, cc = e % HP;
        const float v = cc < H ? P.hbuf[((size_t)blk0 * B + min(r0 + rr, B - 1)) * ldh + cc] : 0.f;
        x3_split(x3_quant(v), hs_hi[rr][cc], hs_lo[rr][cc]);
    }
    float c_st[4], h_st[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const size_t idx = ((size_t)blk0 * B + min(r0 + 16 * rt + 4 * kq + r, B - 1)) * ldh + uc;
        c_st[r] = P.cbuf[idx];
        h_st[r] = P.hbuf[idx];
    }
    __syncthreads();

    // product of this wave's gate pair, both row tiles, one k-step (register slot s = k-step (2j + s) mod 8): three MFMAs
    // per (gate, row tile)
    f32x4 pacc[2][2];
    const int acol = kq * 8;
    auto kstep = [&](int s) {
        const int col = ((2 * j + s) & (KS - 1)) * 32 + acol;
        bf16x8 a_hi[2], a_lo[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            a_hi[q] = *reinterpret_cast<const bf16x8*>(&hs_hi[16 * q + i][col]);
            a_lo[q] = *reinterpret_cast<const bf16x8*>(&hs_lo[16 * q + i][col]);
        }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
            for (int q = 0; q < 2; ++q) pacc[gi][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[q], whi[gi][s], pacc[gi][q], 0, 0, 0);
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
            for (int q = 0; q < 2; ++q) pacc[gi][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[q], whi[gi][s], pacc[gi][q], 0, 0, 0);
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                pacc[gi][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    a_hi[q], s < KR ? wlo[gi][s < KR ? s : 0] : wmine[(gi * kX3FwdLds + (s < KR ? 0 : s - KR)) * 64], pacc[gi][q], 0, 0, 0);
    };
    auto own_part = [&]() {                           // the two k-steps of this workgroup's own units: a fresh accumulator
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
            for (int q = 0; q < 2; ++q) pacc[gi][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        kstep(0);
        kstep(1);
    };
    own_part();

    uint8_t m[4];
    float4 xp[4];
    STAMP_INIT
    for (int step = 0; step < T; ++step) {
        const int t = P.backwards ? (T - 1 - step) : step;
        const int out_blk = t + (P.backwards ? 0 : 1);
        const unsigned tag = tag0 + (unsigned)step;
        unsigned long long* xpar = xb + (size_t)(step & 1) * 16 * HP;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t ridx = (size_t)t * B + min(r0 + 16 * rt + 4 * kq + r, B - 1);
            m[r] = mask_tb[ridx];
            xp[r] = *reinterpret_cast<const float4*>(P.xproj + ridx * ldg + uc * 4);
        }
        // ---- recurrent product, the partners' six k-steps (the own units' two were multiplied while their granules travelled)
#pragma unroll
        for (int s = 2; s < KS; ++s) kstep(s);
        // accumulator lane map = gate-math lane map (unit lane & 15, rows 4 (lane >> 4) ..+3): hand each tile to its wave
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
            for (int q = 0; q < 2; ++q) xacc[(((2 * rt + gi) * 2 + q) * 4 + ut) * 64 + lane] = pacc[gi][q];
        lds_barrier();                                // every wave has read h_{t-1} (the images may be overwritten); tiles are in place
        STAMP(0);
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = xacc[((g * 2 + rt) * 4 + ut) * 64 + lane];
        // ---- gate math; own images and the publication of each row pair first (the partners are waiting for it)
        unsigned q_even = 0u;
        float h_out[4];
        float4 gts[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a_i = xp[r].x + acc[0][r], a_f = xp[r].y + acc[1][r];
            float a_g = xp[r].z + acc[2][r], a_o = xp[r].w + acc[3][r];
            const float c_prev = c_st[r], h_prev = h_st[r];
            if (P.peep) { a_i += c_prev * P.peep[uc]; a_f += c_prev * P.peep[ldh + uc]; }
            const float gi = c_sigmoid(a_i), gf = c_sigmoid(a_f), gg = c_tanh(a_g);
            const float c_new = gf * c_prev + gi * gg;
            if (P.peep) a_o += c_new * P.peep[2 * ldh + uc];
            const float go = c_sigmoid(a_o);
            const float h_new = go * c_tanh(c_new);
            c_st[r] = m[r] ? c_new : c_prev;
            float h_o = m[r] ? h_new : h_prev;
            h_st[r] = h_o;
            if (u >= H) h_o = 0.f;
            h_out[r] = h_o;
            gts[r] = make_float4(gi, gf, gg, go);
            const unsigned qb = x3_quant(h_o);
            const int row = 16 * rt + 4 * kq + r;
            x3_split(qb, hs_hi[row][u], hs_lo[row][u]);
            if (r & 1)
                __hip_atomic_store(xpar + (size_t)(8 * rt + 2 * kq + (r >> 1)) * HP + u,
                                   ((unsigned long long)(qb | (tag >> 8)) << 32) | (q_even | (tag & 255u)), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            else
                q_even = qb;
        }
        STAMP(1);
        // ---- the step's outputs: issued AHEAD of the barrier (round 6) -- they depend on nothing the barrier orders, and the
        //      waves that reach it early spend the wait issuing stores instead of idling
        if (u < H) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = r0 + 16 * rt + 4 * kq + r;
                if (grow < B) {
                    const size_t ridx = (size_t)t * B + grow;
                    const size_t oidx = ((size_t)out_blk * B + grow) * ldh + u;
                    P.cbuf[oidx] = c_st[r];
                    P.hbuf[oidx] = h_out[r];
                    if (P.gates) *reinterpret_cast<float4*>(P.gates + ridx * ldg + u * 4) = gts[r];
                }
            }
        }
        lds_barrier();                                // the own units' h_t is complete in both images
        // ---- the own units' share of the NEXT step's product, while the partners' granules are in flight
        if (step + 1 < T) own_part();
        STAMP(2);
        // ---- gather the partners' h_t: slot k = (workgroup k >> 1, row pair (tid >> 6) + 8 (k & 1)) of this thread's unit
        //      column; the own workgroup's two slots are skipped (already in the images).  Global and LDS addresses are
        //      one base plus compile-time constants.
        if (step + 1 < T) {
            const unsigned long long* p0 = xpar + (size_t)(tid >> 6) * HP + (tid & 63);
            const unsigned long long* p1 = p0 + (size_t)8 * HP;
            __bf16* img = &hs_hi[2 * (tid >> 6)][tid & 63];              // lo image: + kCRows * HS elements
            unsigned long long g[NF];
            unsigned pending = ((1u << NF) - 1u) & ~(3u << (2 * j));
            unsigned long long t_start = 0;
            for (int spin = 0; pending; ++spin) {
                unsigned long long v[NF];
#pragma unroll
                for (int k = 0; k < NF; ++k)
                    if (pending & (1u << k)) v[k] = granule_load(((k & 1) ? p1 : p0) + kCUnits * (k >> 1));
#pragma unroll
                for (int k = 0; k < NF; ++k)
                    if ((pending & (1u << k)) && (((unsigned)v[k] & 255u) | (((unsigned)(v[k] >> 32) & 255u) << 8)) == tag) {
                        g[k] = v[k]; pending &= ~(1u << k);
                    }
                if (pending && (spin & 1023) == 1023) {
                    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    const unsigned long long now = wall_ticks();
                    if (!t_start) t_start = now;
                    else if (now - t_start > kPollTimeoutTicks) { atomicCAS(err, 0, 1 | ((int)(step & 1023) << 4) | ((int)blockIdx.x << 16)); break; }
                }
            }
            STAMP(3);
#pragma unroll
            for (int k = 0; k < NF; ++k) {
                if ((k >> 1) == j) continue;
                __bf16* d = img + (16 * (k & 1)) * HS + kCUnits * (k >> 1);
                x3_split((unsigned)g[k] & ~255u, d[0], d[kCRows * HS]);
                x3_split((unsigned)(g[k] >> 32) & ~255u, d[HS], d[HS + kCRows * HS]);
            }
        }
        lds_barrier();
        STAMP(4);
    }
}

// ---------------------------------------------------------------------------------------------------------
// bf16x3 forward, 256 < H <= 512 ("wide").  hi + lo fragments of a 64-unit slice would be 512 KB: no CU holds that, so a
// group is 16 workgroups of 32 hidden units (256 KB of fragments each, in registers as in the kernel above) and -- to keep
// 16 workgroups per group within the device at the whole-split batches (B = 520: 11 groups = 176 CUs) -- 48 utterances.
//   product     wave w: gate w >> 1, unit tile w & 1, ALL 16 k-steps and the 3 row tiles: 16 x (hi + lo) = 128 VGPRs of
//               fragments (kWFwdLds lo k-steps of them in LDS), 144 MFMAs per step; no fragment is held twice, no partial
//               sums over k to combine
//   gate math   wave w < 6: row tile w >> 1, unit tile w & 1; the accumulators cross through 24 KB of LDS
//   exchange    [2 parities][12 row quads][512 units][2 granules] per group, granules tagged as above; a lane publishes its 4
//               rows (2 granules) with ONE 16-byte write-through store; a thread owns one unit column and polls its 12 quads
//               in three rounds of 4 16-byte sc1 buffer loads, each half validated by its own tag (the own workgroup's 32
//               columns sit out)
// LDS: the two h images 2 x 48 x 520 x 2 B = 97.5 KB + 24 KB + 24 KB of lo fragments.
// ---------------------------------------------------------------------------------------------------------
constexpr int kWRows = 48, kWUnits = 32, kWCWG = 16, kWHP = 512, kWKS = 16, kWHS = kWHP + 8;
constexpr int kWAccLds = 4 * 3 * 2 * 64 * 4;                           // fp32 words: [4 gates][3 row tiles][2 unit tiles][64 lanes] x 4 rows
constexpr int kWFwdLds = 4;                                            // lo k-steps of a wave's W fragments that live in LDS
constexpr size_t kWFwdLdsBytes = (size_t)2 * kWRows * kWHS * 2 + (size_t)kWAccLds * 4 + (size_t)8 * kWFwdLds * 64 * 16;
__global__ __launch_bounds__(512) void lstm_fwd_cluster_x3w_kernel(const LstmClusterX3P L, const uint8_t* __restrict__ mask_tb,
                                                                   int B, int T, int H, int ldh, int ldg, int* err) {
    constexpr int HP = kWHP, KS = kWKS, HS = kWHS, R = kWRows, NF = 4;
    const unsigned tag0 = L.tag0[(L.pair0 + (int)blockIdx.x / kWCWG) / L.groups];
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    __bf16 (*hs_hi)[HS] = reinterpret_cast<__bf16 (*)[HS]>(lds);
    __bf16 (*hs_lo)[HS] = reinterpret_cast<__bf16 (*)[HS]>(lds + R * HS);
    f32x4* xacc = reinterpret_cast<f32x4*>(lds + 2 * R * HS);            // (2 * 48 * 520 * 2 bytes: 16-byte aligned)
    bf16x8* wl = reinterpret_cast<bf16x8*>(lds + 2 * R * HS + kWAccLds * 2);   // [8 waves][kWFwdLds][64 lanes]
    const int pair_ = L.pair0 + (int)blockIdx.x / kWCWG, lstm_ = pair_ / L.groups;
    const LstmStep& P = L.l[lstm_];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int group = pair_ - lstm_ * L.groups, j = blockIdx.x % kWCWG;
    const int r0 = group * R;
    const int rt = wave >> 1, ut = wave & 1;          // gate math: row tile (3: this wave has none), unit tile;  product: gate rt, unit tile ut
    const bool gm = rt < 3;
    const int u = kWUnits * j + 16 * ut + i;
    const int uc = min(u, H - 1);
    unsigned long long* xb = reinterpret_cast<unsigned long long*>(P.xchg) + (size_t)group * 2 * (R / 2) * HP;

    const bf16x8* wsrc_hi = reinterpret_cast<const bf16x8*>(P.W_frag_fwd) + ((size_t)(4 * (2 * j + ut) + rt) * KS) * 64 + lane;
    const bf16x8* wsrc_lo = reinterpret_cast<const bf16x8*>(P.W_frag_fwd_lo) + ((size_t)(4 * (2 * j + ut) + rt) * KS) * 64 + lane;
    constexpr int KR = KS - kWFwdLds;
    bf16x8 whi[KS], wlo[KR];
    bf16x8* wmine = wl + (size_t)wave * kWFwdLds * 64 + lane;
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) {
        whi[s_] = wsrc_hi[(size_t)s_ * 64];
        const bf16x8 lo = wsrc_lo[(size_t)s_ * 64];
        if (s_ < KR) wlo[s_ < KR ? s_ : 0] = lo;
        else wmine[(s_ - KR) * 64] = lo;
    }
    // ---- initial state: both images from the fp32 block
    const int blk0 = P.backwards ? T : 0;
    for (int e = tid; e < R * HP; e += 512) {
        const int rr = e / HP, cc = e % HP;
        const float v = cc < H ? P.hbuf[((size_t)blk0 * B + min(r0 + rr, B - 1)) * ldh + cc] : 0.f;
        x3_split(x3_quant(v), hs_hi[rr][cc], hs_lo[rr][cc]);
    }
    float c_st[4], h_st[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const size_t idx = ((size_t)blk0 * B + min(r0 + 16 * min(rt, 2) + 4 * kq + r, B - 1)) * ldh + uc;
        c_st[r] = P.cbuf[idx];
        h_st[r] = P.hbuf[idx];
    }
    __syncthreads();

    const int acol = kq * 8;
    const int row0 = r0 + 16 * min(rt, 2) + 4 * kq;
    float pw_i = 0.f, pw_f = 0.f, pw_o = 0.f;          // peephole weights of this lane's unit (0: none -- c * 0 adds nothing)
    if (P.peep) { pw_i = P.peep[uc]; pw_f = P.peep[ldh + uc]; pw_o = P.peep[2 * ldh + uc]; }
    uint8_t m[4];
    float4 xp[4];
    STAMP_INIT
    for (int step = 0; step < T; ++step) {
        const int t = P.backwards ? (T - 1 - step) : step;
        const int out_blk = t + (P.backwards ? 0 : 1);
        const unsigned tag = tag0 + (unsigned)step;
        unsigned long long* xpar = xb + (size_t)(step & 1) * (R / 2) * HP;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xpar, 0, (R / 2) * HP * 8, 0x00020000);
        // (the row index is made opaque once per step: the compiler otherwise keeps a 64-bit address per row and array across the
        //  loop -- 16 registers spilled to scratch and reloaded every step)
        int rw = row0;
        asm volatile("" : "+v"(rw));
        int ucl = uc;
        asm volatile("" : "+v"(ucl));
        if (gm) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t ridx = (size_t)t * B + min(rw + r, B - 1);
                m[r] = mask_tb[ridx];
                xp[r] = *reinterpret_cast<const float4*>(P.xproj + ridx * ldg + ucl * 4);
            }
        }
        // ---- recurrent product of this wave's gate: three MFMAs per (row tile, k-step)
        f32x4 pacc[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) pacc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int col = s * 32 + acol;
            bf16x8 a_hi[3], a_lo[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                a_hi[q] = *reinterpret_cast<const bf16x8*>(&hs_hi[16 * q + i][col]);
                a_lo[q] = *reinterpret_cast<const bf16x8*>(&hs_lo[16 * q + i][col]);
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) pacc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[q], whi[s], pacc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 3; ++q) pacc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[q], whi[s], pacc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                pacc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    a_hi[q], s < KR ? wlo[s < KR ? s : 0] : wmine[(s < KR ? 0 : s - KR) * 64], pacc[q], 0, 0, 0);
        }
        // accumulator lane map = gate-math lane map (unit lane & 15, rows 4 (lane >> 4) ..+3): hand each tile to its wave
#pragma unroll
        for (int q = 0; q < 3; ++q) xacc[((rt * 3 + q) * 2 + ut) * 64 + lane] = pacc[q];
        lds_barrier();                                // every wave has read h_{t-1} (the images may be overwritten); tiles are in place
        STAMP(0);
        if (gm) {
            f32x4 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = xacc[((g * 3 + rt) * 2 + ut) * 64 + lane];
            // ---- gate math; own images and the publication of each row pair first (the partners are waiting for it)
            unsigned q_even = 0u;
            unsigned long long g_first = 0ull;
            float h_out[4];
            float4 gts[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a_i = xp[r].x + acc[0][r], a_f = xp[r].y + acc[1][r];
                float a_g = xp[r].z + acc[2][r], a_o = xp[r].w + acc[3][r];
                const float c_prev = c_st[r], h_prev = h_st[r];
                a_i += c_prev * pw_i; a_f += c_prev * pw_f;
                const float gi = c_sigmoid(a_i), gf = c_sigmoid(a_f), gg = c_tanh(a_g);
                const float c_new = gf * c_prev + gi * gg;
                a_o += c_new * pw_o;
                const float go = c_sigmoid(a_o);
                const float h_new = go * c_tanh(c_new);
                c_st[r] = m[r] ? c_new : c_prev;
                float h_o = m[r] ? h_new : h_prev;
                h_st[r] = h_o;
                if (u >= H) h_o = 0.f;
                h_out[r] = h_o;
                gts[r] = make_float4(gi, gf, gg, go);
                const unsigned qb = x3_quant(h_o);
                const int row = 16 * rt + 4 * kq + r;
                x3_split(qb, hs_hi[row][u], hs_lo[row][u]);
                if (r == 1) {
                    g_first = ((unsigned long long)(qb | (tag >> 8)) << 32) | (q_even | (tag & 255u));
                } else if (r == 3) {
                    const u32x4 pk = {(unsigned)g_first, (unsigned)(g_first >> 32), q_even | (tag & 255u), qb | (tag >> 8)};
                    __builtin_amdgcn_raw_buffer_store_b128(pk, rs, (unsigned)(((4 * rt + kq) * HP + u) * 16), 0, 16);
                } else {
                    q_even = qb;
                }
            }
            STAMP(1);
            // ---- the step's outputs
            if (u < H) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int grow = rw + r;
                    if (grow < B) {
                        const size_t ridx = (size_t)t * B + grow;
                        const size_t oidx = ((size_t)out_blk * B + grow) * ldh + u;
                        P.cbuf[oidx] = c_st[r];
                        P.hbuf[oidx] = h_out[r];
                        if (P.gates) *reinterpret_cast<float4*>(P.gates + ridx * ldg + u * 4) = gts[r];
                    }
                }
            }
        }
        STAMP(2);
        // ---- gather the partners' h_t: this thread's unit column, 12 row quads in three rounds of 4
        if (step + 1 < T && (tid >> 5) != j) {
#pragma unroll 1
            for (int rd = 0; rd < 3; ++rd) {
                const unsigned soff = (unsigned)(NF * rd * HP * 16);
                u32x4 v[NF];
                unsigned pending = (1u << NF) - 1u;
                unsigned long long t_start = 0;
                for (int spin = 0; pending; ++spin) {
#pragma unroll
                    for (int k = 0; k < NF; ++k)
                        if (pending & (1u << k)) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)tid * 16u + (unsigned)(k * HP * 16), soff, 16);
#pragma unroll
                    for (int k = 0; k < NF; ++k)
                        if ((pending & (1u << k)) && ((v[k].x & 255u) | ((v[k].y & 255u) << 8)) == tag && ((v[k].z & 255u) | ((v[k].w & 255u) << 8)) == tag)
                            pending &= ~(1u << k);
                    if (pending && (spin & 1023) == 1023) {
                        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                        const unsigned long long now = wall_ticks();
                        if (!t_start) t_start = now;
                        else if (now - t_start > kPollTimeoutTicks) { atomicCAS(err, 0, 1 | ((int)(step & 1023) << 4) | ((int)blockIdx.x << 16)); break; }
                    }
                }
                __bf16* img = &hs_hi[4 * NF * rd][tid];                      // lo image: + R * HS elements
#pragma unroll
                for (int k = 0; k < NF; ++k) {
                    __bf16* d = img + (4 * k) * HS;
                    x3_split(v[k].x & ~255u, d[0], d[R * HS]);
                    x3_split(v[k].y & ~255u, d[HS], d[HS + R * HS]);
                    x3_split(v[k].z & ~255u, d[2 * HS], d[2 * HS + R * HS]);
                    x3_split(v[k].w & ~255u, d[3 * HS], d[3 * HS + R * HS]);
                }
            }
        }
        STAMP(3);
        lds_barrier();
        STAMP(4);
    }
}

// =========================================================================================
// backward (BPTT): see lstm.hip for the per-step math
// =========================================================================================
// Workgroup j keeps dG_{t+1} of ITS gate columns (32 rows x 256, bf16) in LDS together with the 256 matching rows of
// W_hid^T, multiplies them into a partial dh for ALL 256 units, keeps its own 64-unit quarter and sends the other
// three quarters to their owners.  A partial is fp32; two of them (2 rows of one unit) travel in one 8-byte granule
// whose tag is spread over the 4 low mantissa bits of either value (they keep 19 bits: far finer than the bf16
// operands that produced them).  Tag 0 means "empty": owners zero their inbox slots after the last step, so every
// launch starts from clean slots and only has to tell step s from step s - 2 of the same parity.
constexpr int kBxPair = kCRows / 2 * kCUnits;        // granules of one (destination, source) pair: 16 row pairs x 64 units

__device__ __forceinline__ unsigned long long pack_partials(float a, float b, unsigned tag8) {
    const unsigned ua = (__builtin_bit_cast(unsigned, a) & ~15u) | (tag8 & 15u);
    const unsigned ub = (__builtin_bit_cast(unsigned, b) & ~15u) | (tag8 >> 4);
    return ((unsigned long long)ub << 32) | ua;
}

template <int CWG>
__global__ __launch_bounds__(512) void lstm_bwd_cluster_kernel(const LstmClusterP L, const uint8_t* __restrict__ mask_arg,
                                                               int B, int T_arg, int H, int ldh, int ldg, int* err) {
    using G = ClusterGeom<CWG>;
    constexpr int HP = G::HP, KSLB = G::KSLB, KSRB = G::KSRB, NRT = G::NRT, NF = G::NB;
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    __bf16* wl = lds;                                 // [HP/16 unit tiles][KSLB k-steps][64 lanes][8]: W_hid[unit][own gate columns]
    __bf16 (*dgs)[kCDS] = reinterpret_cast<__bf16 (*)[kCDS]>(lds + G::WLdsBwd);            // [32][kCDS] own dG_{t+1}
    float (*part)[kCUnits + 1] = reinterpret_cast<float (*)[kCUnits + 1]>(lds + G::WLdsBwd + kCRows * kCDS);   // [32][65]
    const int pair_ = L.pair0 + (int)blockIdx.x / CWG, lstm_ = pair_ / L.groups;
    const LstmStep& P = L.l[lstm_];
    // (a launch entry may be one LENGTH BUCKET of an LSTM -- model.hip, TmPlan: its own step count and its own rows of the mask)
    const int T = P.T_own ? P.T_own : T_arg;
    const uint8_t* __restrict__ mask_tb = P.mask_own ? P.mask_own : mask_arg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int group = pair_ - lstm_ * L.groups, j = blockIdx.x % CWG;
    const int r0 = group * kCRows;
    const int rt = wave >> 2, ut = wave & 3;          // gate math: row tile, local unit tile
    const int dst = CWG == 4 ? ut : wave;             // MFMA role: destination workgroup (and row tile rt when CWG == 4)
    const int ul = 16 * ut + i;                       // local unit
    const int u = kCUnits * j + ul;
    __bf16* dg16g = reinterpret_cast<__bf16*>(P.dG16);
    // inbox / outbox: [2 parities][CWG destinations][CWG slots][16 row pairs][64 units] granules, after the forward region;
    // source src uses slot (src - dst - 1) mod CWG of destination dst, so that a receiver's addresses are base + constants
    unsigned long long* xb = reinterpret_cast<unsigned long long*>(P.xchg) + (size_t)((B + kCRows - 1) / kCRows) * 2 * 16 * HP +
                             (size_t)group * 2 * CWG * CWG * kBxPair;

    // resident W slice: of every unit tile of the fragment image ([tile][4 HP / 32 k-steps][64][8]) the 8 k-steps
    // [8j, 8j+8) = this workgroup's gate columns; the first KSLB of them in LDS, the others in the registers of the
    // wave that multiplies them (destination dst: unit tiles 4 dst + ct)
    const bf16x8* wsrc = reinterpret_cast<const bf16x8*>(P.W_frag_bwd) + (size_t)(8 * j) * 64;
    for (int e = tid; e < G::WLdsBwd / 8; e += 512) {
        const int l64 = e & 63, s_ = (e >> 6) % KSLB, tile = (e >> 6) / KSLB;
        reinterpret_cast<bf16x8*>(wl)[e] = wsrc[((size_t)tile * (4 * G::KS) + s_) * 64 + l64];
    }
    bf16x8 wreg[4][KSRB > 0 ? KSRB : 1];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int s_ = 0; s_ < KSRB; ++s_)
            wreg[ct][s_] = wsrc[((size_t)(4 * dst + ct) * (4 * G::KS) + KSLB + s_) * 64 + lane];
    for (int e = tid; e < kCRows * kCDS / 8; e += 512) reinterpret_cast<bf16x8*>(&dgs[0][0])[e] = bf16x8{};
    float dh_c[4], dc_s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { dh_c[r] = 0.f; dc_s[r] = 0.f; }
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);   // this lane's share of the bias gradient (its unit, its 4 rows, all steps)
    __syncthreads();
    const bf16x8* wfrag = reinterpret_cast<const bf16x8*>(wl) + lane;

    const int uc = min(u, H - 1);
    // what the gate math of a step reads from HBM is requested at the top of the step: the round trips hide under the
    // recurrent product and the exchange (measured: requesting a step ahead, before or after the polls, is no faster --
    // the step is bound by the exchange hop, ~3 us of the 5-7 us)
    float l_dhs[4], l_ct[4], l_cp[4];
    float4 l_gt[4];
    uint8_t l_m[4];
    auto request_state = [&](int step_) {
        const int t_ = P.backwards ? step_ : (T - 1 - step_);
        const int pb = t_ + (P.backwards ? 1 : 0), ob = t_ + (P.backwards ? 0 : 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int growc = min(r0 + 16 * rt + 4 * kq + r, B - 1);
            const size_t ridx = (size_t)t_ * B + growc;
            l_m[r] = mask_tb[ridx];
            l_dhs[r] = P.dhs[ridx * (P.ld_dhs ? P.ld_dhs : ldh) + uc];
            l_gt[r] = *reinterpret_cast<const float4*>(P.gates + ridx * ldg + uc * 4);
            l_ct[r] = P.cbuf[((size_t)ob * B + growc) * ldh + uc];
            l_cp[r] = P.cbuf[((size_t)pb * B + growc) * ldh + uc];
        }
    };
    STAMP_INIT
    for (int step = 0; step <= T; ++step) {
        const int t = P.backwards ? step : (T - 1 - step);
        request_state(min(step, T - 1));              // unconditional: a conditional request turns the loaded registers
                                                      // into loop-carried values that the compiler copies (and waits for) at once
        float rec[4] = {0.f, 0.f, 0.f, 0.f};          // recurrent part of dh for this lane's 4 (row, unit) pairs
        if (step > 0) {
            const unsigned tag8 = 1u + (unsigned)(step % 255);
            unsigned long long* xpar = xb + (size_t)(step & 1) * CWG * CWG * kBxPair;
            // ---- partial dh for destination dst (= MFMA role of this wave): 4 unit tiles x 8 k-steps per row tile
#pragma unroll
            for (int q = 0; q < NRT; ++q) {
                const int mr = CWG == 4 ? rt : q;
                f32x4 acc[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(&dgs[16 * mr + i][s * 32 + kq * 8]);
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            a, s < KSLB ? wfrag[((4 * dst + ct) * KSLB + s) * 64] : wreg[ct][s < KSLB ? 0 : s - KSLB], acc[ct], 0, 0, 0);
                }
                // accumulator map: unit = 64 dst + 16 ct + (lane & 15), row = 16 mr + 4 kq + r
                if (dst == j) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) part[16 * mr + 4 * kq + r][16 * ct + i] = acc[ct][r];
                } else {
                    unsigned long long* box = xpar + (size_t)(dst * CWG + (j - dst - 1 + CWG) % CWG) * kBxPair;
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                        for (int rp = 0; rp < 2; ++rp)
                            __hip_atomic_store(box + (size_t)(8 * mr + 2 * kq + rp) * kCUnits + 16 * ct + i,
                                               pack_partials(acc[ct][2 * rp], acc[ct][2 * rp + 1], tag8), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            lds_barrier();                            // own part is in `part`; dG_{t+1} has been consumed
            STAMP(5);
            // ---- collect the CWG - 1 foreign parts of this lane's pairs
            const unsigned long long* ptr[NF];
            unsigned long long g[NF];
#pragma unroll
            for (int k = 0; k < NF; ++k) {
                const int slot = k / 2, rp = k & 1;        // source (j + 1 + slot) mod CWG
                ptr[k] = xpar + (size_t)(j * CWG + slot) * kBxPair + (size_t)(8 * rt + 2 * kq + rp) * kCUnits + ul;
            }
            unsigned pending = (1u << NF) - 1u;
            unsigned long long t_start = 0;
            for (int spin = 0; pending; ++spin) {
                unsigned long long v[NF];
#pragma unroll
                for (int k = 0; k < NF; ++k)          // all outstanding requests first: ONE round trip per spin
                    if (pending & (1u << k)) v[k] = granule_load(ptr[k]);
#pragma unroll
                for (int k = 0; k < NF; ++k)
                    if ((pending & (1u << k)) && (((unsigned)v[k] & 15u) | (((unsigned)(v[k] >> 32) & 15u) << 4)) == tag8) {
                        g[k] = v[k]; pending &= ~(1u << k);
                    }
                if (pending && (spin & 1023) == 1023) {
                    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    const unsigned long long now = wall_ticks();
                    if (!t_start) t_start = now;
                    else if (now - t_start > kPollTimeoutTicks) { atomicCAS(err, 0, 2 | (step << 4) | ((int)blockIdx.x << 16)); break; }
                }
            }
            STAMP(6);
#pragma unroll
            for (int r = 0; r < 4; ++r) rec[r] = part[16 * rt + 4 * kq + r][ul];
#pragma unroll
            for (int k = 0; k < NF; ++k) {
                const int rp = k & 1;
                rec[2 * rp] += __builtin_bit_cast(float, (unsigned)g[k] & ~15u);
                rec[2 * rp + 1] += __builtin_bit_cast(float, (unsigned)(g[k] >> 32) & ~15u);
            }
        }
        if (step == T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dh_c[r] += rec[r];
            break;
        }
        // ---- gate math of step t for this lane's unit and 4 rows
        float pw_i = 0.f, pw_f = 0.f, pw_o = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * rt + 4 * kq + r, grow = r0 + row;
            const size_t ridx = (size_t)t * B + min(grow, B - 1);
            const bool ok = grow < B && u < H;
            float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                const float dh = l_dhs[r] + dh_c[r] + rec[r];
                const float dc = dc_s[r];
                if (l_m[r]) {
                    const float4 gt = l_gt[r];
                    const float c_t = l_ct[r], c_prev = l_cp[r];
                    const float tc = c_tanh(c_t);
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
                    dg = make_float4(c_clip5(da_i), c_clip5(da_f), c_clip5(da_g), c_clip5(da_o));
                    dh_c[r] = 0.f;
                    dc_s[r] = dcp;
                } else {
                    dh_c[r] = dh;
                }
                if (!P.dG_fp32_off) *reinterpret_cast<float4*>(P.dG + ridx * ldg + u * 4) = dg;
                bsum.x += dg.x; bsum.y += dg.y; bsum.z += dg.z; bsum.w += dg.w;
            }
            bf16x4 d16;
            d16[0] = (__bf16)dg.x; d16[1] = (__bf16)dg.y; d16[2] = (__bf16)dg.z; d16[3] = (__bf16)dg.w;
            if (ok) *reinterpret_cast<bf16x4*>(dg16g + ridx * ldg + u * 4) = d16;
            *reinterpret_cast<bf16x4*>(&dgs[row][ul * 4]) = d16;
        }
        if (P.dpeep_part) {                          // sum this lane's 4 rows, then the 4 row groups (kq) of the wave
            float si = pw_i, sf = pw_f, so = pw_o;
            si += __shfl_xor(si, 16, 64); si += __shfl_xor(si, 32, 64);
            sf += __shfl_xor(sf, 16, 64); sf += __shfl_xor(sf, 32, 64);
            so += __shfl_xor(so, 16, 64); so += __shfl_xor(so, 32, 64);
            if (kq == 0 && u < H) {
                float* ds = P.det_ws ? P.det_ws + (size_t)(2 * group + rt) * P.det_stride + ldg + 2 * ldh : nullptr;
                group_sum_add(P.dpeep_part + u, ds ? ds + u : nullptr, si);
                group_sum_add(P.dpeep_part + ldh + u, ds ? ds + ldh + u : nullptr, sf);
                group_sum_add(P.dpeep_part + 2 * (size_t)ldh + u, ds ? ds + 2 * ldh + u : nullptr, so);
            }
        }
        lds_barrier();
        STAMP(7);
    }
    // gradient wrt the initial state of every row of this slice
    float sh = 0.f, sc = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int grow = r0 + 16 * rt + 4 * kq + r;
        if (grow < B && u < H) {
            P.dh_carry[(size_t)grow * ldh + u] = dh_c[r];
            P.dc_state[(size_t)grow * ldh + u] = dc_s[r];
            sh += dh_c[r]; sc += dc_s[r];
        }
    }
    // sums over this workgroup's rows: bias gradient (4 gates of the unit) and the learnt initial state; the 4 row
    // groups of a wave combine by shuffle, then one float atomic per value (2 row tiles x 17 groups adds per address)
    if (P.dbias) {
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
            bsum.x += __shfl_xor(bsum.x, o, 64); bsum.y += __shfl_xor(bsum.y, o, 64);
            bsum.z += __shfl_xor(bsum.z, o, 64); bsum.w += __shfl_xor(bsum.w, o, 64);
            sh += __shfl_xor(sh, o, 64); sc += __shfl_xor(sc, o, 64);
        }
        if (kq == 0 && u < H) {
            float* ds = P.det_ws ? P.det_ws + (size_t)(2 * group + rt) * P.det_stride : nullptr;
            group_sum_add(P.dbias + 4 * u, ds ? ds + 4 * u : nullptr, bsum.x); group_sum_add(P.dbias + 4 * u + 1, ds ? ds + 4 * u + 1 : nullptr, bsum.y);
            group_sum_add(P.dbias + 4 * u + 2, ds ? ds + 4 * u + 2 : nullptr, bsum.z); group_sum_add(P.dbias + 4 * u + 3, ds ? ds + 4 * u + 3 : nullptr, bsum.w);
            group_sum_add(P.dhid_init + u, ds ? ds + ldg + u : nullptr, sh); group_sum_add(P.dcell_init + u, ds ? ds + ldg + ldh + u : nullptr, sc);
        }
    }
    // leave this workgroup's inbox empty for the next launch (after EVERY lane has taken its last granules)
    __syncthreads();
    for (int e = tid; e < 2 * CWG * kBxPair; e += 512) {
        const int par = e / (CWG * kBxPair), rest = e % (CWG * kBxPair);
        xb[(size_t)par * CWG * CWG * kBxPair + (size_t)j * CWG * kBxPair + rest] = 0ull;
    }
}

// ---------------------------------------------------------------------------------------------------------
// bf16x3 backward (H <= 256): the schedule of lstm_bwd_cluster_kernel<4> with fp32-grade products
//   dG W^T ~ dG_hi W_hi + dG_lo W_hi + dG_hi W_lo.
// The hi and lo fragments of a workgroup's 256 rows of W_hid^T (2 x 128 KB) fill the registers of its 8 waves exactly once:
// for the product, wave w takes destination w >> 1 and the unit-tile pair 2 (w & 1) of that destination's 64 units, for
// both 16-row tiles (2 tiles x 8 k-steps x (hi + lo) = 128 VGPRs).  LDS holds only the hi / lo images of the own
// dG_{t+1} and the own quarter of the partial dh.  Exchange, tags, inbox hygiene, gate math and the sums are those of
// the bf16 kernel (partials travel as fp32 with 4 tag bits each: 2^-20, finer than the products); no bf16 shadow of dG
// is written.
// ---------------------------------------------------------------------------------------------------------
constexpr int kX3BwdLds = 3;                                           // lo k-steps of a wave's W fragments that live in LDS
constexpr int kX3BwdWOff = (2 * kCRows * kCDS * 2 + kCRows * (kCUnits + 1) * 4 + 15) / 16 * 16 / 2;   // bf16 elements
__global__ __launch_bounds__(512) void lstm_bwd_cluster_x3_kernel(const LstmClusterP L, const uint8_t* __restrict__ mask_arg,
                                                                  int B, int T_arg, int H, int ldh, int ldg, int* err) {
    using G = ClusterGeom<4>;
    constexpr int CWG = 4, HP = G::HP, NF = G::NB;
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    __bf16 (*dgs_hi)[kCDS] = reinterpret_cast<__bf16 (*)[kCDS]>(lds);                          // [32][kCDS] own dG_{t+1}, hi
    __bf16 (*dgs_lo)[kCDS] = reinterpret_cast<__bf16 (*)[kCDS]>(lds + kCRows * kCDS);          // ... lo
    float (*part)[kCUnits + 1] = reinterpret_cast<float (*)[kCUnits + 1]>(lds + 2 * kCRows * kCDS);   // [32][65]
    bf16x8* wl = reinterpret_cast<bf16x8*>(lds + kX3BwdWOff);            // [8 waves][2 tiles][kX3BwdLds k-steps][64 lanes] lo fragments
    const int pair_ = L.pair0 + (int)blockIdx.x / CWG, lstm_ = pair_ / L.groups;
    const LstmStep& P = L.l[lstm_];
    // (a launch entry may be one LENGTH BUCKET of an LSTM -- model.hip, TmPlan: its own step count and its own rows of the mask)
    const int T = P.T_own ? P.T_own : T_arg;
    const uint8_t* __restrict__ mask_tb = P.mask_own ? P.mask_own : mask_arg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int group = pair_ - lstm_ * L.groups, j = blockIdx.x % CWG;
    const int r0 = group * kCRows;
    const int rt = wave >> 2, ut = wave & 3;          // gate math: row tile, local unit tile
    const int dst = wave >> 1, tp = 2 * (wave & 1);   // product: destination workgroup, first of its two unit tiles
    const int ul = 16 * ut + i;
    const int u = kCUnits * j + ul;
    unsigned long long* xb = reinterpret_cast<unsigned long long*>(P.xchg) + (size_t)((B + kCRows - 1) / kCRows) * 2 * 16 * HP +
                             (size_t)group * 2 * CWG * CWG * kBxPair;

    // resident W slice: k-steps [8j, 8j+8) (this workgroup's gate columns) of unit tiles 4 dst + tp + {0, 1}
    const bf16x8* wsrc_hi = reinterpret_cast<const bf16x8*>(P.W_frag_bwd) + (size_t)(8 * j) * 64;
    const bf16x8* wsrc_lo = reinterpret_cast<const bf16x8*>(P.W_frag_bwd_lo) + (size_t)(8 * j) * 64;
    // (the last kX3BwdLds lo k-steps of a wave sit in LDS: with all 128 fragment registers taken the step's other values spill)
    constexpr int KR = 8 - kX3BwdLds;
    bf16x8 whi[2][8], wlo[2][KR];
    bf16x8* wmine = wl + (size_t)wave * 2 * kX3BwdLds * 64 + lane;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            whi[c][s_] = wsrc_hi[((size_t)(4 * dst + tp + c) * (4 * G::KS) + s_) * 64 + lane];
            const bf16x8 lo = wsrc_lo[((size_t)(4 * dst + tp + c) * (4 * G::KS) + s_) * 64 + lane];
            if (s_ < KR) wlo[c][s_ < KR ? s_ : 0] = lo;
            else wmine[(c * kX3BwdLds + (s_ - KR)) * 64] = lo;
        }
    for (int e = tid; e < 2 * kCRows * kCDS / 8; e += 512) reinterpret_cast<bf16x8*>(lds)[e] = bf16x8{};
    float dh_c[4], dc_s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { dh_c[r] = 0.f; dc_s[r] = 0.f; }
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    const int uc = min(u, H - 1);
    // per-lane element offsets of its 4 rows inside one time block (32 bits: the block bases are wave-uniform)
    const int ld_dhs = P.ld_dhs ? P.ld_dhs : ldh;
    unsigned rowc[4], goff[4], hoff[4], doff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rowc[r] = (unsigned)min(r0 + 16 * rt + 4 * kq + r, B - 1);
        goff[r] = rowc[r] * (unsigned)ldg + (unsigned)uc * 4u;
        hoff[r] = rowc[r] * (unsigned)ldh + (unsigned)uc;
        doff[r] = rowc[r] * (unsigned)ld_dhs + (unsigned)uc;
    }
    float l_dhs[4], l_ct[4], l_cp[4];
    float4 l_gt[4];
    uint8_t l_m[4];
    auto request_state = [&](int step_) {
        const int t_ = P.backwards ? step_ : (T - 1 - step_);
        const int pb = t_ + (P.backwards ? 1 : 0), ob = t_ + (P.backwards ? 0 : 1);
        const uint8_t* mk = mask_tb + (size_t)t_ * B;
        const float* dh_ = P.dhs + (size_t)t_ * B * ld_dhs;
        const float* gt_ = P.gates + (size_t)t_ * B * ldg;
        const float* co_ = P.cbuf + (size_t)ob * B * ldh;
        const float* cp_ = P.cbuf + (size_t)pb * B * ldh;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            l_m[r] = mk[rowc[r]];
            l_dhs[r] = dh_[doff[r]];
            l_gt[r] = *reinterpret_cast<const float4*>(gt_ + goff[r]);
            l_ct[r] = co_[hoff[r]];
            l_cp[r] = cp_[hoff[r]];
        }
    };
    for (int step = 0; step <= T; ++step) {
        const int t = P.backwards ? step : (T - 1 - step);
        const unsigned tag8 = 1u + (unsigned)(step % 255);
        unsigned long long* xpar = xb + (size_t)(step & 1) * CWG * CWG * kBxPair;
        float rec[4] = {0.f, 0.f, 0.f, 0.f};
        if (step > 0) {
            // ---- partial dh for destination dst, unit tiles tp, tp + 1, both row tiles
            f32x4 acc[2][2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[c][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                bf16x8 a_hi[2], a_lo[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    a_hi[q] = *reinterpret_cast<const bf16x8*>(&dgs_hi[16 * q + i][s * 32 + kq * 8]);
                    a_lo[q] = *reinterpret_cast<const bf16x8*>(&dgs_lo[16 * q + i][s * 32 + kq * 8]);
                }
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc[c][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[q], whi[c][s], acc[c][q], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc[c][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[q], whi[c][s], acc[c][q], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        acc[c][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            a_hi[q], s < KR ? wlo[c][s < KR ? s : 0] : wmine[(c * kX3BwdLds + (s < KR ? 0 : s - KR)) * 64], acc[c][q], 0, 0, 0);
            }
            // accumulator map: unit = 64 dst + 16 (tp + c) + (lane & 15), row = 16 q + 4 kq + r
            if (dst == j) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int r = 0; r < 4; ++r) part[16 * q + 4 * kq + r][16 * (tp + c) + i] = acc[c][q][r];
            } else {
                unsigned long long* box = xpar + (size_t)(dst * CWG + (j - dst - 1 + CWG) % CWG) * kBxPair;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int rp = 0; rp < 2; ++rp)
                            __hip_atomic_store(box + (size_t)(8 * q + 2 * kq + rp) * kCUnits + 16 * (tp + c) + i,
                                               pack_partials(acc[c][q][2 * rp], acc[c][q][2 * rp + 1], tag8), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // what the gate math reads from HBM is requested AFTER the product (its 128 fragment registers leave no room for
        // 32 more values in flight); the round trips hide under the exchange hop
        request_state(min(step, T - 1));
        if (step > 0) {
            lds_barrier();                            // own part is in `part`; dG_{t+1} has been consumed
            // ---- collect the 3 foreign parts of this lane's pairs
            const unsigned long long* ptr[NF];
            unsigned long long g[NF];
#pragma unroll
            for (int k = 0; k < NF; ++k) {
                const int slot = k / 2, rp = k & 1;        // source (j + 1 + slot) mod CWG
                ptr[k] = xpar + (size_t)(j * CWG + slot) * kBxPair + (size_t)(8 * rt + 2 * kq + rp) * kCUnits + ul;
            }
            unsigned pending = (1u << NF) - 1u;
            unsigned long long t_start = 0;
            for (int spin = 0; pending; ++spin) {
                unsigned long long v[NF];
#pragma unroll
                for (int k = 0; k < NF; ++k)
                    if (pending & (1u << k)) v[k] = granule_load(ptr[k]);
#pragma unroll
                for (int k = 0; k < NF; ++k)
                    if ((pending & (1u << k)) && (((unsigned)v[k] & 15u) | (((unsigned)(v[k] >> 32) & 15u) << 4)) == tag8) {
                        g[k] = v[k]; pending &= ~(1u << k);
                    }
                if (pending && (spin & 1023) == 1023) {
                    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    const unsigned long long now = wall_ticks();
                    if (!t_start) t_start = now;
                    else if (now - t_start > kPollTimeoutTicks) { atomicCAS(err, 0, 2 | (step << 4) | ((int)blockIdx.x << 16)); break; }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) rec[r] = part[16 * rt + 4 * kq + r][ul];
#pragma unroll
            for (int k = 0; k < NF; ++k) {
                const int rp = k & 1;
                rec[2 * rp] += __builtin_bit_cast(float, (unsigned)g[k] & ~15u);
                rec[2 * rp + 1] += __builtin_bit_cast(float, (unsigned)(g[k] >> 32) & ~15u);
            }
        }
        if (step == T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dh_c[r] += rec[r];
            break;
        }
        // ---- gate math of step t for this lane's unit and 4 rows (fp32; see lstm.hip)
        float pw_i = 0.f, pw_f = 0.f, pw_o = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * rt + 4 * kq + r, grow = r0 + row;
            const bool ok = grow < B && u < H;
            float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                const float dh = l_dhs[r] + dh_c[r] + rec[r];
                const float dc = dc_s[r];
                if (l_m[r]) {
                    const float4 gt = l_gt[r];
                    const float c_t = l_ct[r], c_prev = l_cp[r];
                    const float tc = c_tanh(c_t);
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
                    dg = make_float4(c_clip5(da_i), c_clip5(da_f), c_clip5(da_g), c_clip5(da_o));
                    dh_c[r] = 0.f;
                    dc_s[r] = dcp;
                } else {
                    dh_c[r] = dh;
                }
                if (!P.dG16lo) *reinterpret_cast<float4*>(P.dG + (size_t)t * B * ldg + goff[r]) = dg;      // (ok: goff is this row and unit)
                bsum.x += dg.x; bsum.y += dg.y; bsum.z += dg.z; bsum.w += dg.w;
            }
            bf16x4 dhi, dlo;
            dhi[0] = (__bf16)dg.x; dhi[1] = (__bf16)dg.y; dhi[2] = (__bf16)dg.z; dhi[3] = (__bf16)dg.w;
            dlo[0] = (__bf16)(dg.x - (float)dhi[0]); dlo[1] = (__bf16)(dg.y - (float)dhi[1]);
            dlo[2] = (__bf16)(dg.z - (float)dhi[2]); dlo[3] = (__bf16)(dg.w - (float)dhi[3]);
            if (ok && P.dG16lo) {                     // the two planes in place of the fp32 row piece: the same 16 bytes per (row, unit)
                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(P.dG16) + (size_t)t * B * ldg + goff[r]) = dhi;
                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(P.dG16lo) + (size_t)t * B * ldg + goff[r]) = dlo;
            }
            *reinterpret_cast<bf16x4*>(&dgs_hi[row][ul * 4]) = dhi;
            *reinterpret_cast<bf16x4*>(&dgs_lo[row][ul * 4]) = dlo;

        }
        if (P.dpeep_part) {
            float si = pw_i, sf = pw_f, so = pw_o;
            si += __shfl_xor(si, 16, 64); si += __shfl_xor(si, 32, 64);
            sf += __shfl_xor(sf, 16, 64); sf += __shfl_xor(sf, 32, 64);
            so += __shfl_xor(so, 16, 64); so += __shfl_xor(so, 32, 64);
            if (kq == 0 && u < H) {
                float* ds = P.det_ws ? P.det_ws + (size_t)(2 * group + rt) * P.det_stride + ldg + 2 * ldh : nullptr;
                group_sum_add(P.dpeep_part + u, ds ? ds + u : nullptr, si);
                group_sum_add(P.dpeep_part + ldh + u, ds ? ds + ldh + u : nullptr, sf);
                group_sum_add(P.dpeep_part + 2 * (size_t)ldh + u, ds ? ds + 2 * ldh + u : nullptr, so);
            }
        }
        lds_barrier();
    }
    float sh = 0.f, sc = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int grow = r0 + 16 * rt + 4 * kq + r;
        if (grow < B && u < H) {
            P.dh_carry[(size_t)grow * ldh + u] = dh_c[r];
            P.dc_state[(size_t)grow * ldh + u] = dc_s[r];
            sh += dh_c[r]; sc += dc_s[r];
        }
    }
    if (P.dbias) {
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
            bsum.x += __shfl_xor(bsum.x, o, 64); bsum.y += __shfl_xor(bsum.y, o, 64);
            bsum.z += __shfl_xor(bsum.z, o, 64); bsum.w += __shfl_xor(bsum.w, o, 64);
            sh += __shfl_xor(sh, o, 64); sc += __shfl_xor(sc, o, 64);
        }
        if (kq == 0 && u < H) {
            float* ds = P.det_ws ? P.det_ws + (size_t)(2 * group + rt) * P.det_stride : nullptr;
            group_sum_add(P.dbias + 4 * u, ds ? ds + 4 * u : nullptr, bsum.x); group_sum_add(P.dbias + 4 * u + 1, ds ? ds + 4 * u + 1 : nullptr, bsum.y);
            group_sum_add(P.dbias + 4 * u + 2, ds ? ds + 4 * u + 2 : nullptr, bsum.z); group_sum_add(P.dbias + 4 * u + 3, ds ? ds + 4 * u + 3 : nullptr, bsum.w);
            group_sum_add(P.dhid_init + u, ds ? ds + ldg + u : nullptr, sh); group_sum_add(P.dcell_init + u, ds ? ds + ldg + ldh + u : nullptr, sc);
        }
    }
    __syncthreads();
    for (int e = tid; e < 2 * CWG * kBxPair; e += 512) {
        const int par = e / (CWG * kBxPair), rest = e % (CWG * kBxPair);
        xb[(size_t)par * CWG * CWG * kBxPair + (size_t)j * CWG * kBxPair + rest] = 0ull;
    }
}

// ---------------------------------------------------------------------------------------------------------
// bf16x3 backward, 256 < H <= 512 ("wide"): the geometry of lstm_fwd_cluster_x3w_kernel (16 workgroups of 32 units per
// 48-utterance group).  Workgroup j holds the hi + lo fragments of ITS 128 rows of W_hid^T (its gate columns: 4 k-steps) for all
// 512 units (2 x 128 KB: wave w takes unit tiles 4 w ..+3 = destinations 2 w, 2 w + 1; 12 of its 16 lo fragments sit in LDS),
// multiplies its own dG_{t+1} (48 x 128, hi / lo images in LDS) into a partial dh for all units, keeps its 32 and sends the
// other 15 x (48 x 32) to their owners as tagged granules (2 rows of a unit; pack_partials), two granules -- the lane's 4 rows --
// per 16-byte write-through store.  Inbox of a workgroup: [2 parities][16 source slots][12 row quads][32 units][2 granules];
// ALL 512 threads collect: thread = (128 positions x 3 rounds, source quarter), one 16-byte sc1 buffer load per (position,
// source), each half validated by its own tag; the <= 4 sources of a quarter are summed in slot order and the four quarter
// sums handed to the gate-math lanes through LDS (fixed order: deterministic).
// ---------------------------------------------------------------------------------------------------------
constexpr int kWDS = 4 * kWUnits + 8;                                  // LDS row stride of the dG image (own 128 gate columns)
constexpr int kWPair = kWRows / 2 * kWUnits;                           // granules of one (destination, source) pair
constexpr int kWBwdLds = 3;                                            // lo k-steps (of a wave's 4) whose fragments live in LDS
constexpr int kWBwdWOff = (2 * kWRows * kWDS * 2 + 5 * kWRows * (kWUnits + 1) * 4 + 15) / 16 * 16 / 2;   // bf16 elements
constexpr size_t kWBwdLdsBytes = (size_t)kWBwdWOff * 2 + (size_t)8 * 4 * kWBwdLds * 64 * 16;
__global__ __launch_bounds__(512) void lstm_bwd_cluster_x3w_kernel(const LstmClusterP L, const uint8_t* __restrict__ mask_tb,
                                                                   int B, int T, int H, int ldh, int ldg, int* err) {
    constexpr int R = kWRows, CWG = kWCWG, HP = kWHP, NQ = 4;
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    __bf16 (*dgs_hi)[kWDS] = reinterpret_cast<__bf16 (*)[kWDS]>(lds);                            // [48][kWDS] own dG_{t+1}, hi
    __bf16 (*dgs_lo)[kWDS] = reinterpret_cast<__bf16 (*)[kWDS]>(lds + R * kWDS);                 // ... lo
    float (*part)[kWUnits + 1] = reinterpret_cast<float (*)[kWUnits + 1]>(lds + 2 * R * kWDS);  // [48][33] own share of the own partial
    float (*recv)[kWUnits + 1] = part + R;                                                       // [4 source quarters x 48][33] sums of the foreign shares
    bf16x8* wl = reinterpret_cast<bf16x8*>(lds + kWBwdWOff);             // [8 waves][4 tiles][kWBwdLds k-steps][64 lanes] lo fragments
    const int pair_ = L.pair0 + (int)blockIdx.x / CWG, lstm_ = pair_ / L.groups;
    const LstmStep& P = L.l[lstm_];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int group = pair_ - lstm_ * L.groups, j = blockIdx.x % CWG;
    const int r0 = group * R;
    const int rt = wave >> 1, ut = wave & 1;          // gate math: row tile (3: none), local unit tile
    const bool gm = rt < 3;
    const int ul = 16 * ut + i;
    const int u = kWUnits * j + ul;
    unsigned long long* xb = reinterpret_cast<unsigned long long*>(P.xchg) + (size_t)((B + R - 1) / R) * 2 * (R / 2) * HP +
                             (size_t)group * 2 * CWG * CWG * kWPair;

    // resident W slice: k-steps [4j, 4j+4) (this workgroup's gate columns) of unit tiles 4 wave + {0..3}
    const bf16x8* wsrc_hi = reinterpret_cast<const bf16x8*>(P.W_frag_bwd) + (size_t)(4 * j) * 64 + lane;
    const bf16x8* wsrc_lo = reinterpret_cast<const bf16x8*>(P.W_frag_bwd_lo) + (size_t)(4 * j) * 64 + lane;
    constexpr int KR = 4 - kWBwdLds;
    bf16x8 whi[4][4], wlo[4][KR];
    bf16x8* wmine = wl + (size_t)wave * 4 * kWBwdLds * 64 + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            whi[c][s_] = wsrc_hi[((size_t)(4 * wave + c) * (HP / 8) + s_) * 64];
            const bf16x8 lo = wsrc_lo[((size_t)(4 * wave + c) * (HP / 8) + s_) * 64];
            if (s_ < KR) wlo[c][s_ < KR ? s_ : 0] = lo;
            else wmine[(c * kWBwdLds + (s_ - KR)) * 64] = lo;
        }
    for (int e = tid; e < 2 * R * kWDS / 8; e += 512) reinterpret_cast<bf16x8*>(lds)[e] = bf16x8{};
    float dh_c[4], dc_s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { dh_c[r] = 0.f; dc_s[r] = 0.f; }
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    const int uc = min(u, H - 1);
    float pk_i = 0.f, pk_f = 0.f, pk_o = 0.f;          // peephole weights of this lane's unit
    if (P.peep) { pk_i = P.peep[uc]; pk_f = P.peep[ldh + uc]; pk_o = P.peep[2 * ldh + uc]; }
    // (row and unit indices are made opaque once per use: the compiler otherwise keeps a 64-bit offset per row and array live
    //  across the loop and spills them, and the loaded values with them)
    const int ld_dhs = P.ld_dhs ? P.ld_dhs : ldh;
    const int rowb = r0 + 16 * min(rt, 2) + 4 * kq;
    float l_dhs[4], l_ct[4], l_cp[4];
    float4 l_gt[4];
    uint8_t l_m[4];
    auto request_state = [&](int step_) {
        const int t_ = P.backwards ? step_ : (T - 1 - step_);
        const int pb = t_ + (P.backwards ? 1 : 0), ob = t_ + (P.backwards ? 0 : 1);
        const uint8_t* mk = mask_tb + (size_t)t_ * B;
        const float* dh_ = P.dhs + (size_t)t_ * B * ld_dhs;
        const float* gt_ = P.gates + (size_t)t_ * B * ldg;
        const float* co_ = P.cbuf + (size_t)ob * B * ldh;
        const float* cp_ = P.cbuf + (size_t)pb * B * ldh;
        int rb = rowb, ucl = uc;
        asm volatile("" : "+v"(rb), "+v"(ucl));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned rc = (unsigned)min(rb + r, B - 1);
            l_m[r] = mk[rc];
            l_dhs[r] = dh_[rc * (unsigned)ld_dhs + (unsigned)ucl];
            l_gt[r] = *reinterpret_cast<const float4*>(gt_ + (rc * (unsigned)ldg + (unsigned)ucl * 4u));
            l_ct[r] = co_[rc * (unsigned)ldh + (unsigned)ucl];
            l_cp[r] = cp_[rc * (unsigned)ldh + (unsigned)ucl];
        }
    };
    const int pb = tid & 127, sq = tid >> 7;           // collect: position block, source quarter
    const unsigned send_off = (unsigned)((kq * kWUnits + i) * 16);     // sends: lane part of the byte offset inside a (dst, slot) block
    STAMP_INIT
    for (int step = 0; step <= T; ++step) {
        const int t = P.backwards ? step : (T - 1 - step);
        const unsigned tag8 = 1u + (unsigned)(step % 255);
        unsigned long long* xpar = xb + (size_t)(step & 1) * CWG * CWG * kWPair;
        // the exchange runs on sc1 (write-through / L1-bypassing) buffer accesses: 32-bit lane offsets, no address registers
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(xpar, 0, CWG * CWG * kWPair * 8, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(xpar + (size_t)j * CWG * kWPair, 0, CWG * kWPair * 8, 0x00020000);
        float rec[4] = {0.f, 0.f, 0.f, 0.f};
        if (step > 0) {
            // ---- partial dh of unit tiles 4 wave ..+3 (destinations 2 wave, 2 wave + 1), one row tile after the other (the
            //      accumulators of all three at once leave no room: 48 registers on top of the 80 of the fragments)
            const bf16x8* wm = wmine;                 // (opaque: the fragments in LDS are read where they are used, not hoisted
            asm volatile("" : "+v"(wm));              //  into 48 more registers ahead of the loop)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                f32x4 acc[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bf16x8 a_hi = *reinterpret_cast<const bf16x8*>(&dgs_hi[16 * q + i][s * 32 + kq * 8]);
                    const bf16x8 a_lo = *reinterpret_cast<const bf16x8*>(&dgs_lo[16 * q + i][s * 32 + kq * 8]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, whi[c][s], acc[c], 0, 0, 0);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, whi[c][s], acc[c], 0, 0, 0);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            a_hi, s < KR ? wlo[c][s < KR ? s : 0] : wm[(c * kWBwdLds + (s < KR ? 0 : s - KR)) * 64], acc[c], 0, 0, 0);
                }
                // accumulator map: unit = 16 (4 wave + c) + (lane & 15), row = 16 q + 4 kq + r
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int dst = 2 * wave + (c >> 1), tl = c & 1;
                    if (dst == j) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) part[16 * q + 4 * kq + r][16 * tl + i] = acc[c][r];
                    } else {
                        // block (dst, slot): [12 quads][32 units][2 granules]; this lane's 4 rows are quad 4 q + kq
                        const unsigned blk = (unsigned)__builtin_amdgcn_readfirstlane((dst * CWG + (j - dst - 1 + CWG) % CWG) * (kWPair * 8));
                        const unsigned long long g0 = pack_partials(acc[c][0], acc[c][1], tag8), g1 = pack_partials(acc[c][2], acc[c][3], tag8);
                        const u32x4 pk = {(unsigned)g0, (unsigned)(g0 >> 32), (unsigned)g1, (unsigned)(g1 >> 32)};
                        __builtin_amdgcn_raw_buffer_store_b128(pk, rs_out, send_off + (unsigned)((4 * q * kWUnits + 16 * tl) * 16), blk, 16);
                    }
                }
            }
        }
        STAMP(5);
        // what the gate math reads from HBM is requested AFTER the product; the round trips hide under the exchange hop
        if (gm) request_state(min(step, T - 1));
        if (step > 0) {
            // ---- collect: position (quad, unit) = pb + 128 round, sources 4 sq .. 4 sq + 3 (slot 15 does not exist)
#pragma unroll 1
            for (int rd = 0; rd < 3; ++rd) {
                const int pos = pb + 128 * rd;
                const unsigned voff = (unsigned)(pos * 16 + sq * NQ * (kWPair * 8));
                u32x4 v[NQ];
                unsigned pending = sq == 3 ? 0x7u : 0xfu;
                unsigned long long t_start = 0;
                for (int spin = 0; pending; ++spin) {
#pragma unroll
                    for (int k = 0; k < NQ; ++k)
                        if (pending & (1u << k)) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, k * (kWPair * 8), 16);
#pragma unroll
                    for (int k = 0; k < NQ; ++k)
                        if ((pending & (1u << k)) && ((v[k].x & 15u) | ((v[k].y & 15u) << 4)) == tag8 && ((v[k].z & 15u) | ((v[k].w & 15u) << 4)) == tag8)
                            pending &= ~(1u << k);
                    if (pending && (spin & 1023) == 1023) {
                        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                        const unsigned long long now = wall_ticks();
                        if (!t_start) t_start = now;
                        else if (now - t_start > kPollTimeoutTicks) { atomicCAS(err, 0, 2 | (step << 4) | ((int)blockIdx.x << 16)); break; }
                    }
                }
                float sr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < NQ; ++k)
                    if (k < 3 || sq != 3) {
                        sr[0] += __builtin_bit_cast(float, v[k].x & ~15u); sr[1] += __builtin_bit_cast(float, v[k].y & ~15u);
                        sr[2] += __builtin_bit_cast(float, v[k].z & ~15u); sr[3] += __builtin_bit_cast(float, v[k].w & ~15u);
                    }
                float* d = &recv[R * sq + 4 * (pos >> 5)][pos & 31];
#pragma unroll
                for (int r = 0; r < 4; ++r) d[r * (kWUnits + 1)] = sr[r];
            }
            STAMP(6);
            lds_barrier();                            // own share in `part`, the foreign sums in `recv`; dG_{t+1} has been consumed
            if (gm) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * rt + 4 * kq + r;
                    rec[r] = part[row][ul] + recv[row][ul] + recv[R + row][ul] + recv[2 * R + row][ul] + recv[3 * R + row][ul];
                }
            }
        }
        if (step == T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dh_c[r] += rec[r];
            break;
        }
        // ---- gate math of step t for this lane's unit and 4 rows (fp32; see lstm.hip)
        if (gm) {
            float pw_i = 0.f, pw_f = 0.f, pw_o = 0.f;
            int rb = rowb, ucl = uc;
            asm volatile("" : "+v"(rb), "+v"(ucl));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rt + 4 * kq + r, grow = rb + r;
                const bool ok = grow < B && u < H;
                const unsigned goff = (unsigned)grow * (unsigned)ldg + (unsigned)ucl * 4u;       // (used where ok: this row and unit)
                float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) {
                    const float dh = l_dhs[r] + dh_c[r] + rec[r];
                    const float dc = dc_s[r];
                    if (l_m[r]) {
                        const float4 gt = l_gt[r];
                        const float c_t = l_ct[r], c_prev = l_cp[r];
                        const float tc = c_tanh(c_t);
                        const float da_o = dh * tc * gt.w * (1.f - gt.w);
                        float dcn = dc + dh * gt.w * (1.f - tc * tc);
                        if (P.peep) { dcn += da_o * pk_o; pw_o += da_o * c_t; }
                        const float da_i = dcn * gt.z * gt.x * (1.f - gt.x);
                        const float da_f = dcn * c_prev * gt.y * (1.f - gt.y);
                        const float da_g = dcn * gt.x * (1.f - gt.z * gt.z);
                        float dcp = dcn * gt.y;
                        if (P.peep) {
                            dcp += da_i * pk_i + da_f * pk_f;
                            pw_i += da_i * c_prev; pw_f += da_f * c_prev;
                        }
                        dg = make_float4(c_clip5(da_i), c_clip5(da_f), c_clip5(da_g), c_clip5(da_o));
                        dh_c[r] = 0.f;
                        dc_s[r] = dcp;
                    } else {
                        dh_c[r] = dh;
                    }
                    if (!P.dG16lo) *reinterpret_cast<float4*>(P.dG + (size_t)t * B * ldg + goff) = dg;      // (ok: goff is this row and unit)
                    bsum.x += dg.x; bsum.y += dg.y; bsum.z += dg.z; bsum.w += dg.w;
                }
                bf16x4 dhi, dlo;
                dhi[0] = (__bf16)dg.x; dhi[1] = (__bf16)dg.y; dhi[2] = (__bf16)dg.z; dhi[3] = (__bf16)dg.w;
                dlo[0] = (__bf16)(dg.x - (float)dhi[0]); dlo[1] = (__bf16)(dg.y - (float)dhi[1]);
                dlo[2] = (__bf16)(dg.z - (float)dhi[2]); dlo[3] = (__bf16)(dg.w - (float)dhi[3]);
                if (ok && P.dG16lo) {                     // the two planes in place of the fp32 row piece: the same 16 bytes per (row, unit)
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(P.dG16) + (size_t)t * B * ldg + goff) = dhi;
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(P.dG16lo) + (size_t)t * B * ldg + goff) = dlo;
                }
                *reinterpret_cast<bf16x4*>(&dgs_hi[row][ul * 4]) = dhi;
                *reinterpret_cast<bf16x4*>(&dgs_lo[row][ul * 4]) = dlo;
            }
            if (P.dpeep_part) {
                float si = pw_i, sf = pw_f, so = pw_o;
                si += __shfl_xor(si, 16, 64); si += __shfl_xor(si, 32, 64);
                sf += __shfl_xor(sf, 16, 64); sf += __shfl_xor(sf, 32, 64);
                so += __shfl_xor(so, 16, 64); so += __shfl_xor(so, 32, 64);
                if (kq == 0 && u < H) {
                    float* ds = P.det_ws ? P.det_ws + (size_t)(3 * group + rt) * P.det_stride + ldg + 2 * ldh : nullptr;
                    group_sum_add(P.dpeep_part + u, ds ? ds + u : nullptr, si);
                    group_sum_add(P.dpeep_part + ldh + u, ds ? ds + ldh + u : nullptr, sf);
                    group_sum_add(P.dpeep_part + 2 * (size_t)ldh + u, ds ? ds + 2 * ldh + u : nullptr, so);
                }
            }
        }
        lds_barrier();
        STAMP(7);
    }
    if (gm) {
        float sh = 0.f, sc = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int grow = r0 + 16 * rt + 4 * kq + r;
            if (grow < B && u < H) {
                P.dh_carry[(size_t)grow * ldh + u] = dh_c[r];
                P.dc_state[(size_t)grow * ldh + u] = dc_s[r];
                sh += dh_c[r]; sc += dc_s[r];
            }
        }
        if (P.dbias) {
#pragma unroll
            for (int o = 16; o < 64; o <<= 1) {
                bsum.x += __shfl_xor(bsum.x, o, 64); bsum.y += __shfl_xor(bsum.y, o, 64);
                bsum.z += __shfl_xor(bsum.z, o, 64); bsum.w += __shfl_xor(bsum.w, o, 64);
                sh += __shfl_xor(sh, o, 64); sc += __shfl_xor(sc, o, 64);
            }
            if (kq == 0 && u < H) {
                float* ds = P.det_ws ? P.det_ws + (size_t)(3 * group + rt) * P.det_stride : nullptr;
                group_sum_add(P.dbias + 4 * u, ds ? ds + 4 * u : nullptr, bsum.x); group_sum_add(P.dbias + 4 * u + 1, ds ? ds + 4 * u + 1 : nullptr, bsum.y);
                group_sum_add(P.dbias + 4 * u + 2, ds ? ds + 4 * u + 2 : nullptr, bsum.z); group_sum_add(P.dbias + 4 * u + 3, ds ? ds + 4 * u + 3 : nullptr, bsum.w);
                group_sum_add(P.dhid_init + u, ds ? ds + ldg + u : nullptr, sh); group_sum_add(P.dcell_init + u, ds ? ds + ldg + ldh + u : nullptr, sc);
            }
        }
    }
    // leave this workgroup's inbox empty for the next launch (after EVERY lane has taken its last granules)
    __syncthreads();
    for (int e = tid; e < 2 * CWG * kWPair; e += 512) {
        const int par = e / (CWG * kWPair), rest = e % (CWG * kWPair);
        xb[(size_t)par * CWG * CWG * kWPair + (size_t)j * CWG * kWPair + rest] = 0ull;
    }
}

// per-device launcher state (one process may drive several devices: a model polls the error word of ITS device and
// sizes its launches by ITS device's CU count)
constexpr int kMaxDevices = 64;
static unsigned g_cluster_epoch = 1;       // launch tag (any strictly increasing sequence works; shared on purpose)
static int* g_cluster_err[kMaxDevices] = {};   // device words, lazily allocated; polled after the launch by the caller's sync
static int g_cluster_cus[kMaxDevices] = {};

static int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 0;
    return dev;
}

static int cluster_cus() {
    const int dev = current_device();
    if (!g_cluster_cus[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        g_cluster_cus[dev] = prop.multiProcessorCount;
        // ADN_LSTM_CUS=n: size the launches for n CUs (a partitioned device / a CU mask the runtime does not report; the
        // tests use it to drive an LSTM onto the one-workgroup kernels the way a too-small device would)
        if (const char* e = getenv("ADN_LSTM_CUS")) {
            const int n = atoi(e);
            if (n > 0 && n < g_cluster_cus[dev]) g_cluster_cus[dev] = n;
        }
    }
    return g_cluster_cus[dev];
}

int lstm_cluster_cus() { return cluster_cus(); }

// Residency guard.  The weight-stationary kernels are plain launches of one 512-thread workgroup per CU whose workgroups
// wait for their group's partners.  What makes that safe is checked here instead of assumed: (1) the runtime's occupancy
// answer for the kernel with its dynamic LDS is asked once per device and kernel -- a workgroup that cannot become resident
// at all (a driver that reserves LDS, a smaller register file) turns the kernel family off and the LSTM runs on the
// one-workgroup kernels; (2) groups are independent and their workgroups have consecutive blockIdx.x, so with in-order
// dispatch a launch makes progress on any number >= CWG of free CUs: a foreign kernel that holds CUs (a profiler's, another
// tenant's, a collective's) delays the launch, group by group, and does not deadlock it
// (tests/test_gpu_residency.py occupies 240 of the CUs for 20 ms beside a forward / backward pass); (3) polls are
// bounded, so that even a dispatcher that breaks (2) ends in ADN_ERR_STATE, never in a hang.
template <auto Kernel>                             // (the kernel as a template argument: one cache per kernel, not per signature)
static bool cluster_kernel_fits(size_t lds_bytes) {
    static int cached[kMaxDevices] = {};           // 0 = not asked, 1 = fits, -1 = does not
    int& c = cached[current_device()];
    if (!c) {
        int blocks = 0;
        const void* f = reinterpret_cast<const void*>(Kernel);
        const bool ok = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) == hipSuccess &&
                        hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, f, 512, lds_bytes) == hipSuccess && blocks >= 1;
        if (!ok) (void)hipGetLastError();
        c = ok ? 1 : -1;
    }
    return c > 0;
}
template <int CWG, int KXS = 0> static size_t fwd_lds_bytes() {
    using G = ClusterGeom<CWG, KXS>;
    return (size_t)(G::WLdsFwd + kCRows * G::HS) * 2 + (size_t)16 * KXS * 64 * 16 + (KXS ? (size_t)2 * kCRows * (32 * KXS + 8) * 2 : 0);
}
template <int CWG> static size_t bwd_lds_bytes() {
    using G = ClusterGeom<CWG>;
    return (size_t)(G::WLdsBwd + kCRows * kCDS) * 2 + (size_t)kCRows * (kCUnits + 1) * 4;
}

static int cluster_wgs(int H) { return H <= 256 ? 4 : 8; }      // workgroups per group: 64 hidden units each

bool lstm_cluster_supported(const LstmStep* l, int n, int B, int T, int H) {
    if (H > 512 || T >= 1024 || getenv("ADN_LSTM_NO_CLUSTER")) return false;
    if (H > 256 && getenv("ADN_LSTM_NO_WIDE_CLUSTER")) return false;
    for (int k = 0; k < n; ++k)
        if (!l[k].xchg || !l[k].W_frag_fwd || !l[k].W_frag_bwd) return false;
    if (cdiv(B, kCRows) * cluster_wgs(H) > cluster_cus()) return false;     // every workgroup of one LSTM must be resident at once
    const size_t hp = (size_t)cluster_wgs(H) * kCUnits;
    if (lstm_frag_elems(H) != 4 * hp * hp) return false;
    return H <= 256 ? cluster_kernel_fits<&lstm_fwd_cluster_kernel<4, 0>>(fwd_lds_bytes<4>()) &&
                          cluster_kernel_fits<&lstm_bwd_cluster_kernel<4>>(bwd_lds_bytes<4>())
                    : cluster_kernel_fits<&lstm_fwd_cluster_kernel<8, 0>>(fwd_lds_bytes<8>()) &&
                          cluster_kernel_fits<&lstm_bwd_cluster_kernel<8>>(bwd_lds_bytes<8>());
}

size_t lstm_cluster_xchg_bytes(int B, int H) {   // forward region (h granules) + backward region (partial-dh granules)
    const int cwg = cluster_wgs(H);
    const size_t narrow = (size_t)cdiv(B, kCRows) * (2 * 16 * cwg * kCUnits + 2 * cwg * cwg * kBxPair) * 8;
    if (H <= 256) return narrow;
    // 256 < H <= 512 also serves the bf16x3 mode's wide kernels: 16 workgroups per 48-utterance group
    const size_t wide = (size_t)cdiv(B, kWRows) * (2 * (kWRows / 2) * kWHP + 2 * kWCWG * kWCWG * kWPair) * 8;
    return std::max(narrow, wide);
}

int lstm_cluster_error_word(int** out) {
    const int dev = current_device();
    if (!g_cluster_err[dev]) {
        ADN_HIP_CHECK(hipMalloc((void**)&g_cluster_err[dev], sizeof(int)));
        ADN_HIP_CHECK(hipMemset(g_cluster_err[dev], 0, sizeof(int)));
    }
    *out = g_cluster_err[dev];
    return ADN_OK;
}

// ---- folded input projection (KXS > 0 instantiations of the forward kernel) -------------------------------------------------
static int fold_ksteps(int Kx) { return Kx <= 0 ? 0 : (Kx <= 96 ? 3 : (Kx <= 160 ? 5 : 0)); }
template <int KXS> static size_t fwd_fold_lds_bytes() { return fwd_lds_bytes<4, KXS>(); }

size_t lstm_win_frag_elems(int Kx, int H) { return (size_t)(H <= 256 ? 256 : 512) / 16 * 4 * fold_ksteps(Kx) * 512; }

struct PackWinArgs { const float* W[8]; void* out[8]; };
__global__ __launch_bounds__(256) void pack_win_frags_kernel(const PackWinArgs a, int Kx, int H, int ldg, int kxs, int total) {
    const float* __restrict__ W = a.W[blockIdx.y];
    __bf16* __restrict__ out = reinterpret_cast<__bf16*>(a.out[blockIdx.y]);
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int q = e & 7, lane = (e >> 3) & 63, rest = e >> 9;
        const int s_ = rest % kxs, tile = rest / kxs;             // tile = 4 * (16-unit tile) + gate, like the W_hid image
        const int unit = 16 * (tile >> 2) + (lane & 15), col = 4 * unit + (tile & 3);
        const int k = 32 * s_ + 8 * (lane >> 4) + q;
        out[e] = (__bf16)((k < Kx && unit < H) ? W[(size_t)k * ldg + col] : 0.f);
    }
}
int lstm_pack_win_frags(int n, const float* const* W_in, void* const* out, int Kx, int H, hipStream_t s) {
    ADN_CHECK(n >= 1 && n <= 8 && fold_ksteps(Kx) > 0, ADN_ERR_INVALID, "lstm_pack_win_frags: 1..8 matrices of <= 160 input features");
    PackWinArgs a{};
    for (int k = 0; k < n; ++k) { a.W[k] = W_in[k]; a.out[k] = out[k]; }
    const int total = (int)lstm_win_frag_elems(Kx, H);
    hipLaunchKernelGGL(pack_win_frags_kernel, dim3(cdiv(total, 256 * 4), n), dim3(256), 0, s, a, Kx, H, ld_of(4 * H), fold_ksteps(Kx), total);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

// (measured on the bench model, profiles/r04: B = 520 -- projection GEMM 91 us gone, forward LSTM class +4 us: 3.563 -> 3.478 ms
//  per train step; B = 26 -- the GEMM is a 10 us launch there and the step's 20 extra MFMAs per wave sit on the exchange's
//  critical path: 1.246 -> 1.263 ms.  Folded from three groups on; ADN_LSTM_FOLD_MIN_B moves the threshold.)
static bool fold_offered(const LstmStep* l, int n, int H, int B) {
    const char* mb = getenv("ADN_LSTM_FOLD_MIN_B");          // (read per call: the tests move it)
    const int min_b = mb ? atoi(mb) : 65;
    if (H > 256 || B < min_b || getenv("ADN_LSTM_NO_FOLD")) return false;
    const int kxs = fold_ksteps(l[0].Kx);
    if (!kxs) return false;
    for (int k = 0; k < n; ++k)
        if (!l[k].x16 || !l[k].W_in_frag || !l[k].b_in || l[k].Kx != l[0].Kx || l[k].ld_x < 32 * kxs || (l[k].ld_x & 7)) return false;
    return kxs == 3 ? cluster_kernel_fits<&lstm_fwd_cluster_kernel<4, 3>>(fwd_fold_lds_bytes<3>())
                    : cluster_kernel_fits<&lstm_fwd_cluster_kernel<4, 5>>(fwd_fold_lds_bytes<5>());
}

bool lstm_forward_folds_projection(const LstmStep* l, int n, int B, int T, int H, int precision) {
    if (precision != ADN_PRECISION_BF16 || n < 1) return false;
    for (int k = 0; k < n; ++k)
        if (!l[k].W_hid16T || !l[k].h16) return false;            // (lstm_forward's own test for the bf16 kernels)
    return lstm_persistent_supported(H) && lstm_cluster_supported(l, n, B, T, H) && fold_offered(l, n, H, B);
}

// The profiler's units of a launch (SURVEY 8d prices one LSTM step as e(12 B H + 4 H^2) + B bytes and 8 B H^2 flops).  With length
// buckets an entry is a BUCKET of an LSTM: the row terms are booked per entry and per step it runs (row_steps, B = the entry's
// utterances), the W_hid term once per LSTM and step of its LONGEST bucket (weight_steps) -- the buckets of one LSTM are one
// recurrence of the reference, which reads W_hid once per step; booking it per entry would count it nb times.
struct EntrySteps { double rows, weights; };
static EntrySteps entry_steps(const LstmStep* l, int n, int T) {
    EntrySteps s{0.0, 0.0};
    for (int k = 0; k < n; ++k) {
        const int tk = l[k].T_own ? l[k].T_own : T;
        s.rows += tk;
        bool first = true; int tmax = tk;
        for (int j = 0; j < n; ++j)
            if (l[j].W_hid == l[k].W_hid) {
                if (j < k) first = false;
                tmax = std::max(tmax, l[j].T_own ? l[j].T_own : T);
            }
        if (first) s.weights += tmax;
    }
    return s;
}
// the (LSTM, group) pairs of a call over the fewest launches that keep every workgroup of a launch resident, in equal shares
struct PairRange { int pair0, count; };
static std::vector<PairRange> plan_pairs(int n, int groups, int cwg, int cus) {
    const int pairs = n * groups, cap = std::max(1, cus / cwg);
    const int launches = cdiv(pairs, cap);
    std::vector<PairRange> out;
    for (int k = 0, p0 = 0; k < launches; ++k) {
        const int cnt = pairs / launches + (k < pairs % launches ? 1 : 0);
        out.push_back({p0, cnt});
        p0 += cnt;
    }
    return out;
}

template <int CWG, int KXS>
static int forward_cluster(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, hipStream_t s) {
    const int groups = cdiv(B, kCRows), per = groups * CWG, cus = cluster_cus();
    ADN_CHECK(cus >= per, ADN_ERR_STATE, "lstm cluster kernel: one LSTM does not fit the device");
    int* err = nullptr;
    ADN_TRY(lstm_cluster_error_word(&err));
    const int ldh = ld_of(H), ldg = ld_of(4 * H);
    const size_t lds = fwd_lds_bytes<CWG, KXS>();
    static bool attr_set[kMaxDevices] = {};        // (a function attribute is per device)
    bool& attr = attr_set[current_device()];
    if (!attr) {
        ADN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_fwd_cluster_kernel<CWG, KXS>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    // (KXS > 0: the step also multiplies x_t W_in -- its flops are booked with the recurrence's, its bytes replace the xproj read)
    const EntrySteps es = entry_steps(l, n, T);       // (length buckets: every entry its own step count)
    const double bytes = es.rows * (4.0 * 12.0 * B * H + B) + es.weights * 16.0 * H * H,
                 flops = es.rows * (8.0 * B * H * H + (KXS ? 8.0 * B * H * l[0].Kx : 0.0));
    ProfScope prof(PROF_LSTM_FWD, flops, bytes, s, T);
    LstmClusterP L;
    for (int k = 0; k < n; ++k) L.l[k] = l[k];
    L.groups = groups;
    for (const PairRange& r : plan_pairs(n, groups, CWG, cus)) {      // every workgroup of a launch must be resident
        L.pair0 = r.pair0;
        const unsigned tag0 = (g_cluster_epoch++ & 0x3fffffu) * 1024u + 1u;
        hipLaunchKernelGGL((lstm_fwd_cluster_kernel<CWG, KXS>), dim3(r.count * CWG), dim3(512), lds, s, L, mask_tb, B, T, H, ldh, ldg, tag0, err);
        ADN_HIP_CHECK(hipGetLastError());
    }
    return ADN_OK;
}

int lstm_forward_cluster(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, hipStream_t s) {
    if (H > 256) return forward_cluster<8, 0>(l, n, mask_tb, B, T, H, s);
    if (fold_offered(l, n, H, B))
        return fold_ksteps(l[0].Kx) == 3 ? forward_cluster<4, 3>(l, n, mask_tb, B, T, H, s) : forward_cluster<4, 5>(l, n, mask_tb, B, T, H, s);
    return forward_cluster<4, 0>(l, n, mask_tb, B, T, H, s);
}

// bf16x3 forward (lstm_fwd_cluster_x3_kernel): H <= 256, hi and lo fragment images of W_hid, the exchange buffer of the bf16
// kernels (its forward region needs 2 x 32 x 256 granules per group: less than the buffer's forward + backward regions)
bool lstm_cluster_x3_supported(const LstmStep* l, int n, int B, int T, int H) {
    if (H > 256 || T >= 1024 || getenv("ADN_LSTM_NO_CLUSTER") || getenv("ADN_LSTM_NO_X3_CLUSTER")) return false;
    for (int k = 0; k < n; ++k)
        if (!l[k].xchg || !l[k].W_frag_fwd || !l[k].W_frag_fwd_lo) return false;
    if (cdiv(B, kCRows) * 4 > cluster_cus()) return false;
    if (lstm_frag_elems(H) != (size_t)4 * 256 * 256) return false;
    using G = ClusterGeom<4>;
    return cluster_kernel_fits<&lstm_fwd_cluster_x3_kernel>((size_t)2 * kCRows * G::HS * 2 + (size_t)kX3AccLds * 4 + (size_t)8 * 2 * kX3FwdLds * 64 * 16);
}

// Launch number (mod 64) of an exchange buffer: the sequence bits of the x3 kernel's 16-bit tags.  The counter lives with the
// buffer (LstmStep::xchg_seq) and restarts at 0 when the buffer is carved; the scheme relies on the owner ZEROING the buffer
// at that point (model.hip: ensure_workspace memsets the slab), so that no slot can hold a tag of an earlier life of the
// memory -- tags start at 1, a zeroed slot never matches.
static int x3_launch_seq(const LstmStep& l, unsigned* seq) {
    ADN_CHECK(l.xchg && l.xchg_seq, ADN_ERR_INVALID, "bf16x3 LSTM kernel: no exchange buffer / launch counter");
    *seq = (*l.xchg_seq)++ & 63u;
    return ADN_OK;
}

int lstm_forward_cluster_x3(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, hipStream_t s) {
    using G = ClusterGeom<4>;
    const int groups = cdiv(B, kCRows), per = groups * 4, cus = cluster_cus();
    ADN_CHECK(cus >= per, ADN_ERR_STATE, "lstm cluster kernel: one LSTM does not fit the device");
    int* err = nullptr;
    ADN_TRY(lstm_cluster_error_word(&err));
    const int ldh = ld_of(H), ldg = ld_of(4 * H);
    const size_t lds = (size_t)2 * kCRows * G::HS * 2 + (size_t)kX3AccLds * 4 + (size_t)8 * 2 * kX3FwdLds * 64 * 16;
    static bool attr_set[kMaxDevices] = {};
    bool& attr = attr_set[current_device()];
    if (!attr) {
        ADN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_fwd_cluster_x3_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    const EntrySteps es = entry_steps(l, n, T);
    const double bytes = es.rows * (4.0 * 12.0 * B * H + B) + es.weights * 16.0 * H * H, flops = es.rows * 8.0 * B * H * H;
    ProfScope prof(PROF_LSTM_FWD, flops, bytes, s, T);
    LstmClusterX3P L;
    for (int k = 0; k < n; ++k) { L.l[k] = l[k]; L.tag0[k] = 0u; }
    L.groups = groups;
    for (const PairRange& r : plan_pairs(n, groups, 4, cus)) {
        L.pair0 = r.pair0;
        for (int k = r.pair0 / groups; k <= (r.pair0 + r.count - 1) / groups; ++k) {     // the LSTMs this launch touches: a new
            unsigned seq = 0;                                                            // sequence number on their buffers
            ADN_TRY(x3_launch_seq(l[k], &seq));
            L.tag0[k] = seq * 1024u + 1u;
        }
        hipLaunchKernelGGL(lstm_fwd_cluster_x3_kernel, dim3(r.count * 4), dim3(512), lds, s, L, mask_tb, B, T, H, ldh, ldg, err);
        ADN_HIP_CHECK(hipGetLastError());
    }
    return ADN_OK;
}

// bf16x3 forward at 256 < H <= 512 (lstm_fwd_cluster_x3w_kernel): 16 workgroups per 48-utterance group
bool lstm_cluster_x3w_supported(const LstmStep* l, int n, int B, int T, int H) {
    if (H <= 256 || H > kWHP || T >= 1024 || getenv("ADN_LSTM_NO_CLUSTER") || getenv("ADN_LSTM_NO_X3_CLUSTER") ||
        getenv("ADN_LSTM_NO_X3_WIDE")) return false;
    for (int k = 0; k < n; ++k)
        if (!l[k].xchg || !l[k].W_frag_fwd || !l[k].W_frag_fwd_lo) return false;
    if (cdiv(B, kWRows) * kWCWG > cluster_cus()) return false;
    if (lstm_frag_elems(H) != (size_t)4 * kWHP * kWHP) return false;
    return cluster_kernel_fits<&lstm_fwd_cluster_x3w_kernel>(kWFwdLdsBytes);
}

int lstm_forward_cluster_x3w(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, hipStream_t s) {
    const int groups = cdiv(B, kWRows), per = groups * kWCWG, cus = cluster_cus();
    ADN_CHECK(cus >= per, ADN_ERR_STATE, "lstm cluster kernel: one LSTM does not fit the device");
    int* err = nullptr;
    ADN_TRY(lstm_cluster_error_word(&err));
    const int ldh = ld_of(H), ldg = ld_of(4 * H);
    static bool attr_set[kMaxDevices] = {};
    bool& attr = attr_set[current_device()];
    if (!attr) {
        ADN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_fwd_cluster_x3w_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWFwdLdsBytes));
        attr = true;
    }
    const double bytes = (double)n * T * (4.0 * (12.0 * B * H + 4.0 * H * H) + B), flops = (double)n * T * 8.0 * B * H * H;
    ProfScope prof(PROF_LSTM_FWD, flops, bytes, s, T);
    LstmClusterX3P L;
    for (int k = 0; k < n; ++k) { L.l[k] = l[k]; L.tag0[k] = 0u; }
    L.groups = groups;
    for (const PairRange& r : plan_pairs(n, groups, kWCWG, cus)) {
        L.pair0 = r.pair0;
        for (int k = r.pair0 / groups; k <= (r.pair0 + r.count - 1) / groups; ++k) {
            unsigned seq = 0;
            ADN_TRY(x3_launch_seq(l[k], &seq));
            L.tag0[k] = seq * 1024u + 1u;
        }
        hipLaunchKernelGGL(lstm_fwd_cluster_x3w_kernel, dim3(r.count * kWCWG), dim3(512), kWFwdLdsBytes, s, L, mask_tb, B, T, H, ldh, ldg, err);
        ADN_HIP_CHECK(hipGetLastError());
    }
    return ADN_OK;
}

// ... and its backward kernel (the exchange buffer's wide layout: lstm_cluster_xchg_bytes)
bool lstm_cluster_x3w_bwd_supported(const LstmStep* l, int n, int B, int T, int H) {
    if (H <= 256 || H > kWHP || T >= 1024 || getenv("ADN_LSTM_NO_CLUSTER") || getenv("ADN_LSTM_NO_X3_CLUSTER") ||
        getenv("ADN_LSTM_NO_X3_WIDE") || getenv("ADN_LSTM_NO_X3_CLUSTER_BWD")) return false;
    for (int k = 0; k < n; ++k)
        if (!l[k].xchg || !l[k].W_frag_bwd || !l[k].W_frag_bwd_lo) return false;
    if (cdiv(B, kWRows) * kWCWG > cluster_cus()) return false;
    if (lstm_frag_elems(H) != (size_t)4 * kWHP * kWHP) return false;
    return cluster_kernel_fits<&lstm_bwd_cluster_x3w_kernel>(kWBwdLdsBytes);
}

int lstm_backward_cluster_x3w(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, hipStream_t s) {
    const int groups = cdiv(B, kWRows), per = groups * kWCWG, cus = cluster_cus();
    ADN_CHECK(cus >= per, ADN_ERR_STATE, "lstm cluster kernel: one LSTM does not fit the device");
    int* err = nullptr;
    ADN_TRY(lstm_cluster_error_word(&err));
    const int ldh = ld_of(H), ldg = ld_of(4 * H);
    static bool attr_set[kMaxDevices] = {};
    bool& attr = attr_set[current_device()];
    if (!attr) {
        ADN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_cluster_x3w_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWBwdLdsBytes));
        attr = true;
    }
    const EntrySteps es = entry_steps(l, n, T);
    const double bytes = es.rows * 4.0 * 15.0 * B * H + es.weights * 16.0 * H * H, flops = es.rows * 8.0 * B * H * H;
    ProfScope prof(PROF_LSTM_BWD, flops, bytes, s, T + 1);
    LstmClusterP L;
    for (int k = 0; k < n; ++k) L.l[k] = l[k];
    L.groups = groups;
    for (const PairRange& r : plan_pairs(n, groups, kWCWG, cus)) {
        L.pair0 = r.pair0;
        hipLaunchKernelGGL(lstm_bwd_cluster_x3w_kernel, dim3(r.count * kWCWG), dim3(512), kWBwdLdsBytes, s, L, mask_tb, B, T, H, ldh, ldg, err);
        ADN_HIP_CHECK(hipGetLastError());
    }
    return ADN_OK;
}

template <int CWG>
static int backward_cluster(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, hipStream_t s) {
    using G = ClusterGeom<CWG>;
    const int groups = cdiv(B, kCRows), per = groups * CWG, cus = cluster_cus();
    ADN_CHECK(cus >= per, ADN_ERR_STATE, "lstm cluster kernel: one LSTM does not fit the device");
    int* err = nullptr;
    ADN_TRY(lstm_cluster_error_word(&err));
    const int ldh = ld_of(H), ldg = ld_of(4 * H);
    const size_t lds = (size_t)(G::WLdsBwd + kCRows * kCDS) * 2 + (size_t)kCRows * (kCUnits + 1) * 4;
    static bool attr_set[kMaxDevices] = {};        // (a function attribute is per device)
    bool& attr = attr_set[current_device()];
    if (!attr) {
        ADN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_cluster_kernel<CWG>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    const EntrySteps es = entry_steps(l, n, T);
    const double bytes = es.rows * 4.0 * 15.0 * B * H + es.weights * 16.0 * H * H, flops = es.rows * 8.0 * B * H * H;
    ProfScope prof(PROF_LSTM_BWD, flops, bytes, s, T + 1);
    LstmClusterP L;
    for (int k = 0; k < n; ++k) L.l[k] = l[k];
    L.groups = groups;
    for (const PairRange& r : plan_pairs(n, groups, CWG, cus)) {
        L.pair0 = r.pair0;
        hipLaunchKernelGGL(lstm_bwd_cluster_kernel<CWG>, dim3(r.count * CWG), dim3(512), lds, s, L, mask_tb, B, T, H, ldh, ldg, err);
        ADN_HIP_CHECK(hipGetLastError());
    }
    return ADN_OK;
}

int lstm_backward_cluster(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, hipStream_t s) {
    return H <= 256 ? backward_cluster<4>(l, n, mask_tb, B, T, H, s) : backward_cluster<8>(l, n, mask_tb, B, T, H, s);
}

bool lstm_cluster_x3_bwd_supported(const LstmStep* l, int n, int B, int T, int H) {
    if (H > 256 || T >= 1024 || getenv("ADN_LSTM_NO_CLUSTER") || getenv("ADN_LSTM_NO_X3_CLUSTER") ||
        getenv("ADN_LSTM_NO_X3_CLUSTER_BWD")) return false;
    for (int k = 0; k < n; ++k)
        if (!l[k].xchg || !l[k].W_frag_bwd || !l[k].W_frag_bwd_lo) return false;
    if (cdiv(B, kCRows) * 4 > cluster_cus()) return false;
    if (lstm_frag_elems(H) != (size_t)4 * 256 * 256) return false;
    return cluster_kernel_fits<&lstm_bwd_cluster_x3_kernel>((size_t)kX3BwdWOff * 2 + (size_t)8 * 2 * kX3BwdLds * 64 * 16);
}

int lstm_backward_cluster_x3(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, hipStream_t s) {
    const int groups = cdiv(B, kCRows), per = groups * 4, cus = cluster_cus();
    ADN_CHECK(cus >= per, ADN_ERR_STATE, "lstm cluster kernel: one LSTM does not fit the device");
    int* err = nullptr;
    ADN_TRY(lstm_cluster_error_word(&err));
    const int ldh = ld_of(H), ldg = ld_of(4 * H);
    const size_t lds = (size_t)kX3BwdWOff * 2 + (size_t)8 * 2 * kX3BwdLds * 64 * 16;
    static bool attr_set[kMaxDevices] = {};
    bool& attr = attr_set[current_device()];
    if (!attr) {
        ADN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_cluster_x3_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    const EntrySteps es = entry_steps(l, n, T);
    const double bytes = es.rows * 4.0 * 15.0 * B * H + es.weights * 16.0 * H * H, flops = es.rows * 8.0 * B * H * H;
    ProfScope prof(PROF_LSTM_BWD, flops, bytes, s, T + 1);
    LstmClusterP L;
    for (int k = 0; k < n; ++k) L.l[k] = l[k];
    L.groups = groups;
    for (const PairRange& r : plan_pairs(n, groups, 4, cus)) {
        L.pair0 = r.pair0;
        hipLaunchKernelGGL(lstm_bwd_cluster_x3_kernel, dim3(r.count * 4), dim3(512), lds, s, L, mask_tb, B, T, H, ldh, ldg, err);
        ADN_HIP_CHECK(hipGetLastError());
    }
    return ADN_OK;
}

// test hook: n workgroups of 64 threads that each claim `lds_bytes` of LDS (160 KB: nothing else fits on their CU) and spin
// for `ms` milliseconds of wall clock
__global__ __launch_bounds__(64) void occupy_cus_kernel(unsigned long long ticks, int* sink) {
    extern __shared__ int pad[];
    const unsigned long long t0 = wall_ticks();
    pad[threadIdx.x] = (int)t0;
    while (wall_ticks() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (sink && pad[threadIdx.x] == 0x7fffffff) *sink = 1;       // (keeps the LDS allocation alive)
}

}  // namespace adn

extern "C" int adn_debug_occupy_cus(int n_workgroups, int lds_bytes, double ms, void* hip_stream) {
    using namespace adn;
    ADN_CHECK(n_workgroups >= 1 && lds_bytes >= 256 && lds_bytes <= 160 * 1024 && ms >= 0 && ms <= 2000.0, ADN_ERR_INVALID,
              "adn_debug_occupy_cus: bad argument");
    ADN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&occupy_cus_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      lds_bytes));
    hipLaunchKernelGGL(occupy_cus_kernel, dim3(n_workgroups), dim3(64), (size_t)lds_bytes, static_cast<hipStream_t>(hip_stream),
                       (unsigned long long)(ms * 1e5), nullptr);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

extern "C" int adn_debug_lstm_family_counts(int64_t out[4]) {
    if (!out) return ADN_ERR_INVALID;
    for (int k = 0; k < 4; ++k) out[k] = adn::g_lstm_family_forwards[k];
    return ADN_OK;
}

// host logic only (no device call): how a call's (LSTM, group) pairs are dealt over launches
extern "C" int adn_debug_plan_lstm_launches(int n_lstm, int groups, int wg_per_group, int cus, int32_t* pair0, int32_t* count, int max_launches) {
    if (n_lstm < 1 || groups < 1 || wg_per_group < 1 || cus < 1 || !pair0 || !count) return -1;
    const std::vector<adn::PairRange> plan = adn::plan_pairs(n_lstm, groups, wg_per_group, cus);
    for (size_t k = 0; k < plan.size() && (int)k < max_launches; ++k) { pair0[k] = plan[k].pair0; count[k] = plan[k].count; }
    return (int)plan.size();
}

extern "C" int adn_debug_lstm_backward_family_counts(int64_t out[4]) {
    if (!out) return ADN_ERR_INVALID;
    for (int k = 0; k < 4; ++k) out[k] = adn::g_lstm_family_backwards[k];
    return ADN_OK;
}

#ifdef ADN_LSTM_STAMPS
extern "C" int adn_debug_lstm_stamps(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(adn::g_stamps), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(adn::g_stamps), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif
