// Fused-plane persistent GEMM for the bf16x3 mode (and, as PLANES = false, the plain bf16 product on the same body).
//
// bf16x3: a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi over the operands' hi / lo planes (gemm_bf16.hip, split_hilo).  The ping-pong
// kernel of gemm_bf16.hip walks that sum as THREE K-segments, i.e. it stages A_hi and B_hi twice: six operand tiles through the
// LDS-DMA ring and six sets of fragment reads per three MFMA products.  This kernel stages the four planes ONCE per K-unit and
// issues the three products from the four fragment sets: a third less DMA and fragment traffic per flop, a third fewer
// barriers and waits per flop.
//
// What makes that fit is the unit, not the tile: a K-unit is 16 k of ALL FOUR planes of a 256 x 256 tile
//     [A_hi 256 x 16 | A_lo 256 x 16 | B_hi 16 x 256 | B_lo 16 x 256]  =  4 x 8 KB  =  32 KB
// -- the stage size of the ping-pong kernel, so the ring stays 4 deep with three units in flight (a K = 32 stage of four planes
// would be 64 KB: two stages, one in flight) -- multiplied by v_mfma_f32_32x32x16_bf16, whose K is that 16: a wave's 64 x 128
// block is 2 x 4 accumulator tiles of 32 x 32 (128 registers, as before), its fragments of a unit 2 x 2 (A) + 4 x 2 (B) sets of
// 4 registers = 48 registers (as before), and 24 MFMAs of 32 cycles per unit and wave where the ping-pong kernel issues 32 of 16
// cycles per K = 32 stage of one segment.  A 32x32x16 MFMA also holds the SIMD's vector issue port for 8 of its 32 cycles (a
// 16x16x32 for 8 of 16: MI355X_MICROARCH.md), which leaves the partner wave's LDS reads and DMA issue three slots out of four
// instead of one out of two.
//
// Everything else is the ping-pong kernel's design (gemm_bf16.hip, which documents the hazards): ONE persistent 512-thread
// workgroup per CU walking an XCD-aware tile list, 8 waves as 4 (m) x 2 (n), global_load_lds_dwordx4 into the ring across tile
// boundaries, the two waves of a SIMD half a unit apart around one s_barrier per unit, hand-counted vmcnt, swapped MFMA
// operands (C^T in the accumulators: a lane holds 4 consecutive columns of a row), the epilogue of a tile at the top of the next
// iteration, split-K through partial slabs, up to 4 same-shape problems per launch.
//
// LDS images (a DMA wave-instruction writes 1 KiB linearly, so conflicts are removed by permuting which 16-byte chunk of global
// memory a lane fetches; the fragment reads apply the same permutation):
//   k-contiguous plane  [256 rows][16 k]  (32-byte rows, one piece = 32 rows): chunk c (0 / 1) of row r lives in slot
//                        c ^ ((r >> 4) & 1): the four 16-lane groups of a ds_read_b128 (rows {0-3, 12-15, 20-27} ...) then
//                        touch every bank once;
//   k-strided plane     [16 k][256 columns] (512-byte rows, one piece = 2 k-rows): chunk c (0..31) of k-row kr lives in slot
//                        c ^ ((kr & 3) << 2): the four k-rows a ds_read_b64_tr_b16 half-wave reads (64 bytes each) fall into
//                        the four 64-byte quarters of a 256-byte bank row.
// PLANES = false: the same unit holds 32 k of ONE plane per operand -- sub-image 0 = k 0..15, sub-image 1 = k 16..31 -- and a
// unit is two products per accumulator tile; nothing else changes.
#include "gemm_common.h"
#include "pp_dma.h"
#include <algorithm>
#include <type_traits>

namespace adn {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kFBM = 256, kFBN = 256, kFNS = 4, kFD = 3;
constexpr int kFPlane = 8192;                 // bytes of one sub-image: 256 x 16 bf16
constexpr int kFSlot = 4 * kFPlane;           // one K-unit
constexpr int kFPW = 4;                       // 1-KiB DMA pieces per wave and unit (one per sub-image)

__device__ __forceinline__ uint2 pack4(const float4& v) {
    bf16x4 r;
    r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
    return __builtin_bit_cast(uint2, r);
}
__device__ __forceinline__ float4 unpack4(const uint2& u) {
    const bf16x4 r = __builtin_bit_cast(bf16x4, u);
    return make_float4((float)r[0], (float)r[1], (float)r[2], (float)r[3]);
}

// Lanes l < 32 hold X = columns 0..3 and Y = columns 8..11 of a row's 16-column stretch, lanes l + 32 columns 4..7 and 12..15.
// Returns for l < 32: (X, partner's X) = columns 0..7; for l >= 32: (partner's Y, Y) = columns 8..15 -- 16 bytes of bf16 each.
__device__ __forceinline__ uint4 xchg32(const uint2& X, const uint2& Y, bool upper) {
    const uint2 give = upper ? X : Y;
    uint2 take;
    take.x = __shfl_xor(give.x, 32, 64); take.y = __shfl_xor(give.y, 32, 64);
    return upper ? make_uint4(take.x, take.y, Y.x, Y.y) : make_uint4(X.x, X.y, take.x, take.y);
}

}  // namespace

// optional phase timing (build with -DADN_GEMM_STAMPS; read with adn_debug_x3f_stamps): shader-clock cycles that wave 0 (early
// half) and wave 4 (late half) of workgroup 3 spend in [0] fragment-read issue, [1] DMA issue, [2] lgkmcnt wait, [3] vmcnt wait,
// [4] barrier behind L, [5] MFMAs, [6] epilogue, [7] barrier behind C; slots 8.. the same for wave 4
#ifdef ADN_GEMM_STAMPS
__device__ unsigned long long g_fstamps[18];      // [16], [17]: s_memtime / s_memrealtime ticks of wave 0 over its whole run (the in-kernel clock)
#define FSTAMP(k) do { if (stamping) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); facc_[k] += n_ - fl_; fl_ = n_; } } while (0)
#define FSTAMP_INIT const bool stamping = blockIdx.x == 3 && blockIdx.y == 0 && (wave == 0 || wave == 4); \
    unsigned long long facc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long fl_ = __builtin_amdgcn_s_memtime(); \
    const unsigned long long ft0_ = fl_, fr0_ = __builtin_amdgcn_s_memrealtime();
#define FSTAMP_FLUSH do { if (stamping && lane == 0) { for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&g_fstamps[k_ + (late ? 8 : 0)], facc_[k_]); \
    if (!late) { atomicAdd(&g_fstamps[16], __builtin_amdgcn_s_memtime() - ft0_); atomicAdd(&g_fstamps[17], __builtin_amdgcn_s_memrealtime() - fr0_); } } } while (0)
#else
#define FSTAMP(k) do {} while (0)
#define FSTAMP_INIT
#define FSTAMP_FLUSH do {} while (0)
#endif

template <bool A_KC, bool SPLIT, bool PLANES>
__global__ __launch_bounds__(512) void gemm_x3f_kernel(const GemmParams p) {
    constexpr int BM = kFBM, BN = kFBN, NS = kFNS, D = kFD, PW = kFPW;
    constexpr int UK = PLANES ? 16 : 32;                           // k per unit
    constexpr int TM = 2, TN = 4;                                  // 32 x 32 accumulator tiles per wave: 64 rows x 128 columns
    __shared__ __attribute__((aligned(1024))) char smem[NS * kFSlot];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bool late = wave >= 4;                                   // the half that runs one barrier behind
    const int per_group = p.tiles_m * p.tiles_n;
    const int ntiles = per_group * p.ngroups;
    const int G = (int)gridDim.x;
    int bid = (int)blockIdx.x, slice = (int)blockIdx.y;
    if (SPLIT && p.xcd_slices) {                                   // K-slice = function of the workgroup's XCD (gemm_bf16_pp_kernel)
        const int S = (int)gridDim.y, lin = (int)blockIdx.y * G + (int)blockIdx.x, xcd = lin & 7, q = lin >> 3;
        if (S >= 8) { const int m_ = S >> 3; slice = xcd * m_ + q % m_; bid = q / m_; }
        else { const int d_ = 8 / S; slice = xcd % S; bid = q * d_ + xcd / S; }
    }
    const int my_tiles = (ntiles - bid + G - 1) / G;
    const int kbeg = slice * p.k_chunk;
    const int kend = min(p.K, kbeg + p.k_chunk);
    const int nk = (kend - kbeg + UK - 1) / UK;
    const int ktail = (kend - kbeg) - (nk - 1) * UK;               // valid k of the last unit (1..UK)
    const bool has_tail = ktail < UK;
    const int total = my_tiles * nk;                               // units of this workgroup's whole tile list
    if (total <= 0) return;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_void_t*)smem;

    // ---- DMA side: wave w moves piece w of each of the four sub-images of a unit ------------------------
    const int kr_dma = 2 * wave + (lane >> 5);                     // k-strided sub-image: this lane's k-row inside it
    const int c_dma = (lane & 31) ^ ((kr_dma & 3) << 2);           // ... and the 16-byte chunk of that row it fetches
    const int ch_kc = (lane & 1) ^ (lane >> 5);                    // k-contiguous sub-image: the chunk (0 / 1) of row 32 w + (lane >> 1)
    const char *bA0 = nullptr, *bA1 = nullptr, *bB0 = nullptr, *bB1 = nullptr;
    unsigned offA = 0, offB = 0;
    auto tile_of = [&](int ord, int& grp, int& tm, int& tn) {      // ord-th tile of this workgroup
        const int q = xcd_tile(bid + ord * G, ntiles);
        grp = q / per_group;
        tile_coords(p, q - grp * per_group, tm, tn);
    };
    auto setup_src = [&](int ord) {
        int grp, tm, tn;
        tile_of(ord, grp, tm, tn);
        const GemmGroup sg = pick_group(p, grp);
        const int m0 = tm * BM, n0 = tn * BN;
        const size_t a_first = A_KC ? ((size_t)m0 * p.lda + kbeg) * 2 : (size_t)kbeg * p.lda * 2;
        const size_t b_first = (size_t)kbeg * p.ldb * 2;
        bA0 = reinterpret_cast<const char*>(sg.A16) + a_first;
        bB0 = reinterpret_cast<const char*>(sg.B16) + b_first;
        if (PLANES) {
            bA1 = reinterpret_cast<const char*>(sg.A16lo) + a_first;
            bB1 = reinterpret_cast<const char*>(sg.B16lo) + b_first;
        } else {
            bA1 = bA0 + (A_KC ? (size_t)16 * 2 : (size_t)16 * p.lda * 2);
            bB1 = bB0 + (size_t)16 * p.ldb * 2;
        }
        if (A_KC) {
            const int row = min(m0 + 32 * wave + (lane >> 1), p.M - 1) - m0;       // clamped: feeds dropped outputs only
            offA = (unsigned)(row * p.lda + 8 * ch_kc) * 2u;
        } else {
            int col = m0 + 8 * c_dma;
            if (col + 8 > p.lda) col = 0;
            offA = (unsigned)(kr_dma * p.lda + col) * 2u;
        }
        int colb = n0 + 8 * c_dma;
        if (colb + 8 > p.ldb) colb = 0;
        offB = (unsigned)(kr_dma * p.ldb + colb) * 2u;
    };
    const size_t a_step = A_KC ? (size_t)UK * 2 : (size_t)UK * p.lda * 2;      // bytes per unit
    const size_t b_step = (size_t)UK * p.ldb * 2;
    const unsigned dst_w = __builtin_amdgcn_readfirstlane(lds_base + wave * 1024);
    int is_k = 0, is_ord = 0, is_slot = 0;                         // next unit to issue: index inside its tile, tile, ring slot
    auto issue_one = [&](auto tail_c) {
        constexpr bool tail = decltype(tail_c)::value;
        const unsigned d = __builtin_amdgcn_readfirstlane(dst_w + (unsigned)(is_slot * kFSlot));
        if (tail) {
            // k >= kend: re-read the last valid k (finite data; the A fragments are masked in registers)
            const int k0 = kbeg + is_k * UK, k1 = PLANES ? k0 : k0 + 16;          // first k of sub-images 0 / 1
            const char *gA0 = bA0 + offA, *gA1 = bA1 + offA, *gB0 = bB0 + offB, *gB1 = bB1 + offB;
            if (A_KC) {
                const int ka = 8 * ch_kc;
                if (k0 + ka >= kend) gA0 -= (size_t)(k0 + ka - (kend - 8)) * 2;
                if (k1 + ka >= kend) gA1 -= (size_t)(k1 + ka - (kend - 8)) * 2;
            } else {
                if (k0 + kr_dma >= kend) gA0 -= (size_t)(k0 + kr_dma - (kend - 1)) * p.lda * 2;
                if (k1 + kr_dma >= kend) gA1 -= (size_t)(k1 + kr_dma - (kend - 1)) * p.lda * 2;
            }
            if (k0 + kr_dma >= kend) gB0 -= (size_t)(k0 + kr_dma - (kend - 1)) * p.ldb * 2;
            if (k1 + kr_dma >= kend) gB1 -= (size_t)(k1 + kr_dma - (kend - 1)) * p.ldb * 2;
            glds16(gA0, d);
            glds16(gA1, __builtin_amdgcn_readfirstlane(d + kFPlane));
            glds16(gB0, __builtin_amdgcn_readfirstlane(d + 2 * kFPlane));
            glds16(gB1, __builtin_amdgcn_readfirstlane(d + 3 * kFPlane));
        } else {
            glds16_s<0>(offA, bA0, d);
            glds16_s<kFPlane>(offA, bA1, d);
            glds16_s<2 * kFPlane>(offB, bB0, d);
            glds16_s<3 * kFPlane>(offB, bB1, d);
        }
        bA0 += a_step; bA1 += a_step; bB0 += b_step; bB1 += b_step;
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

    // ---- per-lane fragment addresses (LDS byte addresses inside ring slot 0, sub-image 0 of the operand) -------
    // 32x32x16 operand: lane l holds index i = l & 31 (row of A / column of B) and k = 8 (l >> 5) .. + 7
    const int i32 = lane & 31, h = lane >> 5;
    const int gq = (lane & 15) >> 2, gp4 = lane & 3, g16 = (lane >> 4) & 1;   // transposing reads: k-row q, 8-byte piece p of the 16-lane group
    unsigned a_off[TM], b_off[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int r0 = wm * 64 + 32 * t;
        if (A_KC) a_off[t] = lds_base + (r0 + i32) * 32 + ((h ^ g16) << 4);
        else a_off[t] = lds_base + (8 * h + gq) * 512 + ((((r0 + 16 * g16) >> 3) + (gp4 >> 1)) ^ (gq << 2)) * 16 + (gp4 & 1) * 8;
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int c0 = wn * 128 + 32 * t;
        b_off[t] = lds_base + 2 * kFPlane + (8 * h + gq) * 512 + ((((c0 + 16 * g16) >> 3) + (gp4 >> 1)) ^ (gq << 2)) * 16 + (gp4 & 1) * 8;
    }
    // k >= K mask of the A fragments in the last unit: element j of this lane is k = 8 h + j of its sub-image
    u32x4 amask0, amask1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = 8 * h + 2 * j;
        amask0[j] = ((k < ktail) ? 0x0000FFFFu : 0u) | ((k + 1 < ktail) ? 0xFFFF0000u : 0u);
        const int k2 = PLANES ? k : k + 16;
        amask1[j] = ((k2 < ktail) ? 0x0000FFFFu : 0u) | ((k2 + 1 < ktail) ? 0xFFFF0000u : 0u);
    }

    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    typedef __attribute__((address_space(3))) bf16x8 lds_bf16x8;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // wait until this wave's pieces of unit s + 1 have landed; `fresh_epi`: an epilogue's stores are in the queue
    auto wait_next = [&](int s, bool fresh_epi) {
        int younger = min(D - 1, total - 2 - s);                   // younger units this wave has issued
        if (younger < 0) return;                                   // there is no unit s + 1
        if (fresh_epi) younger = min(younger, 1);
        if (younger >= 2) wait_vmcnt<2 * PW>();
        else if (younger == 1) wait_vmcnt<PW>();
        else wait_vmcnt<0>();
    };

    // ---- prologue: D units in flight, unit 0 landed for everyone ------------------------------------------
    setup_src(0);
    for (int s = 0; s < D && s < total; ++s) issue_next();
    wait_next(-1, false);
    __builtin_amdgcn_s_barrier();

    int kt = 0, ord = 0;
    int grp, tile_m, tile_n;
    tile_of(0, grp, tile_m, tile_n);
    bf16x8 fa0[TM], fa1[TM], fb0[TN], fb1[TN];
    auto tr_frag = [&](unsigned addr) __attribute__((always_inline)) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(addr));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(addr + 4 * 512));
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto read_frags = [&](unsigned so) __attribute__((always_inline)) {       // so: byte offset of the ring slot
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            if (A_KC) {
                fa0[t] = *(lds_bf16x8*)(uintptr_t)(a_off[t] + so);
                fa1[t] = *(lds_bf16x8*)(uintptr_t)(a_off[t] + so + kFPlane);
            } else {
                fa0[t] = tr_frag(a_off[t] + so);
                fa1[t] = tr_frag(a_off[t] + so + kFPlane);
            }
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            fb0[t] = tr_frag(b_off[t] + so);
            fb1[t] = tr_frag(b_off[t] + so + kFPlane);
        }
    };
    // operands swapped (B fragment first): C^T in the accumulators
    auto mfmas = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb0[b], fa0[a], acc[a][b], 0, 0, 0);
        if (PLANES) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb1[b], fa0[a], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb0[b], fa1[a], acc[a][b], 0, 0, 0);
        } else {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb1[b], fa1[a], acc[a][b], 0, 0, 0);
        }
    };
    int rd_slot = 0;
    FSTAMP_INIT
    for (int s = 0;;) {
        if (kt == 0 && s > 0) {
            // ---- epilogue of tile `ord`, straight from the transposed 32 x 32 accumulators: lane = row (lane & 31) of a tile,
            //      register 4 j + e = column 8 j + 4 h + e: four float4 per accumulator tile
            {
                const GemmGroup gp = pick_group(p, grp);
                float* Cg = SPLIT ? p.partial + ((size_t)(grp * (int)gridDim.y + slice) * p.M) * p.ldc : gp.C;
                const int wr0 = tile_m * BM + wm * 64, wc0 = tile_n * BN + wn * 128;          // the wave's corner (uniform)
                const int cl = 4 * h;
                const int rlim = p.M - wr0, clim = p.N - wc0;                                // rows / columns of it inside the matrix
                const unsigned lo32 = (unsigned)(wr0 + i32) * (unsigned)p.ldc + (unsigned)(wc0 + cl);
                const unsigned ly32 = (unsigned)(wr0 + i32) * (unsigned)p.ldy + (unsigned)(wc0 + cl);
                const unsigned rstep = 32u * (unsigned)p.ldc, ystep = 32u * (unsigned)p.ldy;
                __bf16* C16m = (!SPLIT && gp.C16) ? reinterpret_cast<__bf16*>(gp.C16) : nullptr;
                __bf16* C16lm = (!SPLIT && gp.C16lo) ? reinterpret_cast<__bf16*>(gp.C16lo) : nullptr;
                // rectifier bit images (gemm_common.h GemmGroup::Cbits / Ybits): one uint4 per thread and tile, bit 4 q + e = column e of
                // quad q = (a TN + b) 4 + j -- this kernel's own order: a launch only reads what a launch of this kernel wrote
                // (over planes only: in the one-plane instantiations -- ADN_GEMM_PP=8 -- the four words cost a spill, and scratch traffic
                //  sits in the vmcnt queue this kernel counts by hand; the host offers no images there)
                const uint4* Ybm = (PLANES && !SPLIT && gp.Ybits) ? reinterpret_cast<const uint4*>(gp.Ybits) : nullptr;
                uint4* Cbm = (PLANES && !SPLIT && gp.Cbits) ? reinterpret_cast<uint4*>(gp.Cbits) : nullptr;
                const size_t bits_at = ((size_t)tile_m * p.tiles_n + tile_n) * 512 + tid;
                unsigned yb[4] = {0u, 0u, 0u, 0u}, cb[4] = {0u, 0u, 0u, 0u};
                if (Ybm) { const uint4 t4 = Ybm[bits_at]; yb[0] = t4.x; yb[1] = t4.y; yb[2] = t4.z; yb[3] = t4.w; }
                const __bf16* Ym = (!SPLIT && gp.Y16 && !Ybm) ? reinterpret_cast<const __bf16*>(gp.Y16) : nullptr;
                const float* biasm = (!SPLIT && gp.bias) ? gp.bias : nullptr;
                const float lower = (!SPLIT && p.act == ADN_ACT_RECTIFY) ? 0.f : -3.0e38f;
                const bool upper = h != 0;
                // what the epilogue READS is requested in two bursts per tile, each ahead of its half's first store (see
                // gemm_bf16_pp_kernel): 2 x 2 x 4 8-byte act'(Y) masks, or 2 x 4 float4 of bias -- never both
                uint2 pre[16];
#pragma clang loop unroll(full)
                for (int half = 0; half < 2; ++half) {
                    const int b0 = 2 * half;
                    if (Ym) {
#pragma clang loop unroll(full)
                        for (int bb = 0; bb < 2; ++bb)
#pragma clang loop unroll(full)
                            for (int a = 0; a < TM; ++a)
#pragma clang loop unroll(full)
                                for (int j = 0; j < 4; ++j) {
                                    const int c = 32 * (b0 + bb) + 8 * j;
                                    const bool ok = i32 + 32 * a < rlim && cl + c < clim;
                                    pre[(a * 2 + bb) * 4 + j] = *reinterpret_cast<const uint2*>(Ym + (ok ? ly32 + a * ystep + (unsigned)c : 0u));
                                }
                    } else if (biasm) {
#pragma clang loop unroll(full)
                        for (int bb = 0; bb < 2; ++bb)
#pragma clang loop unroll(full)
                            for (int j = 0; j < 4; ++j) {
                                const int c = cl + 32 * (b0 + bb) + 8 * j;
                                const float4 bv = *reinterpret_cast<const float4*>(biasm + (c < clim ? wc0 + c : 0));
                                pre[(bb * 4 + j) * 2] = make_uint2(__builtin_bit_cast(unsigned, bv.x), __builtin_bit_cast(unsigned, bv.y));
                                pre[(bb * 4 + j) * 2 + 1] = make_uint2(__builtin_bit_cast(unsigned, bv.z), __builtin_bit_cast(unsigned, bv.w));
                            }
                    } else {
#pragma clang loop unroll(full)
                        for (int k = 0; k < 16; ++k) pre[k] = make_uint2(0u, 0u);
                    }
#pragma clang loop unroll(full)
                    for (int bb = 0; bb < 2; ++bb) {
                        const int b = b0 + bb;
                        float4 cs[4];
#pragma clang loop unroll(full)
                        for (int j = 0; j < 4; ++j) cs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma clang loop unroll(full)
                        for (int a = 0; a < TM; ++a) {
                            __builtin_amdgcn_sched_barrier(0);     // (keeps the scheduler from interleaving every block: spills)
                            const bool rok = i32 + 32 * a < rlim;
#pragma clang loop unroll(full)
                            for (int jp = 0; jp < 2; ++jp) {
                                const int jA = 2 * jp, jB = jA + 1;
                                const int cA = 32 * b + 8 * jA;                    // column of block A relative to wc0 + cl; block B: + 8
                                const bool okA = rok && cl + cA < clim, okB = rok && cl + cA + 8 < clim;
                                float4 vA = make_float4(acc[a][b][4 * jA], acc[a][b][4 * jA + 1], acc[a][b][4 * jA + 2], acc[a][b][4 * jA + 3]);
                                float4 vB = make_float4(acc[a][b][4 * jB], acc[a][b][4 * jB + 1], acc[a][b][4 * jB + 2], acc[a][b][4 * jB + 3]);
                                const unsigned oA = lo32 + a * rstep + (unsigned)cA, oB = oA + 8u;
                                if (!SPLIT) {
                                    if (!Ym) {                   // (zeros without a bias)
                                        const uint2 p0 = pre[(bb * 4 + jA) * 2], p1 = pre[(bb * 4 + jA) * 2 + 1];
                                        const uint2 q0 = pre[(bb * 4 + jB) * 2], q1 = pre[(bb * 4 + jB) * 2 + 1];
                                        vA.x += __builtin_bit_cast(float, p0.x); vA.y += __builtin_bit_cast(float, p0.y);
                                        vA.z += __builtin_bit_cast(float, p1.x); vA.w += __builtin_bit_cast(float, p1.y);
                                        vB.x += __builtin_bit_cast(float, q0.x); vB.y += __builtin_bit_cast(float, q0.y);
                                        vB.z += __builtin_bit_cast(float, q1.x); vB.w += __builtin_bit_cast(float, q1.y);
                                    }
                                    vA.x = fmaxf(vA.x, lower); vA.y = fmaxf(vA.y, lower); vA.z = fmaxf(vA.z, lower); vA.w = fmaxf(vA.w, lower);
                                    vB.x = fmaxf(vB.x, lower); vB.y = fmaxf(vB.y, lower); vB.z = fmaxf(vB.z, lower); vB.w = fmaxf(vB.w, lower);
                                    if (Ym) {                    // rectify'(Y) from the bf16 copy of Y
                                        const float4 yA = unpack4(pre[(a * 2 + bb) * 4 + jA]), yB = unpack4(pre[(a * 2 + bb) * 4 + jB]);
                                        vA.x = yA.x > 0.f ? vA.x : 0.f; vA.y = yA.y > 0.f ? vA.y : 0.f; vA.z = yA.z > 0.f ? vA.z : 0.f; vA.w = yA.w > 0.f ? vA.w : 0.f;
                                        vB.x = yB.x > 0.f ? vB.x : 0.f; vB.y = yB.y > 0.f ? vB.y : 0.f; vB.z = yB.z > 0.f ? vB.z : 0.f; vB.w = yB.w > 0.f ? vB.w : 0.f;
                                    } else if (Ybm) {
                                        const int qA = (a * TN + b) * 4 + jA, qB = qA + 1;      // (constants once the loops are unrolled)
                                        const unsigned mA = yb[qA >> 3] >> ((qA & 7) * 4), mB = yb[qB >> 3] >> ((qB & 7) * 4);
                                        vA.x = (mA & 1u) ? vA.x : 0.f; vA.y = (mA & 2u) ? vA.y : 0.f; vA.z = (mA & 4u) ? vA.z : 0.f; vA.w = (mA & 8u) ? vA.w : 0.f;
                                        vB.x = (mB & 1u) ? vB.x : 0.f; vB.y = (mB & 2u) ? vB.y : 0.f; vB.z = (mB & 4u) ? vB.z : 0.f; vB.w = (mB & 8u) ? vB.w : 0.f;
                                    }
                                    if (Cbm) {
                                        const int qA = (a * TN + b) * 4 + jA, qB = qA + 1;
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
                                    if (okA) { cs[jA].x += vA.x; cs[jA].y += vA.y; cs[jA].z += vA.z; cs[jA].w += vA.w; }
                                    if (okB) { cs[jB].x += vB.x; cs[jB].y += vB.y; cs[jB].z += vB.z; cs[jB].w += vB.w; }
                                }
                                if (C16m) {
                                    // after the exchange a lane holds 8 consecutive columns: lower lanes block A's 0..7, upper lanes block B's 0..7
                                    const int csx = cA + (upper ? 8 : 0);                          // relative to wc0
                                    const bool c16 = csx + 8 <= clim, c8 = csx + 4 <= clim;
                                    const unsigned l16 = lo32 + (upper ? 4u : 0u) + a * rstep + (unsigned)cA;
                                    const uint2 pA = pack4(vA), pB = pack4(vB);
                                    const uint4 out = xchg32(pA, pB, upper);
                                    __bf16* dst = C16m + l16;
                                    if (rok && c16) *reinterpret_cast<uint4*>(dst) = out;
                                    else if (rok && c8) *reinterpret_cast<uint2*>(dst) = make_uint2(out.x, out.y);
                                    if (C16lm) {                 // bf16x3: the lo plane bf16(v - bf16(v)) beside it, same exchange
                                        const float4 hA = unpack4(pA), hB = unpack4(pB);
                                        const float4 rA = make_float4(vA.x - hA.x, vA.y - hA.y, vA.z - hA.z, vA.w - hA.w);
                                        const float4 rB = make_float4(vB.x - hB.x, vB.y - hB.y, vB.z - hB.z, vB.w - hB.w);
                                        const uint4 outl = xchg32(pack4(rA), pack4(rB), upper);
                                        __bf16* dl = C16lm + l16;
                                        if (rok && c16) *reinterpret_cast<uint4*>(dl) = outl;
                                        else if (rok && c8) *reinterpret_cast<uint2*>(dl) = make_uint2(outl.x, outl.y);
                                    }
                                }
                            }
                        }
                        if (Cbm && half == 1 && bb == 1) Cbm[bits_at] = make_uint4(cb[0], cb[1], cb[2], cb[3]);
                        if (!SPLIT && gp.colsum) {
                            // column sums over this wave's 64 rows: a reduce-scatter butterfly over the 32 row lanes -- every step
                            // halves what a lane carries; lane bits 4..1 end up selecting (j, e), bit 0 pairs add
                            float v8[8], v4[4], v2[2], v1;
                            const bool t4 = lane & 16, t3 = lane & 8, t2 = lane & 4, t1 = lane & 2;
                            const float c16v[16] = {cs[0].x, cs[0].y, cs[0].z, cs[0].w, cs[1].x, cs[1].y, cs[1].z, cs[1].w,
                                                    cs[2].x, cs[2].y, cs[2].z, cs[2].w, cs[3].x, cs[3].y, cs[3].z, cs[3].w};
#pragma unroll
                            for (int i = 0; i < 8; ++i) v8[i] = (t4 ? c16v[8 + i] : c16v[i]) + __shfl_xor(t4 ? c16v[i] : c16v[8 + i], 16, 64);
#pragma unroll
                            for (int i = 0; i < 4; ++i) v4[i] = (t3 ? v8[4 + i] : v8[i]) + __shfl_xor(t3 ? v8[i] : v8[4 + i], 8, 64);
#pragma unroll
                            for (int i = 0; i < 2; ++i) v2[i] = (t2 ? v4[2 + i] : v4[i]) + __shfl_xor(t2 ? v4[i] : v4[2 + i], 4, 64);
                            v1 = (t1 ? v2[1] : v2[0]) + __shfl_xor(t1 ? v2[0] : v2[1], 2, 64);
                            v1 += __shfl_xor(v1, 1, 64);
                            const int col = 32 * b + 8 * ((t4 ? 2 : 0) + (t3 ? 1 : 0)) + cl + (t2 ? 2 : 0) + (t1 ? 1 : 0);
                            if (!(lane & 1) && col < clim) gp.colsum[(size_t)(tile_m * 4 + wm) * p.colsum_ld + wc0 + col] = v1;
                        }
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
            if (++ord < my_tiles) tile_of(ord, grp, tile_m, tile_n);
            // a compiler-visible vmcnt(0) behind the epilogue (see gemm_bf16_pp_kernel: without it hipcc drains the ring at the
            // head of the interior loop)
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        FSTAMP(6);
        if (s == total) break;
        // ---------------- interior units of a tile: the unit issued (s + D) lies inside this tile and ahead of its masked last
        //      one, this unit is neither the first behind an epilogue nor masked, D - 1 younger units are in flight
        while (kt >= 1 && kt + D < nk - 1) {
            read_frags((unsigned)(rd_slot * kFSlot));
            if (++rd_slot == NS) rd_slot = 0;
            FSTAMP(0);
            issue_one(std::false_type{});
            if (++is_slot == NS) is_slot = 0;
            ++is_k;
            FSTAMP(1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            FSTAMP(2);
            wait_vmcnt<(D - 1) * PW>();
            FSTAMP(3);
            if (late) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            FSTAMP(4);
            mfmas();
            ++kt;
            __builtin_amdgcn_sched_barrier(0);
            FSTAMP(5);
            if (!late) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            FSTAMP(7);
            ++s;
        }
        // ---------------- the general unit.  L(s): fragments of unit s -> registers, DMA for unit s + D
        read_frags((unsigned)(rd_slot * kFSlot));
        if (++rd_slot == NS) rd_slot = 0;
        FSTAMP(0);
        if (s + D < total) issue_next();
        FSTAMP(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        FSTAMP(2);
        if (has_tail && kt == nk - 1) {
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                fa0[a] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, fa0[a]) & amask0);
                fa1[a] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, fa1[a]) & amask1);
            }
        }
        wait_next(s, kt == 0 && s > 0);
        FSTAMP(3);
        if (late) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        FSTAMP(4);
        // ---------------- C(s)
        mfmas();
        if (__builtin_expect(kt == nk - 1, 0)) kt = 0;       // the tile is complete: its epilogue opens the next iteration
        else ++kt;
        __builtin_amdgcn_sched_barrier(0);
        FSTAMP(5);
        if (!late) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        FSTAMP(7);
        ++s;
    }
    FSTAMP_FLUSH;
}

// launch: grid = (workgroups, K-slices); splits > 1: partial slabs + the ping-pong kernel's reduce pass
void launch_gemm_x3f(const GemmParams& p, int layout, bool planes, int splits, dim3 grid, hipStream_t s, bool reduce) {
    const bool split = splits > 1, kc = layout == GEMM_NN;
#define ADN_X3F_LAUNCH(KC, SP, PL) hipLaunchKernelGGL((gemm_x3f_kernel<KC, SP, PL>), grid, dim3(512), 0, s, p)
    if (planes) {
        if (kc) { if (split) ADN_X3F_LAUNCH(true, true, true); else ADN_X3F_LAUNCH(true, false, true); }
        else { if (split) ADN_X3F_LAUNCH(false, true, true); else ADN_X3F_LAUNCH(false, false, true); }
    } else {
        if (kc) { if (split) ADN_X3F_LAUNCH(true, true, false); else ADN_X3F_LAUNCH(true, false, false); }
        else { if (split) ADN_X3F_LAUNCH(false, true, false); else ADN_X3F_LAUNCH(false, false, false); }
    }
#undef ADN_X3F_LAUNCH
    if (split && reduce) launch_splitk_reduce(p, splits, s);
}

}  // namespace adn

#ifdef ADN_GEMM_STAMPS
extern "C" int adn_debug_x3f_stamps(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(adn::g_fstamps), sizeof(unsigned long long) * 18) != hipSuccess) return 1;
    if (reset) { unsigned long long z[18] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(adn::g_fstamps), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif
