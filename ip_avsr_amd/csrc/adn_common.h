// Internal helpers shared by the HIP sources of libadenet_hip.so (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/adenet.h"

namespace adn {

void set_error(const std::string& msg);
// Deterministic mode (ADN_DETERMINISTIC=1 or adn_set_deterministic): every reduction whose order would otherwise depend on the
// arrival order of float atomics runs in a fixed order -- column / scalar sums in one block per column tile, the register-staged
// GEMMs without their atomic split-K, the LSTM kernels' group sums (bias, learnt initial state, peephole gradients) through
// per-(group, row tile) slots and a fixed-order pass (lstm.hip).  Two runs from the same seed then give the same bits; slower.
bool deterministic();
void set_deterministic(bool on);

#define ADN_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ::adn::set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) +  \
                             " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")");    \
            return ADN_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

#define ADN_CHECK(cond, code, msg)                                                       \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            ::adn::set_error(std::string(msg) + " [" #cond "] (" + __FILE__ + ":" +      \
                             std::to_string(__LINE__) + ")");                            \
            return (code);                                                               \
        }                                                                                \
    } while (0)

#define ADN_TRY(expr)                                                                    \
    do {                                                                                 \
        int _s = (expr);                                                                 \
        if (_s != ADN_OK) return _s;                                                     \
    } while (0)

// ---------------------------------------------------------------------------------------
// in-stream kernel timing (HIP events on the stream the kernels are launched on); used by bench.py
// to measure per-kernel-class durations live.  Disabled (zero cost) unless adn_profile_enable(m, 1).
// ---------------------------------------------------------------------------------------
enum ProfClass {
    PROF_GEMM_NN = 0, PROF_GEMM_NT, PROF_GEMM_TN, PROF_LSTM_FWD, PROF_LSTM_BWD, PROF_DELTA_FWD, PROF_DELTA_BWD,
    PROF_ADAM, PROF_SOFTMAX_LOSS, PROF_COUNT
};
struct ProfScope {
    // one scope may bracket several back-to-back launches of the class (`launches`): short kernels are timed as
    // a sequence so that the event records do not sit between them
    ProfScope(int cls, double flops, double bytes, hipStream_t s, int launches = 1);
    ~ProfScope();
    int slot;
    hipStream_t stream;
};

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
// leading dimension (in floats) used for every matrix in HBM: rows start 32-byte aligned
// row stride of every matrix the model owns: rows start on 256-byte (fp32) / 128-byte (bf16 shadow) boundaries,
// so that tile rows and split-K atomics cover whole cache lines (measured +5..30 % on the weight-gradient GEMMs
// over rounding to 8)
static inline int ld_of(int cols) { return (int)round_up(cols, 64); }

// ---------------------------------------------------------------------------------------
// GEMM:  C[M,N] (+)= op(A) * op(B)   fp32 in / fp32 accumulate on the f32 MFMA pipe
// ---------------------------------------------------------------------------------------
enum GemmLayout {
    GEMM_NN = 0,  // A[M][K] (lda), B[K][N] (ldb)           forward:      Y  = X  * W
    GEMM_NT = 1,  // A[M][K] (lda), B[N][K] (ldb)           input grad:   dX = dY * W^T
    GEMM_TN = 2,  // A[K][M] (lda), B[K][N] (ldb)           weight grad:  dW = X^T * dY
};

struct GemmArgs {
    int layout = GEMM_NN;
    int M = 0, N = 0, K = 0;
    const float* A = nullptr; int lda = 0;
    const float* B = nullptr; int ldb = 0;
    float* C = nullptr;       int ldc = 0;
    const float* bias = nullptr;   // per output column, added before the activation
    int act = ADN_ACT_LINEAR;      // activation applied to the result
    int accumulate = 0;            // C += result (instead of C = result)
    int no_split = 0;              // never split K (float atomics add in arrival order: forward-pass GEMMs set this so
                                   // that two evaluations of the same batch give the same bits)
    const float* Y = nullptr; int ldy = 0; int act_grad = ADN_ACT_LINEAR;
                                   // result *= act_grad'(Y[row][col])  (back-prop through the
                                   // activation of the layer that produced Y)
    int precision = ADN_PRECISION_F32;
    // bf16 mode only: shadow copies of A / B (same leading dimensions and element offsets); when both are
    // given the kernel loads bf16 directly instead of converting fp32 in flight.  C16: also store the result
    // as bf16 (non-split-K launches only).
    const void* A16 = nullptr; const void* B16 = nullptr; void* C16 = nullptr;
    // colsum[col] += sum over rows of the FINAL C values (bias gradients ride on the GEMM that produces dZ
    // instead of re-reading it); honoured by the bf16 kernels' coalesced epilogue, non-split launches only --
    // gemm() reports through *colsum_done whether it was.
    float* colsum = nullptr; int* colsum_done = nullptr;
    // workspace for the fused column sums: every workgroup stores the sums of its tile's rows with plain stores
    // ([m-tiles][round_up(N,4)] floats) and a small follow-up kernel adds the m-tiles into `colsum`.  (Float
    // atomics execute at the memory side and every workgroup adding into ONE row is the slow case: measured
    // 50-80 us per GEMM.)  Fusion is skipped when the workspace is missing or too small.
    float* colsum_ws = nullptr; size_t colsum_ws_floats = 0;
    // when set and the fused path ran, the reduction of the partial rows is queued here instead of launched
    struct ColSumBatch* colsum_batch = nullptr;
    // bf16 kernels with shadows only: C may be null when C16 is given (the fp32 copy is simply not written; the
    // launch is then never split over K), and Y16 may replace Y (bf16 copy of the activation, same ld)
    const void* Y16 = nullptr;
    // bf16x3 mode: the lo planes of A and B (bf16(x - bf16(x)), layouts of A16 / B16, which then hold the hi planes).  With all
    // four planes at hand the product runs over the planes (three K-segments in the ping-pong kernel) instead of over split
    // images made for this one launch
    const void* A16lo = nullptr; const void* B16lo = nullptr;
    // ... for an NT problem (B given as [N][K]): planes of B^T ([K][ldbT], k-strided), with which the product runs as NN
    const void* BT16 = nullptr; const void* BT16lo = nullptr; int ldbT = 0;
    // ... and the result's own planes: C16 (hi) and C16lo, written by the kernel that multiplies over planes; *planes_done reports
    // whether they were (otherwise the caller splits C itself)
    void* C16lo = nullptr; int* planes_done = nullptr;
    // rectifier bit images (gemm_common.h GemmGroup): a rectify launch may leave "C > 0" per (tile, thread) in Cbits -- *bits_done
    // then holds the tile grid (tiles_m << 16 | tiles_n), else 0 --, and an act'(Y) launch reads Ybits instead of Y16 when its own
    // tile grid is Ybits_tiles (and falls back to Y16 otherwise): same M x N, same kernel, same thread layout
    void* Cbits = nullptr; int* bits_done = nullptr;
    const void* Ybits = nullptr; int Ybits_tiles = 0;
    // B as a k-contiguous [N][K] bf16 matrix (hi / lo planes), leading dimension ldbkc: for an NN problem the transposed copy of the
    // weights the model keeps, for an NT problem B itself.  The skinny kernels (gemm_skinny.hip) read B in this form
    const void* Bkc16 = nullptr; const void* Bkc16lo = nullptr; int ldbkc = 0;
    int no_planes = 0;                   // model.hip: nobody reads this result's hi / lo planes (the encoder's output): do not make them
    int hi_product = 0;                // ADN_PRECISION_MIXED, back-propagation: ONE bf16 product over the hi planes (A16 / B16 given,
                                       // no lo planes, precision = bf16) whose result is still offered as planes (C16 / C16lo, lean_ok)
    int hi_result = 0;                 // ... and whose readers take the hi plane alone: C16lo is NOT written, planes_done / lean_ok go by
                                       // C16 (the caller marks the fp32 tensor "hi plane only")
    int b_pad_zero = 0;                // the columns of B behind N (up to ldb) hold zeros: a kernel may then compute (and write zeros
                                       // into) the pad columns of C up to round_up(N, 4)
    int* fp32_skipped = nullptr;       // (out) lean_ok was used: C was NOT written, the result lives in its planes only
    int lean_ok = 0;               // ... and nobody reads the fp32 C then: the kernel that writes both planes skips it
    int pp_force = 0;              // take the persistent ping-pong kernel wherever it CAN run (validity checks only), not only where
                                   // the selection measured on the AdeNet shapes says it pays: narrow outputs under very many rows
                                   // (the conv auto-encoder's 129024 x 152 x 2504), whose alternative is the register-staged kernel
    // split-K workspace of the persistent ping-pong kernel (partial tiles as plain stores + a reduce pass instead of
    // float atomics); without it large weight-gradient GEMMs stay on the atomic split-K kernels
    float* splitk_ws = nullptr; size_t splitk_ws_floats = 0;
};
int gemm(const GemmArgs& g, hipStream_t stream);
// streaming kernels for the skinny shapes (N <= 64 / K <= 64 / narrow weight gradients); *used says whether one took the launch
int gemm_skinny_try(const GemmArgs* gs, int n, hipStream_t stream, bool* used, bool dry);
constexpr int kMaxGemmGroups = 4;
// n <= kMaxGemmGroups problems of identical shape, layout and flags (different buffers) as ONE launch where the persistent kernel
// applies (their tiles share one list: fuller last round); otherwise n launches
int gemm_grouped(const GemmArgs* g, int n, hipStream_t stream);
bool gemm_planes_would_run(const GemmArgs* g, int n);      // bf16x3: would these problems run over their hi / lo planes?
// dst[i] = bf16(src[i]), n a multiple of 8
int to_bf16(const float* src, void* dst, size_t n, hipStream_t s);
// out[c * ldT + r] = bf16(W[r * ld + c]) for r < rows, c < cols (LDS-tiled transpose)
int transpose_to_bf16(const float* W, int rows, int cols, int ld, void* out, int ldT, hipStream_t s);
// A-stationary NT product, K <= 256, bf16-only output (gemm_bf16.hip)
bool gemm_nt_astat_takes(int M, int N, int K, int lda, int ldb, int ldc, const void* A16, const void* B16, const void* C16);
int gemm_nt_astat(const void* A16, int lda, const void* B16, int ldb, void* C16, int ldc, int M, int N, int K, hipStream_t s);
// several column sums in one launch (out[c] += sum_r in[r][c]); items by value in the kernel arguments
struct ColSumItem { const float* in; float* out; int ld, rows, cols, ctiles, splits, rps, block_end, ws_off; };
struct ColSumBatch { ColSumItem it[8]; int n = 0; float* ws = nullptr; };      // ws: deterministic mode's per-split partial sums
void col_sum_batch_add(ColSumBatch& b, const float* in, int ld, int rows, int cols, float* out);
int col_sum_batch(ColSumBatch& b, hipStream_t s);          // launches (if any item is pending) and clears the batch

// the same for a table of matrices in ONE launch (items: device array; block_end[k] = running total of 32x32 tiles)
// Internal activation code (never in an adn_config): the rectifier whose derivative AT a pre-activation of exactly zero is 0.5 --
// Theano's rectify is 0.5 (x + |x|), d|x|/dx = sgn(x), sgn(0) = 0 -- instead of this build's default 0 (adn_set_relu_grad_at_zero).
// Forward: a zero pre-activation leaves as -0.0f (a value equal to 0 in every product and sum downstream, and in the bf16 copy /
// the hi plane: bf16(-0.0f) = 0x8000), anything negative as +0.0f; backward: the mask reads y > 0 ? 1 : (y is -0.0 ? 0.5 : 0).
// Only the generic epilogues know the code (act_apply / act_grad_from_output): the specialised kernels decline it.
constexpr int kActRectifyHalf = 64;
struct TransposeItem { const float* W; void* out; int rows, cols, ld, ldT, block_end; };
int transpose_to_bf16_batch(const TransposeItem* dev_items, int n, int total_blocks, hipStream_t s, int lo_part = 0);
int split_hilo(const float* src, void* hi, void* lo, size_t n, hipStream_t s);   // fp32 -> the bf16x3 mode's two planes
constexpr int kMaxSplitJobs = 8;
int split_hilo_batch(const float* const* src, void* const* hi, void* const* lo, int n, size_t count, hipStream_t s);   // n tensors of `count` elements, one launch
int join_hilo(const void* hi, const void* lo, float* dst, size_t n, hipStream_t s);   // ... and back (hi + lo)
// frame compaction of the encoder path (compact.hip): row maps, gathers, the expand / compact-and-sum passes at the delta layer
int compact_build_maps(const int32_t* d_lens, const int32_t* d_prefix, int B, int T, int Z, int32_t* comp_of_full, int32_t* full_of_comp,
                       hipStream_t s);
int compact_gather_rows16(const void* src, int ld_src, void* dst, int ld_dst, const int32_t* full_of_comp, int Nc, int cols, hipStream_t s);
constexpr int kMaxGatherJobs = 8;
int compact_gather_rows_f32_batch(const float* const* src, void* const* dst_hi, void* const* dst_lo, int n, int ld_src, int ld_dst,
                                  const int32_t* full_of_comp, int Nc, int cols, hipStream_t s);     // n <= 4 float32 matrices, one launch
int compact_gather_rows16_batch(const void* const* src, void* const* dst, int n, int ld_src, int ld_dst, const int32_t* full_of_comp, int Nc, int cols,
                                hipStream_t s);     // n matrices of one geometry, one launch
// ... from float32 rows, converted on the way (dst_lo: the lo plane, or nullptr for the bf16 copy alone)
int compact_gather_rows_f32(const float* src, int ld_src, void* dst_hi, void* dst_lo, int ld_dst, const int32_t* full_of_comp, int Nc, int cols,
                            hipStream_t s);
int compact_check_padding32(const float* src, int ld_src, const int32_t* comp_of_full, int N, int cols, int Z, int* flag, int bit, hipStream_t s);
// raises `bit` of *flag when a padding row (comp_of_full[r] == Z) of the 16-bit matrix holds anything but zeros
int compact_check_padding16(const void* src, int ld_src, const int32_t* comp_of_full, int N, int cols, int Z, int* flag, int bit, hipStream_t s);
// the zero-input row's gradient from the per-utterance padding sums the delta layer's backward kernel left (DeltaJob::pad_partial):
// comp[zrow][0 .. ld) (+ its 16-bit copies), up to kMaxPadFinishJobs rows per launch
constexpr int kMaxPadFinishJobs = 4;
struct PadFinishJob { const float* partial; int nparts; float* comp; int ld; int cols; int zrow; void* c16; void* c16lo; };
int compact_pad_finish(const PadFinishJob* jobs, int n, hipStream_t s);

// ---------------------------------------------------------------------------------------
// element-wise / HBM-bound kernels (elementwise.hip)
// ---------------------------------------------------------------------------------------
// batch-major (B,T,F) -> time-major (T,B,[F|dF|ddF]) with optional delta/delta-delta append
// (out16 / din16: optional bf16 copy of the result, same leading dimension -- saves the separate conversion)
int delta_forward(const float* in, int ld_in, float* out, int ld_out, int B, int T, int F, int theta,
                  int append, hipStream_t s, void* out16 = nullptr);
// adjoint: time-major gradient (T,B,3F or F) -> batch-major (B,T,F)
int delta_backward(const float* dout, int ld_out, float* din, int ld_in, int B, int T, int F, int theta,
                   int append, hipStream_t s, void* din16 = nullptr);
// out[r][c] = sum_k alpha_k * in_k[r][c]  (alpha = device scalars or null for 1)
int sum_k(int n_in, const float* const* in, const float* const* alpha, int ld_in, float* out, int ld_out,
          int rows, int cols, hipStream_t s, void* out16 = nullptr, void* out16lo = nullptr);   // (out16lo: with out16 the hi / lo planes)
// out[r][c] = alpha[0] * in[r][c]   (alpha device scalar)
int scale_by(const float* in, int ld_in, const float* alpha, float* out, int ld_out, int rows, int cols,
             hipStream_t s);
// bf16: out[r][j*cols + c] = in_j[r][c] (n <= 4 inputs with the same ld; cols % 8 == 0)
// (in_lo / out_lo: the lo planes of the same matrices, concatenated by the same launch)
int concat_cols_bf16(int n, const void* const* in, int ld_in, void* out, int ld_out, int rows, int cols, hipStream_t s,
                     const void* const* in_lo = nullptr, void* out_lo = nullptr);
// dst[(j*rows_valid + k)*ld + c] += src[(j*rows_pad + k)*ld + c]
int add_row_blocks(const float* src, float* dst, int ld, int nblk, int rows_valid, int rows_pad, int cols, hipStream_t s);
int add_row_blocks_batch(const float* const* src, float* const* dst, int n, int ld, int nblk, int rows_valid, int rows_pad, int cols, hipStream_t s);   // n <= 4 pairs, one launch
// out[c] (+)= sum_r in[r][c]
int col_sum(const float* in, int ld, int rows, int cols, float* out, int accumulate, hipStream_t s);
// out[0] (+)= sum_{r,c} a[r][c]*b[r][c]
int dot_all(const float* a, int lda, const float* b, int ldb, int rows, int cols, float* out,
            float* scratch, hipStream_t s);
// dz = dy * act'(y) in place on dy
int act_backward(float* dy, int ld_dy, const float* y, int ld_y, int rows, int cols, int act, hipStream_t s);
// mask (B,T) uint8 batch-major -> (T,B) time-major; total[0] = number of valid frames
// lens / flag (optional): raises `bit` of *flag unless the mask is the prefix mask of lens (mask[b][t] != 0 <=> t < lens[b])
// tm_row0 / tm_T / tm_stride (optional, length buckets -- DeltaJob): the time-major copy goes to row tm_row0[b] + t tm_stride, t < tm_T[b]
int mask_prepare(const uint8_t* mask_bt, uint8_t* mask_tb, int B, int T, float* total, hipStream_t s, const int32_t* lens = nullptr,
                 int* flag = nullptr, int bit = 0, const int32_t* tm_row0 = nullptr, const int32_t* tm_T = nullptr, int tm_stride = 0);
// rows [0,B) of dst = vec (broadcast of a (1,H) init vector)
int broadcast_rows(const float* vec, float* dst, int ld, int rows, int cols, hipStream_t s);
// h[r][:] = hid, c[r][:] = cell for r < rows (pad columns 0), h16 = optional bf16 copy of h
int lstm_init_state_rows(const float* hid, const float* cell, float* h, float* c, void* h16, int ld, int rows, int cols, hipStream_t s);
// housekeeping of several same-shape tensors in ONE launch (the S input streams' delta layers, the LSTMs' initial states)
constexpr int kMaxDeltaJobs = 4, kMaxInitJobs = 12;      // (12: three stream LSTMs x four length buckets)
struct DeltaJob {
    const float* src; int ld_src; float* dst; int ld_dst; int F; int append; void* dst16;
    void* dst16lo = nullptr;              // bf16x3 / mixed: dst16 is the hi plane of the result, this its lo plane
    // frame compaction (compact.hip): the batch-major side is the compact matrix -- frame (b, t) in row row_map[b T + t].  Backward:
    // rows mapped to zrow (the padding frames) are summed per utterance into pad_partial[b][F] instead of stored
    const int32_t* row_map = nullptr; int zrow = -1; float* pad_partial = nullptr;
    // length buckets (model.hip, TmPlan): the time-major side keeps frame (b, t) in row tm_row0[b] + t * tm_stride for t < tm_T[b];
    // later frames of the utterance have no row (forward: not stored; backward: their gradient is zero).  Null: row t B + b.
    const int32_t* tm_row0 = nullptr; const int32_t* tm_T = nullptr; int tm_stride = 0;
};
struct LstmInitJob { const float* hid; const float* cell; float* h; float* c; void* h16; };
int delta_forward_batch(const DeltaJob* jobs, int n, int B, int T, int theta, hipStream_t s);    // dst = [x | dx | ddx] (append) or a copy
int delta_backward_batch(const DeltaJob* jobs, int n, int B, int T, int theta, hipStream_t s);   // src = d[x | dx | ddx], dst = dx
int lstm_init_state_batch(const LstmInitJob* jobs, int n, int ld, int rows, int cols, hipStream_t s);
// softmax classifier head + double-softmax temporal loss (custom/objectives.py:4-39)
//   z (T*B rows, time-major, ldz) -> probs_bt (B,T,C) batch-major dense (may be null),
//   row_loss[r] = -mask*log softmax(softmax(z))[y]  (if y != null), dz (may be null)
int softmax_loss(const float* z, int ldz, int B, int T, int C, const uint8_t* mask_tb, const int32_t* y_bt,
                 const float* total, float* probs_bt, float* row_loss, float* dz, int lddz, hipStream_t s, void* dz16 = nullptr,
                 const int32_t* bt_of_row = nullptr, int table_rows = 0,     // (bt_of_row: the frame b T + t of each of table_rows rows, -1 = none)
                 void* dz16lo = nullptr);                                    // (dz16lo: with dz16 the hi / lo planes of dz)
// out[0] = (sum_i v[i]) / total[0], fixed summation order
int reduce_loss(const float* v, int n, const float* total, float* out, hipStream_t s);
// p16: optional bf16 shadow of the parameters, written with the update
// poison / sticky (all three update rules): when *poison != 0 the kernel changes nothing and raises *sticky
int adam_update(float* p, const float* g, float* m, float* v, int64_t n, float a_t, float beta1,
                float beta2, float eps, hipStream_t s, void* p16 = nullptr, const float* poison = nullptr, int* sticky = nullptr,
                void* p16lo = nullptr);     // (p16lo: with p16 the hi / lo planes of the bf16x3 mode)
constexpr int kMaxAdamRanges = 16;
struct AdamRanges { int64_t begin[kMaxAdamRanges], end[kMaxAdamRanges]; };    // float offsets into the flat buffers (multiples of 8)
int adam_update_ranges(float* p, const float* g, float* m, float* v, const int64_t* begin, const int64_t* end, int n_ranges, float a_t,
                       float beta1, float beta2, float eps, hipStream_t s, void* p16, const float* poison, int* sticky, void* p16lo);
int poison_tail(const int* err_word, float* tail1, hipStream_t s, const float* loss = nullptr, float* tail0 = nullptr);
int copy_bench(const float* src, float* dst, int64_t n, int repeats, hipStream_t s, float* ms);
// lasagne.updates.sgd (momentum == 0) / momentum / nesterov_momentum; adadelta
int sgd_update(float* p, const float* g, float* vel, int64_t n, float lr, float momentum, int nesterov, hipStream_t s,
               const float* poison = nullptr, int* sticky = nullptr);
int adadelta_update(float* p, const float* g, float* accu, float* delta, int64_t n, float lr, float rho, float eps, hipStream_t s,
                    const float* poison = nullptr, int* sticky = nullptr);
// DropoutLayer on a time-major [T*B][ld] matrix that is the column block [off, off+cols) of a `width`-wide (B,T,width)
// tensor; in == out allowed; the mask is a hash of (seed, counter, layer, element) shared with the oracle
int dropout_apply(const float* in, int ld_in, float* out, int ld_out, int B, int T, int cols, int width, int off, float p,
                  uint32_t seed, uint32_t counter, uint32_t layer, hipStream_t s);
// last-timestep head: softmax + categorical cross-entropy over B rows (label = y_bt[b*T]); dz scaled by 1/total[0]
int softmax_ce(const float* z, int ldz, int B, int T, int C, const int32_t* y_bt, const float* total, float* probs,
               float* row_loss, float* dz, int lddz, hipStream_t s);

// ---------------------------------------------------------------------------------------
// BatchNormLayer over the rows of a [rows][C] matrix (batchnorm.hip)
// ---------------------------------------------------------------------------------------
constexpr float kBnEps = 1e-4f, kBnAlpha = 0.1f;          // lasagne.layers.BatchNormLayer defaults
size_t batchnorm_ws_bytes(int C);
int batchnorm_forward_train(const float* x, int ldx, float* y, int ldy, int rows, int C, const float* gamma, const float* beta,
                            float eps, float alpha, float* save_mean, float* save_inv_std, float* run_mean, float* run_inv_std,
                            void* ws, hipStream_t s, void* y16 = nullptr);
int batchnorm_forward_eval(const float* x, int ldx, float* y, int ldy, int rows, int C, const float* gamma, const float* beta,
                           const float* run_mean, const float* run_inv_std, hipStream_t s, void* y16 = nullptr);
// batch_stats: the forward pass used batch statistics (mean / inv_std = the saved ones); otherwise running averages
// (then they are constants of the graph and the mean / variance terms of the adjoint drop out)
int batchnorm_backward(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, int rows, int C, const float* gamma,
                       const float* mean, const float* inv_std, int batch_stats, float* dgamma, float* dbeta, void* ws,
                       hipStream_t s, void* dx16 = nullptr);

// ---------------------------------------------------------------------------------------
// LSTM recurrence (lstm.hip).  All matrices time-major; gate columns interleaved (unit, gate).
// ---------------------------------------------------------------------------------------
struct LstmStep {          // one LSTM instance taking part in a (possibly multi-LSTM) step launch
    const float* W_hid;    // [H][ldg]   W_hid[k][4*u+g]
    const float* peep;     // [3][ldh]   w_ci, w_cf, w_co  (null: no peepholes)
    const float* xproj;    // [T*B][ldg] x*W_in + b, time-major
    float* hbuf;           // [(T+1)*B][ldh]  see lstm.hip for the block convention
    float* cbuf;           // [(T+1)*B][ldh]
    float* gates;          // [T*B][ldg] post-activation i,f,g,o (saved for backward; may be null)
    // backward only
    float* dG;             // [T*B][ldg] clipped gradient wrt the gate pre-activations
    const float* dhs;      // [T*B][ld_dhs] gradient wrt the layer output
    int ld_dhs = 0;        // row stride of dhs (0: ldh) -- a column block of a wider matrix when the consumer was a concat
    float* dh_carry;       // [B][ldh]
    float* dc_state;       // [B][ldh]
    float* dpeep_part;     // [3][ldh] accumulated peephole-weight gradients (null: none)
    int backwards;
    // bf16 mode (null in f32 mode): bf16 copies the step kernels read / write directly
    const void* W_hid16T;  // [ldg][ldk] bf16: W_hid transposed (row = gate column, k contiguous, zero padded)
    const void* W_hid16;   // [H][ldg]   bf16: W_hid as stored (row = k of the forward, contiguous gate columns)
    const void* W_frag_fwd;  // W_hid in MFMA-fragment order for the persistent kernels (lstm_persistent.hip)
    const void* W_frag_bwd;
    const void* W_frag_fwd_lo = nullptr;   // bf16x3 mode: the fragment image of W_hid - bf16(W_hid) (W_frag_fwd then holds the hi part)
    const void* W_frag_bwd_lo = nullptr;   // ... of the backward image
    void* h16;             // [(T+1)*B][ldh] bf16 shadow of hbuf
    void* dG16;            // [T*B][ldg]     bf16 shadow of dG
    void* dG16lo = nullptr;  // bf16x3 mode: with dG16 the hi / lo planes of dG; the weight-stationary backward kernel then writes the
                             // planes INSTEAD of the fp32 matrix (same bytes; the caller's readers take the planes)
    int dG_fp32_off = 0;     // bf16 mode: the weight-stationary backward kernel writes dG16 only (every reader takes the bf16 copy)
    void* xchg = nullptr;  // exchange buffer of the weight-stationary kernels (lstm_cluster.hip), lstm_cluster_xchg_bytes(B)
    // HOST pointer: launch counter of `xchg`, owned by whoever owns the buffer and reset to 0 whenever the buffer is (re)made
    // and zeroed -- the bf16x3 kernel's 16-bit tags carry it (mod 64); not read on the device
    unsigned* xchg_seq = nullptr;
    // optional (backward): gradients that are plain sums of what the kernel already holds in registers -- the bias
    // (column sums of dG over all frames) and the learnt initial state (sums of dh_carry / dc_state over the batch).
    // Kernels that add them report it through lstm_backward()'s `sums_done`; otherwise the caller runs col_sum.
    // optional (forward, bf16 mode, H <= 256): the kernel computes the input projection itself (lstm_cluster.hip, KXS > 0) --
    // x16: the LSTM's input as a bf16 matrix [T*B][ld_x], time-major, Kx <= 160 features; W_in_frag: lstm_pack_win_frags' image of
    // W_in; b_in: the bias [ldg].  Offered together with xproj; lstm_forward_folds_projection() tells which one will be read.
    const void* x16 = nullptr; int ld_x = 0; int Kx = 0;
    const void* W_in_frag = nullptr;
    const float* b_in = nullptr;
    // deterministic mode (lstm.hip fills these in): [slots][det_stride] zeroed floats, one block
    // [dbias ldg | dhid_init ldh | dcell_init ldh | dpeep 3 ldh] per slot; a kernel adds what it would add atomically into the
    // block of ITS slot (the weight-stationary kernels: slot = 2 group + row tile; the others: slot = row slice = blockIdx.x) with
    // plain adds -- one writer per address -- and lstm_det_reduce sums the slots in order into the gradients
    float* det_ws = nullptr; int det_stride = 0;
    float* dbias = nullptr;      // [ldg]  += sum_{t,b} dG
    float* dhid_init = nullptr;  // [ldh]  += sum_b dh_carry
    float* dcell_init = nullptr; // [ldh]  += sum_b dc_state
    // optional (weight-stationary kernels at H <= 256 only -- lstm_cluster.hip): this entry is one LENGTH BUCKET of an LSTM (model.hip,
    // TmPlan): its pointers name the bucket's first row, it runs T_own steps (0: the launch's T) over mask_own ([T_own][B] bytes;
    // null: the launch's mask).  Every other kernel family rejects such entries (lstm_forward / lstm_backward check).
    int T_own = 0;
    const uint8_t* mask_own = nullptr;
};
constexpr int kMaxLstmPerLaunch = 12;      // (3 stream LSTMs x 4 length buckets; the launch descriptors stay below the 4 KB of kernel arguments)
// runs all T steps of n (<= kMaxLstmPerLaunch) independent LSTMs of identical (B,T,H) concurrently
int lstm_forward(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, int precision, hipStream_t s);
// W [H][ldg] fp32 -> out [ldg][ldk] bf16 (ldk = round_up(H,32)), zero padded
int lstm_pack_whid_t(const float* W, void* out, int H, hipStream_t s);
// persistent variants (lstm_persistent.hip): one launch for all T steps, bf16 mode, H <= 512
bool lstm_persistent_supported(int H);
size_t lstm_frag_elems(int H);
int lstm_pack_frags(const float* W, void* fwd, void* bwd, int H, hipStream_t s);
// the same for n <= 8 LSTMs in one launch
int lstm_pack_frags_batch(int n, const float* const* W, void* const* fwd, void* const* bwd, int H, hipStream_t s, int lo_part = 0);
int repack_rows_bf16_lo(int n, const float* const* in, void* const* out, int nblk, int rows, int rows_pad, int ld, hipStream_t s);
int repack_rows_bf16(int n, const float* const* in, void* const* out, int nblk, int rows, int rows_pad, int ld, hipStream_t s);
int lstm_forward_persistent(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, hipStream_t s);
int lstm_backward_persistent(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, hipStream_t s,
                             bool* sums_done = nullptr);
// weight-stationary variants (lstm_cluster.hip): groups of 4 workgroups share a 32-utterance slice, W_hid stays in LDS
// forward passes dispatched per kernel family since the library was loaded: [per-step launches, one-workgroup persistent,
// weight-stationary, weight-stationary bf16x3] (adn_debug_lstm_family_counts: lets a test see WHICH family ran)
extern long long g_lstm_family_forwards[4];
extern long long g_lstm_family_backwards[4];
bool lstm_cluster_supported(const LstmStep* l, int n, int B, int T, int H);
// true when lstm_forward(l, n, ...) will compute every LSTM's input projection inside the weight-stationary kernel (all of them
// offer x16 / W_in_frag / b_in with the same Kx <= 160): the caller then skips the projection GEMM, xproj is not read
bool lstm_forward_folds_projection(const LstmStep* l, int n, int B, int T, int H, int precision);
// W_in [Kx][ldg] fp32 (gate columns interleaved) -> MFMA B-fragment image [HP/16 unit tiles x 4 gates][KXS][64 lanes][8] bf16 of
// lstm_win_frag_elems(Kx, H) elements, zero beyond Kx features / H units; n <= 8 matrices per launch
size_t lstm_win_frag_elems(int Kx, int H);
int lstm_pack_win_frags(int n, const float* const* W_in, void* const* out, int Kx, int H, hipStream_t s);
size_t lstm_cluster_xchg_bytes(int B, int H);
int lstm_cluster_cus();               // CUs the weight-stationary launches are sized for (ADN_LSTM_CUS)
// whether lstm_forward / lstm_backward would run these entries on a kernel that takes LstmStep::T_own / mask_own (length buckets:
// the weight-stationary kernels at H <= 256 in the bf16 and bf16x3 arithmetics, outside the deterministic mode)
bool lstm_takes_length_buckets(const LstmStep* l, int n, int B, int T, int H, int precision, bool backward);
int lstm_forward_cluster(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, hipStream_t s);
bool lstm_cluster_x3_supported(const LstmStep* l, int n, int B, int T, int H);
int lstm_forward_cluster_x3(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, hipStream_t s);
bool lstm_cluster_x3w_supported(const LstmStep* l, int n, int B, int T, int H);      // 256 < H <= 512
int lstm_forward_cluster_x3w(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, hipStream_t s);
bool lstm_cluster_x3w_bwd_supported(const LstmStep* l, int n, int B, int T, int H);
int lstm_backward_cluster_x3w(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, hipStream_t s);
bool lstm_cluster_x3_bwd_supported(const LstmStep* l, int n, int B, int T, int H);
int lstm_backward_cluster_x3(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, hipStream_t s);
int lstm_backward_cluster(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, hipStream_t s);
int lstm_cluster_error_word(int** out);   // device word raised by a poll that gave up (checked at synchronisation)
static inline int lstm_ldk(int H) { return (int)((H + 31) / 32 * 32); }
// BPTT; on return dG holds d(gates) for every step, dh_carry / dc_state the gradient wrt the
// initial state (per batch row)
int lstm_backward(const LstmStep* l, int n, const uint8_t* mask_tb, int B, int T, int H, int precision, hipStream_t s,
                  bool* sums_done = nullptr);

// ---------------------------------------------------------------------------------------
// feature front-end on the GPU (prep.hip; reference utils/preprocessing.py)
// ---------------------------------------------------------------------------------------
int prep_seq_deltas(const float* in, int ld_in, float* out, int ld_out, const int* first, const int* last, int n_frames, int F,
                    int w, hipStream_t s);
int prep_diff_images(const float* in, float* out, int ld, const int* first, const int* last, int n_frames, int D, hipStream_t s);
int prep_mean_image_subtraction(const float* in, float* out, int ld, const int* starts, const int* lens, int n_utt, int D,
                                hipStream_t s);
int prep_normalize_rows(float* x, int ld, int rows, int cols, hipStream_t s);
int prep_column_stats(const float* x, int ld, int rows, int cols, double* ws, float* mean, float* std, hipStream_t s);
int prep_apply_column_norm(const float* x, float* out, int ld, int rows, int cols, const float* mean, const float* std,
                           hipStream_t s);
int prep_gather_columns(const float* in, int ld_in, float* out, int ld_out, const int* perm, int rows, int cols, hipStream_t s);
constexpr int kLcnMaxK = 15, kLcnMaxPerThread = 32;
int prep_lcn(const float* x, float* y, int n_images, int H, int W, const float* filter_host, int k, float threshold, hipStream_t s);

}  // namespace adn
