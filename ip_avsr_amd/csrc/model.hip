// adn_model: the graph executor behind the C ABI of include/adenet.h.
//
// One model = the whole reference graph:
//   x_s (B,T,D_s) -> [Dense+act]* -> DeltaLayer -> LSTM_s (or summed BLSTM)      s = 1..S
//   fuse (sum | adasum | concat) -> aggregation LSTM / summed BLSTM -> Dense(C)+softmax per frame
// plus temporal_softmax_loss, full back-propagation and Adam, all on one HIP stream.
//
// HBM layout
//   * parameters, gradients, Adam m and v are four flat fp32 buffers with IDENTICAL layout (one Adam
//     kernel sweeps them; the gradient buffer is what data-parallel ranks all-reduce).  Every matrix
//     has a leading dimension rounded up to 8 floats; pad columns are zero and stay zero.
//   * LSTM matrices are stored stacked with gate-interleaved columns (column 4u+g), see lstm.hip.
//     The Lasagne per-gate tensors (W_in_to_ingate, ...) exposed through adn_read/write_tensor are
//     strided views of those.
//   * activations: encoder side batch-major (B*T rows in the caller's (b,t) order), recurrent side
//     time-major (row t*B+b).  The delta kernel converts between the two for free.
//   * the workspace is one slab carved per (B,T) shape and zero-filled when the shape changes.
#include "adn_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace adn {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

static int g_deterministic = -1;                     // -1: not decided yet (ADN_DETERMINISTIC at first use)
bool deterministic() {
    if (g_deterministic < 0) { const char* e = getenv("ADN_DETERMINISTIC"); g_deterministic = (e && *e && *e != '0') ? 1 : 0; }
    return g_deterministic > 0;
}
void set_deterministic(bool on) { g_deterministic = on ? 1 : 0; }

// ------------------------------------------------------------------------------------------
// profiler
// ------------------------------------------------------------------------------------------
struct Profiler {
    struct Rec { hipEvent_t a, b; int cls; double flops, bytes; int launches; };
    bool enabled = false;
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    double ms[PROF_COUNT] = {0}, flops[PROF_COUNT] = {0}, bytes[PROF_COUNT] = {0};
    int64_t launches[PROF_COUNT] = {0};
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
    void drain() {                      // caller has synchronised the stream
        for (auto& r : recs) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) ms[r.cls] += t;
            flops[r.cls] += r.flops; bytes[r.cls] += r.bytes; launches[r.cls] += r.launches;
            pool.push_back(r.a); pool.push_back(r.b);
        }
        recs.clear();
    }
    void reset() {
        drain();
        for (int k = 0; k < PROF_COUNT; ++k) { ms[k] = flops[k] = bytes[k] = 0; launches[k] = 0; }
    }
    ~Profiler() { drain(); for (auto e : pool) (void)hipEventDestroy(e); }
};
static thread_local Profiler* g_prof = nullptr;

ProfScope::ProfScope(int cls, double flops, double bytes, hipStream_t s, int launches) : slot(-1), stream(s) {
    Profiler* p = g_prof;
    if (!p || !p->enabled) return;
    Profiler::Rec r{p->get(), p->get(), cls, flops, bytes, launches};
    (void)hipEventRecord(r.a, s);
    p->recs.push_back(r);
    slot = (int)p->recs.size() - 1;
}
ProfScope::~ProfScope() {
    if (slot >= 0 && g_prof) (void)hipEventRecord(g_prof->recs[slot].b, stream);
}

namespace {

// (gemm_*: by the layout the caller asked for; lstm_*: one time step of one group launch, whichever kernel family ran)
const char* const kProfNames[PROF_COUNT] = {"gemm_nn", "gemm_nt", "gemm_tn", "lstm_fwd_step",
                                            "lstm_bwd_step", "delta_fwd", "delta_bwd", "adam", "softmax_loss"};

constexpr float kBeta1 = 0.9f, kBeta2 = 0.999f, kEps = 1e-8f;
constexpr size_t kAuxFloats = 8;   // [0] = this rank's share of the cost (summed by the DP all-reduce)
const char* const kGateNames[4] = {"ingate", "forgetgate", "cell", "outgate"};

struct ParamDesc {
    std::string name;
    int ndim = 0;
    int64_t dims[2] = {1, 1};
    size_t off = 0;       // float offset of the element (0,0) inside the flat buffers
    int ld = 0;           // physical row stride
    int col_stride = 1;   // physical column step (4 for per-gate views of interleaved matrices)
    size_t phys_begin = 0, phys_end = 0;   // float range of the physical tensor this view lives in
    int64_t numel() const { return dims[0] * dims[1]; }
    int rows() const { return ndim == 2 ? (int)dims[0] : 1; }
    int cols() const { return ndim == 2 ? (int)dims[1] : (ndim == 1 ? (int)dims[0] : 1); }
};

struct LstmParams {      // physical tensors (float offsets into the flat buffers)
    char* whid16t = nullptr;   // bf16 mode: transposed bf16 copy of W_hid, refreshed with the parameter shadow
    char* wfrag_fwd_lo = nullptr;   // bf16x3 mode: fragment image of W_hid - bf16(W_hid) (wfrag_fwd = the hi part)
    char* wfrag_bwd_lo = nullptr;   // ... of the backward image
    char* wfrag_fwd = nullptr; // ... and the two MFMA-fragment-ordered copies the persistent kernels stream
    char* wfrag_bwd = nullptr;
    char* win_frag = nullptr;  // bf16 mode, stream LSTMs of <= 160 input features: MFMA-fragment image of W_in for the kernels that
                               // compute their own input projection (lstm_pack_win_frags)
    char* wcat16 = nullptr;    // concat consumers: bf16 W_in with every input block padded to ldh rows ([S*ldh][ldg])
    char* wcat16lo = nullptr;  // ... its lo plane (bf16x3 mode)
    int fin = 0;
    size_t W_in = 0, W_hid = 0, b = 0, peep = 0, cell_init = 0, hid_init = 0;
    bool peepholes = false;
    bool backwards = false;
};

// Length buckets: the time-major (recurrent) side of a train step without most of its padding frames.  The utterances of the call,
// sorted by length, are cut into nb buckets of Bb = ceil(B / nb); bucket k keeps Tk[k] steps (its longest utterance) of Bb rows.  The
// buckets follow one another along the TIME axis -- bucket k starts at block c[k] = sum_{k' < k} (Tk[k'] + 1), one spare block behind
// each -- so every time-major tensor is an ordinary [Tt][Bb] one with Tt = c[nb] - 1: GEMMs, sums, concat and the loss only see a
// row count (Bb Tt instead of B T: 15 860 instead of 20 800 at the bench lengths), the state histories keep their T + 1 block
// convention PER BUCKET (block c[k] / c[k] + Tk[k] is the bucket's initial state), and the weight-stationary LSTM kernels run each
// (LSTM, bucket) as an entry of its own -- the bucket's first row in every pointer, LstmStep::T_own steps, its own rows of the mask.
// Rows no frame lives in (the spare blocks, the slots behind the last utterance of the last bucket): mask 0, zero rows in every
// gradient matrix (softmax_loss writes them for dz, zero_spare_blocks() for the gate gradients the LSTM kernels never touch there),
// finite anything elsewhere.  Frames (b, t >= Tk[bucket of b]) have no row at all: padding of every utterance of the bucket.
constexpr int kMaxBuckets = 4;
struct TmPlan {
    bool on = false;
    int nb = 1, Bb = 0, Tt = 0, Tmax = 0;
    int c[kMaxBuckets + 1] = {0, 0, 0, 0, 0};
    int Tk[kMaxBuckets] = {0, 0, 0, 0};
};

struct LstmWork {        // per-shape workspace pointers
    float *xproj = nullptr, *gates = nullptr, *dG = nullptr, *hbuf = nullptr, *cbuf = nullptr;
    float *dh_carry = nullptr, *dc_state = nullptr;
    void* xchg = nullptr;    // lstm_cluster.hip exchange buffer
    unsigned xchg_seq = 0;   // launches of the bf16x3 kernel on `xchg` since it was carved (and zeroed with the slab)
    float* out(int B, int ldh, bool backwards) const { return hbuf + (backwards ? 0 : (size_t)B * ldh); }
    float* prev(int B, int ldh, bool backwards) const { return hbuf + (backwards ? (size_t)B * ldh : 0); }
};

struct StreamState {
    adn_stream_config cfg;
    std::vector<size_t> encW, encb;   // offsets
    std::vector<int> enc_in;          // input width of each encoder layer
    int feat_dim = 0;                 // LSTM input width (3E / E / 3D / D, + aux_dim)
    int delta_dim = 0;                // ... of which the delta layer produces the first columns
    int enc_out = 0;                  // width entering the delta layer
    size_t bn_beta = 0, bn_gamma = 0, bn_mean = 0, bn_inv_std = 0;   // BatchNormLayer behind the encoder (cfg.batchnorm)
    float *bn_out = nullptr, *bn_save_mean = nullptr, *bn_save_inv_std = nullptr; char* bn_ws = nullptr;
    float* aux_stage = nullptr;       // staged auxiliary input (batch-major, ld_of(aux_dim))
    std::vector<LstmParams> lstm;     // 1 or 2
    // workspace
    const float* x = nullptr; int ldx = 0;     // staged input (batch-major)
    float* xstage = nullptr;
    void* x16 = nullptr;                       // bf16 copy of x when x is the caller's own device buffer
    void* x16lo = nullptr;                     // ... and its lo plane (bf16x3 mode)
    bool x_convert_pending = false;            // float32 device input: its 16-bit copies (x16 / x16lo) are not made yet -- a compacted call
                                               // converts while it gathers (one pass over the valid rows), any other call converts in full
    std::vector<float*> act;                   // encoder activations (batch-major)
    // rectifier bit images (gemm_common.h GemmGroup::Cbits): per layer whose output is rectified and feeds another layer, what the
    // forward epilogue left for the input-gradient epilogue of the same tile grid (bits_tiles: that grid, 0 = not written this pass)
    std::vector<char*> relu_bits; std::vector<int> bits_tiles;
    float* feat = nullptr;                     // time-major LSTM input
    std::vector<LstmWork> lw;
    float* hsum = nullptr;                     // stream output when bidirectional (else alias of lw[0].out)
    float* dout_buf = nullptr;                 // own buffer for the gradient wrt the stream output
    float* out_drop = nullptr;                 // stream output after the fused-tensor dropout (concat / single stream)
    float *pingA = nullptr, *pingB = nullptr;  // dZ ping-pong of this stream's encoder back-propagation (own copy: the
    int ping_ld = 0;                           // streams back-propagate concurrently on their own HIP streams)
    float* colsum_ws = nullptr; size_t colsum_ws_floats = 0;
    int dout_ld = 0;                           // row stride of `dout` (0: ldh)
    float* dout = nullptr;                     // ... the buffer actually holding it (may be a shared one)
    float* dfeat = nullptr;
    float* dE = nullptr;
    // frame compaction (compact.hip): the first layer's operand gathered to the valid frames + one zero row, the encoder output
    // expanded back to B T rows for the delta layer, the delta layer's gradient compacted (padding rows summed into the zero row)
    float* xc = nullptr; float* dEc = nullptr; float* compact_ws = nullptr;
    float* out_ptr = nullptr;
    size_t param_begin = 0;                    // this stream's parameters start here in the flat buffers
};

}  // namespace
}  // namespace adn

using namespace adn;

struct adn_model {
    adn_config cfg;
    hipStream_t stream = nullptr;
    int H = 0, C = 0, ldh = 0, ldg = 0, ldc = 0, S = 0;

    std::vector<ParamDesc> params;
    size_t flat_floats = 0;
    float* flat[4] = {nullptr, nullptr, nullptr, nullptr};   // param, grad, m, v
    int adam_t = 0;
    bool grads_valid = false;
    int* poison_sticky = nullptr;      // device word: an optimiser kernel skipped its update because a rank's gradients were invalid
    // the word behind it: what the device found wrong with a call's inputs (frame compaction, compact.hip).  kInputLens: the mask is not
    // the prefix mask of the announced lengths; kInputPadding: a padding frame of a stream input is not zero
    int* input_flags() const { return poison_sticky + 1; }
    float* poison_word() const { return flat[ADN_BUF_GRAD] + flat_floats + 1; }      // tail[1] of the gradient buffer
    // stochastic layers: masks are a hash of (seed, counter, layer, element); `stochastic` is set per call
    uint32_t drop_seed = 0x5EED1234u, drop_counter = 0;
    bool stochastic = false;
    bool training = false;            // this pass is get_output(deterministic=False): BatchNorm uses batch statistics
    int stream_units() const { return cfg.stream_lstm_units > 0 ? cfg.stream_lstm_units : cfg.lstm_size; }
    int n_aux() const { int n = 0; for (int k = 0; k < cfg.n_streams; ++k) n += cfg.streams[k].aux_dim > 0; return n; }
    bool head_last() const { return cfg.head == ADN_HEAD_LAST; }
    bool has_dropout() const {
        if (cfg.agg_dropout_p > 0.f) return true;
        for (int k = 0; k < cfg.n_streams; ++k) if (cfg.streams[k].dropout_p > 0.f) return true;
        return false;
    }

    std::vector<StreamState> st;
    std::vector<LstmParams> agg;       // 0, 1 or 2
    std::vector<LstmWork> aggw;
    size_t tail_begin = 0;             // first float of the [fuse | agg | softmax] parameters
    std::vector<hipEvent_t> bucket_events;   // caller-owned; in the order of adn_grad_buckets (bucket_ranges)
    // bf16x3 over planes: results whose fp32 copy was not written (their readers take the planes); a reader that needs fp32 after
    // all -- a GEMM the ping-pong kernel declines, adn_read_encoder_activation -- gets hi + lo written back first
    std::vector<std::pair<const float*, size_t>> fp32_stale;
    // housekeeping launches of one phase queued for ONE batched launch each (the S streams' delta layers, the LSTMs' initial
    // states): flushed ahead of the first reader
    std::vector<DeltaJob> delta_q; int delta_q_theta = 0; bool delta_q_fwd = true;
    std::vector<LstmInitJob> init_q;
    int dp_order = -1;                       // -1: not latched yet; 0: layer-major buckets / back-propagation, 1: stream-major
    float adam_a_t = 0.f; bool adam_open = false;   // step size of the optimiser step opened by adn_adam_begin
    size_t adacoeff = 0;               // S scalars (one 8-float block)
    size_t smW = 0, smb = 0;
    int fused_dim = 0;

    // workspace slab
    char* slab = nullptr;
    size_t slab_bytes = 0;
    int wsB = 0, wsT = 0;
    bool ws_host_inputs = false;
    // shared workspace tensors
    uint8_t *mask_bt = nullptr, *mask_tb = nullptr;
    int32_t* y_bt = nullptr;
    const int32_t* y_src = nullptr;          // the call's targets as the kernels read them: the caller's device array in place, or y_bt
    float *total = nullptr, *loss = nullptr, *row_loss = nullptr, *probs_bt = nullptr;
    float *z = nullptr, *dz = nullptr, *cls_in = nullptr, *dcls = nullptr, *fused = nullptr, *dfused = nullptr;
    // The S input streams are independent up to the fusion (and again below it in back-propagation): optionally each
    // runs on its own HIP stream, forked from / joined into the model's stream with events (see streams_concurrent).
    hipStream_t side[ADN_MAX_STREAMS] = {};
    hipEvent_t fork_ev = nullptr, join_ev[ADN_MAX_STREAMS] = {};
    bool side_ready = false;
    // concat fusion in bf16 mode: the aggregation LSTMs read ONE materialised [N][S*ldh] bf16 matrix, so their input
    // projection, dW_in and the gradient wrt the concat are one GEMM each per LSTM instead of S
    char* cat16 = nullptr; float* dcat = nullptr; float* wcat_tmp = nullptr; size_t wcat_tmp_slots = 1;
    char* cat16lo = nullptr;           // bf16x3 mode: the concat's lo plane
    bool cat_planes_ok = false;        // bf16x3 mode: the last forward pass took the concat path (all of its GEMMs run over planes)
    // partial slabs of the split-K weight-gradient GEMMs (gemm_bf16_pp_kernel): one workgroup = one 256 x 256 fp32 tile
    float* splitk_ws = nullptr; size_t splitk_ws_floats = 0;
    int lastB = 0, lastT = 0;
    // frame compaction: lengths announced by adn_set_batch_lengths (host), the row maps of the batch on the device, Nc = valid + 1
    std::vector<int32_t> batch_lens, maps_lens;
    std::vector<const float*> ones_cols;      // matrices of the slab whose first pad column holds 1.0 (bias gradients on the dW GEMM)
    std::vector<int32_t> call_lens;          // the lengths of the call in flight: the announcement, or read off a host mask
    bool call_lens_auto = false;             // ... read off the mask: nobody promised zero padding frames, the device looks first
    bool auto_compact = true;                // adn_set_auto_compaction
    // adn_set_relu_grad_at_zero(0.5): the encoders' rectifiers run as kActRectifyHalf (adn_common.h) -- the generic epilogues
    bool relu0_half = false;
    int act_code(int act) const { return (relu0_half && act == ADN_ACT_RECTIFY) ? kActRectifyHalf : act; }
    struct PinSlot { int32_t* host = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; };
    PinSlot pin[4]; int pin_next = 0;        // pinned staging of lengths + prefix sums (a ring: the copies are asynchronous)
    int32_t *d_lens = nullptr, *d_prefix = nullptr, *comp_of_full = nullptr, *full_of_comp = nullptr;
    std::vector<int32_t> h_comp_of_full, h_prefix;
    int Nc = 0; bool compact = false; int maps_T = 0;
    Profiler prof;
    // bf16 shadow copies of every GEMM operand (bf16 mode): fp32 range -> bf16 buffer with the same layout
    struct ShadowRange { const float* base; size_t n; char* shadow; char* shadow_lo; };
    std::vector<ShadowRange> shadows;
    std::vector<const float*> fused_in;     // what the aggregation layer read in the last forward pass (after dropout)
    char* params16 = nullptr;
    char* params16lo = nullptr;           // bf16x3 mode: bf16(p - bf16(p)), the lo plane of the parameter buffer (params16 = hi)
    bool params16_dirty = true;
    bool params16_values_fresh = false;   // Adam wrote the bf16 parameter shadow itself; only the derived images are stale
    void mark_params_dirty() { params16_dirty = true; params16_values_fresh = false; }
    // transposed bf16 copies of the weights that input-gradient GEMMs read as "B given [N][K]": with W^T [K][N]
    // at hand the same product runs through the faster k-strided-B kernel (measured 1.2-1.4x)
    struct TransW { const float* key; char* buf; int ldT; bool lo_fwd; };   // lo_fwd: a forward skinny problem may read the lo plane
    std::vector<TransW> transw;
    char* transw_slab = nullptr;
    size_t transw_slab_bytes = 0;               // one plane; bf16x3 mode keeps a second (lo) plane right behind it
    TransposeItem* transw_items = nullptr;      // device table for the one-launch refresh
    TransposeItem* transw_items_lo = nullptr;   // ... of the lo planes
    TransposeItem* transw_items_lo_fwd = nullptr;   // ... of the lo planes a FORWARD product reads (mixed arithmetic: the others are never read)
    int transw_n_lo_fwd = 0, transw_blocks_lo_fwd = 0;
    int transw_blocks = 0;
    bool packed_for_persistent = false;         // which LSTM weight images the last refresh produced
    // ADN_PRECISION_MIXED: cfg.precision is ADN_PRECISION_BF16X3 and back-propagation's GEMMs take the hi planes only (m_gemm)
    bool bwd_hi_only = false;
    bool in_backward = false;
    // length buckets (TmPlan): asked for by the train-step entry points, decided per call by setup_buckets(); the tables live in the slab
    TmPlan tm; bool want_buckets = false, buckets_allowed = true, probs_partial = false;
    int32_t *tm_row0 = nullptr, *tm_T = nullptr, *tm_bt = nullptr;      // [B] first row / steps of an utterance; [rows] frame of a row (-1: none)
    TmPlan tm_cached;                                                  // the plan the tables on the device were made for ...
    std::vector<int32_t> tm_lens; int tm_key_T = 0, tm_key_prec = -1, tm_key_mixed = -1;     // ... and the call it was made for
    PinSlot pin_tm[4]; int pin_tm_next = 0;
    int xchg_key = 0;                       // utterances per launch entry the LSTM exchange buffers were last used with (0: the whole batch)
    bool bf16() const { return cfg.precision == ADN_PRECISION_BF16; }
    // bf16x3 mode keeps TWO bf16 planes (hi = bf16(x), lo = bf16(x - hi)) of every GEMM operand -- written once per tensor by a
    // split pass behind its producer -- and its large GEMMs run over the planes (three K-segments in the ping-pong kernel)
    // instead of over [hi | hi | lo] / [hi ; lo ; hi] images made for every single use (ADN_X3_NO_PLANES: the image path)
    bool planes() const { return cfg.precision == ADN_PRECISION_BF16X3 && !getenv("ADN_X3_NO_PLANES"); }

    // bf16x3 mode is fp32 mode everywhere but inside gemm(): fp32 activations, no shadows, the fp32 recurrent kernels
    int lstm_precision() const { return bf16() ? ADN_PRECISION_BF16 : ADN_PRECISION_F32; }
    bool keep_fp32 = false;        // debug: also write the fp32 copies that bf16 mode normally skips
    void* shadow_of(const float* p) const {
        if (!p) return nullptr;
        if (p >= flat[ADN_BUF_PARAM] && p < flat[ADN_BUF_PARAM] + flat_floats)
            return params16 ? params16 + 2 * (size_t)(p - flat[ADN_BUF_PARAM]) : nullptr;
        for (const auto& st_ : st)
            if (st_.x16 && p == st_.x) return st_.x16;
        for (const auto& r : shadows)
            if (p >= r.base && p < r.base + r.n) return r.shadow + 2 * (size_t)(p - r.base);
        return nullptr;
    }
    void* shadow_lo_of(const float* p) const {
        if (!p) return nullptr;
        if (p >= flat[ADN_BUF_PARAM] && p < flat[ADN_BUF_PARAM] + flat_floats)
            return params16lo ? params16lo + 2 * (size_t)(p - flat[ADN_BUF_PARAM]) : nullptr;
        for (const auto& st_ : st)
            if (st_.x16lo && p == st_.x) return st_.x16lo;
        for (const auto& r : shadows)
            if (p >= r.base && p < r.base + r.n) return r.shadow_lo + 2 * (size_t)(p - r.base);
        return nullptr;
    }

    float* P(size_t off) const { return flat[ADN_BUF_PARAM] + off; }
    float* G(size_t off) const { return flat[ADN_BUF_GRAD] + off; }
};

namespace adn {
namespace {

// ------------------------------------------------------------------------------------------
// parameter table
// ------------------------------------------------------------------------------------------
struct Builder {
    adn_model* m;
    size_t cursor = 0;
    std::vector<std::pair<size_t, size_t>> ranges;   // physical tensors handed out so far
    size_t alloc(size_t floats) {
        size_t o = cursor; cursor += (size_t)round_up((int64_t)floats, 8);
        ranges.push_back({o, cursor});
        return o;
    }
    void add(const std::string& name, int ndim, int64_t d0, int64_t d1, size_t off, int ld, int col_stride = 1) {
        ParamDesc p;
        p.name = name; p.ndim = ndim; p.dims[0] = d0; p.dims[1] = d1; p.off = off; p.ld = ld; p.col_stride = col_stride;
        for (auto& r : ranges) if (off >= r.first && off < r.second) { p.phys_begin = r.first; p.phys_end = r.second; }
        if (ndim < 2) p.dims[1] = 1;
        if (ndim < 1) p.dims[0] = 1;
        m->params.push_back(p);
    }
    // Lasagne registration order inside an LSTMLayer
    // units / fin_view: logical sizes of the parameter views when they are narrower than the physical tensors
    // (adn_config.stream_lstm_units): the surplus rows / columns are never written and stay zero
    LstmParams add_lstm(const std::string& prefix, int fin, bool peep, bool backwards, int units = 0, int fin_view = 0) {
        const int ldg = m->ldg, ldh = m->ldh;
        const int H = units > 0 ? units : m->H;
        const int fin_phys = fin;
        if (fin_view > 0) fin = fin_view;
        LstmParams lp;
        lp.fin = fin_phys; lp.peepholes = peep; lp.backwards = backwards;
        lp.W_in = alloc((size_t)fin_phys * ldg);
        lp.W_hid = alloc((size_t)m->H * ldg);
        lp.b = alloc(ldg);
        if (peep) lp.peep = alloc((size_t)3 * ldh);
        lp.cell_init = alloc(ldh);
        lp.hid_init = alloc(ldh);
        for (int g = 0; g < 4; ++g) {
            add(prefix + ".W_in_to_" + kGateNames[g], 2, fin, H, lp.W_in + g, ldg, 4);
            add(prefix + ".W_hid_to_" + kGateNames[g], 2, H, H, lp.W_hid + g, ldg, 4);
            add(prefix + ".b_" + kGateNames[g], 1, H, 1, lp.b + g, ldg, 4);
        }
        if (peep) {
            add(prefix + ".W_cell_to_ingate", 1, H, 1, lp.peep, ldh);
            add(prefix + ".W_cell_to_forgetgate", 1, H, 1, lp.peep + ldh, ldh);
            add(prefix + ".W_cell_to_outgate", 1, H, 1, lp.peep + 2 * (size_t)ldh, ldh);
        }
        add(prefix + ".cell_init", 2, 1, H, lp.cell_init, ldh);
        add(prefix + ".hid_init", 2, 1, H, lp.hid_init, ldh);
        return lp;
    }
};

int validate(const adn_config& c) {
    ADN_CHECK(c.n_streams >= 1 && c.n_streams <= ADN_MAX_STREAMS, ADN_ERR_INVALID, "n_streams out of range");
    ADN_CHECK(c.lstm_size >= 1 && c.lstm_size <= 4096, ADN_ERR_INVALID, "lstm_size out of range");
    ADN_CHECK(c.classes >= 1 && c.classes <= ADN_MAX_CLASSES, ADN_ERR_INVALID, "classes out of range");
    ADN_CHECK(c.fusion >= ADN_FUSE_NONE && c.fusion <= ADN_FUSE_CONCAT, ADN_ERR_INVALID, "unknown fusion type");
    ADN_CHECK(c.head == ADN_HEAD_FRAMES || c.head == ADN_HEAD_LAST, ADN_ERR_INVALID, "unknown classifier head");
    ADN_CHECK(c.agg_dropout_p >= 0.f && c.agg_dropout_p < 1.f, ADN_ERR_INVALID, "dropout probability must be in [0, 1)");
    for (int k = 0; k < c.n_streams; ++k)
        ADN_CHECK(c.streams[k].dropout_p >= 0.f && c.streams[k].dropout_p < 1.f, ADN_ERR_INVALID,
                  "dropout probability must be in [0, 1)");
    ADN_CHECK(c.agg >= 0 && c.agg <= 2, ADN_ERR_INVALID, "agg must be 0, 1 or 2");
    ADN_CHECK(c.precision >= ADN_PRECISION_F32 && c.precision <= ADN_PRECISION_MIXED, ADN_ERR_INVALID,
              "unsupported precision");
    if (c.fusion == ADN_FUSE_NONE) ADN_CHECK(c.n_streams == 1, ADN_ERR_INVALID, "fusion 'none' needs exactly one stream");
    if (c.fusion == ADN_FUSE_CONCAT && c.n_streams > 1)
        ADN_CHECK(c.agg != 0, ADN_ERR_INVALID, "concat fusion needs an aggregation LSTM");
    ADN_CHECK(c.stream_lstm_units >= 0 && c.stream_lstm_units <= c.lstm_size, ADN_ERR_INVALID,
              "stream_lstm_units must be 0 or at most lstm_size");
    if (c.stream_lstm_units > 0 && c.stream_lstm_units < c.lstm_size)
        ADN_CHECK(c.fusion != ADN_FUSE_CONCAT && c.agg > 0, ADN_ERR_INVALID,
                  "narrower stream LSTMs need an aggregation LSTM and a fusion other than concat");
    int n_sub = 0;
    for (int s = 0; s < c.n_streams; ++s) {
        const adn_stream_config& sc = c.streams[s];
        ADN_CHECK(sc.input_dim >= 1, ADN_ERR_INVALID, "stream input_dim must be positive");
        ADN_CHECK(sc.aux_dim >= 0 && sc.aux_dim <= 65536, ADN_ERR_INVALID, "aux_dim out of range");
        ADN_CHECK(!sc.batchnorm || sc.n_enc > 0, ADN_ERR_INVALID, "the BatchNorm layer sits behind an encoder");
        ADN_CHECK(sc.n_enc >= 0 && sc.n_enc <= ADN_MAX_ENC_LAYERS, ADN_ERR_INVALID, "n_enc out of range");
        for (int l = 0; l < sc.n_enc; ++l) {
            ADN_CHECK(sc.enc_units[l] >= 1, ADN_ERR_INVALID, "encoder layer width must be positive");
            ADN_CHECK(sc.enc_act[l] >= ADN_ACT_LINEAR && sc.enc_act[l] <= ADN_ACT_SCALED_TANH_LECUN, ADN_ERR_INVALID,
                      "unsupported encoder nonlinearity");
        }
        n_sub += sc.bidirectional ? 2 : 1;
    }
    ADN_CHECK(n_sub <= 2 * ADN_MAX_STREAMS, ADN_ERR_INVALID, "too many stream LSTMs");
    return ADN_OK;
}

int build_params(adn_model* m) {
    const adn_config& c = m->cfg;
    Builder b{m};
    m->st.resize(m->S);
    for (int s = 0; s < m->S; ++s) {
        StreamState& st = m->st[s];
        st.cfg = c.streams[s];
        st.param_begin = b.cursor;
        const std::string sp = "stream" + std::to_string(s);
        int d = st.cfg.input_dim;
        for (int l = 0; l < st.cfg.n_enc; ++l) {
            const int u = st.cfg.enc_units[l], ld = ld_of(u);
            const size_t w = b.alloc((size_t)d * ld), bb = b.alloc(ld);
            st.encW.push_back(w); st.encb.push_back(bb); st.enc_in.push_back(d);
            b.add(sp + ".enc" + std::to_string(l) + ".W", 2, d, u, w, ld);
            b.add(sp + ".enc" + std::to_string(l) + ".b", 1, u, 1, bb, ld);
            d = u;
        }
        st.enc_out = d;
        if (st.cfg.batchnorm) {                   // lasagne registration order: beta, gamma, mean, inv_std
            const int ld = ld_of(d);
            st.bn_beta = b.alloc(ld); st.bn_gamma = b.alloc(ld); st.bn_mean = b.alloc(ld); st.bn_inv_std = b.alloc(ld);
            b.add(sp + ".bn.beta", 1, d, 1, st.bn_beta, ld);
            b.add(sp + ".bn.gamma", 1, d, 1, st.bn_gamma, ld);
            b.add(sp + ".bn.mean", 1, d, 1, st.bn_mean, ld);
            b.add(sp + ".bn.inv_std", 1, d, 1, st.bn_inv_std, ld);
        }
        st.delta_dim = st.cfg.use_delta ? 3 * d : d;
        st.feat_dim = st.delta_dim + st.cfg.aux_dim;
        const int ndir = st.cfg.bidirectional ? 2 : 1;
        const int units = m->stream_units() < m->H ? m->stream_units() : 0;
        for (int k = 0; k < ndir; ++k)
            st.lstm.push_back(b.add_lstm(sp + ".lstm" + std::to_string(k), st.feat_dim, st.cfg.peepholes != 0, k == 1, units));
    }
    m->tail_begin = b.cursor;                 // fusion coefficients, aggregation LSTMs, classifier
    if (c.fusion == ADN_FUSE_ADASUM) {
        m->adacoeff = b.alloc(8);
        for (int s = 0; s < m->S; ++s) b.add("fuse.adacoeff" + std::to_string(s), 0, 1, 1, m->adacoeff + s, 1);
    }
    m->fused_dim = (c.fusion == ADN_FUSE_CONCAT) ? m->S * m->H : m->H;
    for (int k = 0; k < c.agg; ++k)     // (narrower stream LSTMs: only the first stream_units rows of W_in are exposed)
        m->agg.push_back(b.add_lstm("agg" + std::to_string(k), m->fused_dim, c.agg_peepholes != 0, k == 1, 0,
                                    m->stream_units() < m->H ? m->stream_units() : 0));
    m->smW = b.alloc((size_t)m->H * m->ldc);
    m->smb = b.alloc(m->ldc);
    b.add("softmax.W", 2, m->H, m->C, m->smW, m->ldc);
    b.add("softmax.b", 1, m->C, 1, m->smb, m->ldc);
    m->flat_floats = b.cursor;
    return ADN_OK;
}

// ------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------
struct Carver {
    char* base; size_t cursor = 0;
    template <typename T> T* take(size_t count) {
        T* p = base ? reinterpret_cast<T*>(base + cursor) : nullptr;
        cursor += (size_t)round_up((int64_t)(count * sizeof(T)), 256);
        return p;
    }
};

// carve an fp32 matrix that is consumed by GEMMs together with its bf16 shadow
float* take_shadowed(adn_model* m, Carver& cv, size_t floats) {
    floats = (size_t)round_up((int64_t)floats, 8);
    float* p = cv.take<float>(floats);
    char* sh = cv.take<char>(floats * 2);
    char* lo = cv.take<char>(floats * 2);
    if (p) m->shadows.push_back({p, floats, sh, lo});
    return p;
}

void carve_lstm(adn_model* m, Carver& cv, LstmWork& w, int B, int T, int ldh, int ldg) {
    const size_t N = (size_t)B * T;
    w.xproj = cv.take<float>(N * ldg);
    w.gates = cv.take<float>(N * ldg);
    w.dG = take_shadowed(m, cv, N * ldg);
    w.hbuf = take_shadowed(m, cv, (size_t)(T + 1) * B * ldh);
    w.cbuf = cv.take<float>((size_t)(T + 1) * B * ldh);
    // (length buckets: nb Bb <= B + nb - 1 utterance slots, and up to one more 32-utterance group per bucket in the exchange buffer)
    w.dh_carry = cv.take<float>((size_t)(B + 2 * kMaxBuckets) * ldh);
    w.dc_state = cv.take<float>((size_t)(B + 2 * kMaxBuckets) * ldh);
    w.xchg = cv.take<char>(lstm_cluster_xchg_bytes(B + 32 * kMaxBuckets, m->H));
    w.xchg_seq = 0;          // ensure_workspace zeroes the whole slab behind every carve: all tag slots read 0
}

// frame compaction (compact.hip): whether a call of N rows could run compacted at all -- decides what carve() takes for it
constexpr int kInputLens = 1, kInputPadding = 2;
static int64_t compact_min_rows() { const char* e = getenv("ADN_COMPACT_MIN_ROWS"); return e ? atoll(e) : 8192; }
bool compaction_possible(const adn_model* m, size_t N) {
    static const bool off = getenv("ADN_NO_COMPACT") != nullptr;
    return !off && m->cfg.precision != ADN_PRECISION_F32 && (int64_t)N >= compact_min_rows();
}

size_t carve(adn_model* m, char* base, int B, int T, bool host_inputs) {
    Carver cv{base};
    m->shadows.clear();
    m->fp32_stale.clear();
    m->ones_cols.clear();
    const size_t N = (size_t)B * T;
    const int ldh = m->ldh, ldg = m->ldg;
    m->mask_bt = cv.take<uint8_t>(N);
    m->mask_tb = cv.take<uint8_t>(N);
    m->d_lens = cv.take<int32_t>((size_t)B); m->d_prefix = cv.take<int32_t>((size_t)B + 1);
    m->comp_of_full = cv.take<int32_t>(N); m->full_of_comp = cv.take<int32_t>(N + 8);
    m->maps_lens.clear(); m->compact = false;          // (the maps live in the slab: re-made after every carve)
    m->tm_row0 = cv.take<int32_t>((size_t)B); m->tm_T = cv.take<int32_t>((size_t)B); m->tm_bt = cv.take<int32_t>(N + 8);
    m->tm_lens.clear(); m->tm = TmPlan{}; m->tm_cached = TmPlan{}; m->xchg_key = 0;
    m->y_bt = cv.take<int32_t>(N);
    m->total = cv.take<float>(8);
    m->loss = cv.take<float>(8);
    m->row_loss = cv.take<float>(N);
    m->probs_bt = cv.take<float>(N * m->C);
    m->z = cv.take<float>(N * m->ldc);
    m->dz = take_shadowed(m, cv, N * m->ldc);
    m->cls_in = take_shadowed(m, cv, N * ldh);
    m->dcls = cv.take<float>(N * ldh);
    m->fused = take_shadowed(m, cv, N * ldh);
    m->dfused = cv.take<float>(N * ldh);
    int maxw = 8;
    for (auto& st : m->st) {
        // staged copy of the input: always (host inputs need one; device inputs whose width is not a
        // multiple of 4 floats cannot feed the GEMM loader directly)
        st.xstage = take_shadowed(m, cv, N * ld_of(st.cfg.input_dim));
        st.act.resize(st.cfg.n_enc);
        st.relu_bits.assign(st.cfg.n_enc, nullptr); st.bits_tiles.assign(st.cfg.n_enc, 0);
        for (int l = 0; l < st.cfg.n_enc; ++l) {
            st.act[l] = take_shadowed(m, cv, N * ld_of(st.cfg.enc_units[l]));
            maxw = std::max(maxw, ld_of(st.cfg.enc_units[l]));
            if (l + 1 < st.cfg.n_enc && st.cfg.enc_act[l] == ADN_ACT_RECTIFY && m->cfg.precision != ADN_PRECISION_F32)
                st.relu_bits[l] = cv.take<char>((size_t)cdiv((int)N, 256) * cdiv(st.cfg.enc_units[l], 256) * 512 * 16);
        }
        if (st.cfg.batchnorm) {
            st.bn_out = cv.take<float>(N * ld_of(st.enc_out));
            st.bn_save_mean = cv.take<float>(ld_of(st.enc_out));
            st.bn_save_inv_std = cv.take<float>(ld_of(st.enc_out));
            st.bn_ws = cv.take<char>(batchnorm_ws_bytes(st.enc_out));
        }
        if (st.cfg.aux_dim > 0) st.aux_stage = cv.take<float>(N * ld_of(st.cfg.aux_dim));
        st.feat = take_shadowed(m, cv, N * ld_of(st.feat_dim));
        st.dfeat = cv.take<float>(N * ld_of(st.feat_dim));
        st.dE = take_shadowed(m, cv, N * ld_of(st.enc_out));
        st.xc = nullptr; st.dEc = nullptr; st.compact_ws = nullptr;
        if (st.cfg.n_enc > 0 && compaction_possible(m, N)) {     // frame compaction (row counts <= N): only where a call of this shape can compact
            st.xc = take_shadowed(m, cv, N * ld_of(st.cfg.input_dim));
            st.dEc = take_shadowed(m, cv, N * ld_of(st.enc_out));
            st.compact_ws = cv.take<float>((size_t)B * st.enc_out);      // the padding frames' gradient, summed per utterance (delta layer)
        }
        st.lw.resize(st.lstm.size());
        for (auto& w : st.lw) carve_lstm(m, cv, w, B, T, ldh, ldg);
        st.hsum = take_shadowed(m, cv, N * ldh);
        st.dout_buf = cv.take<float>(N * ldh);
        st.dout = st.dout_buf;
        {
            int w = 64;
            for (int l = 0; l < st.cfg.n_enc; ++l) w = std::max(w, ld_of(st.cfg.enc_units[l]));
            st.ping_ld = w;
            if (st.cfg.n_enc > 1) {
                st.pingA = take_shadowed(m, cv, N * w);
                st.pingB = take_shadowed(m, cv, N * w);
                // per layer (the reductions are batched): one row of partial sums per 64 rows -- the register-staged kernels' 64-row
                // tiles, the persistent kernels' 4 wave rows per 256-row tile, whose last tile may add up to 3 rows more
                // (N = 5200: 84 rows against 82: the fused bias gradient was declined and the lean dZ had no sums -- B = 130)
                st.colsum_ws_floats = (size_t)(cdiv((int)N, 64) + 4) * w;
                st.colsum_ws = cv.take<float>(st.colsum_ws_floats * st.cfg.n_enc);
            }
        }
        if (m->cfg.agg_dropout_p > 0.f) st.out_drop = take_shadowed(m, cv, N * ldh);
    }
    m->aggw.resize(m->agg.size());
    for (auto& w : m->aggw) carve_lstm(m, cv, w, B, T, ldh, ldg);
    (void)maxw;
    if (m->cfg.fusion == ADN_FUSE_CONCAT && m->S > 1 && !m->agg.empty()) {
        m->cat16 = cv.take<char>(N * (size_t)m->S * ldh * 2);
        m->cat16lo = cv.take<char>(N * (size_t)m->S * ldh * 2);
        m->dcat = cv.take<float>(N * (size_t)m->S * ldh);
        m->wcat_tmp_slots = std::max<size_t>(1, std::min<size_t>(m->agg.size(), kMaxGemmGroups));
        m->wcat_tmp = cv.take<float>(m->wcat_tmp_slots * (size_t)m->S * ldh * ldg);
    }
    (void)host_inputs;
    return cv.cursor;
}

// Bias gradients on the weight-gradient GEMM.  db_l = 1^T dZ_l is one more row of dW_l = A_l^T dZ_l when A_l (the layer's input: the
// compacted stream input, or the activation below) carries a column of ones -- and it has a place for one: leading dimensions are
// rounded up to 64, the first pad column of a width that is no multiple of 64 is free, and [W_l | b_l] are adjacent in the flat
// buffers (build_params: b_l IS row in_w of W_l's matrix).  The MFMAs sum the column for nothing (1200 / 2000 / 1000 / 500 rows
// + 1 fill the same 256-row tiles), where the input-gradient GEMM of the layer above paid ~60 us of epilogue VALU + shuffles for
// its fused column sums at the bench size.  What keeps the column alive: every writer of these matrices stores columns < width
// only (GEMM epilogues and the gather in whole 8-column groups: width % 8 == 0 is required), whole-buffer conversions
// (refresh / restore_fp32) map 1.0 -> hi 1.0, lo 0 -> 1.0, and no reader depends on it being ZERO -- the kernels mask k >= K
// in registers (gemm_bf16.hip header; "the columns of A behind K may hold anything").  The bf16 arithmetic only: there the sum runs
// over the bf16 dZ the weight-gradient GEMM reads anyway; the parity-grade arithmetics (bf16x3, mixed) keep the sums of the fp32
// accumulators (measured on both: 5.87 -> 5.89 ms and nothing -- and their bits, which the accuracy runs of tests/test_gpu_accuracy.py
// are pinned on, stay what they were).  ADN_NO_BIAS_ON_DW=1 restores the fused / separate column sums everywhere.
static bool bias_on_dw_enabled(const adn_model* m) {
    static const bool off = getenv("ADN_NO_BIAS_ON_DW") != nullptr;
    return !off && m->bf16();
}
static bool ones_col_fits(int width) { return width % 8 == 0 && ld_of(width) > width; }
static bool has_ones_col(const adn_model* m, const float* A) {
    for (const float* p : m->ones_cols) if (p == A) return true;
    return false;
}
__global__ __launch_bounds__(256) void fill_ones_col_kernel(float* __restrict__ f32, uint16_t* __restrict__ hi, int ld, int64_t rows, int col) {
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
        f32[r * ld + col] = 1.f;
        if (hi) hi[r * ld + col] = 0x3F80u;          // bf16(1.0); the lo plane keeps its zero
    }
}
static int fill_ones_col(adn_model* m, float* A, int width, int64_t rows) {
    if (!A || !ones_col_fits(width)) return ADN_OK;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((rows + 255) / 256, 1024));
    hipLaunchKernelGGL(fill_ones_col_kernel, dim3(grid), dim3(256), 0, m->stream, A, static_cast<uint16_t*>(m->shadow_of(A)), ld_of(width), rows, width);
    ADN_HIP_CHECK(hipGetLastError());
    m->ones_cols.push_back(A);
    return ADN_OK;
}

int ensure_workspace(adn_model* m, int B, int T) {
    if (B == m->wsB && T == m->wsT && m->slab) return ADN_OK;
    const size_t need = carve(m, nullptr, B, T, true);
    if (need > m->slab_bytes) {
        if (m->slab) { ADN_HIP_CHECK(hipStreamSynchronize(m->stream)); ADN_HIP_CHECK(hipFree(m->slab)); m->slab = nullptr; }
        const size_t want = need + need / 8;
        ADN_HIP_CHECK(hipMalloc((void**)&m->slab, want));
        m->slab_bytes = want;
    }
    carve(m, m->slab, B, T, true);
    ADN_HIP_CHECK(hipMemsetAsync(m->slab, 0, need, m->stream));   // pad columns must read as zero
    if (bias_on_dw_enabled(m))                                    // ... but for the ones behind the inputs of the encoder layers
        for (auto& st : m->st) {
            ADN_TRY(fill_ones_col(m, st.xc, st.cfg.input_dim, (int64_t)B * T));
            for (int l = 0; l + 1 < st.cfg.n_enc; ++l) ADN_TRY(fill_ones_col(m, st.act[l], st.cfg.enc_units[l], (int64_t)B * T));
        }
    m->wsB = B; m->wsT = T;
    return ADN_OK;
}

int refresh(adn_model* m, const float* p, size_t floats);
int refresh_params(adn_model* m);

// ------------------------------------------------------------------------------------------
// staging of the caller's arrays
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void widen_bf16_rows_kernel(const uint16_t* __restrict__ src, int ld_src, float* __restrict__ dst,
                                                              int ld_dst, int64_t rows, int cols) {
    const int64_t total = rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / cols; const int c = (int)(e - r * cols);
        dst[r * ld_dst + c] = __uint_as_float((uint32_t)src[r * ld_src + c] << 16);
    }
}

// bf16 rows [rows][cols] (row stride ld_src elements, device memory) -> fp32 rows with stride ld_dst
int widen_bf16_rows(const void* src, int ld_src, float* dst, int ld_dst, int64_t rows, int cols, hipStream_t s) {
    const int64_t total = rows * cols;
    if (total <= 0) return ADN_OK;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 8192));
    hipLaunchKernelGGL(widen_bf16_rows_kernel, dim3(grid), dim3(256), 0, s, static_cast<const uint16_t*>(src), ld_src, dst, ld_dst,
                       rows, cols);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int setup_compaction(adn_model* m, int B, int T, bool dev);
int setup_buckets(adn_model* m, int B, int T);
int read_input_flags(adn_model* m, int* out);
bool streams_concurrent(const adn_model* m);
int stage_inputs(adn_model* m, const void* const* inputs, const int32_t* targets, const uint8_t* mask, int B, int T,
                 int flags) {
    const size_t N = (size_t)B * T;
    const bool dev = flags & ADN_FLAG_DEVICE_INPUTS;
    // Frame compaction: the lengths of THIS call's batch.  An announcement (adn_set_batch_lengths) is used up here, whatever becomes
    // of the call, and checked before anything else; with a host mask the lengths are read off the mask -- the reference's
    // train(*inputs, targets, mask, window) carries no lengths (runners/3stream.py:309-320,370) -- and an announcement must agree with it.
    m->call_lens.clear(); m->call_lens_auto = false;
    {
        std::vector<int32_t> lens; lens.swap(m->batch_lens);
        if (!lens.empty()) {
            ADN_CHECK((int)lens.size() == B, ADN_ERR_INVALID, "adn_set_batch_lengths: lengths of another batch size than this call's");
            for (int b = 0; b < B; ++b) ADN_CHECK(lens[b] >= 1 && lens[b] <= T, ADN_ERR_INVALID, "adn_set_batch_lengths: a length outside [1, T]");
        }
        if (!dev && mask && (!lens.empty() || (m->auto_compact && compaction_possible(m, N)))) {
            std::vector<int32_t> seen((size_t)B);
            bool prefix = true;
            for (int b = 0; b < B; ++b) {
                const uint8_t* row = mask + (size_t)b * T;
                int t = 0;
                while (t < T && row[t]) ++t;
                seen[b] = t;
                for (; t < T; ++t) if (row[t]) { prefix = false; break; }
                if (seen[b] < 1) prefix = false;
            }
            if (!lens.empty()) ADN_CHECK(prefix && seen == lens, ADN_ERR_INVALID, "adn_set_batch_lengths: the mask of this call is not the prefix mask of the announced lengths");
            else if (prefix) { lens.swap(seen); m->call_lens_auto = true; }
        }
        m->call_lens.swap(lens);
    }
    const bool in16 = flags & ADN_FLAG_BF16_INPUTS;              // stream (and auxiliary) inputs arrive as bf16 arrays
    const bool in_planes = flags & ADN_FLAG_PLANE_INPUTS;        // ... as their hi / lo planes (inputs[S + s] = the lo plane of stream s)
    // (a staging buffer listed as "fp32 copy not written" by an earlier call with plane inputs is about to be re-decided)
    for (auto& st : m->st)
        for (const float* gone : {(const float*)st.xstage, (const float*)st.xc})
            for (size_t k = 0; k < m->fp32_stale.size(); ++k)
                if (gone && m->fp32_stale[k].first == gone) { m->fp32_stale.erase(m->fp32_stale.begin() + (long)k); break; }
    if (in_planes) {
        ADN_CHECK(dev && m->planes() && !in16, ADN_ERR_INVALID, "plane inputs: device arrays in the bf16x3 / mixed arithmetic only");
        for (auto& st : m->st) ADN_CHECK(st.cfg.aux_dim <= 0, ADN_ERR_INVALID, "plane inputs: auxiliary inputs are not supported");
    }
    const hipMemcpyKind kind = dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    ADN_CHECK(inputs && mask, ADN_ERR_INVALID, "null inputs / mask");
    for (int s = 0; s < m->S; ++s) {
        StreamState& st = m->st[s];
        ADN_CHECK(inputs[s], ADN_ERR_INVALID, "null stream input");
        const int D = st.cfg.input_dim;
        st.x16 = nullptr; st.x16lo = nullptr; st.x_convert_pending = false;
        if (in_planes) {
            // the caller's planes ARE the first GEMM's operands.  st.x names the fp32 staging buffer (dense rows of D), which holds
            // nothing: it is listed as stale, so that a reader that does not run over planes gets hi + lo written there first
            const void* lo = inputs[m->S + s];
            ADN_CHECK(lo && st.cfg.n_enc > 0 && D % 8 == 0 && ((uintptr_t)inputs[s]) % 16 == 0 && ((uintptr_t)lo) % 16 == 0, ADN_ERR_INVALID,
                      "plane inputs: every stream needs an encoder, D % 8 == 0 and 16-byte aligned planes");
            st.x = st.xstage; st.ldx = D;
            st.x16 = const_cast<void*>(inputs[s]); st.x16lo = const_cast<void*>(lo);
            bool listed = false;
            for (auto& e : m->fp32_stale) if (e.first == st.x) { e.second = N * (size_t)D; listed = true; }
            if (!listed) m->fp32_stale.push_back({st.x, N * (size_t)D});
            continue;
        }
        if (in16) {
            // bf16 mode, an encoder in front (every consumer of the input is a GEMM reading bf16 operands): the caller's
            // device array IS the operand -- no copy, no conversion; st.x only names it (shadow_of), it is never read
            const bool direct16 = dev && m->bf16() && st.cfg.n_enc > 0 && D % 8 == 0 && ((uintptr_t)inputs[s]) % 16 == 0 &&
                                  !getenv("ADN_BF16_NO_SHADOW");
            if (direct16) {
                st.x = static_cast<const float*>(inputs[s]); st.ldx = D;
                st.x16 = const_cast<void*>(inputs[s]);
                continue;
            }
            // otherwise widened into the fp32 staging buffer (exact), through the staging buffer's bf16 shadow for host arrays
            const int ld = ld_of(D);
            const void* src16 = inputs[s]; int ld16 = D;
            if (!dev) {
                void* sh = m->shadow_of(st.xstage);
                ADN_CHECK(sh, ADN_ERR_STATE, "internal: the input staging buffer has no bf16 shadow");
                ADN_HIP_CHECK(hipMemcpy2DAsync(sh, (size_t)ld * 2, inputs[s], (size_t)D * 2, (size_t)D * 2, N, kind, m->stream));
                src16 = sh; ld16 = ld;
            }
            ADN_TRY(widen_bf16_rows(src16, ld16, st.xstage, ld, (int64_t)N, D, m->stream));
            st.x = st.xstage; st.ldx = ld;
            continue;
        }
        // device inputs are used in place when the GEMM loader can read them directly; in bf16 mode they
        // are copied into the staging buffer instead, which owns a bf16 shadow
        const bool direct = dev && (D % (m->bf16() ? 8 : 4) == 0) && (((uintptr_t)inputs[s]) % 16 == 0);
        if (direct) {
            st.x = static_cast<const float*>(inputs[s]); st.ldx = D;
            // (the 16-bit copies are made behind setup_compaction(): a compacted call converts only the rows it gathers)
            if (m->bf16()) {                   // the staging buffer's shadow holds the bf16 copy (ld_of(D) == D)
                st.x16 = m->shadow_of(st.xstage);
                st.x_convert_pending = true;
            } else if (m->planes() && D % 8 == 0) {     // ... its two planes the hi / lo parts (dense rows of D, like x)
                st.x16 = m->shadow_of(st.xstage); st.x16lo = m->shadow_lo_of(st.xstage);
                st.x_convert_pending = true;
            }
        } else {
            const int ld = ld_of(D);
            ADN_HIP_CHECK(hipMemcpy2DAsync(st.xstage, (size_t)ld * 4, inputs[s], (size_t)D * 4, (size_t)D * 4, N, kind,
                                           m->stream));
            st.x = st.xstage; st.ldx = ld;
        }
    }
    int aux_k = 0;                               // auxiliary inputs follow the S stream inputs, in stream order
    for (auto& st : m->st) {
        if (st.cfg.aux_dim <= 0) continue;
        const void* src = inputs[m->S + aux_k++];
        ADN_CHECK(src, ADN_ERR_INVALID, "null auxiliary input");
        const int ad = st.cfg.aux_dim, ald = ld_of(ad);
        if (in16) {
            ADN_CHECK(dev, ADN_ERR_INVALID, "bf16 auxiliary inputs must be device arrays");
            ADN_TRY(widen_bf16_rows(src, ad, st.aux_stage, ald, (int64_t)N, ad, m->stream));
        } else {
            ADN_HIP_CHECK(hipMemcpy2DAsync(st.aux_stage, (size_t)ald * 4, src, (size_t)ad * 4, (size_t)ad * 4, N, kind, m->stream));
        }
    }
    for (auto& st : m->st) if (!st.x16) ADN_TRY(refresh(m, st.x, N * st.ldx));
    ADN_TRY(refresh_params(m));
    // (device arrays are read in place -- like the stream inputs: the caller keeps them alive until the stream has run the call --,
    //  host arrays through the slab's copies)
    const uint8_t* mask_src = mask;
    m->y_src = targets;
    if (!dev) {
        ADN_HIP_CHECK(hipMemcpyAsync(m->mask_bt, mask, N, kind, m->stream));
        mask_src = m->mask_bt;
        if (targets) { ADN_HIP_CHECK(hipMemcpyAsync(m->y_bt, targets, N * sizeof(int32_t), kind, m->stream)); m->y_src = m->y_bt; }
    }
    ADN_TRY(setup_compaction(m, B, T, dev));
    for (auto& st : m->st) {                     // float32 device inputs that were not gathered: their 16-bit copies in full
        if (!st.x_convert_pending) continue;
        st.x_convert_pending = false;
        if (st.x16 && st.x16lo) ADN_TRY(split_hilo(st.x, st.x16, st.x16lo, N * (size_t)st.cfg.input_dim, m->stream));
        else if (st.x16) ADN_TRY(to_bf16(st.x, st.x16, N * (size_t)st.cfg.input_dim, m->stream));
    }
    // (a device mask is compared with the announced lengths by the kernel that walks it anyway; the word is read at the call's next
    //  synchronisation point -- check_device_errors() -- or right here under ADN_CHECK_PADDING=1)
    const bool verify = m->compact && dev;
    ADN_TRY(setup_buckets(m, B, T));
    ADN_TRY(mask_prepare(mask_src, m->mask_tb, B, T, m->total, m->stream, verify ? m->d_lens : nullptr, verify ? m->input_flags() : nullptr, kInputLens,
                         m->tm.on ? m->tm_row0 : nullptr, m->tm.on ? m->tm_T : nullptr, m->tm.on ? m->tm.Bb : 0));
    if (verify && getenv("ADN_CHECK_PADDING")) {
        int f = 0;
        ADN_TRY(read_input_flags(m, &f));
        ADN_CHECK(!(f & kInputLens), ADN_ERR_INVALID, "adn_set_batch_lengths: the mask of this call is not the prefix mask of the announced lengths");
    }
    return ADN_OK;
}

// synchronises the model's stream, reads the input-check word and clears it
int read_input_flags(adn_model* m, int* out) {
    ADN_HIP_CHECK(hipStreamSynchronize(m->stream));
    ADN_HIP_CHECK(hipMemcpy(out, m->input_flags(), sizeof(int), hipMemcpyDeviceToHost));
    if (*out) ADN_HIP_CHECK(hipMemset(m->input_flags(), 0, sizeof(int)));
    return ADN_OK;
}

bool shadows_on(const adn_model* m) { return m->bf16() && !getenv("ADN_BF16_NO_SHADOW"); }

// Frame compaction (compact.hip): decided per call.  Needs the batch's lengths on the host (announced -- adn_set_batch_lengths: which
// also promises that the padding frames of the inputs are zero, as the reference's generators make them -- or read off a host mask by
// stage_inputs), a 16-bit arithmetic (the operands are gathered as bf16 / planes), encoders that end in a linear layer with no
// BatchNorm behind them (whose batch statistics would see the padding rows), and enough padding to pay for the gather (>= 10 % of
// the rows).  What the caller promised is checked where that is cheap: the mask against the lengths always (stage_inputs), the
// padding frames against zero whenever the arrays came from the host (a pass over a third of the bytes just uploaded, and the one
// case where nobody promised anything: lengths read off the mask -- a non-zero padding frame then simply leaves the call padded) and
// for device arrays under ADN_CHECK_PADDING=1.
int setup_compaction(adn_model* m, int B, int T, bool dev) {
    const size_t N = (size_t)B * T;
    m->compact = false; m->Nc = (int)N;
    std::vector<int32_t> lens; lens.swap(m->call_lens);
    const bool auto_lens = m->call_lens_auto; m->call_lens_auto = false;
    if (lens.empty() || !compaction_possible(m, N) || !(shadows_on(m) || m->planes()) || m->keep_fp32 || streams_concurrent(m)) return ADN_OK;
    bool any = false;
    for (auto& st : m->st) {
        if (st.cfg.n_enc == 0) continue;
        if (st.cfg.batchnorm || st.cfg.enc_act[st.cfg.n_enc - 1] != ADN_ACT_LINEAR || st.cfg.input_dim % 8 || st.ldx % 8 || !st.xc) return ADN_OK;
        if (!m->shadow_of(st.x) || (m->planes() && !m->shadow_lo_of(st.x))) return ADN_OK;
        any = true;
    }
    if (!any) return ADN_OK;
    int64_t valid = 0;
    for (int b = 0; b < B; ++b) valid += lens[b];                // (each in [1, T]: stage_inputs)
    if ((double)(valid + 1) > 0.9 * (double)N) return ADN_OK;
    // (small batches stay padded -- compaction_possible(): the encoder GEMMs are latency-bound and the dozen extra launches cost more
    //  than the rows save; the reference's 26-utterance minibatch: 29.5 -> 30.3 ms per epoch of 20 steps with it)
    const int Z = (int)valid;
    if (m->maps_lens != lens || m->maps_T != T) {
        // (uploads ON the model's stream, through pinned buffers the model owns -- a ring, one event per slot: the host runs ahead
        //  of the device, and the previous call's map kernel -- still queued, perhaps -- must read the previous call's lengths)
        m->maps_lens = lens; m->maps_T = T;
        std::vector<int32_t>& prefix = m->h_prefix;
        prefix.assign((size_t)B + 1, 0);
        for (int b = 0; b < B; ++b) prefix[b + 1] = prefix[b] + lens[b];
        adn_model::PinSlot& slot = m->pin[m->pin_next++ & 3];
        const size_t want = 2 * (size_t)B + 1;
        if (slot.ev) ADN_HIP_CHECK(hipEventSynchronize(slot.ev));
        else ADN_HIP_CHECK(hipEventCreateWithFlags(&slot.ev, hipEventDisableTiming));
        if (slot.cap < want) {
            if (slot.host) ADN_HIP_CHECK(hipHostFree(slot.host));
            slot.host = nullptr; slot.cap = 0;
            ADN_HIP_CHECK(hipHostMalloc((void**)&slot.host, (want + want / 2) * sizeof(int32_t), hipHostMallocDefault));
            slot.cap = want + want / 2;
        }
        memcpy(slot.host, lens.data(), (size_t)B * 4);
        memcpy(slot.host + B, prefix.data(), ((size_t)B + 1) * 4);
        ADN_HIP_CHECK(hipMemcpyAsync(m->d_lens, slot.host, (size_t)B * 4, hipMemcpyHostToDevice, m->stream));
        ADN_HIP_CHECK(hipMemcpyAsync(m->d_prefix, slot.host + B, ((size_t)B + 1) * 4, hipMemcpyHostToDevice, m->stream));
        ADN_HIP_CHECK(hipEventRecord(slot.ev, m->stream));
        ADN_TRY(compact_build_maps(m->d_lens, m->d_prefix, B, T, Z, m->comp_of_full, m->full_of_comp, m->stream));
        m->h_comp_of_full.assign(N, Z);
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < lens[b]; ++t) m->h_comp_of_full[(size_t)b * T + t] = prefix[b] + t;
    }
    if (auto_lens || !dev || getenv("ADN_CHECK_PADDING")) {
        for (auto& st : m->st) {
            if (st.cfg.n_enc == 0) continue;
            if (st.x_convert_pending) ADN_TRY(compact_check_padding32(st.x, st.ldx, m->comp_of_full, (int)N, st.cfg.input_dim, Z, m->input_flags(), kInputPadding, m->stream));
            else ADN_TRY(compact_check_padding16(m->shadow_of(st.x), st.ldx, m->comp_of_full, (int)N, st.cfg.input_dim, Z, m->input_flags(), kInputPadding, m->stream));
        }
        int f = 0;
        ADN_TRY(read_input_flags(m, &f));
        if (f & kInputPadding) {
            if (auto_lens) return ADN_OK;          // nobody said the padding frames were zero: this batch runs padded
            set_error("adn_set_batch_lengths: a padding frame of a stream input is not zero (the announcement promises zero frames behind every "
                      "utterance, as utils/datagen.py:129-142 pads them)");
            return ADN_ERR_INVALID;
        }
    }
    m->Nc = Z + 1; m->compact = true;
    // (the 16-bit gathers of streams of one geometry -- and their lo planes -- go out as ONE launch)
    const void* gsrc[kMaxGatherJobs]; void* gdst[kMaxGatherJobs];
    int gn = 0, g_ld_src = 0, g_ld = 0, g_D = 0;
    auto flush_gathers = [&]() -> int {
        if (!gn) return ADN_OK;
        const int rc = compact_gather_rows16_batch(gsrc, gdst, gn, g_ld_src, g_ld, m->full_of_comp, m->Nc, g_D, m->stream);
        gn = 0;
        return rc;
    };
    auto queue_gather = [&](const void* src, int ld_src, void* dst, int ld, int D) -> int {
        if (gn && (gn == kMaxGatherJobs || ld_src != g_ld_src || ld != g_ld || D != g_D)) ADN_TRY(flush_gathers());
        g_ld_src = ld_src; g_ld = ld; g_D = D;
        gsrc[gn] = src; gdst[gn] = dst; ++gn;
        return ADN_OK;
    };
    const float* fsrc[4]; void* fhi[4]; void* flo[4];
    int fn = 0, f_ld_src = 0, f_ld = 0, f_D = 0;
    auto flush_f32 = [&]() -> int {
        if (!fn) return ADN_OK;
        const int rc = compact_gather_rows_f32_batch(fsrc, fhi, flo, fn, f_ld_src, f_ld, m->full_of_comp, m->Nc, f_D, m->stream);
        fn = 0;
        return rc;
    };
    for (auto& st : m->st) {
        if (st.cfg.n_enc == 0) continue;
        const int D = st.cfg.input_dim, ld = ld_of(D);
        const void* src_hi = m->shadow_of(st.x); const void* src_lo = m->planes() ? m->shadow_lo_of(st.x) : nullptr;
        const float* src_f32 = st.x_convert_pending ? st.x : nullptr;         // float32 device rows: converted while they are gathered
        const int ld_src = st.ldx;
        st.x = st.xc; st.ldx = ld; st.x16 = nullptr; st.x16lo = nullptr; st.x_convert_pending = false;      // (the staged names go: shadow_of(xc) is the slab's)
        if (src_f32) {
            if (fn && (fn == 4 || ld_src != f_ld_src || ld != f_ld || D != f_D)) ADN_TRY(flush_f32());
            f_ld_src = ld_src; f_ld = ld; f_D = D;
            fsrc[fn] = src_f32; fhi[fn] = m->shadow_of(st.xc); flo[fn] = m->planes() ? m->shadow_lo_of(st.xc) : nullptr; ++fn;
        } else {
            ADN_TRY(queue_gather(src_hi, ld_src, m->shadow_of(st.xc), ld, D));
            if (src_lo) ADN_TRY(queue_gather(src_lo, ld_src, m->shadow_lo_of(st.xc), ld, D));
        }
        if (m->planes()) {                    // the fp32 matrix behind the planes holds nothing: a reader that wants it gets hi + lo first
            bool listed = false;
            for (auto& e : m->fp32_stale) if (e.first == st.xc) { e.second = (size_t)m->Nc * ld; listed = true; }
            if (!listed) m->fp32_stale.push_back({st.xc, (size_t)m->Nc * ld});
        }
    }
    ADN_TRY(flush_gathers());
    ADN_TRY(flush_f32());
    return ADN_OK;
}

// Length buckets (TmPlan), decided per call: a train step (the entry point asked) of a compacted batch -- the lengths are on the host
// and the mask is checked against them --, the weight-stationary LSTM kernels of H <= 256 on every recurrence, nothing in the model
// that walks the time-major tensors by (b, t) outside the kernels that take the tables (no dropout, no last-timestep head, no
// adaptive fusion, no auxiliary inputs, one HIP stream), and >= 10 % fewer rows.  ADN_NO_LENGTH_BUCKETS=1 / adn_set_length_buckets(0): never.
LstmStep make_step(const adn_model* m, const LstmParams& lp, const LstmWork& w, const float* dhs, bool grads);
bool streams_concurrent(const adn_model* m);
static std::vector<LstmStep> expand_entries(const adn_model* m, const TmPlan& p, const LstmStep* l, int n);
static int decide_buckets(adn_model* m, int B, int T);
static int zero_spare_blocks(adn_model* m);
int setup_buckets(adn_model* m, int B, int T) {
    ADN_TRY(decide_buckets(m, B, T));
    // The exchange buffers of the weight-stationary LSTM kernels are cut by the launch geometry: one region per entry, sized by the
    // entry's utterances (forward granules, then the backward inboxes, which the kernels expect -- and leave -- EMPTY).  When the
    // geometry changes (B x T rows <-> buckets, or another bucket size) what was somebody's forward region becomes somebody's inbox:
    // no slot may keep a granule of the earlier cut (lstm_cluster.hip; the bf16x3 kernels' launch counters restart with the buffer).
    const int key = m->tm.on ? m->tm.Bb : 0;
    if (key != m->xchg_key) {
        m->xchg_key = key;
        auto reset_xchg = [&](LstmWork& w) -> int {
            ADN_HIP_CHECK(hipMemsetAsync(w.xchg, 0, lstm_cluster_xchg_bytes(B + 32 * kMaxBuckets, m->H), m->stream));
            w.xchg_seq = 0;
            return ADN_OK;
        };
        for (auto& st : m->st) for (auto& w : st.lw) ADN_TRY(reset_xchg(w));
        for (auto& w : m->aggw) ADN_TRY(reset_xchg(w));
    }
    return ADN_OK;
}
static int decide_buckets(adn_model* m, int B, int T) {
    const bool was_on = m->tm.on;
    m->tm.on = false;
    static const bool off = getenv("ADN_NO_LENGTH_BUCKETS") != nullptr;
    if (off || !m->buckets_allowed || !m->want_buckets || !m->compact || m->maps_lens.size() != (size_t)B) return ADN_OK;
    if (m->head_last() || m->has_dropout() || m->cfg.fusion == ADN_FUSE_ADASUM || streams_concurrent(m) || deterministic() || m->H > 256 ||
        m->keep_fp32) return ADN_OK;
    size_t n_stream_lstm = 0;
    for (auto& st : m->st) {
        if (st.cfg.aux_dim > 0 || st.cfg.n_enc == 0 || st.lstm.empty()) return ADN_OK;
        n_stream_lstm += st.lstm.size();
    }
    const int n_l = (int)std::max(n_stream_lstm, m->agg.size());       // LSTMs of the widest launch
    const std::vector<int32_t>& lens = m->maps_lens;
    // the batch of the last bucketed call again (an epoch driver's evaluation in between, the benchmark's loop): the plan stands and
    // its tables are still on the device; only the mask's empty rows need their zeros back when another layout was there since
    if (m->tm_cached.on && m->tm_lens == lens && m->tm_key_T == T && m->tm_key_prec == m->cfg.precision && m->tm_key_mixed == (int)m->bwd_hi_only) {
        m->tm = m->tm_cached;
        if (!was_on) {                 // (a call over B x T rows wrote the mask's and the gate gradients' rows no frame lives in)
            ADN_HIP_CHECK(hipMemsetAsync(m->mask_tb, 0, (size_t)B * T, m->stream));
            ADN_TRY(zero_spare_blocks(m));
        }
        return ADN_OK;
    }
    m->tm_cached.on = false;
    std::vector<int32_t> order((size_t)B);
    for (int b = 0; b < B; ++b) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return lens[a] > lens[b]; });
    // the bucket count: fewest rows among the counts whose launches still hold every LSTM of a group at once
    TmPlan best; int64_t best_rows = (int64_t)B * T;
    for (int nb = 2; nb <= kMaxBuckets; ++nb) {
        const int Bb = cdiv(B, nb);
        if (n_l * nb > kMaxLstmPerLaunch || (int64_t)n_l * nb * cdiv(Bb, 32) * 4 > lstm_cluster_cus()) continue;
        TmPlan p; p.nb = nb; p.Bb = Bb;
        for (int k = 0; k < nb; ++k) {
            p.Tk[k] = (k * Bb < B) ? lens[order[(size_t)k * Bb]] : 1;      // (a bucket of phantom slots only: one masked step)
            p.c[k + 1] = p.c[k] + p.Tk[k] + 1;
            p.Tmax = std::max(p.Tmax, p.Tk[k]);
        }
        p.Tt = p.c[nb] - 1;
        if ((int64_t)p.Bb * p.Tt < best_rows) { best_rows = (int64_t)p.Bb * p.Tt; best = p; }
    }
    if (best.nb < 2 || (double)best_rows > 0.9 * (double)B * T) return ADN_OK;
    best.on = true;
    // every recurrence of the model on a kernel that takes the entries? (asked of the dispatcher's own predicate, per launch)
    {
        const int per = kMaxLstmPerLaunch / best.nb;
        const int prec = m->cfg.precision == ADN_PRECISION_BF16X3 ? ADN_PRECISION_BF16X3 : m->lstm_precision();
        std::vector<LstmStep> all[2];
        for (auto& st : m->st) for (size_t k = 0; k < st.lstm.size(); ++k) all[0].push_back(make_step(m, st.lstm[k], st.lw[k], nullptr, true));
        for (size_t k = 0; k < m->agg.size(); ++k) all[1].push_back(make_step(m, m->agg[k], m->aggw[k], nullptr, true));
        for (auto& steps : all)
            for (size_t i = 0; i < steps.size(); i += (size_t)per) {
                const int n = (int)std::min<size_t>((size_t)per, steps.size() - i);
                std::vector<LstmStep> ex = expand_entries(m, best, steps.data() + i, n);
                if (!lstm_takes_length_buckets(ex.data(), (int)ex.size(), best.Bb, best.Tmax, m->H, prec, false)) return ADN_OK;
                // (mixed: run_lstm_group back-propagates on the bf16 kernel where that one takes its entries and on this one otherwise)
                if (!lstm_takes_length_buckets(ex.data(), (int)ex.size(), best.Bb, best.Tmax, m->H, prec, true)) return ADN_OK;
            }
    }
    m->tm = best; m->tm_cached = best;
    {   // the tables of this cut
        m->tm_lens = lens; m->tm_key_T = T; m->tm_key_prec = m->cfg.precision; m->tm_key_mixed = (int)m->bwd_hi_only;
        const size_t rows = (size_t)best.Bb * best.Tt, want = 2 * (size_t)B + rows;
        adn_model::PinSlot& slot = m->pin_tm[m->pin_tm_next++ & 3];
        if (slot.ev) ADN_HIP_CHECK(hipEventSynchronize(slot.ev));
        else ADN_HIP_CHECK(hipEventCreateWithFlags(&slot.ev, hipEventDisableTiming));
        if (slot.cap < want) {
            if (slot.host) ADN_HIP_CHECK(hipHostFree(slot.host));
            slot.host = nullptr; slot.cap = 0;
            ADN_HIP_CHECK(hipHostMalloc((void**)&slot.host, (want + want / 2) * sizeof(int32_t), hipHostMallocDefault));
            slot.cap = want + want / 2;
        }
        int32_t* row0 = slot.host; int32_t* tw = slot.host + B; int32_t* bt = slot.host + 2 * (size_t)B;
        for (size_t r = 0; r < rows; ++r) bt[r] = -1;
        for (int r = 0; r < B; ++r) {
            const int b = order[r], k = r / best.Bb, j = r % best.Bb;
            row0[b] = best.c[k] * best.Bb + j; tw[b] = best.Tk[k];
            for (int t = 0; t < best.Tk[k]; ++t) bt[(size_t)row0[b] + (size_t)t * best.Bb] = b * T + t;
        }
        ADN_HIP_CHECK(hipMemcpyAsync(m->tm_row0, row0, (size_t)B * 4, hipMemcpyHostToDevice, m->stream));
        ADN_HIP_CHECK(hipMemcpyAsync(m->tm_T, tw, (size_t)B * 4, hipMemcpyHostToDevice, m->stream));
        ADN_HIP_CHECK(hipMemcpyAsync(m->tm_bt, bt, rows * 4, hipMemcpyHostToDevice, m->stream));
        ADN_HIP_CHECK(hipEventRecord(slot.ev, m->stream));
        // rows no frame lives in keep a zero mask (mask_prepare writes the frames' rows only) and zero gate gradients
        ADN_HIP_CHECK(hipMemsetAsync(m->mask_tb, 0, (size_t)B * T, m->stream));
        ADN_TRY(zero_spare_blocks(m));
    }
    return ADN_OK;
}

int ensure_splitk_ws(adn_model* m) {
    if (m->splitk_ws || m->cfg.precision == ADN_PRECISION_F32) return ADN_OK;
    // every (tile, K-slice) workgroup of a launch owns one slab piece: <= CUs x 256 x 256 floats, + row padding
    m->splitk_ws_floats = (size_t)24 << 20;
    ADN_HIP_CHECK(hipMalloc((void**)&m->splitk_ws, m->splitk_ws_floats * sizeof(float)));
    return ADN_OK;
}

bool streams_concurrent(const adn_model* m);

// fills in the bf16 operand copies of one GEMM of the model (bf16 mode)
// lean: the fp32 copy of C is not needed by anyone (bf16 mode: every consumer reads the shadow) -> skip writing it
void mgemm_prepare(adn_model* m, GemmArgs& g, bool lean) {
    g.precision = m->cfg.precision;
    // (the slab workspace is ONE buffer: with the streams on forked HIP streams their split-K GEMMs would share it
    //  concurrently -- those runs keep the atomic split-K of the register-staged kernel)
    const bool shared_ws_ok = !streams_concurrent(m);
    g.splitk_ws = shared_ws_ok ? m->splitk_ws : nullptr; g.splitk_ws_floats = shared_ws_ok ? m->splitk_ws_floats : 0;
    if (m->cfg.precision == ADN_PRECISION_BF16X3 && !getenv("ADN_X3_NO_MASK_SHADOWS")) {
        // A rectifier's mask is the sign pattern of its output, and bf16(y) > 0 <=> y > 0: the forward producers of rectified
        // activations also write the bf16 copy, and the masked input-gradient GEMMs read THAT as their mask (half the bytes,
        // and the form the ping-pong kernel's epilogue takes: the streams' input gradients then go out as grouped launches
        // like in bf16 mode).  Values and products stay fp32-grade; only the mask's carrier changes.
        if (g.act == ADN_ACT_RECTIFY && g.C && !g.accumulate) g.C16 = m->shadow_of(g.C);
        if (g.Y && g.act_grad == ADN_ACT_RECTIFY) g.Y16 = m->shadow_of(g.Y);
    }
    // B as a k-contiguous [N][K] matrix for the skinny kernels (gemm_skinny.hip): an NT problem's B is that already (the weights
    // of dX = dZ W^T), an NN problem's is the transposed copy of its weights where the model keeps one
    if (shadows_on(m) || m->planes()) {
        if (g.layout == GEMM_NT) { g.Bkc16 = m->shadow_of(g.B); g.Bkc16lo = m->planes() ? m->shadow_lo_of(g.B) : nullptr; g.ldbkc = g.ldb; }
        else if (g.layout == GEMM_NN)
            for (const auto& t : m->transw)
                if (t.key == g.B) {
                    // (mixed arithmetic: only the lo planes a forward skinny product can take are kept current, refresh_transposed())
                    if (m->planes() && m->bwd_hi_only && !t.lo_fwd) break;
                    g.Bkc16 = t.buf; g.Bkc16lo = m->planes() ? t.buf + m->transw_slab_bytes : nullptr; g.ldbkc = t.ldT; break;
                }
    }
    if (m->planes()) {
        // every GEMM operand of this mode has its two planes (refresh() behind non-GEMM producers, planes_of_output() behind
        // GEMMs, refresh_params_x3() for the weights): handed to gemm(), which multiplies over them where the ping-pong
        // kernel takes the shape and otherwise falls back to split images of the fp32 operands
        g.A16 = m->shadow_of(g.A); g.A16lo = m->shadow_lo_of(g.A);
        g.B16 = m->shadow_of(g.B); g.B16lo = m->shadow_lo_of(g.B);
        // (the fp32 copy of a result whose planes the kernel writes is skipped; m_gemm() keeps a list of such tensors and writes
        //  hi + lo back ahead of a reader that does not run over planes -- the 50-unit bottleneck's GEMMs take the image path.
        //  ADN_X3_NO_LEAN=1 keeps every fp32 copy)
        g.lean_ok = lean && !m->keep_fp32 && !getenv("ADN_X3_NO_LEAN");
        if (!g.A16 || !g.A16lo || !g.B16 || !g.B16lo) { g.A16 = g.B16 = g.A16lo = g.B16lo = nullptr; }
        else if (g.layout == GEMM_NT) {                    // dX = dZ W^T: the transposed plane copies of W (k-strided B), offered
                                                           // beside the NT operands -- gemm() switches to NN only if it uses them
            g.B16 = g.B16lo = nullptr;
            for (const auto& t : m->transw)
                if (t.key == g.B) { g.BT16 = t.buf; g.BT16lo = t.buf + m->transw_slab_bytes; g.ldbT = t.ldT; break; }
            if (!g.BT16) { g.A16 = g.A16lo = nullptr; }
        }
    }
    if (shadows_on(m)) {                                  // env switch: convert-in-flight reference path
        g.A16 = m->shadow_of(g.A);
        g.B16 = m->shadow_of(g.B);
        g.C16 = m->shadow_of(g.C);
        if (g.Y) g.Y16 = m->shadow_of(g.Y);
        if (g.layout == GEMM_NT && g.A16 && !getenv("ADN_NO_TRANSW"))
            for (const auto& t : m->transw)
                if (t.key == g.B) { g.layout = GEMM_NN; g.B16 = t.buf; g.ldb = t.ldT; g.b_pad_zero = 1; break; }
        if (lean && g.A16 && g.B16 && g.C16 && !g.accumulate && g.N % 4 == 0 && g.ldc % 4 == 0 && !m->keep_fp32)
            g.C = nullptr;
    }
}

int planes_of_output(adn_model* m, const GemmArgs& g);
int restore_fp32(adn_model* m, const float* p);
int m_gemm(adn_model* m, const GemmArgs& g);
int mgemm(adn_model* m, GemmArgs& g, bool lean = false) {
    mgemm_prepare(m, g, lean);
    return m_gemm(m, g);
}

int m_gemm(adn_model* m, const GemmArgs& g);
int m_gemm_grouped(adn_model* m, const GemmArgs* gs, int n);

// bring the bf16 shadow of an fp32 matrix written by a non-GEMM kernel up to date (bf16 mode only)
int refresh(adn_model* m, const float* p, size_t floats) {
    if (m->planes()) {                                     // bf16x3: both planes
        void* hi = m->shadow_of(p); void* lo = m->shadow_lo_of(p);
        if (!hi || !lo) return ADN_OK;
        return split_hilo(p, hi, lo, (size_t)round_up((int64_t)floats, 8), m->stream);
    }
    if (!m->bf16()) return ADN_OK;
    void* sh = m->shadow_of(p);
    if (!sh) return ADN_OK;
    return to_bf16(p, sh, (size_t)round_up((int64_t)floats, 8), m->stream);
}

// bf16x3 through planes: the output of a GEMM that later GEMMs read gets its planes here (whole rows: the pad columns are
// zero in fp32 and stay zero in both planes)
int planes_of_output(adn_model* m, const GemmArgs& g) {
    if (!m->planes() || !g.C || g.accumulate || g.no_planes) return ADN_OK;
    if (g.planes_done && *g.planes_done) return ADN_OK;             // the kernel wrote both planes in its epilogue
    if (!m->shadow_of(g.C)) return ADN_OK;
    return refresh(m, g.C, (size_t)g.M * g.ldc);
}
// (planes mode) offer the result's planes to the kernel: the ping-pong kernel writes them from its epilogue
void offer_output_planes(adn_model* m, GemmArgs& g, int* done) {
    *done = 0;
    if (!m->planes() || !g.C || g.accumulate || g.ldc % 8 || g.no_planes) return;
    void* hi = m->shadow_of(g.C); void* lo = m->shadow_lo_of(g.C);
    if (!hi || !lo) return;
    g.C16 = hi; g.C16lo = lo; g.planes_done = done;
}
// fp32 copies that were skipped (see mgemm_prepare): bookkeeping around every GEMM of the model
static void stale_forget(adn_model* m, const float* p) {
    for (size_t k = 0; k < m->fp32_stale.size(); ++k)
        if (m->fp32_stale[k].first == p) { m->fp32_stale.erase(m->fp32_stale.begin() + (long)k); return; }
}
// (an entry's count with kStaleHiOnly set: only the hi plane holds the tensor -- the mixed mode's LSTM back-propagation writes dG
//  as bf16 alone -- and the fp32 values are that plane's)
constexpr size_t kStaleHiOnly = (size_t)1 << 62;
int restore_fp32(adn_model* m, const float* p) {
    for (size_t k = 0; k < m->fp32_stale.size(); ++k)
        if (m->fp32_stale[k].first == p) {
            const size_t n = m->fp32_stale[k].second & ~kStaleHiOnly;
            const bool hi_only = (m->fp32_stale[k].second & kStaleHiOnly) != 0;
            m->fp32_stale.erase(m->fp32_stale.begin() + (long)k);
            return join_hilo(m->shadow_of(p), hi_only ? nullptr : m->shadow_lo_of(p), const_cast<float*>(p), (size_t)round_up((int64_t)n, 8), m->stream);
        }
    return ADN_OK;
}
static bool reads_hi_planes_only(const GemmArgs& g);
static int operands_ready(adn_model* m, const GemmArgs* gs, int n) {
    if (m->fp32_stale.empty()) return ADN_OK;
    if (gemm_planes_would_run(gs, n)) return ADN_OK;               // the whole group reads planes
    for (int k = 0; k < n; ++k) {
        if (n > 1 && gemm_planes_would_run(&gs[k], 1)) continue;    // (a declined group is retried problem by problem)
        if (reads_hi_planes_only(gs[k])) {                          // (every bf16 kernel reads the copies it is handed)
            if (gs[k].Y && !gs[k].Y16) ADN_TRY(restore_fp32(m, gs[k].Y));
            continue;
        }
        ADN_TRY(restore_fp32(m, gs[k].A));
        ADN_TRY(restore_fp32(m, gs[k].B));
        if (gs[k].Y && !gs[k].Y16) ADN_TRY(restore_fp32(m, gs[k].Y));
    }
    return ADN_OK;
}
static void result_written(adn_model* m, const GemmArgs& g, int skipped) {
    if (!g.C) return;
    if (skipped) {
        const size_t count = ((size_t)g.M * g.ldc) | ((g.hi_product && g.hi_result) ? kStaleHiOnly : 0);
        for (auto& e : m->fp32_stale) if (e.first == g.C) { e.second = count; return; }
        m->fp32_stale.push_back({g.C, count});
    } else if (!m->fp32_stale.empty()) stale_forget(m, g.C);
}
// ADN_PRECISION_MIXED: a GEMM of back-propagation whose operands have their planes runs as ONE bf16 product over the hi planes
// (= the bf16 copies of bf16 mode).  Its result is still written as two planes where the kernel can (the readers further down
// the backward pass take planes), otherwise as fp32 + a split pass -- the bookkeeping of m_gemm() is unchanged.
static void mixed_backward(const adn_model* m, GemmArgs& g) {
    if (!m->bwd_hi_only || !m->in_backward || g.precision != ADN_PRECISION_BF16X3 || !g.A16 || !g.A16lo) return;
    if (g.layout == GEMM_NT) {
        if (!g.BT16 || !g.BT16lo) return;
        g.layout = GEMM_NN; g.B16 = g.BT16; g.ldb = g.ldbT; g.b_pad_zero = 1;
    } else if (!g.B16 || !g.B16lo) return;
    g.A16lo = g.B16lo = nullptr; g.BT16 = g.BT16lo = nullptr;
    g.precision = ADN_PRECISION_BF16; g.hi_product = 1;
    static const bool both = getenv("ADN_MIXED_BOTH_PLANES") != nullptr;        // (A/B: write the lo planes nobody reads)
    g.hi_result = both ? 0 : 1;      // every reader of a back-propagated tensor's planes is such a product: the lo plane is not written
}
static bool reads_hi_planes_only(const GemmArgs& g) { return g.hi_product && g.A16 && g.B16; }

int m_gemm(adn_model* m, const GemmArgs& g0) {
    GemmArgs g = g0; int done = 0, skipped = 0;
    mixed_backward(m, g);
    offer_output_planes(m, g, &done);
    g.fp32_skipped = &skipped;
    ADN_TRY(operands_ready(m, &g, 1));
    if (g.accumulate) ADN_TRY(restore_fp32(m, g.C));
    ADN_TRY(gemm(g, m->stream));
    result_written(m, g, skipped);
    return planes_of_output(m, g);
}
int m_gemm_grouped(adn_model* m, const GemmArgs* gs0, int n) {
    GemmArgs gs[kMaxGemmGroups]; int done[kMaxGemmGroups], skipped[kMaxGemmGroups];
    if (n > kMaxGemmGroups) {                                       // (never: the callers batch at most kMaxGemmGroups)
        for (int k = 0; k < n; ++k) ADN_TRY(m_gemm(m, gs0[k]));
        return ADN_OK;
    }
    for (int k = 0; k < n; ++k) {
        gs[k] = gs0[k]; mixed_backward(m, gs[k]);
        offer_output_planes(m, gs[k], &done[k]); skipped[k] = 0; gs[k].fp32_skipped = &skipped[k];
    }
    ADN_TRY(operands_ready(m, gs, n));
    for (int k = 0; k < n; ++k) if (gs[k].accumulate) ADN_TRY(restore_fp32(m, gs[k].C));
    ADN_TRY(gemm_grouped(gs, n, m->stream));
    for (int k = 0; k < n; ++k) { result_written(m, gs[k], skipped[k]); ADN_TRY(planes_of_output(m, gs[k])); }
    return ADN_OK;
}

// W^T copies: encoder weights of layers >= 1, every LSTM's W_in (per input block), the classifier weights
int refresh_transposed(adn_model* m) {
    // ldT / col_off: an aggregation LSTM fed by a concat keeps the transposed copies of its S input blocks side by side
    // in ONE [4H][S*ldh] matrix (block j at column j*ldh), which serves the block-wise products and the fused one alike
    struct Item { const float* W; int rows, cols, ld, ldT, col_off; };
    std::vector<Item> items;
    for (auto& st : m->st) {
        for (int l = 1; l < st.cfg.n_enc; ++l)
            items.push_back({m->P(st.encW[l]), st.enc_in[l], st.cfg.enc_units[l], ld_of(st.cfg.enc_units[l]), ld_of(st.enc_in[l]), 0});
        if (st.cfg.n_enc > 0)
            for (auto& lp : st.lstm) items.push_back({m->P(lp.W_in), lp.fin, 4 * m->H, m->ldg, ld_of(lp.fin), 0});
    }
    const int nblk = (m->cfg.fusion == ADN_FUSE_CONCAT) ? m->S : 1;
    for (auto& lp : m->agg)
        for (int j = 0; j < nblk; ++j)
            items.push_back({m->P(lp.W_in) + (size_t)j * m->H * m->ldg, m->H, 4 * m->H, m->ldg, nblk * m->ldh, j * m->ldh});
    items.push_back({m->P(m->smW), m->H, m->C, m->ldc, ld_of(m->H), 0});
    if (m->transw.empty()) {
        size_t bytes = 0;
        for (auto& it : items)
            if (it.col_off == 0) bytes += (size_t)round_up((int64_t)it.cols * it.ldT * 2, 256);
        // (two planes -- the second one only written in bf16x3 mode -- each with zero slack behind it: a k-strided operand is
        //  read in whole 32-row stages)
        m->transw_slab_bytes = (size_t)round_up((int64_t)(bytes + ((size_t)1 << 20)), 256);
        ADN_HIP_CHECK(hipMalloc((void**)&m->transw_slab, 2 * m->transw_slab_bytes));
        ADN_HIP_CHECK(hipMemsetAsync(m->transw_slab, 0, 2 * m->transw_slab_bytes, m->stream));
        size_t cur = 0; char* base = nullptr;
        for (auto& it : items) {
            if (it.col_off == 0) { base = m->transw_slab + cur; cur += (size_t)round_up((int64_t)it.cols * it.ldT * 2, 256); }
            // (what gemm_skinny.hip takes with B as a k-contiguous transpose: N <= 160 columns, or K <= 64 rows)
            m->transw.push_back({it.W, base + (size_t)it.col_off * 2, it.ldT, it.cols <= 160 || it.rows <= 64});
        }
    }
    if (!m->transw_items) {
        std::vector<TransposeItem> tab(items.size());
        int total = 0;
        for (size_t k = 0; k < items.size(); ++k) {
            total += cdiv(items[k].rows, 32) * cdiv(items[k].cols, 32);
            tab[k] = {items[k].W, m->transw[k].buf, items[k].rows, items[k].cols, items[k].ld, m->transw[k].ldT, total};
        }
        ADN_HIP_CHECK(hipMalloc((void**)&m->transw_items, tab.size() * sizeof(TransposeItem)));
        ADN_HIP_CHECK(hipMemcpy(m->transw_items, tab.data(), tab.size() * sizeof(TransposeItem), hipMemcpyHostToDevice));
        for (auto& t : tab) t.out = static_cast<char*>(t.out) + m->transw_slab_bytes;        // the same table for the lo planes
        ADN_HIP_CHECK(hipMalloc((void**)&m->transw_items_lo, tab.size() * sizeof(TransposeItem)));
        ADN_HIP_CHECK(hipMemcpy(m->transw_items_lo, tab.data(), tab.size() * sizeof(TransposeItem), hipMemcpyHostToDevice));
        m->transw_blocks = total;
        std::vector<TransposeItem> fwd;
        int total_fwd = 0;
        for (size_t k = 0; k < items.size(); ++k) {
            if (!m->transw[k].lo_fwd) continue;
            total_fwd += cdiv(items[k].rows, 32) * cdiv(items[k].cols, 32);
            fwd.push_back({items[k].W, m->transw[k].buf + m->transw_slab_bytes, items[k].rows, items[k].cols, items[k].ld, m->transw[k].ldT, total_fwd});
        }
        if (!fwd.empty()) {
            ADN_HIP_CHECK(hipMalloc((void**)&m->transw_items_lo_fwd, fwd.size() * sizeof(TransposeItem)));
            ADN_HIP_CHECK(hipMemcpy(m->transw_items_lo_fwd, fwd.data(), fwd.size() * sizeof(TransposeItem), hipMemcpyHostToDevice));
        }
        m->transw_n_lo_fwd = (int)fwd.size(); m->transw_blocks_lo_fwd = total_fwd;
    }
    ADN_TRY(transpose_to_bf16_batch(m->transw_items, (int)items.size(), m->transw_blocks, m->stream));
    // The lo planes.  Back-propagation reads the transposed copies as dX = dZ W^T; in the mixed arithmetic that is one product over
    // the hi planes.  The FORWARD pass reads them too, though: the skinny kernels take a narrow layer's weights (the 50-unit
    // bottleneck, the classifier) as this k-contiguous transpose, and the forward pass of the mixed arithmetic is a bf16x3 pass --
    // so those items' lo planes are kept current in every planes mode; the rest only where back-propagation runs over planes
    // (mgemm_prepare() does not hand out a lo plane that is not kept).
    if (m->planes() && !m->bwd_hi_only) ADN_TRY(transpose_to_bf16_batch(m->transw_items_lo, (int)items.size(), m->transw_blocks, m->stream, 1));
    else if (m->planes()) ADN_TRY(transpose_to_bf16_batch(m->transw_items_lo_fwd, m->transw_n_lo_fwd, m->transw_blocks_lo_fwd, m->stream, 1));
    return ADN_OK;
}

// bf16x3 mode: hi / lo fragment images of every W_hid for the weight-stationary kernels (H <= 512)
int refresh_transposed(adn_model* m);
constexpr size_t kPlaneSlack = (size_t)1 << 20;       // bytes behind a bf16 plane that k-strided stage reads may touch: zero
int ensure_params16(adn_model* m) {
    if (m->params16) return ADN_OK;
    ADN_HIP_CHECK(hipMalloc((void**)&m->params16, m->flat_floats * 2 + kPlaneSlack));
    ADN_HIP_CHECK(hipMemsetAsync(m->params16, 0, m->flat_floats * 2 + kPlaneSlack, m->stream));
    return ADN_OK;
}
int refresh_params_x3(adn_model* m) {
    if (!m->params16_dirty) return ADN_OK;
    if (m->planes()) {               // hi / lo planes of the whole parameter buffer + of the transposed copies (dX = dZ W^T as NN)
        ADN_TRY(ensure_params16(m));
        if (!m->params16lo) {        // (+ zero slack: a k-strided B operand is read in whole 32-row stages past a short last one)
            ADN_HIP_CHECK(hipMalloc((void**)&m->params16lo, m->flat_floats * 2 + kPlaneSlack));
            ADN_HIP_CHECK(hipMemsetAsync(m->params16lo, 0, m->flat_floats * 2 + kPlaneSlack, m->stream));
        }
        if (!m->params16_values_fresh)      // (Adam writes both planes with the update)
            ADN_TRY(split_hilo(m->flat[ADN_BUF_PARAM], m->params16, m->params16lo, (size_t)round_up((int64_t)m->flat_floats, 8), m->stream));
        m->params16_values_fresh = false;
        ADN_TRY(refresh_transposed(m));
        if (m->cfg.fusion == ADN_FUSE_CONCAT && m->S > 1 && m->S <= 4) {      // the concat consumers' padded W_in, both planes
            const size_t blk = (size_t)m->ldh * m->ldg;
            std::vector<const float*> win; std::vector<void*> whi, wlo;
            for (auto& lp : m->agg) {
                for (char** pl : {&lp.wcat16, &lp.wcat16lo})
                    if (!*pl) {
                        ADN_HIP_CHECK(hipMalloc((void**)pl, (size_t)m->S * blk * 2 + kPlaneSlack));
                        ADN_HIP_CHECK(hipMemsetAsync(*pl, 0, (size_t)m->S * blk * 2 + kPlaneSlack, m->stream));
                    }
                win.push_back(m->P(lp.W_in)); whi.push_back(lp.wcat16); wlo.push_back(lp.wcat16lo);
            }
            for (size_t k0 = 0; k0 < win.size(); k0 += 4) {
                const int nn = (int)std::min<size_t>(4, win.size() - k0);
                ADN_TRY(repack_rows_bf16(nn, win.data() + k0, whi.data() + k0, m->S, m->H, m->ldh, m->ldg, m->stream));
                ADN_TRY(repack_rows_bf16_lo(nn, win.data() + k0, wlo.data() + k0, m->S, m->H, m->ldh, m->ldg, m->stream));
            }
        }
    }
    if (m->H > 512) { m->params16_dirty = false; return ADN_OK; }
    // (256 < H <= 512: the wide forward kernel reads the forward images; the backward ones are packed with them)
    std::vector<const float*> fw; std::vector<void*> fhi, flo, bhi, blo;
    auto add = [&](LstmParams& lp) -> int {
        if (!lp.wfrag_fwd) ADN_HIP_CHECK(hipMalloc((void**)&lp.wfrag_fwd, lstm_frag_elems(m->H) * 2));
        if (!lp.wfrag_bwd) ADN_HIP_CHECK(hipMalloc((void**)&lp.wfrag_bwd, lstm_frag_elems(m->H) * 2));
        if (!lp.wfrag_fwd_lo) ADN_HIP_CHECK(hipMalloc((void**)&lp.wfrag_fwd_lo, lstm_frag_elems(m->H) * 2));
        if (!lp.wfrag_bwd_lo) ADN_HIP_CHECK(hipMalloc((void**)&lp.wfrag_bwd_lo, lstm_frag_elems(m->H) * 2));
        fw.push_back(m->P(lp.W_hid)); fhi.push_back(lp.wfrag_fwd); flo.push_back(lp.wfrag_fwd_lo);
        bhi.push_back(lp.wfrag_bwd); blo.push_back(lp.wfrag_bwd_lo);
        return ADN_OK;
    };
    for (auto& st : m->st) for (auto& lp : st.lstm) ADN_TRY(add(lp));
    for (auto& lp : m->agg) ADN_TRY(add(lp));
    for (size_t k0 = 0; k0 < fw.size(); k0 += 8) {
        const int n = (int)std::min<size_t>(8, fw.size() - k0);
        ADN_TRY(lstm_pack_frags_batch(n, fw.data() + k0, fhi.data() + k0, bhi.data() + k0, m->H, m->stream, 0));
        ADN_TRY(lstm_pack_frags_batch(n, fw.data() + k0, flo.data() + k0, blo.data() + k0, m->H, m->stream, 1));
    }
    m->params16_dirty = false;
    return ADN_OK;
}

int refresh_params(adn_model* m) {
    if (m->cfg.precision == ADN_PRECISION_BF16X3) return refresh_params_x3(m);
    if (!m->bf16()) return ADN_OK;
    ADN_TRY(ensure_splitk_ws(m));
    const bool persistent = lstm_persistent_supported(m->H);        // (an environment switch can flip it between calls)
    if (persistent != m->packed_for_persistent) m->params16_dirty = true;
    if (!m->params16_dirty) return ADN_OK;
    m->packed_for_persistent = persistent;
    ADN_TRY(ensure_params16(m));
    if (!m->params16_values_fresh) ADN_TRY(to_bf16(m->flat[ADN_BUF_PARAM], m->params16, m->flat_floats, m->stream));
    m->params16_values_fresh = false;
    std::vector<const float*> fw; std::vector<void*> ff, fb;       // fragment images to (re)build
    auto pack = [&](LstmParams& lp) -> int {
        if (!lp.whid16t) {           // + tail: the persistent kernel reads whole 32-k steps past a short last row
            const size_t bytes = ((size_t)m->ldg * lstm_ldk(m->H) + 1024) * 2;
            ADN_HIP_CHECK(hipMalloc((void**)&lp.whid16t, bytes));
            ADN_HIP_CHECK(hipMemsetAsync(lp.whid16t, 0, bytes, m->stream));
        }
        if (persistent) {            // the step kernels' transposed image is not read in this mode -- except at H > 256,
                                     // where they take over whenever a launch cannot keep a whole LSTM resident
            if (!lp.wfrag_fwd) {
                ADN_HIP_CHECK(hipMalloc((void**)&lp.wfrag_fwd, lstm_frag_elems(m->H) * 2));
                ADN_HIP_CHECK(hipMalloc((void**)&lp.wfrag_bwd, lstm_frag_elems(m->H) * 2));
            }
            fw.push_back(m->P(lp.W_hid)); ff.push_back(lp.wfrag_fwd); fb.push_back(lp.wfrag_bwd);
            if (m->H <= 256) return ADN_OK;
        }
        return lstm_pack_whid_t(m->P(lp.W_hid), lp.whid16t, m->H, m->stream);
    };
    for (auto& st : m->st) for (auto& lp : st.lstm) ADN_TRY(pack(lp));
    for (auto& lp : m->agg) ADN_TRY(pack(lp));
    // W_in images of the stream LSTMs whose forward kernel can multiply x_t W_in itself (lstm_cluster.hip, KXS > 0): one launch
    // per input width
    if (persistent && m->H <= 256) {
        std::vector<int> widths;
        for (auto& st : m->st) for (auto& lp : st.lstm) {
            if (lstm_win_frag_elems(lp.fin, m->H) == 0) continue;
            if (!lp.win_frag) ADN_HIP_CHECK(hipMalloc((void**)&lp.win_frag, lstm_win_frag_elems(lp.fin, m->H) * 2));
            if (std::find(widths.begin(), widths.end(), lp.fin) == widths.end()) widths.push_back(lp.fin);
        }
        for (int wdt : widths) {
            std::vector<const float*> wi; std::vector<void*> wo;
            for (auto& st : m->st) for (auto& lp : st.lstm) if (lp.fin == wdt && lp.win_frag) { wi.push_back(m->P(lp.W_in)); wo.push_back(lp.win_frag); }
            for (size_t k0 = 0; k0 < wi.size(); k0 += 8)
                ADN_TRY(lstm_pack_win_frags((int)std::min<size_t>(8, wi.size() - k0), wi.data() + k0, wo.data() + k0, wdt, m->H, m->stream));
        }
    }
    for (size_t k0 = 0; k0 < fw.size(); k0 += 8)                 // every fragment image in one launch (per 8 LSTMs)
        ADN_TRY(lstm_pack_frags_batch((int)std::min<size_t>(8, fw.size() - k0), fw.data() + k0, ff.data() + k0, fb.data() + k0,
                                      m->H, m->stream));
    if (m->cfg.fusion == ADN_FUSE_CONCAT && m->S > 1) {
        const size_t blk = (size_t)m->ldh * m->ldg;              // elements of one padded input block
        std::vector<const float*> win; std::vector<void*> wout;
        for (auto& lp : m->agg) {
            if (!lp.wcat16) {
                ADN_HIP_CHECK(hipMalloc((void**)&lp.wcat16, (size_t)m->S * blk * 2 + kPlaneSlack));
                ADN_HIP_CHECK(hipMemsetAsync(lp.wcat16, 0, (size_t)m->S * blk * 2 + kPlaneSlack, m->stream));
            }
            win.push_back(m->P(lp.W_in)); wout.push_back(lp.wcat16);
        }
        for (size_t k0 = 0; k0 < win.size(); k0 += 4)            // W_in [S*H][ldg] -> bf16 [S][ldh][ldg], all LSTMs at once
            ADN_TRY(repack_rows_bf16((int)std::min<size_t>(4, win.size() - k0), win.data() + k0, wout.data() + k0, m->S, m->H,
                                     m->ldh, m->ldg, m->stream));
    }
    ADN_TRY(refresh_transposed(m));
    m->params16_dirty = false;
    return ADN_OK;
}

LstmStep make_step(const adn_model* m, const LstmParams& lp, const LstmWork& w, const float* dhs, bool grads) {
    LstmStep s;
    s.W_hid = m->P(lp.W_hid);
    s.peep = lp.peepholes ? m->P(lp.peep) : nullptr;
    s.xproj = w.xproj; s.hbuf = w.hbuf; s.cbuf = w.cbuf; s.gates = w.gates;
    s.dG = w.dG; s.dhs = dhs; s.dh_carry = w.dh_carry; s.dc_state = w.dc_state;
    s.dpeep_part = (grads && lp.peepholes) ? m->G(lp.peep) : nullptr;
    s.backwards = lp.backwards ? 1 : 0;
    const bool b16 = m->bf16();      // the step kernels maintain their own bf16 copies (h16, dG16)
    s.W_hid16T = b16 ? lp.whid16t : nullptr;
    const bool x3 = m->cfg.precision == ADN_PRECISION_BF16X3 && m->H <= 512;
    s.W_frag_fwd = (b16 || x3) ? lp.wfrag_fwd : nullptr;
    s.W_frag_bwd = (b16 || x3) ? lp.wfrag_bwd : nullptr;
    s.W_frag_fwd_lo = x3 ? lp.wfrag_fwd_lo : nullptr;
    s.W_frag_bwd_lo = x3 ? lp.wfrag_bwd_lo : nullptr;
    s.W_hid16 = b16 ? m->shadow_of(m->P(lp.W_hid)) : nullptr;
    s.h16 = b16 ? m->shadow_of(w.hbuf) : nullptr;
    s.dG16 = b16 ? m->shadow_of(w.dG) : nullptr;
    static const bool dg_fp32 = getenv("ADN_LSTM_DG_FP32") != nullptr;       // (A/B: keep the fp32 copy of dG)
    if (grads && shadows_on(m) && s.dG16 && !m->keep_fp32 && !dg_fp32) s.dG_fp32_off = 1;       // bf16 mode: the bf16 copy is the only one read
    if (grads && x3 && m->planes() && !m->keep_fp32 && !dg_fp32) { s.dG16 = m->shadow_of(w.dG); s.dG16lo = m->shadow_lo_of(w.dG); if (!s.dG16 || !s.dG16lo) s.dG16 = s.dG16lo = nullptr; }
    // (dG leaves the weight-stationary backward kernels in the form its readers take -- the bf16 copy in bf16 mode, the two planes
    //  in bf16x3 mode -- INSTEAD of as fp32: the same or fewer bytes per step from the kernel, no split pass behind it.  Writing
    //  the planes in ADDITION to fp32 cost the bf16x3 step 0.3 - 0.7 us of its 6.4 and put the kernel below the 40 % line.)
    s.xchg = (b16 || x3) ? w.xchg : nullptr;
    s.xchg_seq = const_cast<unsigned*>(&w.xchg_seq);
    if (grads) { s.dbias = m->G(lp.b); s.dhid_init = m->G(lp.hid_init); s.dcell_init = m->G(lp.cell_init); }
    return s;
}

// Length buckets: the launch entries of n LSTMs -- one per (LSTM, bucket), the bucket's first row in every pointer (TmPlan)
static std::vector<LstmStep> expand_entries(const adn_model* m, const TmPlan& p, const LstmStep* l, int n) {
    std::vector<LstmStep> out;
    if (!p.on || p.nb < 2) { out.assign(l, l + n); return out; }
    const size_t ldh = (size_t)m->ldh, ldg = (size_t)m->ldg;
    const size_t xchg_bytes = lstm_cluster_xchg_bytes(p.Bb, m->H);
    auto rows_f = [](auto* q, size_t rows, size_t ld) { return q ? q + rows * ld : q; };
    auto rows_16 = [](auto* q, size_t rows, size_t ld) -> decltype(q) {
        typedef decltype(q) P;
        return q ? reinterpret_cast<P>(reinterpret_cast<uintptr_t>(q) + rows * ld * 2) : q;
    };
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < p.nb; ++k) {
            LstmStep e = l[i];
            const size_t r = (size_t)p.c[k] * p.Bb;
            e.xproj = rows_f(e.xproj, r, ldg); e.gates = rows_f(e.gates, r, ldg); e.dG = rows_f(e.dG, r, ldg);
            e.hbuf = rows_f(e.hbuf, r, ldh); e.cbuf = rows_f(e.cbuf, r, ldh);
            e.dhs = rows_f(e.dhs, r, e.ld_dhs ? (size_t)e.ld_dhs : ldh);
            e.dh_carry = rows_f(e.dh_carry, (size_t)k * p.Bb, ldh); e.dc_state = rows_f(e.dc_state, (size_t)k * p.Bb, ldh);
            e.h16 = rows_16(e.h16, r, ldh); e.dG16 = rows_16(e.dG16, r, ldg); e.dG16lo = rows_16(e.dG16lo, r, ldg);
            e.x16 = rows_16(e.x16, r, (size_t)e.ld_x);
            if (e.xchg) e.xchg = static_cast<char*>(e.xchg) + (size_t)k * xchg_bytes;
            e.T_own = p.Tk[k]; e.mask_own = m->mask_tb + r;
            out.push_back(e);
        }
    return out;
}
// LSTMs per launch: all kMaxLstmPerLaunch entries, or as many whole LSTMs as their buckets leave room for
static int lstms_per_launch(const adn_model* m) { return m->tm.on ? std::max(1, kMaxLstmPerLaunch / m->tm.nb) : kMaxLstmPerLaunch; }
// the gate-gradient rows of the spare blocks: never written by the LSTM kernels, read by every GEMM over all rows -- zero, in the
// copies the step's GEMMs read (rows of an earlier cut, or of a call over B x T rows, may lie there)
struct ZeroBlocksArgs { void* p[3 * 8]; int row_bytes[3 * 8]; int n; int blk[kMaxBuckets]; int nblk; int Bb; };
__global__ __launch_bounds__(256) void zero_spare_blocks_kernel(const ZeroBlocksArgs a) {
    const int q = blockIdx.y;
    if (q >= a.n) return;
    const size_t per = (size_t)a.Bb * a.row_bytes[q] / 16;                  // uint4 per block
    for (int k = 0; k < a.nblk; ++k) {
        uint4* dst = reinterpret_cast<uint4*>(static_cast<char*>(a.p[q]) + (size_t)a.blk[k] * a.Bb * a.row_bytes[q]);
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < per; e += (size_t)gridDim.x * 256) dst[e] = make_uint4(0u, 0u, 0u, 0u);
    }
}
// ... of every LSTM of the model, in every copy a GEMM may read (fp32, bf16 copy / hi plane, lo plane).  Called where the layout
// changes -- new tables, or a call over B x T rows in between, whose gate gradients fill those rows --, not per step: inside a run of
// bucketed steps nothing writes there (the LSTM kernels store their own blocks; whole-buffer conversions map zeros to zeros).
static int zero_spare_blocks(adn_model* m) {
    if (!m->tm.on || m->tm.nb < 2) return ADN_OK;
    ZeroBlocksArgs a{};
    a.Bb = m->tm.Bb;
    for (int k = 0; k + 1 < m->tm.nb; ++k) a.blk[a.nblk++] = m->tm.c[k] + m->tm.Tk[k];
    auto flush = [&]() -> int {
        if (!a.n) return ADN_OK;
        hipLaunchKernelGGL(zero_spare_blocks_kernel, dim3(32, a.n), dim3(256), 0, m->stream, a);
        ADN_HIP_CHECK(hipGetLastError());
        a.n = 0;
        return ADN_OK;
    };
    auto add = [&](LstmWork& w) -> int {
        if (a.n + 3 > 3 * 8) ADN_TRY(flush());
        a.p[a.n] = w.dG; a.row_bytes[a.n++] = m->ldg * 4;
        if (void* h = m->shadow_of(w.dG)) { a.p[a.n] = h; a.row_bytes[a.n++] = m->ldg * 2; }
        if (void* l = m->shadow_lo_of(w.dG)) { a.p[a.n] = l; a.row_bytes[a.n++] = m->ldg * 2; }
        return ADN_OK;
    };
    for (auto& st : m->st) for (auto& w : st.lw) ADN_TRY(add(w));
    for (auto& w : m->aggw) ADN_TRY(add(w));
    return flush();
}

// sums_done: the backward kernels already added the bias / initial-state gradients of every LSTM of the group
int flush_init_states(adn_model* m, int B);
int run_lstm_group(adn_model* m, std::vector<LstmStep>& steps, int B, int T, bool backward, bool* sums_done = nullptr) {
    ADN_TRY(flush_init_states(m, B));
    bool all = true;
    // (B, T: the time-major geometry -- with length buckets [Tt][Bb]; the launches then run Bb utterances over at most Tmax steps)
    const int per = lstms_per_launch(m), Bk = m->tm.on ? m->tm.Bb : B, Tk = m->tm.on ? m->tm.Tmax : T;
    std::vector<char> planes_done(steps.size(), 0);
    // ADN_PRECISION_MIXED: back-propagation through the recurrences on the bf16 mode's weight-stationary kernel -- one bf16 product
    // dG W_hid^T per step over the hi image of W_hid (the image bf16 mode packs), like the mode's backward GEMMs; dG leaves as its hi
    // plane alone, which is all those GEMMs read.  Saved state (gates, c: fp32) and the exchange protocol are the two kernels' common
    // ground; where the bf16 kernel cannot run (too many groups for the device, a switch) the bf16x3 one does as before.
    static const bool mixed_x3 = getenv("ADN_MIXED_LSTM_X3") != nullptr, dg_fp32 = getenv("ADN_LSTM_DG_FP32") != nullptr;   // (A/B)
    bool mixed16 = backward && m->bwd_hi_only && m->planes() && !m->keep_fp32 && !mixed_x3 && !dg_fp32 && lstm_persistent_supported(m->H) &&
                   !getenv("ADN_LSTM_NO_CLUSTER_BWD");
    std::vector<LstmStep> alt;
    if (mixed16) {
        alt = steps;
        for (auto& q : alt) {
            q.W_hid16 = m->shadow_of(q.W_hid); q.h16 = m->shadow_of(q.hbuf); q.dG16 = m->shadow_of(q.dG); q.dG16lo = nullptr;
            q.dG_fp32_off = 1; q.W_frag_fwd_lo = q.W_frag_bwd_lo = nullptr;
            if (!q.W_hid16 || !q.dG16 || !q.W_frag_bwd) mixed16 = false;
        }
        for (size_t i = 0; mixed16 && i < alt.size(); i += (size_t)per) {
            const std::vector<LstmStep> ex = expand_entries(m, m->tm, alt.data() + i, (int)std::min<size_t>((size_t)per, alt.size() - i));
            mixed16 = lstm_cluster_supported(ex.data(), (int)ex.size(), Bk, Tk, m->H);
        }
    }
    for (size_t i = 0; i < steps.size(); i += (size_t)per) {
        const int n = (int)std::min<size_t>((size_t)per, steps.size() - i);
        bool done = false;
        // (length buckets: one entry per (LSTM, bucket), Bb utterances each, its own step count -- the launch's T is the longest)
        const std::vector<LstmStep> ex = expand_entries(m, m->tm, (backward && mixed16 ? alt.data() : steps.data()) + i, n);
        const int ne = (int)ex.size();
        if (backward && mixed16) {
            ADN_TRY(lstm_backward(ex.data(), ne, m->mask_tb, Bk, Tk, m->H, ADN_PRECISION_BF16, m->stream, &done));
            ADN_CHECK(done, ADN_ERR_STATE, "mixed mode: the weight-stationary bf16 backward kernel was expected to run");
        } else if (backward)
            ADN_TRY(lstm_backward(ex.data(), ne, m->mask_tb, Bk, Tk, m->H,
                                  m->cfg.precision == ADN_PRECISION_BF16X3 ? ADN_PRECISION_BF16X3 : m->lstm_precision(), m->stream, &done));
        else ADN_TRY(lstm_forward(ex.data(), ne, m->mask_tb, Bk, Tk, m->H,
                                  m->cfg.precision == ADN_PRECISION_BF16X3 ? ADN_PRECISION_BF16X3 : m->lstm_precision(), m->stream));
        ADN_CHECK(!m->tm.on || !backward || done, ADN_ERR_STATE, "length buckets: the weight-stationary backward kernel was expected to run");
        all = all && done;
        // (bf16x3: `done` <=> the weight-stationary hi / lo kernel ran, which also writes the planes of dG when they are offered)
        for (int k = 0; k < n; ++k) planes_done[i + k] = backward && done && (mixed16 || (steps[i + k].dG16 && steps[i + k].dG16lo));
    }
    if (sums_done) *sums_done = backward && all;
    if (m->planes() && !backward && !streams_concurrent(m)) {
        // the state histories of the group as planes (they are the next GEMMs' operands): one launch for all of them
        const size_t count = (size_t)round_up((int64_t)(T + 1) * B * m->ldh, 8);
        for (size_t q0 = 0; q0 < steps.size(); q0 += kMaxSplitJobs) {
            const float* src[kMaxSplitJobs]; void* hi[kMaxSplitJobs]; void* lo[kMaxSplitJobs];
            int n = 0;
            for (size_t q = q0; q < std::min(steps.size(), q0 + kMaxSplitJobs); ++q) {
                void* h = m->shadow_of(steps[q].hbuf); void* l = m->shadow_lo_of(steps[q].hbuf);
                if (!h || !l) continue;
                src[n] = steps[q].hbuf; hi[n] = h; lo[n] = l; ++n;
            }
            if (n) ADN_TRY(split_hilo_batch(src, hi, lo, n, count, m->stream));
        }
    } else if (m->planes())          // (bf16 mode: the kernels write the 16-bit copies themselves)
        for (size_t q = 0; q < steps.size(); ++q) {
            const LstmStep& st_ = steps[q];
            if (backward) {
                if (!planes_done[q]) ADN_TRY(refresh(m, st_.dG, (size_t)B * T * m->ldg));
                else {                                   // fp32 dG was not written: a reader that needs it asks restore_fp32()
                    bool have = false;
                    const size_t count = ((size_t)B * T * m->ldg) | (mixed16 ? kStaleHiOnly : 0);
                    for (auto& e : m->fp32_stale) if (e.first == st_.dG) { e.second = count; have = true; }
                    if (!have) m->fp32_stale.push_back({st_.dG, count});
                }
            }
            else ADN_TRY(refresh(m, st_.hbuf, (size_t)(T + 1) * B * m->ldh));
        }
    return ADN_OK;
}

// a list of prepared GEMMs as grouped launches: neighbours in the list that share shape, strides and flags go out together
int issue_grouped(adn_model* m, std::vector<GemmArgs>& list) {
    std::vector<char> done(list.size(), 0);
    for (size_t i = 0; i < list.size(); ++i) {
        if (done[i]) continue;
        GemmArgs batch[kMaxGemmGroups];
        int nb = 0;
        batch[nb++] = list[i]; done[i] = 1;
        for (size_t j = i + 1; j < list.size() && nb < kMaxGemmGroups; ++j) {
            const GemmArgs& a = list[i]; const GemmArgs& b = list[j];
            if (done[j] || a.layout != b.layout || a.M != b.M || a.N != b.N || a.K != b.K || a.lda != b.lda || a.ldb != b.ldb ||
                a.ldc != b.ldc || a.accumulate != b.accumulate || (a.bias == nullptr) != (b.bias == nullptr) ||
                (a.A16 == nullptr) != (b.A16 == nullptr) || (a.B16 == nullptr) != (b.B16 == nullptr) || a.C == b.C)
                continue;
            batch[nb++] = b; done[j] = 1;
        }
        ADN_TRY(m_gemm_grouped(m, batch, nb));
    }
    return ADN_OK;
}

// x*W_in + b for one LSTM whose input is the (virtual) concatenation of `nblk` matrices of width `blkw`
int lstm_project(adn_model* m, const LstmParams& lp, const LstmWork& w, const float* const* in, const int* ld_in,
                 int nblk, int blkw, int rows) {
    for (int j = 0; j < nblk; ++j) {
        GemmArgs g;
        g.layout = GEMM_NN; g.M = rows; g.N = 4 * m->H; g.K = blkw;
        g.A = in[j]; g.lda = ld_in[j];
        g.B = m->P(lp.W_in) + (size_t)j * blkw * m->ldg; g.ldb = m->ldg;
        g.C = w.xproj; g.ldc = m->ldg;
        g.bias = (j == 0) ? m->P(lp.b) : nullptr;
        g.accumulate = j > 0;
        g.no_split = 1;                                          // forward pass: reproducible bits
        ADN_TRY(mgemm(m, g));
    }
    return ADN_OK;
}

// bf16 mode, concat fusion: the aggregation LSTMs consume one materialised [N][S*ldh] bf16 matrix
bool cat_path(const adn_model* m) {
    return (shadows_on(m) || m->planes()) && m->cfg.fusion == ADN_FUSE_CONCAT && m->S > 1 && m->S <= 4 && !m->agg.empty() &&
           m->cat16 && !getenv("ADN_NO_CAT");
}

// x*W_in + b with x = the materialised concat (ONE GEMM, K = S*ldh; the pad rows of wcat16 are zero)
GemmArgs lstm_project_cat_args(adn_model* m, const LstmParams& lp, const LstmWork& w, int rows) {
    GemmArgs g;
    g.layout = GEMM_NN; g.M = rows; g.N = 4 * m->H; g.K = m->S * m->ldh;
    g.A = reinterpret_cast<const float*>(m->cat16); g.lda = m->S * m->ldh;      // (only the bf16 operands are read)
    g.B = m->P(lp.W_in); g.ldb = m->ldg;
    g.A16 = m->cat16; g.B16 = lp.wcat16;
    if (m->planes()) { g.A16lo = m->cat16lo; g.B16lo = lp.wcat16lo; }
    g.C = w.xproj; g.ldc = m->ldg; g.bias = m->P(lp.b);
    g.precision = m->cfg.precision;
    g.no_split = 1;                                              // forward pass: reproducible bits
    return g;
}

// the concat path's backward GEMMs of aggregation LSTM k: dW_in for all S blocks (cat^T dG into a scratch matrix), and
// d(concat) (+)= dG W_in^T through the side-by-side W^T copies
GemmArgs cat_dw_args(adn_model* m, size_t k, bool group_dw, int N) {
    const int ldcat = m->S * m->ldh;
    const LstmWork& w = m->aggw[k];
    GemmArgs g;
    g.layout = GEMM_TN; g.M = ldcat; g.N = 4 * m->H; g.K = N;
    g.A = reinterpret_cast<const float*>(m->cat16); g.lda = ldcat; g.A16 = m->cat16;
    g.B = w.dG; g.ldb = m->ldg; g.B16 = m->shadow_of(w.dG);
    if (m->planes()) { g.A16lo = m->cat16lo; g.B16lo = m->shadow_lo_of(w.dG); }
    g.C = m->wcat_tmp + (group_dw ? k * (size_t)ldcat * m->ldg : 0); g.ldc = m->ldg; g.precision = m->cfg.precision;
    g.splitk_ws = m->splitk_ws; g.splitk_ws_floats = m->splitk_ws_floats;      // (main stream: before the fork)
    return g;
}
GemmArgs cat_dx_args(adn_model* m, size_t k, int N) {
    const int ldcat = m->S * m->ldh;
    const LstmParams& lp = m->agg[k]; const LstmWork& w = m->aggw[k];
    GemmArgs d;
    d.layout = GEMM_NN; d.M = N; d.N = ldcat; d.K = 4 * m->H;
    d.A = w.dG; d.lda = m->ldg; d.A16 = m->shadow_of(w.dG);
    d.B = m->P(lp.W_in); d.ldb = ldcat;
    for (const auto& t : m->transw) if (t.key == d.B) { d.B16 = t.buf; break; }
    if (m->planes()) { d.A16lo = m->shadow_lo_of(w.dG); d.B16lo = d.B16 ? static_cast<const char*>(d.B16) + m->transw_slab_bytes : nullptr; }
    d.C = m->dcat; d.ldc = ldcat; d.accumulate = k > 0; d.precision = m->cfg.precision;
    return d;
}

// Batched housekeeping pays where its launches are latency-bound -- the reference's own minibatches: B = 26, 1.276 -> 1.251 ms per
// train step -- and costs at the whole-split batch (B = 520: 3.880 -> 3.913 ms, three times measured): small batches only.
bool batched_housekeeping(const adn_model* m, int B, int T) {
    static const bool off = getenv("ADN_NO_BATCHED_HOUSEKEEPING") != nullptr;
    return !off && !streams_concurrent(m) && (int64_t)B * T <= 8192;
}
// queued: run_lstm_group() flushes the queue as one launch ahead of the LSTM kernels (with the streams on forked HIP streams every
// job goes out at once, on its stream)
int flush_init_states(adn_model* m, int B) {
    if (m->init_q.empty()) return ADN_OK;
    const int rc = lstm_init_state_batch(m->init_q.data(), (int)m->init_q.size(), m->ldh, B, m->H, m->stream);
    m->init_q.clear();
    return rc;
}
int lstm_init_state(adn_model* m, const LstmParams& lp, const LstmWork& w, int B, int T) {
    char* h16 = m->bf16() ? static_cast<char*>(m->shadow_of(w.hbuf)) : nullptr;     // bf16 copy of the initial-state block
    // (length buckets: B = Bb, and every bucket has an initial-state block of its own -- its first, or its last for a backwards LSTM)
    for (int k = 0; k < (m->tm.on ? m->tm.nb : 1); ++k) {
        const int first = m->tm.on ? m->tm.c[k] : 0, steps_k = m->tm.on ? m->tm.Tk[k] : T;
        const size_t blk = (size_t)(first + (lp.backwards ? steps_k : 0)) * B * m->ldh;
        m->init_q.push_back(LstmInitJob{m->P(lp.hid_init), m->P(lp.cell_init), w.hbuf + blk, w.cbuf + blk, h16 ? h16 + blk * 2 : nullptr});
    }
    // (queued at every batch size since round 6: the jobs are 5 us of launch latency each whatever B is, and their first reader is
    //  the LSTM launch that run_lstm_group() puts behind the flush; forked streams keep one job per stream)
    if (streams_concurrent(m)) return flush_init_states(m, B);
    return ADN_OK;
}
// the delta layers of the streams: queued while consecutive jobs share direction and window, flushed by the first reader
int flush_deltas(adn_model* m, int B, int T) {
    if (m->delta_q.empty()) return ADN_OK;
    int rc = ADN_OK;
    for (size_t k0 = 0; k0 < m->delta_q.size() && rc == ADN_OK; k0 += kMaxDeltaJobs) {
        const int n = (int)std::min<size_t>(kMaxDeltaJobs, m->delta_q.size() - k0);
        rc = m->delta_q_fwd ? delta_forward_batch(m->delta_q.data() + k0, n, B, T, m->delta_q_theta, m->stream)
                            : delta_backward_batch(m->delta_q.data() + k0, n, B, T, m->delta_q_theta, m->stream);
    }
    m->delta_q.clear();
    return rc;
}
// (the streams' delta layers as ONE launch at every batch size since round 6: at B = 520 three launches of 12 us each are latency-
//  and tail-bound too -- 2.877 -> 2.856 ms per bf16 step, 5.694 -> 5.645 bf16x3, alternating runs on one box)
static bool deltas_batched(const adn_model* m) {
    static const bool off = getenv("ADN_NO_BATCHED_HOUSEKEEPING") != nullptr;
    return !off && !streams_concurrent(m);
}
int queue_delta(adn_model* m, bool fwd, const DeltaJob& j, int B, int T, int theta, bool flush_now) {
    if (!m->delta_q.empty() && (m->delta_q_fwd != fwd || m->delta_q_theta != theta)) ADN_TRY(flush_deltas(m, B, T));
    m->delta_q_fwd = fwd; m->delta_q_theta = theta;
    m->delta_q.push_back(j);
    if (flush_now || !deltas_batched(m)) return flush_deltas(m, B, T);
    return ADN_OK;
}

// ------------------------------------------------------------------------------------------
// per-input-stream concurrency: fork the model's stream into S side streams and join them again
// ------------------------------------------------------------------------------------------
// Opt-in (ADN_STREAMS=1): measured on the benchmark workload the kernels do overlap (9.1 ms of kernel time in a 5.1 ms
// step, 4 hardware queues) but the step is not shorter -- the work is throughput-bound, the overlapped kernels just
// run slower -- and per-kernel event timings lose their meaning, so the default keeps everything on one stream.
bool streams_concurrent(const adn_model* m) { return m->S > 1 && getenv("ADN_STREAMS") != nullptr; }

int ensure_side_streams(adn_model* m) {
    if (m->side_ready) return ADN_OK;
    ADN_HIP_CHECK(hipEventCreateWithFlags(&m->fork_ev, hipEventDisableTiming));
    for (int k = 0; k < m->S; ++k) {
        ADN_HIP_CHECK(hipStreamCreateWithFlags(&m->side[k], hipStreamNonBlocking));
        ADN_HIP_CHECK(hipEventCreateWithFlags(&m->join_ev[k], hipEventDisableTiming));
    }
    m->side_ready = true;
    return ADN_OK;
}

// every side stream waits for what the model's stream has enqueued so far
int fork_streams(adn_model* m) {
    if (!streams_concurrent(m)) return ADN_OK;
    ADN_TRY(ensure_side_streams(m));
    ADN_HIP_CHECK(hipEventRecord(m->fork_ev, m->stream));
    for (int k = 0; k < m->S; ++k) ADN_HIP_CHECK(hipStreamWaitEvent(m->side[k], m->fork_ev, 0));
    return ADN_OK;
}

// the model's stream waits for everything the side streams have enqueued
int join_streams(adn_model* m) {
    if (!streams_concurrent(m)) return ADN_OK;
    for (int k = 0; k < m->S; ++k) {
        ADN_HIP_CHECK(hipEventRecord(m->join_ev[k], m->side[k]));
        ADN_HIP_CHECK(hipStreamWaitEvent(m->stream, m->join_ev[k], 0));
    }
    return ADN_OK;
}

// work enqueued while one of these is alive goes to input stream k's side stream (every helper reads m->stream)
struct OnSideStream {
    adn_model* m; hipStream_t saved;
    OnSideStream(adn_model* m_, int k) : m(m_), saved(m_->stream) { if (streams_concurrent(m)) m->stream = m->side[k]; }
    ~OnSideStream() { m->stream = saved; }
};

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
int forward_pass(adn_model* m, int B0, int T0, int theta, bool want_loss, bool want_dz) {
    // (B0, T0): the call's batch.  (B, T): the geometry of the time-major tensors -- the same, or with length buckets (TmPlan) Bb rows
    // per block over Tt chained blocks; everything behind the delta layer only ever sees that.  The encoders' own row count is Nc
    // then (a bucketed call is a compacted call), the delta layer's kernels take (B0, T0) and the tables.
    const int B = m->tm.on ? m->tm.Bb : B0, T = m->tm.on ? m->tm.Tt : T0;
    const int N = B * T, H = m->H, ldh = m->ldh;
    m->probs_partial = m->tm.on;
    auto tm_job = [&](DeltaJob& dj) { if (m->tm.on) { dj.tm_row0 = m->tm_row0; dj.tm_T = m->tm_T; dj.tm_stride = m->tm.Bb; } };
    hipStream_t s = m->stream;
    std::vector<LstmStep> steps;
    ADN_TRY(ensure_splitk_ws(m));                                // (bf16x3 mode: the weight gradients' partial slabs)
    ADN_TRY(fork_streams(m));
    auto enc_gemm = [&](StreamState& st, int l) {                // modelzoo/pretrained_encoder.py:4-9
        GemmArgs g;
        g.layout = GEMM_NN; g.M = m->compact ? m->Nc : (int)N; g.N = st.cfg.enc_units[l]; g.K = st.enc_in[l];      // (compact.hip: valid frames + the zero row)
        g.A = l ? st.act[l - 1] : st.x; g.lda = l ? ld_of(st.enc_in[l]) : st.ldx;
        g.B = m->P(st.encW[l]); g.ldb = ld_of(g.N);
        g.C = st.act[l]; g.ldc = ld_of(g.N); g.bias = m->P(st.encb[l]); g.act = m->act_code(st.cfg.enc_act[l]);
        g.no_split = 1;                                          // forward pass: reproducible bits
        // (the encoder's OUTPUT feeds no GEMM -- the delta layer / BatchNorm read it in fp32 --: no planes of it; round 5 split it anyway)
        g.no_planes = (l + 1 == st.cfg.n_enc) ? 1 : 0;
        // (bf16: the rectifier's mask for this layer's input gradient leaves with the activation, as one bit per element in the
        //  kernel's own thread order -- read back by the input-gradient launch of the same tile grid instead of the bf16 activation)
        st.bits_tiles[l] = 0;
        if (st.relu_bits[l] && (m->bf16() || m->planes()) && g.act == ADN_ACT_RECTIFY) { g.Cbits = st.relu_bits[l]; g.bits_done = &st.bits_tiles[l]; }
        mgemm_prepare(m, g, /*lean=*/l + 1 < st.cfg.n_enc);      // the delta layer reads the last one in fp32
        // (bf16x3: a narrow next layer -- the 50-unit bottleneck -- reads the planes too since round 5 (gemm_skinny.hip); with those
        //  kernels switched off it multiplies over split images of the fp32 values, and writing them here is cheaper than writing
        //  hi + lo back ahead of that layer)
        static const bool no_skinny = getenv("ADN_GEMM_NO_SKINNY") != nullptr;
        if (no_skinny && m->planes() && l + 1 < st.cfg.n_enc && st.cfg.enc_units[l + 1] < 128) g.lean_ok = 0;
        return g;
    };
    // Encoders layer by layer; the streams whose layer l has the same geometry go out as ONE grouped launch (their
    // tiles share a list: the AVLetters model's 3 x 656 tiles fill 7.7 rounds of 256 CUs, one stream alone 2.6).
    const bool grouped = !streams_concurrent(m);
    if (grouped) {
        int max_enc = 0;
        for (auto& st : m->st) max_enc = std::max(max_enc, st.cfg.n_enc);
        for (int l = 0; l < max_enc; ++l) {
            std::vector<char> done(m->st.size(), 0);
            for (size_t i = 0; i < m->st.size(); ++i) {
                if (done[i] || m->st[i].cfg.n_enc <= l) continue;
                GemmArgs batch[kMaxGemmGroups];
                int nb = 0;
                batch[nb++] = enc_gemm(m->st[i], l);
                done[i] = 1;
                for (size_t j = i + 1; j < m->st.size() && nb < kMaxGemmGroups; ++j) {
                    StreamState& o = m->st[j];
                    if (done[j] || o.cfg.n_enc <= l) continue;
                    GemmArgs c = enc_gemm(o, l);
                    const GemmArgs& r = batch[0];
                    if (c.N == r.N && c.K == r.K && c.lda == r.lda && c.act == r.act && (c.C == nullptr) == (r.C == nullptr) &&
                        (c.A16 == nullptr) == (r.A16 == nullptr) && (o.cfg.n_enc == l + 1) == (m->st[i].cfg.n_enc == l + 1)) {
                        batch[nb++] = c; done[j] = 1;
                    }
                }
                ADN_TRY(m_gemm_grouped(m, batch, nb));
            }
        }
    }
    std::vector<GemmArgs> stream_proj;
    for (auto& st : m->st) {
        OnSideStream on(m, (int)(&st - m->st.data()));           // encoder, delta layer, input projection of this stream
        const float* a = st.x; int lda = st.ldx;
        for (int l = 0; l < st.cfg.n_enc; ++l) {
            if (!grouped) { GemmArgs g = enc_gemm(st, l); ADN_TRY(m_gemm(m, g)); }
            a = st.act[l]; lda = ld_of(st.cfg.enc_units[l]);
        }
        if (st.cfg.batchnorm) {                                  // BatchNormLayer on the (B*T, E) encoder output (adenet_v1.py:82)
            const int ldE = ld_of(st.enc_out);
            if (m->training)
                ADN_TRY(batchnorm_forward_train(a, lda, st.bn_out, ldE, N, st.enc_out, m->P(st.bn_gamma), m->P(st.bn_beta), kBnEps,
                                                kBnAlpha, st.bn_save_mean, st.bn_save_inv_std, m->P(st.bn_mean), m->P(st.bn_inv_std),
                                                st.bn_ws, m->stream));
            else
                ADN_TRY(batchnorm_forward_eval(a, lda, st.bn_out, ldE, N, st.enc_out, m->P(st.bn_gamma), m->P(st.bn_beta),
                                               m->P(st.bn_mean), m->P(st.bn_inv_std), m->stream));
            a = st.bn_out; lda = ldE;
        }
        // (compact.hip) the delta layer spans B T frames and reads the compact encoder output through the row map: every padding
        // frame sees row Z = enc(0) -- no expanded copy
        const int32_t* enc_rows = (m->compact && st.cfg.n_enc > 0) ? m->comp_of_full : nullptr;
        const bool drop = m->stochastic && st.cfg.dropout_p > 0.f;
        // the 16-bit copies of the LSTM input are written by the delta kernel itself: the bf16 copy, or both planes
        void* feat16 = (m->bf16() && !drop) ? m->shadow_of(st.feat) : nullptr;
        void* feat16lo = nullptr;
        if (m->planes() && !drop && m->shadow_of(st.feat) && m->shadow_lo_of(st.feat)) { feat16 = m->shadow_of(st.feat); feat16lo = m->shadow_lo_of(st.feat); }
        // (queued: the streams' delta layers go out as ONE launch ahead of their first reader -- the dropout / bf16-copy pass
        //  right below where a stream has one, else the grouped input projections behind this loop)
        {
            DeltaJob dj{a, lda, st.feat, ld_of(st.feat_dim), st.enc_out, st.cfg.use_delta, feat16};
            dj.dst16lo = feat16lo; dj.row_map = enc_rows; tm_job(dj);
            ADN_TRY(queue_delta(m, true, dj, B0, T0, theta, false));
        }
        if (st.cfg.aux_dim > 0) {                                // ConcatLayer([l_delta, l_dct], axis=2): columns behind the deltas
            DeltaJob dj{st.aux_stage, ld_of(st.cfg.aux_dim), st.feat + st.delta_dim, ld_of(st.feat_dim), st.cfg.aux_dim, 0,
                        feat16 ? static_cast<void*>(static_cast<char*>(feat16) + 2 * (size_t)st.delta_dim) : nullptr};
            dj.dst16lo = feat16lo ? static_cast<void*>(static_cast<char*>(feat16lo) + 2 * (size_t)st.delta_dim) : nullptr;
            ADN_TRY(queue_delta(m, true, dj, B0, T0, theta, false));
        }
        if (drop || !feat16 || !grouped) ADN_TRY(flush_deltas(m, B0, T0));
        if (drop)                                                // DropoutLayer ahead of the LSTM (adenet_v3.py:112,123,134)
            ADN_TRY(dropout_apply(st.feat, ld_of(st.feat_dim), st.feat, ld_of(st.feat_dim), B, T, st.feat_dim, st.feat_dim, 0,
                                  st.cfg.dropout_p, m->drop_seed, m->drop_counter, (uint32_t)(&st - m->st.data()), m->stream));
        if (!feat16) ADN_TRY(refresh(m, st.feat, (size_t)N * ld_of(st.feat_dim)));
        for (size_t k = 0; k < st.lstm.size(); ++k) {
            const float* in[1] = {st.feat}; const int ld[1] = {ld_of(st.feat_dim)};
            if (grouped) {                                       // the streams' projections go out together below
                GemmArgs g;
                g.layout = GEMM_NN; g.M = N; g.N = 4 * m->H; g.K = st.feat_dim;
                g.A = st.feat; g.lda = ld[0]; g.B = m->P(st.lstm[k].W_in); g.ldb = m->ldg;
                g.C = st.lw[k].xproj; g.ldc = m->ldg; g.bias = m->P(st.lstm[k].b); g.no_split = 1;
                mgemm_prepare(m, g, false);
                stream_proj.push_back(g);
            } else {
                ADN_TRY(lstm_project(m, st.lstm[k], st.lw[k], in, ld, 1, st.feat_dim, N));
            }
            ADN_TRY(lstm_init_state(m, st.lstm[k], st.lw[k], B, T));
            steps.push_back(make_step(m, st.lstm[k], st.lw[k], nullptr, false));
            if (grouped && m->bf16() && st.lstm[k].win_frag && m->shadow_of(st.feat)) {
                // offer: the forward kernel may multiply x_t W_in + b itself (decided per launch group below)
                LstmStep& q = steps.back();
                q.x16 = m->shadow_of(st.feat); q.ld_x = ld[0]; q.Kx = st.feat_dim;
                q.W_in_frag = st.lstm[k].win_frag; q.b_in = m->P(st.lstm[k].b);
            }
        }
    }
    ADN_TRY(flush_deltas(m, B0, T0));
    ADN_TRY(join_streams(m));
    if (grouped && stream_proj.size() == steps.size()) {
        // launch groups as run_lstm_group forms them: where the kernel folds the projection in, its GEMM is dropped; elsewhere
        // the offer is withdrawn, so that exactly one of the two computes it
        std::vector<GemmArgs> kept;
        const size_t per = (size_t)lstms_per_launch(m);
        for (size_t i = 0; i < steps.size(); i += per) {
            const int n = (int)std::min<size_t>(per, steps.size() - i);
            const std::vector<LstmStep> ex = expand_entries(m, m->tm, steps.data() + i, n);      // (the entries run_lstm_group will launch)
            const bool fold = lstm_forward_folds_projection(ex.data(), (int)ex.size(), m->tm.on ? m->tm.Bb : B, m->tm.on ? m->tm.Tmax : T, m->H, m->lstm_precision());
            for (int k = 0; k < n; ++k) {
                if (fold) continue;
                steps[i + k].x16 = nullptr; steps[i + k].W_in_frag = nullptr;
                kept.push_back(stream_proj[i + k]);
            }
        }
        stream_proj.swap(kept);
    } else {
        for (auto& q : steps) { q.x16 = nullptr; q.W_in_frag = nullptr; }
    }
    ADN_TRY(issue_grouped(m, stream_proj));
    ADN_TRY(run_lstm_group(m, steps, B, T, false));
    for (auto& st : m->st) {
        if (st.lstm.size() == 2) {                               // summed BLSTM sub-stream
            const float* in[2] = {st.lw[0].out(B, ldh, false), st.lw[1].out(B, ldh, true)};
            // (the 16-bit copies the next GEMMs read leave with the sum: the bf16 copy, or both planes)
            void* s16 = (m->bf16() || m->planes()) ? m->shadow_of(st.hsum) : nullptr;
            void* s16lo = (m->planes() && s16) ? m->shadow_lo_of(st.hsum) : nullptr;
            if (m->planes() && !s16lo) s16 = nullptr;
            ADN_TRY(sum_k(2, in, nullptr, ldh, st.hsum, ldh, N, H, s, s16, s16lo));
            if (!s16) ADN_TRY(refresh(m, st.hsum, (size_t)N * ldh));
            st.out_ptr = st.hsum;
        } else {
            st.out_ptr = st.lw[0].out(B, ldh, false);
        }
    }
    // fusion (modelzoo/adenet_v2.py:68-75; custom/layers.py:178-228)
    std::vector<const float*> fin; std::vector<int> fld;
    const int fusion = m->cfg.fusion;
    if (fusion == ADN_FUSE_CONCAT || fusion == ADN_FUSE_NONE || m->S == 1) {
        if (fusion == ADN_FUSE_ADASUM) {                         // single stream, still scaled
            ADN_TRY(scale_by(m->st[0].out_ptr, ldh, m->P(m->adacoeff), m->fused, ldh, N, H, s));
            ADN_TRY(refresh(m, m->fused, (size_t)N * ldh));
            fin.push_back(m->fused); fld.push_back(ldh);
        } else {
            for (auto& st : m->st) { fin.push_back(st.out_ptr); fld.push_back(ldh); }
        }
    } else {
        const float* in[ADN_MAX_STREAMS]; const float* al[ADN_MAX_STREAMS];
        for (int k = 0; k < m->S; ++k) {
            in[k] = m->st[k].out_ptr;
            al[k] = fusion == ADN_FUSE_ADASUM ? m->P(m->adacoeff + k) : nullptr;
        }
        ADN_TRY(sum_k(m->S, in, al, ldh, m->fused, ldh, N, H, s));
        ADN_TRY(refresh(m, m->fused, (size_t)N * ldh));
        fin.push_back(m->fused); fld.push_back(ldh);
    }
    if (m->stochastic && m->cfg.agg_dropout_p > 0.f) {          // dropout_agg on the fused tensor (adenet_v3.py:154)
        const int W = (int)fin.size() * H;
        for (size_t j = 0; j < fin.size(); ++j) {
            // a stream output is also the LSTM's own state history: the dropped copy goes to a buffer of its own
            float* dst = (fin[j] == m->fused) ? m->fused : m->st[j].out_drop;
            ADN_TRY(dropout_apply(fin[j], ldh, dst, ldh, B, T, H, W, (int)j * H, m->cfg.agg_dropout_p, m->drop_seed,
                                  m->drop_counter, 100u, s));
            ADN_TRY(refresh(m, dst, (size_t)N * ldh));
            fin[j] = dst;
        }
    }
    m->fused_in = fin;
    const float* cls = nullptr;
    if (!m->agg.empty()) {                                       // custom/layers.py:55-80
        steps.clear();
        bool cat = cat_path(m) && (int)fin.size() == m->S;
        m->cat_planes_ok = false;
        if (cat && m->planes()) {
            // the concat exists only as planes: the path is taken when every GEMM that reads it (projection, weight gradient,
            // input gradient with and without accumulation) runs over planes at this shape -- none may fall back to images
            GemmArgs pr[kMaxGemmGroups];
            const int np = (int)std::min<size_t>(m->agg.size(), kMaxGemmGroups);
            for (int k = 0; k < np; ++k) pr[k] = lstm_project_cat_args(m, m->agg[k], m->aggw[k], N);
            for (int k = 0; k < np; ++k) pr[k].precision = ADN_PRECISION_BF16X3;
            cat = gemm_planes_would_run(pr, np) && m->agg.size() <= (size_t)kMaxGemmGroups;
            const bool group_dw = m->agg.size() >= 2 && m->agg.size() <= m->wcat_tmp_slots;
            for (int k = 0; k < np && cat; ++k) pr[k] = cat_dw_args(m, (size_t)k, group_dw, N);
            cat = cat && gemm_planes_would_run(pr, group_dw ? np : 1);
            for (size_t k = 0; k < m->agg.size() && cat; ++k) { GemmArgs d = cat_dx_args(m, k, N); cat = d.B16 && gemm_planes_would_run(&d, 1); }
            m->cat_planes_ok = cat;
        }
        if (cat) {
            const void* in16[4]; const void* in16lo[4];
            for (int j = 0; j < m->S; ++j) { in16[j] = m->shadow_of(fin[j]); in16lo[j] = m->planes() ? m->shadow_lo_of(fin[j]) : nullptr; }
            // (bf16x3 / mixed: the hi and the lo concat in one launch)
            ADN_TRY(concat_cols_bf16(m->S, in16, ldh, m->cat16, m->S * ldh, N, ldh, s, m->planes() ? in16lo : nullptr, m->planes() ? m->cat16lo : nullptr));
        }
        if (cat) {                                               // the pair's projections share the concat: one grouped launch
            GemmArgs gs[kMaxGemmGroups];
            size_t k = 0;
            while (k < m->agg.size()) {
                int n = 0;
                for (; k < m->agg.size() && n < kMaxGemmGroups; ++k) gs[n++] = lstm_project_cat_args(m, m->agg[k], m->aggw[k], N);
                ADN_TRY(m_gemm_grouped(m, gs, n));
            }
        }
        for (size_t k = 0; k < m->agg.size(); ++k) {
            if (!cat) ADN_TRY(lstm_project(m, m->agg[k], m->aggw[k], fin.data(), fld.data(), (int)fin.size(), H, N));
            ADN_TRY(lstm_init_state(m, m->agg[k], m->aggw[k], B, T));
            steps.push_back(make_step(m, m->agg[k], m->aggw[k], nullptr, false));
        }
        ADN_TRY(run_lstm_group(m, steps, B, T, false));
        if (m->agg.size() == 2) {
            const float* in[2] = {m->aggw[0].out(B, ldh, false), m->aggw[1].out(B, ldh, true)};
            void* c16 = (m->bf16() || m->planes()) ? m->shadow_of(m->cls_in) : nullptr;
            void* c16lo = (m->planes() && c16) ? m->shadow_lo_of(m->cls_in) : nullptr;
            if (m->planes() && !c16lo) c16 = nullptr;
            ADN_TRY(sum_k(2, in, nullptr, ldh, m->cls_in, ldh, N, H, s, c16, c16lo));
            if (!c16) ADN_TRY(refresh(m, m->cls_in, (size_t)N * ldh));
            cls = m->cls_in;
        } else {
            cls = m->aggw[0].out(B, ldh, false);
        }
    } else {
        ADN_CHECK(fin.size() == 1, ADN_ERR_INVALID, "classifier needs a single fused tensor");
        cls = fin[0];
    }
    if (m->head_last()) {
        // SliceLayer(-1) + Dense(C) + softmax on the LAST row of the padded tensor (adenet_v3.py:180-186, App. E-3);
        // time-major: block T-1 is B contiguous rows
        GemmArgs g;
        g.layout = GEMM_NN; g.M = B; g.N = m->C; g.K = H; g.A = cls + (size_t)(T - 1) * B * ldh; g.lda = ldh;
        g.B = m->P(m->smW); g.ldb = m->ldc; g.C = m->z; g.ldc = m->ldc; g.bias = m->P(m->smb); g.no_split = 1;
        ADN_TRY(mgemm(m, g));
        ADN_TRY(softmax_ce(m->z, m->ldc, B, T, m->C, want_loss ? m->y_src : nullptr, m->total, m->probs_bt,
                           want_loss ? m->row_loss : nullptr, want_dz ? m->dz : nullptr, m->ldc, s));
        if (want_dz) ADN_TRY(refresh(m, m->dz, (size_t)B * m->ldc));
        if (want_loss) ADN_TRY(reduce_loss(m->row_loss, B, m->total, m->loss, s));
        m->lastB = B0; m->lastT = T0;
        return ADN_OK;
    }
    {   // Dense(C) + softmax per frame (modelzoo/adenet_v2.py:89-92)
        GemmArgs g;
        g.layout = GEMM_NN; g.M = N; g.N = m->C; g.K = H; g.A = cls; g.lda = ldh;
        g.B = m->P(m->smW); g.ldb = m->ldc; g.C = m->z; g.ldc = m->ldc; g.bias = m->P(m->smb); g.no_split = 1;
        ADN_TRY(mgemm(m, g));
    }
    // (length buckets: which frame a row holds comes from the table; rows without one get a zero gradient and no loss)
    const bool dz_planes = m->planes() && m->shadow_of(m->dz) && m->shadow_lo_of(m->dz);      // bf16x3 / mixed: dz leaves as fp32 + both planes
    ADN_TRY(softmax_loss(m->z, m->ldc, B0, T0, m->C, m->mask_tb, want_loss ? m->y_src : nullptr, m->total, m->probs_bt,
                         want_loss ? m->row_loss : nullptr, want_dz ? m->dz : nullptr, m->ldc, s,
                         (want_dz && (m->bf16() || dz_planes)) ? m->shadow_of(m->dz) : nullptr,      // (pad columns of dz stay zero in every copy)
                         m->tm.on ? m->tm_bt : nullptr, N, (want_dz && dz_planes) ? m->shadow_lo_of(m->dz) : nullptr));
    if (want_dz && !m->bf16() && !dz_planes) ADN_TRY(refresh(m, m->dz, (size_t)N * m->ldc));
    if (want_loss) ADN_TRY(reduce_loss(m->row_loss, N, m->total, m->loss, s));
    m->lastB = B0; m->lastT = T0;
    return ADN_OK;
}

const float* classifier_input(const adn_model* m, int B) {
    if (!m->agg.empty()) return m->agg.size() == 2 ? m->cls_in : m->aggw[0].out(B, m->ldh, false);
    if (m->cfg.fusion == ADN_FUSE_ADASUM || (m->S > 1)) return m->fused;
    return m->st[0].out_ptr;
}

// weight / bias / initial-state gradients of one LSTM after its BPTT sweep
int lstm_param_grads(adn_model* m, const LstmParams& lp, const LstmWork& w, const float* const* in, const int* ld_in,
                     int nblk, int blkw, int B, int T, bool sums_done) {
    const int N = B * T, H = m->H, ldh = m->ldh, ldg = m->ldg;
    hipStream_t s = m->stream;
    for (int j = 0; j < nblk; ++j) {                             // dW_in = X^T dG
        GemmArgs g;
        g.layout = GEMM_TN; g.M = blkw; g.N = 4 * H; g.K = N;
        g.A = in[j]; g.lda = ld_in[j]; g.B = w.dG; g.ldb = ldg;
        g.C = m->G(lp.W_in) + (size_t)j * blkw * ldg; g.ldc = ldg; g.accumulate = 1;
        ADN_TRY(mgemm(m, g));
    }
    {                                                            // dW_hid = H_prev^T dG  (one GEMM over all steps)
        GemmArgs g;
        g.layout = GEMM_TN; g.M = H; g.N = 4 * H; g.K = N;
        g.A = w.prev(B, ldh, lp.backwards); g.lda = ldh; g.B = w.dG; g.ldb = ldg;
        g.C = m->G(lp.W_hid); g.ldc = ldg; g.accumulate = 1;
        ADN_TRY(mgemm(m, g));
    }
    if (!sums_done) {
        ADN_TRY(restore_fp32(m, w.dG));
        ADN_TRY(col_sum(w.dG, ldg, N, 4 * H, m->G(lp.b), 1, s));
        ADN_TRY(col_sum(w.dh_carry, ldh, B, H, m->G(lp.hid_init), 1, s));
        ADN_TRY(col_sum(w.dc_state, ldh, B, H, m->G(lp.cell_init), 1, s));
    }
    return ADN_OK;
}

// the same for several LSTMs with ONE input block each: their dW_in GEMMs and their dW_hid GEMMs go out as grouped launches
// (equal shapes share a tile list in the ping-pong kernel; anything else falls back to one launch each)
struct LstmGradJob { const LstmParams* lp; const LstmWork* w; const float* in; int ld_in; int blkw; };
int lstm_param_grads_grouped(adn_model* m, const std::vector<LstmGradJob>& jobs, int B, int T, bool sums_done) {
    const int N = B * T, H = m->H, ldh = m->ldh, ldg = m->ldg;
    for (int pass = 0; pass < 2; ++pass) {                       // 0: dW_in = X^T dG, 1: dW_hid = H_prev^T dG
        size_t i = 0;
        while (i < jobs.size()) {
            GemmArgs gs[kMaxGemmGroups];
            int n = 0;
            const size_t first = i;
            for (; i < jobs.size() && n < kMaxGemmGroups; ++i) {
                const LstmGradJob& jb = jobs[i];
                if (pass == 0 && !jb.in) continue;               // (a concat-fused aggregation LSTM: its dW_in is made elsewhere)
                if (n && pass == 0 && (jb.blkw != jobs[first].blkw || jb.ld_in != jobs[first].ld_in)) break;
                GemmArgs& g = gs[n++];
                g.layout = GEMM_TN; g.N = 4 * H; g.K = N; g.B = jb.w->dG; g.ldb = ldg; g.ldc = ldg; g.accumulate = 1;
                if (pass == 0) { g.M = jb.blkw; g.A = jb.in; g.lda = jb.ld_in; g.C = m->G(jb.lp->W_in); }
                else { g.M = H; g.A = jb.w->prev(B, ldh, jb.lp->backwards); g.lda = ldh; g.C = m->G(jb.lp->W_hid); }
                mgemm_prepare(m, g, false);
            }
            if (n) ADN_TRY(m_gemm_grouped(m, gs, n));
        }
    }
    if (!sums_done)
        for (const LstmGradJob& jb : jobs) {
            ADN_TRY(col_sum(jb.w->dG, ldg, N, 4 * H, m->G(jb.lp->b), 1, m->stream));
            ADN_TRY(col_sum(jb.w->dh_carry, ldh, B, H, m->G(jb.lp->hid_init), 1, m->stream));
            ADN_TRY(col_sum(jb.w->dc_state, ldh, B, H, m->G(jb.lp->cell_init), 1, m->stream));
        }
    return ADN_OK;
}

// dX (+)= dG W_in^T for input block j
int lstm_input_grad(adn_model* m, const LstmParams& lp, const LstmWork& w, int j, int blkw, float* dx, int lddx, int rows,
                    bool accumulate) {
    GemmArgs g;
    g.layout = GEMM_NT; g.M = rows; g.N = blkw; g.K = 4 * m->H;
    g.A = w.dG; g.lda = m->ldg; g.B = m->P(lp.W_in) + (size_t)j * blkw * m->ldg; g.ldb = m->ldg;
    g.C = dx; g.ldc = lddx; g.accumulate = accumulate;
    return mgemm(m, g);
}

// ------------------------------------------------------------------------------------------
// backward (theano.grad of the loss wrt every parameter)
// ------------------------------------------------------------------------------------------
// ---- gradient buckets (data parallel).  One bucket per [fusion | aggregation | classifier | tail], per stream's "top" (everything
// behind its encoder: BatchNorm, LSTMs; the whole stream when it has no encoder) and per (stream, encoder layer) = [W_l | b_l].
// The list is in the order the ranges become final, which is the order an in-order communication stream must take them in:
//   layer-major (default):  tail, top_0 .. top_{S-1}, then depth by depth (depth d = layer L_s - 1 - d of stream s) every stream
//   stream-major:           tail, then per stream its top and its layers from the last to the first
// Which of the two a model uses is decided ONCE (adn_grad_buckets / adn_set_bucket_events, whichever comes first) from the
// same predicate back-propagation uses, and kept.
bool streams_concurrent(const adn_model* m);
int dp_stream_major(adn_model* m) {
    if (m->dp_order < 0)
        m->dp_order = (getenv("ADN_DP_STREAM_MAJOR") || getenv("ADN_NO_GROUPED_BACKWARD") || streams_concurrent(m)) ? 1 : 0;
    return m->dp_order;
}
size_t bucket_of_top(const adn_model* m, size_t si) {
    if (!m->dp_order) return 1 + si;
    size_t k = 1;
    for (size_t q = 0; q < si; ++q) k += 1 + (size_t)m->st[q].cfg.n_enc;
    return k;
}
// layer-major back-propagation, depth d: the streams whose layer L_s - 1 - d exists, grouped by geometry (one grouped
// launch per group) in the order the groups are issued
std::vector<std::vector<size_t>> depth_groups(const adn_model* m, int d) {
    std::vector<std::vector<size_t>> out;
    std::vector<char> done(m->st.size(), 0);
    for (size_t i = 0; i < m->st.size(); ++i) {
        const StreamState& a = m->st[i];
        if (done[i] || d >= a.cfg.n_enc) continue;
        std::vector<size_t> sis{i};
        done[i] = 1;
        const int la = a.cfg.n_enc - 1 - d;
        for (size_t j = i + 1; j < m->st.size() && sis.size() < (size_t)kMaxGemmGroups; ++j) {
            const StreamState& o = m->st[j];
            if (done[j] || d >= o.cfg.n_enc) continue;
            const int lo = o.cfg.n_enc - 1 - d;
            if (o.cfg.enc_units[lo] == a.cfg.enc_units[la] && o.enc_in[lo] == a.enc_in[la] && (lo == 0) == (la == 0) &&
                (lo == 0 || o.cfg.enc_act[lo - 1] == a.cfg.enc_act[la - 1])) { sis.push_back(j); done[j] = 1; }
        }
        out.push_back(sis);
    }
    return out;
}
size_t bucket_of_layer(const adn_model* m, size_t si, int l) {
    const int depth = m->st[si].cfg.n_enc - 1 - l;
    if (m->dp_order) return bucket_of_top(m, si) + 1 + (size_t)depth;
    size_t k = 1 + (size_t)m->S;
    for (int d = 0; d <= depth; ++d)
        for (const auto& grp : depth_groups(m, d))
            for (size_t q : grp) {
                if (d == depth && q == si) return k;
                ++k;
            }
    return k;
}
size_t bucket_count(const adn_model* m) {
    size_t n = 1 + (size_t)m->S;
    for (int s = 0; s < m->S; ++s) n += (size_t)m->st[s].cfg.n_enc;
    return n;
}

int backward_pass(adn_model* m, int B0, int T0, int theta) {
    struct InBackward { adn_model* m; explicit InBackward(adn_model* m_) : m(m_) { m->in_backward = true; } ~InBackward() { m->in_backward = false; } } in_backward(m);
    // (B, T): the time-major geometry, as in forward_pass -- with length buckets [Tt][Bb]; (B0, T0), the call's batch, is what the delta
    // layer's kernels and the padding sums walk.  (Everything between them and the encoders' GEMMs -- Nc rows -- is time-major.)
    const int B = m->tm.on ? m->tm.Bb : B0, T = m->tm.on ? m->tm.Tt : T0;
    const int N = B * T, H = m->H, ldh = m->ldh;
    auto tm_job = [&](DeltaJob& dj) { if (m->tm.on) { dj.tm_row0 = m->tm_row0; dj.tm_T = m->tm_T; dj.tm_stride = m->tm.Bb; } };
    hipStream_t s = m->stream;
    ADN_HIP_CHECK(hipMemsetAsync(m->flat[ADN_BUF_GRAD], 0, (m->flat_floats + kAuxFloats) * sizeof(float), s));
    // (the cost share, tail[0]: with the exchange status where a weight-stationary LSTM kernel may have run -- one launch for both)
    if (m->cfg.precision == ADN_PRECISION_F32)
        ADN_HIP_CHECK(hipMemcpyAsync(m->flat[ADN_BUF_GRAD] + m->flat_floats, m->loss, sizeof(float), hipMemcpyDeviceToDevice, s));
    const float* cls = classifier_input(m, B);
    if (m->head_last()) {   // classifier on the last time step only: every other row of d(cls) is zero
        const float* cl = cls + (size_t)(T - 1) * B * ldh;
        float* dl = m->dcls + (size_t)(T - 1) * B * ldh;
        GemmArgs g;
        g.layout = GEMM_TN; g.M = H; g.N = m->C; g.K = B; g.A = cl; g.lda = ldh; g.B = m->dz; g.ldb = m->ldc;
        g.C = m->G(m->smW); g.ldc = m->ldc; g.accumulate = 1;
        ADN_TRY(mgemm(m, g));
        ADN_TRY(col_sum(m->dz, m->ldc, B, m->C, m->G(m->smb), 1, s));
        ADN_HIP_CHECK(hipMemsetAsync(m->dcls, 0, (size_t)N * ldh * sizeof(float), s));
        GemmArgs d;
        d.layout = GEMM_NT; d.M = B; d.N = H; d.K = m->C; d.A = m->dz; d.lda = m->ldc; d.B = m->P(m->smW); d.ldb = m->ldc;
        d.C = dl; d.ldc = ldh;
        ADN_TRY(mgemm(m, d));
    } else
    {   // classifier
        GemmArgs g;
        g.layout = GEMM_TN; g.M = H; g.N = m->C; g.K = N; g.A = cls; g.lda = ldh; g.B = m->dz; g.ldb = m->ldc;
        g.C = m->G(m->smW); g.ldc = m->ldc; g.accumulate = 1;
        ADN_TRY(mgemm(m, g));
        ADN_TRY(col_sum(m->dz, m->ldc, N, m->C, m->G(m->smb), 1, s));
        GemmArgs d;
        d.layout = GEMM_NT; d.M = N; d.N = H; d.K = m->C; d.A = m->dz; d.lda = m->ldc; d.B = m->P(m->smW); d.ldb = m->ldc;
        d.C = m->dcls; d.ldc = ldh;
        ADN_TRY(mgemm(m, d));
    }
    const int fusion = m->cfg.fusion;
    const bool per_stream_fused = (fusion == ADN_FUSE_CONCAT || fusion == ADN_FUSE_NONE || m->S == 1) &&
                                  fusion != ADN_FUSE_ADASUM;
    std::vector<const float*> fin; std::vector<int> fld;
    if (per_stream_fused) for (auto& st : m->st) { fin.push_back(st.out_ptr); fld.push_back(ldh); }
    else { fin.push_back(m->fused); fld.push_back(ldh); }
    if (m->fused_in.size() == fin.size()) fin = m->fused_in;   // (after the fused-tensor dropout, when it was active)

    // gradient wrt the fused tensor(s)
    std::vector<float*> dfin;          // one per entry of fin
    if (!m->agg.empty()) {
        std::vector<LstmStep> steps;
        for (size_t k = 0; k < m->agg.size(); ++k) steps.push_back(make_step(m, m->agg[k], m->aggw[k], m->dcls, true));
        bool sums_done = false;
        ADN_TRY(run_lstm_group(m, steps, B, T, true, &sums_done));
        const bool cat = cat_path(m) && (int)fin.size() == m->S && per_stream_fused && (!m->planes() || m->cat_planes_ok);
        for (auto& st : m->st) st.dout_ld = 0;
        if (cat) {
            const int ldcat = m->S * ldh;
            // dW_in for all S blocks: cat^T dG into a scratch matrix per LSTM (pad rows dropped by add_row_blocks); the LSTMs'
            // GEMMs share the concat: one grouped launch when the scratch holds them all
            const size_t wcat_elems = (size_t)ldcat * m->ldg;
            const bool group_dw = m->agg.size() >= 2 && m->agg.size() <= (size_t)kMaxGemmGroups && m->agg.size() <= m->wcat_tmp_slots;
            GemmArgs dws[kMaxGemmGroups];
            for (size_t k = 0; k < m->agg.size(); ++k) {
                GemmArgs g = cat_dw_args(m, k, group_dw, N);
                mixed_backward(m, g);
                if (group_dw) { dws[k] = g; continue; }
                ADN_TRY(gemm(g, s));
                ADN_TRY(add_row_blocks(m->wcat_tmp, m->G(m->agg[k].W_in), m->ldg, m->S, H, ldh, 4 * H, s));
            }
            if (group_dw) {
                ADN_TRY(gemm_grouped(dws, (int)m->agg.size(), s));
                for (size_t k0 = 0; k0 < m->agg.size(); k0 += 4) {        // (the LSTMs' scratch matrices into their gradients: one launch)
                    const float* src[4]; float* dst[4];
                    const int n = (int)std::min<size_t>(4, m->agg.size() - k0);
                    for (int k = 0; k < n; ++k) { src[k] = m->wcat_tmp + (k0 + k) * wcat_elems; dst[k] = m->G(m->agg[k0 + k].W_in); }
                    ADN_TRY(add_row_blocks_batch(src, dst, n, m->ldg, m->S, H, ldh, 4 * H, s));
                }
            }
            for (size_t k = 0; k < m->agg.size(); ++k) {
                GemmArgs d = cat_dx_args(m, k, N);
                ADN_CHECK(d.B16, ADN_ERR_STATE, "internal: transposed copy of an aggregation W_in is missing");
                mixed_backward(m, d);
                ADN_TRY(gemm(d, s));
            }
            {                                                 // dW_hid (+ sums) of the aggregation LSTMs: one grouped launch
                std::vector<LstmGradJob> jobs;
                for (size_t k = 0; k < m->agg.size(); ++k) jobs.push_back(LstmGradJob{&m->agg[k], &m->aggw[k], nullptr, 0, 0});
                ADN_TRY(lstm_param_grads_grouped(m, jobs, B, T, sums_done));
            }
            for (int j = 0; j < m->S; ++j) { dfin.push_back(m->dcat + (size_t)j * ldh); m->st[j].dout_ld = ldcat; }
        } else {
            // Per aggregation LSTM: the weight gradients of its input blocks and of W_hid are GEMMs of one shape that share
            // dG, and so are the input gradients of its blocks -- grouped launches (one shared tile list; in bf16x3 mode
            // the shared operand is also split only once).  Where grouping does not apply they go out one by one.
            const bool group = (int)fin.size() + 1 <= kMaxGemmGroups && !streams_concurrent(m);
            for (size_t k = 0; k < m->agg.size(); ++k) {
                const LstmParams& lp = m->agg[k]; const LstmWork& w = m->aggw[k];
                if (!group) {
                    ADN_TRY(lstm_param_grads(m, lp, w, fin.data(), fld.data(), (int)fin.size(), H, B, T, sums_done));
                    continue;
                }
                GemmArgs gs[kMaxGemmGroups];
                int n = 0;
                for (size_t j = 0; j <= fin.size(); ++j) {        // dW_in of every block, then dW_hid (all [H] x [4H] over K = N)
                    GemmArgs& g = gs[n++];
                    g.layout = GEMM_TN; g.M = H; g.N = 4 * H; g.K = N; g.B = w.dG; g.ldb = m->ldg; g.ldc = m->ldg; g.accumulate = 1;
                    if (j < fin.size()) { g.A = fin[j]; g.lda = fld[j]; g.C = m->G(lp.W_in) + j * (size_t)H * m->ldg; }
                    else { g.A = w.prev(B, ldh, lp.backwards); g.lda = ldh; g.C = m->G(lp.W_hid); }
                    mgemm_prepare(m, g, false);
                }
                bool same = true;
                for (int j = 1; j < n; ++j) same = same && gs[j].lda == gs[0].lda && (gs[j].A16 == nullptr) == (gs[0].A16 == nullptr);
                if (same) ADN_TRY(m_gemm_grouped(m, gs, n));
                else for (int j = 0; j < n; ++j) ADN_TRY(m_gemm(m, gs[j]));
                if (!sums_done) {
                    ADN_TRY(restore_fp32(m, w.dG));
                    ADN_TRY(col_sum(w.dG, m->ldg, N, 4 * H, m->G(lp.b), 1, s));
                    ADN_TRY(col_sum(w.dh_carry, ldh, B, H, m->G(lp.hid_init), 1, s));
                    ADN_TRY(col_sum(w.dc_state, ldh, B, H, m->G(lp.cell_init), 1, s));
                }
            }
            std::vector<float*> dsts;
            for (size_t j = 0; j < fin.size(); ++j) dsts.push_back(per_stream_fused ? m->st[j].dout_buf : m->dfused);
            for (size_t k = 0; k < m->agg.size(); ++k) {
                if (!group || fin.size() < 2) {
                    for (size_t j = 0; j < fin.size(); ++j)
                        ADN_TRY(lstm_input_grad(m, m->agg[k], m->aggw[k], (int)j, H, dsts[j], ldh, N, k > 0));
                    continue;
                }
                GemmArgs gs[kMaxGemmGroups];
                int n = 0;
                for (size_t j = 0; j < fin.size(); ++j) {         // dX_j (+)= dG W_in[block j]^T
                    GemmArgs& g = gs[n++];
                    g.layout = GEMM_NT; g.M = N; g.N = H; g.K = 4 * H;
                    g.A = m->aggw[k].dG; g.lda = m->ldg; g.B = m->P(m->agg[k].W_in) + j * (size_t)H * m->ldg; g.ldb = m->ldg;
                    g.C = dsts[j]; g.ldc = ldh; g.accumulate = k > 0;
                    mgemm_prepare(m, g, false);
                }
                bool same = true;
                for (int j = 1; j < n; ++j)
                    same = same && gs[j].layout == gs[0].layout && gs[j].ldb == gs[0].ldb && (gs[j].B16 == nullptr) == (gs[0].B16 == nullptr);
                if (same) ADN_TRY(m_gemm_grouped(m, gs, n));
                else for (int j = 0; j < n; ++j) ADN_TRY(m_gemm(m, gs[j]));
            }
            for (size_t j = 0; j < fin.size(); ++j) dfin.push_back(dsts[j]);
        }
    } else {
        dfin.push_back(m->dcls);
    }
    if (m->stochastic && m->cfg.agg_dropout_p > 0.f && !m->agg.empty()) {
        const int W = (int)dfin.size() * H;
        for (size_t j = 0; j < dfin.size(); ++j) {
            const int ldd = (per_stream_fused && m->st[j].dout_ld) ? m->st[j].dout_ld : ldh;
            ADN_TRY(dropout_apply(dfin[j], ldd, dfin[j], ldd, B, T, H, W, (int)j * H, m->cfg.agg_dropout_p, m->drop_seed,
                                  m->drop_counter, 100u, s));
        }
    }
    // un-fuse: gradient wrt each stream's output
    for (int k = 0; k < m->S; ++k) {
        StreamState& st = m->st[k];
        if (per_stream_fused) {
            st.dout = dfin[k];
        } else if (fusion == ADN_FUSE_SUM) {
            st.dout = dfin[0];                                   // shared, read-only from here on
        } else {                                                 // adasum: d alpha_k = <dfused, out_k>, dout_k = alpha_k dfused
            ADN_TRY(dot_all(dfin[0], ldh, st.out_ptr, ldh, N, H, m->G(m->adacoeff + k), nullptr, s));
            ADN_TRY(scale_by(dfin[0], ldh, m->P(m->adacoeff + k), st.dout_buf, ldh, N, H, s));
            st.dout = st.dout_buf;
        }
    }
    auto bucket_ready = [&](size_t k) -> int {
        // (on the stream the bucket's last gradient was enqueued on: an input stream's side stream below the fork)
        if (k < m->bucket_events.size() && m->bucket_events[k]) ADN_HIP_CHECK(hipEventRecord(m->bucket_events[k], m->stream));
        return ADN_OK;
    };
    // [fuse | agg | softmax] gradients and the cost share are final here, but their bucket is only released BEHIND the
    // stream LSTMs' backward launch: that launch needs (nearly) every CU resident at once, and an all-reduce kernel
    // that starts beside it could leave both half-scheduled on several GPUs at the same time, each waiting for CUs the
    // other holds.  Released after it, the transfer still hides under the encoders' backward GEMMs, and no
    // resident-workgroup launch ever runs next to a collective.
    // stream LSTMs
    bool stream_sums_done = false;
    {
        std::vector<LstmStep> steps;
        for (auto& st : m->st)
            for (size_t k = 0; k < st.lstm.size(); ++k) {
                steps.push_back(make_step(m, st.lstm[k], st.lw[k], st.dout, true));
                steps.back().ld_dhs = st.dout_ld;
            }
        ADN_TRY(run_lstm_group(m, steps, B, T, true, &stream_sums_done));
        // Every weight-stationary LSTM launch of this step (forward, aggregation backward, stream backward) is enqueued by
        // now: this device's exchange status goes into tail[1] HERE, ahead of bucket 0's release -- tail[1] lies inside
        // bucket 0, and a word written after the release would be reduced as 0 (and overwritten under the collective).
        if (m->cfg.precision != ADN_PRECISION_F32) {   // (bf16 and bf16x3 run the weight-stationary LSTM kernels)
            int* word = nullptr;
            ADN_TRY(lstm_cluster_error_word(&word));
            ADN_TRY(poison_tail(word, m->poison_word(), m->stream, m->loss, m->flat[ADN_BUF_GRAD] + m->flat_floats));
        }
        ADN_TRY(bucket_ready(0));
    }
    ADN_TRY(fork_streams(m));                 // below the stream LSTMs the S streams back-propagate independently
    // Per-stream walk state of the encoder's back-propagation.
    struct Walk {
        float* dZ = nullptr; int lddz = 0; int bias_done = 0; int L = 0;
        bool active = false;
    };
    std::vector<Walk> walk(m->st.size());
    ColSumBatch bias_sums;                     // bias reductions queued by the layer in flight, all streams: ONE launch per flush
    const bool stream_major = dp_stream_major(m) != 0;
    // everything of stream si above its encoder: LSTM parameter / input gradients, dropout, delta layer, BatchNorm, act'
    std::vector<char> first_dx_done(m->st.size(), 0);          // layer-major: the first LSTM's input gradient went out grouped
    std::vector<PadFinishJob> pad_rows;        // compact.hip: zero-input rows still to be summed from the delta kernels' partial sums
    auto flush_pad_rows = [&]() -> int {
        if (pad_rows.empty()) return ADN_OK;
        const int rc = compact_pad_finish(pad_rows.data(), (int)pad_rows.size(), m->stream);
        pad_rows.clear();
        return rc;
    };
    auto stream_head = [&](size_t si, bool lstm_grads_done) -> int {
        StreamState& st = m->st[si];
        Walk& w = walk[si];
        const int ldf = ld_of(st.feat_dim);
        const float* in[1] = {st.feat}; const int ld[1] = {ldf};
        for (size_t k = 0; k < st.lstm.size() && !lstm_grads_done; ++k)
            ADN_TRY(lstm_param_grads(m, st.lstm[k], st.lw[k], in, ld, 1, st.feat_dim, B, T, stream_sums_done));
        w.L = st.cfg.n_enc;
        if (st.cfg.n_enc == 0) { ADN_TRY(bucket_ready(bucket_of_top(m, si))); return ADN_OK; }   // nothing trainable below the LSTM
        for (size_t k = first_dx_done[si] ? 1 : 0; k < st.lstm.size(); ++k)
            ADN_TRY(lstm_input_grad(m, st.lstm[k], st.lw[k], 0, st.feat_dim, st.dfeat, ldf, N, k > 0));
        if (m->stochastic && st.cfg.dropout_p > 0.f)
            ADN_TRY(dropout_apply(st.dfeat, ldf, st.dfeat, ldf, B, T, st.feat_dim, st.feat_dim, 0, st.cfg.dropout_p,
                                  m->drop_seed, m->drop_counter, (uint32_t)si, m->stream));
        const int ldE = ld_of(st.enc_out);
        // encoder: dZ_l = dA_l * act_l'(A_l);  dW_l = A_{l-1}^T dZ_l;  dA_{l-1} = dZ_l W_l^T
        const int L = st.cfg.n_enc;
        const bool last_linear = L > 0 && st.cfg.enc_act[L - 1] == ADN_ACT_LINEAR;
        void* dE16 = (m->bf16() && (L == 0 || last_linear)) ? m->shadow_of(st.dE) : nullptr;   // bf16 copy straight from the kernel
        // (an auxiliary input sits in the columns behind the delta features: data, no gradient)
        // (queued: where nothing of this stream reads dE before the encoder's back-propagation starts -- no BatchNorm, a linear
        //  bottleneck, the bf16 copy written by the kernel -- the streams' delta layers go out as ONE launch behind this loop)
        if (m->compact) {
            // compact.hip: the delta layer's backward kernel stores a valid frame's gradient at its compact row -- with its 16-bit
            // copies -- and sums the padding frames' per utterance; the zero-input row is finished from those sums ahead of the
            // first reader (flush_pad_rows)
            DeltaJob dj{st.dfeat, ldf, st.dEc, ldE, st.enc_out, st.cfg.use_delta, nullptr};
            dj.row_map = m->comp_of_full; dj.zrow = m->Nc - 1; dj.pad_partial = st.compact_ws; tm_job(dj);
            if (m->bf16() || m->planes()) dj.dst16 = m->shadow_of(st.dEc);
            if (m->planes() && dj.dst16) dj.dst16lo = m->shadow_lo_of(st.dEc);
            // (queued like the padded path's: flushed behind the streams' loop in the layer-major order; a stream that back-propagates
            //  on its own right away, or reads the fp32 result here, takes it now)
            ADN_TRY(queue_delta(m, false, dj, B0, T0, theta, stream_major || !dj.dst16));
            pad_rows.push_back(PadFinishJob{st.compact_ws, B0, st.dEc, ldE, st.enc_out, m->Nc - 1, dj.dst16, dj.dst16lo});
            if (!dj.dst16) { ADN_TRY(flush_pad_rows()); ADN_TRY(refresh(m, st.dEc, (size_t)m->Nc * ldE)); }
            w.dZ = st.dEc; w.lddz = ldE; w.bias_done = 0; w.active = true;
            return bucket_ready(bucket_of_top(m, si));
        }
        const bool dE_read_here = st.cfg.batchnorm || !last_linear || !dE16 || stream_major;
        ADN_TRY(queue_delta(m, false, DeltaJob{st.dfeat, ldf, st.dE, ldE, st.enc_out, st.cfg.use_delta, st.cfg.batchnorm ? nullptr : dE16},
                            B, T, theta, dE_read_here));
        if (st.cfg.batchnorm)                     // through the BatchNormLayer: d(bn_out) -> d(encoder output), dgamma, dbeta
            ADN_TRY(batchnorm_backward(st.act[L - 1], ldE, st.dE, ldE, st.dE, ldE, N, st.enc_out, m->P(st.bn_gamma),
                                       m->training ? st.bn_save_mean : m->P(st.bn_mean),
                                       m->training ? st.bn_save_inv_std : m->P(st.bn_inv_std), m->training ? 1 : 0,
                                       m->G(st.bn_gamma), m->G(st.bn_beta), st.bn_ws, m->stream, dE16));
        if (!last_linear)
            ADN_TRY(act_backward(st.dE, ldE, st.act[L - 1], ldE, N, st.enc_out, m->act_code(st.cfg.enc_act[L - 1]), m->stream));
        if (!dE16) ADN_TRY(refresh(m, st.dE, (size_t)N * ldE));
        w.dZ = st.dE; w.lddz = ldE; w.bias_done = 0; w.active = true;
        // this stream's "top" bucket (BatchNorm, LSTMs) is final: its last writers are the launches above
        return bucket_ready(bucket_of_top(m, si));
    };
    // encoder layer L - 1 - depth of the streams `sis` (same geometry when more than one): weight gradients, bias
    // gradients, input gradients -- each kind as ONE grouped launch where the ping-pong kernel takes it
    auto layer_step = [&](const std::vector<size_t>& sis, int depth) -> int {
        const int n = (int)sis.size();
        const int Ne = m->compact ? m->Nc : N;       // rows of the encoder's matrices (compact.hip)
        GemmArgs gws[kMaxGemmGroups], gxs[kMaxGemmGroups];
        // db_l as row in_w of dW_l where the layer's input carries its column of ones (ensure_workspace)
        auto bias_rides = [&](const StreamState& st, int l) { return has_ones_col(m, l > 0 ? st.act[l - 1] : st.x); };
        for (int q = 0; q < n; ++q) {
            StreamState& st = m->st[sis[q]]; Walk& w = walk[sis[q]];
            const int l = w.L - 1 - depth;
            const int out_w = st.cfg.enc_units[l], in_w = st.enc_in[l];
            GemmArgs& gw = gws[q];
            gw.layout = GEMM_TN; gw.M = in_w + (bias_rides(st, l) ? 1 : 0); gw.N = out_w; gw.K = Ne; gw.A = l > 0 ? st.act[l - 1] : st.x;
            gw.lda = l > 0 ? ld_of(in_w) : st.ldx;
            gw.B = w.dZ; gw.ldb = w.lddz; gw.C = m->G(st.encW[l]); gw.ldc = ld_of(out_w); gw.accumulate = 1;
            mgemm_prepare(m, gw, false);
        }
        ADN_TRY(m_gemm_grouped(m, gws, n));
        bool any_dx = false;
        for (int q = 0; q < n; ++q) {
            StreamState& st = m->st[sis[q]]; Walk& w = walk[sis[q]];
            const int l = w.L - 1 - depth;
            const int out_w = st.cfg.enc_units[l];
            if (!w.bias_done && !bias_rides(st, l)) {           // b_l rode neither on dW_l nor on the input-gradient GEMM of the layer above: summed from dZ now
                if ((m->bf16() || m->planes()) && (w.dZ == st.dE || w.dZ == st.dEc) && bias_sums.n < 8) col_sum_batch_add(bias_sums, w.dZ, w.lddz, Ne, out_w, m->G(st.encb[l]));   // (dE / dEc hold their fp32 values until the step ends: the delta layer wrote them, nothing reuses them)
                else ADN_TRY(col_sum(w.dZ, w.lddz, Ne, out_w, m->G(st.encb[l]), 1, m->stream));
            }
            w.bias_done = 0;
        }
        // [W_l | b_l] of these streams is final behind the weight-gradient launch above and the queued bias reductions
        // (partial column sums of the layer above's input-gradient GEMM, or the sums just queued): one launch, then the
        // buckets are released -- AHEAD of this layer's input-gradient GEMM, which then covers their transfer
        // (without bucket events the queue is only drained when it runs full, and once at the end)
        if (!m->bucket_events.empty() || bias_sums.n + n > 8) ADN_TRY(col_sum_batch(bias_sums, m->stream));
        for (int q = 0; q < n; ++q) ADN_TRY(bucket_ready(bucket_of_layer(m, sis[q], walk[sis[q]].L - 1 - depth)));
        for (int q = 0; q < n; ++q) {
            StreamState& st = m->st[sis[q]]; Walk& w = walk[sis[q]];
            const int l = w.L - 1 - depth;
            const int out_w = st.cfg.enc_units[l], in_w = st.enc_in[l];
            if (l == 0) continue;
            any_dx = true;
            float* dst = (w.dZ == st.pingA) ? st.pingB : st.pingA;
            GemmArgs& gx = gxs[q];
            gx.layout = GEMM_NT; gx.M = Ne; gx.N = in_w; gx.K = out_w; gx.A = w.dZ; gx.lda = w.lddz;
            gx.B = m->P(st.encW[l]); gx.ldb = ld_of(out_w); gx.C = dst; gx.ldc = st.ping_ld;
            gx.Y = st.act[l - 1]; gx.ldy = ld_of(in_w); gx.act_grad = m->act_code(st.cfg.enc_act[l - 1]);
            if (st.relu_bits[l - 1] && st.bits_tiles[l - 1] && gx.act_grad == ADN_ACT_RECTIFY) { gx.Ybits = st.relu_bits[l - 1]; gx.Ybits_tiles = st.bits_tiles[l - 1]; }
            if (!bias_rides(st, l - 1)) {
                gx.colsum = m->G(st.encb[l - 1]); gx.colsum_done = &w.bias_done;     // db_{l-1} rides on this GEMM
                gx.colsum_ws = st.colsum_ws + (size_t)l * st.colsum_ws_floats; gx.colsum_ws_floats = st.colsum_ws_floats;
                gx.colsum_batch = &bias_sums;
            }
            mgemm_prepare(m, gx, /*lean=*/true);
        }
        if (!any_dx) return ADN_OK;                 // (streams of one group share l == 0)
        ADN_TRY(m_gemm_grouped(m, gxs, n));
        for (int q = 0; q < n; ++q) {
            StreamState& st = m->st[sis[q]]; Walk& w = walk[sis[q]];
            const int l = w.L - 1 - depth, in_w = st.enc_in[l];
            float* dst = (w.dZ == st.pingA) ? st.pingB : st.pingA;
            if (!w.bias_done && !bias_rides(st, l - 1) && shadows_on(m) && !m->keep_fp32 && in_w % 4 == 0 && m->shadow_of(dst)) {
                // fp32 dZ was skipped but the fused column sum did not run: cannot happen for in_w % 4 == 0
                set_error("internal: lean dZ without fused bias gradient"); return ADN_ERR_STATE;
            }
            w.dZ = dst; w.lddz = st.ping_ld;
        }
        return ADN_OK;
    };
    // Layer-major with grouped launches (the streams' layers of equal geometry share tile lists, like the forward pass);
    // stream-major when the streams run on forked HIP streams.  Data parallel runs (gradient buckets released to an
    // all-reduce as they complete) are layer-major too: every stream's [layers >= 1 + LSTM] bucket becomes final just ahead
    // of the grouped layer-0 weight-gradient launch and travels under it, the [layer 0] buckets go last -- more of the
    // transfer is exposed than in the stream-major order (where a whole stream's buckets hide under the next stream's
    // GEMMs), but that order costs 0.5 ms of GEMM time per step (3.9 -> 4.4 ms at B = 520), more than the ~0.2 ms of
    // transfer it hides.  ADN_DP_STREAM_MAJOR=1 selects it for data parallel runs, ADN_NO_GROUPED_BACKWARD=1 always.
    const bool layer_major = !stream_major;
    if (layer_major) {
        int max_depth = 0;
        {                                      // parameter gradients of all stream LSTMs: grouped launches
            std::vector<LstmGradJob> jobs;
            for (auto& st : m->st)
                for (size_t k = 0; k < st.lstm.size(); ++k)
                    jobs.push_back(LstmGradJob{&st.lstm[k], &st.lw[k], st.feat, ld_of(st.feat_dim), st.feat_dim});
            ADN_TRY(lstm_param_grads_grouped(m, jobs, B, T, stream_sums_done));
        }
        {                                      // dfeat = dG W_in^T of every stream's first LSTM: grouped launches
            std::vector<GemmArgs> dx;
            for (size_t si = 0; si < m->st.size(); ++si) {
                StreamState& st = m->st[si];
                if (st.cfg.n_enc == 0 || st.lstm.empty()) continue;
                GemmArgs g;
                g.layout = GEMM_NT; g.M = N; g.N = st.feat_dim; g.K = 4 * m->H;
                g.A = st.lw[0].dG; g.lda = m->ldg; g.B = m->P(st.lstm[0].W_in); g.ldb = m->ldg;
                g.C = st.dfeat; g.ldc = ld_of(st.feat_dim);
                mgemm_prepare(m, g, false);
                dx.push_back(g);
                first_dx_done[si] = 1;
            }
            ADN_TRY(issue_grouped(m, dx));
        }
        for (size_t si = 0; si < m->st.size(); ++si) { ADN_TRY(stream_head(si, true)); max_depth = std::max(max_depth, walk[si].active ? walk[si].L : 0); }
        ADN_TRY(flush_deltas(m, B0, T0));
        ADN_TRY(flush_pad_rows());
        for (int d = 0; d < max_depth; ++d)
            for (const auto& sis : depth_groups(m, d)) ADN_TRY(layer_step(sis, d));
        ADN_TRY(col_sum_batch(bias_sums, m->stream));
    } else {
        for (size_t si = 0; si < m->st.size(); ++si) {
            OnSideStream on(m, (int)si);
            ADN_TRY(stream_head(si, false));
            ADN_TRY(flush_pad_rows());
            if (!walk[si].active) continue;
            for (int d = 0; d < walk[si].L; ++d) ADN_TRY(layer_step(std::vector<size_t>{si}, d));
            ADN_TRY(col_sum_batch(bias_sums, m->stream));
        }
    }
    ADN_TRY(join_streams(m));
    m->grads_valid = true;
    return ADN_OK;
}

__global__ void set_scalar_kernel(float* p, float v) { *p = v; }

// the loss normaliser: valid frames of the batch (already there after mask_prepare), B for the last-timestep head, or
// the caller's global count (data parallel)
int set_loss_normaliser(adn_model* m, int B, double override_total) {
    const double v = override_total > 0 ? override_total : (m->head_last() ? (double)B : 0.0);
    if (v > 0) {
        hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, m->stream, m->total, (float)v);
        ADN_HIP_CHECK(hipGetLastError());
    }
    return ADN_OK;
}

int check_shape(const adn_model* m, int B, int T, int theta) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    ADN_CHECK(B >= 1 && T >= 1, ADN_ERR_INVALID, "empty batch (B and T must be >= 1)");
    ADN_CHECK((int64_t)B * T < (1 << 30), ADN_ERR_INVALID, "batch too large");
    ADN_CHECK(theta >= 0 && theta < 4096, ADN_ERR_INVALID, "delta window out of range");
    return ADN_OK;
}

// the weight-stationary LSTM kernels raise a device word when a workgroup gave up waiting for its partners
// (lstm_cluster.hip); call after the stream has been synchronised
int check_device_errors(adn_model* m) {
    if (m->cfg.precision == ADN_PRECISION_F32) return ADN_OK;   // (only the weight-stationary LSTM kernels / a compacted call can raise a word)
    int words[2] = {0, 0};
    ADN_HIP_CHECK(hipMemcpy(words, m->poison_sticky, 2 * sizeof(int), hipMemcpyDeviceToHost));
    const int sticky = words[0];
    if (words[0] || words[1]) ADN_HIP_CHECK(hipMemset(m->poison_sticky, 0, 2 * sizeof(int)));
    if (words[1] & kInputLens) {
        set_error("adn_set_batch_lengths: the mask of a call is not the prefix mask of the lengths announced for it; the encoders ran over "
                  "the announced frames -- results since then are invalid");
        return ADN_ERR_INVALID;
    }
    int* word = nullptr;
    ADN_TRY(lstm_cluster_error_word(&word));
    int v = 0;
    ADN_HIP_CHECK(hipMemcpy(&v, word, sizeof(int), hipMemcpyDeviceToHost));
    if (v) {
        ADN_HIP_CHECK(hipMemset(word, 0, sizeof(int)));
        char msg[200];
        snprintf(msg, sizeof(msg), "LSTM exchange timed out (%s kernel, step tag %d, workgroup %d): a workgroup never "
                 "received its partners' state; results are invalid", (v & 15) == 1 ? "forward" : "backward", (v >> 4) & 4095,
                 (v >> 16) & 1023);
        set_error(msg);
        return ADN_ERR_STATE;
    }
    if (sticky) {
        set_error("an optimiser step was skipped: the LSTM exchange of a data-parallel peer timed out and its gradients were "
                  "invalid (the flag travels in the gradient buffer's tail, so every rank skips the same step)");
        return ADN_ERR_STATE;
    }
    return ADN_OK;
}

int fetch(adn_model* m, void* dst, const void* src, size_t bytes, bool to_device) {
    if (to_device) {
        ADN_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, m->stream));
    } else {
        ADN_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, m->stream));
        ADN_HIP_CHECK(hipStreamSynchronize(m->stream));
        ADN_TRY(check_device_errors(m));
    }
    return ADN_OK;
}

int tensor_io(adn_model* m, int buffer, int index, float* host, bool write) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    ADN_CHECK(buffer >= 0 && buffer < 4, ADN_ERR_INVALID, "bad buffer id");
    ADN_CHECK(index >= 0 && index < (int)m->params.size(), ADN_ERR_INVALID, "parameter index out of range");
    ADN_CHECK(host, ADN_ERR_INVALID, "null host pointer");
    const ParamDesc& p = m->params[index];
    const int rows = p.rows(), cols = p.cols();
    float* base = m->flat[buffer] + p.off;
    ADN_HIP_CHECK(hipStreamSynchronize(m->stream));
    ADN_TRY(check_device_errors(m));
    if (p.col_stride == 1) {
        if (write) {
            ADN_HIP_CHECK(hipMemcpy2D(base, (size_t)p.ld * 4, host, (size_t)cols * 4, (size_t)cols * 4, rows,
                                      hipMemcpyHostToDevice));
            if (buffer == ADN_BUF_PARAM) m->mark_params_dirty();
        }
        else ADN_HIP_CHECK(hipMemcpy2D(host, (size_t)cols * 4, base, (size_t)p.ld * 4, (size_t)cols * 4, rows,
                                       hipMemcpyDeviceToHost));
        return ADN_OK;
    }
    // strided (per-gate) view: stage the enclosing rows through the host
    const size_t span = (size_t)(rows - 1) * p.ld + (size_t)(cols - 1) * p.col_stride + 1;
    std::vector<float> tmp(span);
    ADN_HIP_CHECK(hipMemcpy(tmp.data(), base, span * 4, hipMemcpyDeviceToHost));
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            float& phys = tmp[(size_t)r * p.ld + (size_t)c * p.col_stride];
            if (write) phys = host[(size_t)r * cols + c]; else host[(size_t)r * cols + c] = phys;
        }
    if (write) ADN_HIP_CHECK(hipMemcpy(base, tmp.data(), span * 4, hipMemcpyHostToDevice));
    if (write && buffer == ADN_BUF_PARAM) m->mark_params_dirty();
    return ADN_OK;
}

}  // namespace
}  // namespace adn

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

const char* adn_version(void) { return "adenet-hip 0.1.0 (gfx950)"; }
const char* adn_last_error(void) { return g_last_error.c_str(); }
void adn_abi_sizes(int32_t out[4]) {
    out[0] = (int32_t)sizeof(adn_stream_config); out[1] = (int32_t)sizeof(adn_config);
    out[2] = (int32_t)sizeof(adn_param_info_t); out[3] = (int32_t)sizeof(adn_profile_entry);
}

int adn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

int adn_create(const adn_config* cfg, adn_model** out) {
    ADN_CHECK(cfg && out, ADN_ERR_INVALID, "null argument");
    *out = nullptr;
    ADN_TRY(validate(*cfg));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device visible: libadenet_hip needs an MI355X (gfx950); there is no CPU fallback");
        return ADN_ERR_NO_DEVICE;
    }
    int dev = 0;
    ADN_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    ADN_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error(std::string("current device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
        return ADN_ERR_NO_DEVICE;
    }
    adn_model* m = new adn_model();
    m->cfg = *cfg;
    m->bwd_hi_only = cfg->precision == ADN_PRECISION_MIXED;
    if (m->bwd_hi_only) m->cfg.precision = ADN_PRECISION_BF16X3;
    m->S = cfg->n_streams; m->H = cfg->lstm_size; m->C = cfg->classes;
    m->ldh = ld_of(m->H); m->ldg = ld_of(4 * m->H); m->ldc = ld_of(m->C);
    int st = build_params(m);
    if (st != ADN_OK) { delete m; return st; }
    for (int k = 0; k < 4; ++k) {     // + kAuxFloats: tail slot carried through the gradient all-reduce
        if (hipMalloc((void**)&m->flat[k], (m->flat_floats + kAuxFloats) * sizeof(float)) != hipSuccess ||
            hipMemset(m->flat[k], 0, (m->flat_floats + kAuxFloats) * sizeof(float)) != hipSuccess) {
            set_error("hipMalloc of the parameter buffers failed");
            adn_destroy(m);
            return ADN_ERR_HIP;
        }
    }
    if (hipMalloc((void**)&m->poison_sticky, 2 * sizeof(int)) != hipSuccess || hipMemset(m->poison_sticky, 0, 2 * sizeof(int)) != hipSuccess) {
        set_error("hipMalloc of the model's status word failed");
        adn_destroy(m);
        return ADN_ERR_HIP;
    }
    *out = m;
    return ADN_OK;
}

void adn_destroy(adn_model* m) {
    if (!m) return;
    (void)hipStreamSynchronize(m->stream);
    if (g_prof == &m->prof) g_prof = nullptr;
    for (int k = 0; k < 4; ++k) if (m->flat[k]) (void)hipFree(m->flat[k]);
    if (m->params16) (void)hipFree(m->params16);
    if (m->transw_slab) (void)hipFree(m->transw_slab);
    if (m->transw_items) (void)hipFree(m->transw_items);
    if (m->transw_items_lo) (void)hipFree(m->transw_items_lo);
    if (m->transw_items_lo_fwd) (void)hipFree(m->transw_items_lo_fwd);
    if (m->params16lo) (void)hipFree(m->params16lo);
    if (m->side_ready) {
        for (int k = 0; k < m->S; ++k) {
            if (m->side[k]) { (void)hipStreamSynchronize(m->side[k]); (void)hipStreamDestroy(m->side[k]); }
            if (m->join_ev[k]) (void)hipEventDestroy(m->join_ev[k]);
        }
        if (m->fork_ev) (void)hipEventDestroy(m->fork_ev);
    }
    auto free_lp = [](LstmParams& lp) {
        if (lp.whid16t) (void)hipFree(lp.whid16t);
        if (lp.wcat16) (void)hipFree(lp.wcat16);
        if (lp.wcat16lo) (void)hipFree(lp.wcat16lo);
        if (lp.wfrag_fwd) (void)hipFree(lp.wfrag_fwd);
        if (lp.win_frag) (void)hipFree(lp.win_frag);
        if (lp.wfrag_bwd) (void)hipFree(lp.wfrag_bwd);
        if (lp.wfrag_fwd_lo) (void)hipFree(lp.wfrag_fwd_lo);
        if (lp.wfrag_bwd_lo) (void)hipFree(lp.wfrag_bwd_lo);
    };
    for (auto& st : m->st) for (auto& lp : st.lstm) free_lp(lp);
    for (auto& lp : m->agg) free_lp(lp);
    if (m->slab) (void)hipFree(m->slab);
    if (m->splitk_ws) (void)hipFree(m->splitk_ws);
    if (m->poison_sticky) (void)hipFree(m->poison_sticky);
    for (auto& slot : m->pin) { if (slot.host) (void)hipHostFree(slot.host); if (slot.ev) (void)hipEventDestroy(slot.ev); }
    for (auto& slot : m->pin_tm) { if (slot.host) (void)hipHostFree(slot.host); if (slot.ev) (void)hipEventDestroy(slot.ev); }
    delete m;
}

int adn_set_stream(adn_model* m, void* hip_stream) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    m->stream = static_cast<hipStream_t>(hip_stream);
    return ADN_OK;
}

int adn_set_precision(adn_model* m, int precision) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    ADN_CHECK(precision >= ADN_PRECISION_F32 && precision <= ADN_PRECISION_MIXED, ADN_ERR_INVALID,
              "unsupported precision");
    m->bwd_hi_only = precision == ADN_PRECISION_MIXED;
    m->cfg.precision = m->bwd_hi_only ? (int)ADN_PRECISION_BF16X3 : precision;
    m->mark_params_dirty();
    m->wsB = 0;                       // input staging differs between the modes: re-carve on the next call
    return ADN_OK;
}

int adn_set_batch_lengths(adn_model* m, const int32_t* lengths, int B) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    ADN_CHECK(B >= 0 && (lengths || B == 0), ADN_ERR_INVALID, "adn_set_batch_lengths: null lengths");
    if (B == 0) m->batch_lens.clear(); else m->batch_lens.assign(lengths, lengths + B);
    return ADN_OK;
}

int adn_get_compact_rows(const adn_model* m) { return (m && m->compact) ? m->Nc : 0; }

int adn_set_relu_grad_at_zero(adn_model* m, float value) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    ADN_CHECK(value == 0.f || value == 0.5f, ADN_ERR_INVALID, "adn_set_relu_grad_at_zero: 0 (this build's default) or 0.5 (Theano's 0.5 (x + |x|))");
    m->relu0_half = value == 0.5f;
    return ADN_OK;
}

int adn_get_bucket_rows(const adn_model* m) { return (m && m->tm.on) ? m->tm.Bb * m->tm.Tt : 0; }

int adn_set_length_buckets(adn_model* m, int on) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    m->buckets_allowed = on != 0;
    return ADN_OK;
}

int adn_set_auto_compaction(adn_model* m, int on) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    m->auto_compact = on != 0;
    return ADN_OK;
}

int adn_num_params(const adn_model* m) { return m ? (int)m->params.size() : 0; }

int adn_param_info(const adn_model* m, int index, adn_param_info_t* info) {
    ADN_CHECK(m && info, ADN_ERR_INVALID, "null argument");
    ADN_CHECK(index >= 0 && index < (int)m->params.size(), ADN_ERR_INVALID, "parameter index out of range");
    const ParamDesc& p = m->params[index];
    memset(info, 0, sizeof(*info));
    strncpy(info->name, p.name.c_str(), sizeof(info->name) - 1);
    info->ndim = p.ndim; info->dims[0] = p.dims[0]; info->dims[1] = p.dims[1]; info->numel = p.numel();
    return ADN_OK;
}

int64_t adn_total_param_count(const adn_model* m) {
    int64_t n = 0;
    if (m) for (auto& p : m->params) n += p.numel();
    return n;
}

int adn_read_tensor(adn_model* m, int buffer, int index, float* host_dst) { return tensor_io(m, buffer, index, host_dst, false); }
int adn_write_tensor(adn_model* m, int buffer, int index, const float* host_src) {
    return tensor_io(m, buffer, index, const_cast<float*>(host_src), true);
}

int adn_flat_buffer(adn_model* m, int buffer, void** device_ptr, size_t* bytes) {
    ADN_CHECK(m && device_ptr && bytes, ADN_ERR_INVALID, "null argument");
    ADN_CHECK(buffer >= 0 && buffer < 4, ADN_ERR_INVALID, "bad buffer id");
    if (buffer == ADN_BUF_PARAM) m->mark_params_dirty();      // the caller may write through the pointer
    *device_ptr = m->flat[buffer];
    *bytes = (m->flat_floats + kAuxFloats) * sizeof(float);
    return ADN_OK;
}

int adn_flat_buffer_const(const adn_model* m, int buffer, const void** device_ptr, size_t* bytes) {
    ADN_CHECK(m && device_ptr && bytes, ADN_ERR_INVALID, "null argument");
    ADN_CHECK(buffer >= 0 && buffer < 4, ADN_ERR_INVALID, "bad buffer id");
    *device_ptr = m->flat[buffer];
    *bytes = (m->flat_floats + kAuxFloats) * sizeof(float);
    return ADN_OK;
}

// Bucket list in the order the ranges become final (see "gradient buckets" above backward_pass)
static void bucket_ranges(adn_model* m, std::vector<std::pair<size_t, size_t>>& out) {
    (void)dp_stream_major(m);                      // latch the order
    out.assign(bucket_count(m), std::make_pair((size_t)0, (size_t)0));
    out[0] = std::make_pair(m->tail_begin, m->flat_floats + kAuxFloats);
    for (int s = 0; s < m->S; ++s) {
        const StreamState& st = m->st[s];
        const size_t begin = st.param_begin, end = s + 1 < m->S ? m->st[s + 1].param_begin : m->tail_begin;
        const int L = st.cfg.n_enc;
        // behind the encoder: [end of the last layer's bias, end) -- the first tensor behind the encoder starts there
        size_t top_begin = begin;
        if (L > 0) top_begin = st.cfg.batchnorm ? st.bn_beta : st.lstm[0].W_in;
        out[bucket_of_top(m, (size_t)s)] = std::make_pair(top_begin, end);
        for (int l = 0; l < L; ++l)
            out[bucket_of_layer(m, (size_t)s, l)] = std::make_pair(st.encW[l], l + 1 < L ? st.encW[l + 1] : top_begin);
    }
}

int adn_grad_buckets(const adn_model* cm, int max_buckets, int64_t* begin_floats, int64_t* end_floats, int* n_out) {
    adn_model* m = const_cast<adn_model*>(cm);     // (latches the bucket order)
    ADN_CHECK(m && begin_floats && end_floats && n_out, ADN_ERR_INVALID, "null argument");
    std::vector<std::pair<size_t, size_t>> r;
    bucket_ranges(m, r);
    ADN_CHECK(max_buckets >= (int)r.size(), ADN_ERR_INVALID, "bucket arrays too small");
    for (size_t k = 0; k < r.size(); ++k) { begin_floats[k] = (int64_t)r[k].first; end_floats[k] = (int64_t)r[k].second; }
    *n_out = (int)r.size();
    return ADN_OK;
}

int adn_grad_bucket_groups(const adn_model* cm, int max_buckets, int* group_of_bucket, int* n_out) {
    adn_model* m = const_cast<adn_model*>(cm);
    ADN_CHECK(m && group_of_bucket && n_out, ADN_ERR_INVALID, "null argument");
    (void)dp_stream_major(m);
    const size_t n = bucket_count(m);
    ADN_CHECK((int)n <= max_buckets, ADN_ERR_INVALID, "bucket table too small");
    for (size_t k = 0; k < n; ++k) group_of_bucket[k] = (int)k;              // stream-major: every bucket on its own
    if (!m->dp_order) {
        int g = 0;
        group_of_bucket[0] = g++;
        for (int s = 0; s < m->S; ++s) group_of_bucket[bucket_of_top(m, (size_t)s)] = g;   // the tops: back to back
        ++g;
        int max_depth = 0;
        for (int s = 0; s < m->S; ++s) max_depth = std::max(max_depth, m->st[s].cfg.n_enc);
        for (int d = 0; d < max_depth; ++d)
            for (const auto& grp : depth_groups(m, d)) {
                for (size_t si : grp) group_of_bucket[bucket_of_layer(m, si, m->st[si].cfg.n_enc - 1 - d)] = g;
                ++g;
            }
    }
    *n_out = (int)n;
    return ADN_OK;
}

int adn_set_bucket_events(adn_model* m, void* const* hip_events, int n) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    std::vector<std::pair<size_t, size_t>> r;
    bucket_ranges(m, r);
    ADN_CHECK(n == 0 || (hip_events && n == (int)r.size()), ADN_ERR_INVALID, "expected one event per gradient bucket");
    m->bucket_events.clear();
    for (int k = 0; k < n; ++k) m->bucket_events.push_back(static_cast<hipEvent_t>(hip_events[k]));
    return ADN_OK;
}

int adn_forward(adn_model* m, const void* const* inputs, const uint8_t* mask, int B, int T, int theta, int flags,
                float* probs) {
    ADN_TRY(check_shape(m, B, T, theta));
    ADN_CHECK(probs, ADN_ERR_INVALID, "null output");
    ADN_TRY(ensure_workspace(m, B, T));
    ADN_TRY(stage_inputs(m, inputs, nullptr, mask, B, T, flags));
    m->stochastic = false; m->training = false;
    ADN_TRY(forward_pass(m, B, T, theta, false, false));
    const size_t rows = m->head_last() ? (size_t)B : (size_t)B * T;
    return fetch(m, probs, m->probs_bt, rows * m->C * sizeof(float), flags & ADN_FLAG_DEVICE_OUTPUTS);
}

int adn_loss(adn_model* m, const void* const* inputs, const int32_t* targets, const uint8_t* mask, int B, int T,
             int theta, int flags, float* loss) {
    ADN_TRY(check_shape(m, B, T, theta));
    ADN_CHECK(targets && loss, ADN_ERR_INVALID, "null targets / loss");
    ADN_TRY(ensure_workspace(m, B, T));
    ADN_TRY(stage_inputs(m, inputs, targets, mask, B, T, flags));
    ADN_TRY(set_loss_normaliser(m, B, 0.0));
    m->training = (flags & ADN_FLAG_STOCHASTIC) != 0;          // compute_train_cost: get_output(deterministic=False)
    m->stochastic = m->training && m->has_dropout();
    ADN_TRY(forward_pass(m, B, T, theta, true, false));
    if (m->stochastic) m->drop_counter += 1;
    return fetch(m, loss, m->loss, sizeof(float), flags & ADN_FLAG_DEVICE_OUTPUTS);
}

int adn_read_probs(adn_model* m, int B, int T, int flags, float* probs) {
    ADN_CHECK(m && probs, ADN_ERR_INVALID, "null argument");
    ADN_CHECK(m->lastB == B && m->lastT == T && B >= 1, ADN_ERR_STATE, "adn_read_probs: no forward pass of this (B, T) to read from");
    ADN_CHECK(!m->probs_partial, ADN_ERR_STATE, "adn_read_probs: the last pass was a train step over length buckets -- its probabilities cover each "
              "utterance's bucket only; run adn_forward / adn_loss for the batch's (adn_set_length_buckets(0) keeps train steps unbucketed)");
    const size_t rows = m->head_last() ? (size_t)B : (size_t)B * T;
    return fetch(m, probs, m->probs_bt, rows * m->C * sizeof(float), flags & ADN_FLAG_DEVICE_OUTPUTS);
}

int adn_compute_grads(adn_model* m, const void* const* inputs, const int32_t* targets, const uint8_t* mask, int B,
                      int T, int theta, int flags, double total_frames, float* loss) {
    ADN_TRY(check_shape(m, B, T, theta));
    ADN_CHECK(targets, ADN_ERR_INVALID, "null targets");
    ADN_TRY(ensure_workspace(m, B, T));
    {   // a train step may keep its time-major side in length buckets (TmPlan): nobody reads per-frame outputs of this call
        struct Ask { adn_model* m; explicit Ask(adn_model* m_) : m(m_) { m->want_buckets = true; } ~Ask() { m->want_buckets = false; } } ask(m);
        ADN_TRY(stage_inputs(m, inputs, targets, mask, B, T, flags));
    }
    ADN_TRY(set_loss_normaliser(m, B, total_frames));
    m->training = !(flags & ADN_FLAG_DETERMINISTIC);
    m->stochastic = m->training && m->has_dropout();
    ADN_TRY(forward_pass(m, B, T, theta, true, true));
    ADN_TRY(backward_pass(m, B, T, theta));
    if (m->stochastic) m->drop_counter += 1;
    if (loss) return fetch(m, loss, m->loss, sizeof(float), flags & ADN_FLAG_DEVICE_OUTPUTS);
    return ADN_OK;
}

int adn_zero_grads(adn_model* m) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    ADN_HIP_CHECK(hipMemsetAsync(m->flat[ADN_BUF_GRAD], 0, (m->flat_floats + kAuxFloats) * sizeof(float), m->stream));
    for (auto ev : m->bucket_events)
        if (ev) ADN_HIP_CHECK(hipEventRecord(ev, m->stream));
    m->grads_valid = true;
    return ADN_OK;
}

int adn_apply_adam(adn_model* m, float learning_rate) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    ADN_CHECK(m->grads_valid, ADN_ERR_STATE, "adn_apply_adam called without gradients (call adn_compute_grads first)");
    m->adam_t += 1;
    const float t = (float)m->adam_t;
    const float a_t = learning_rate * sqrtf(1.f - powf(kBeta2, t)) / (1.f - powf(kBeta1, t));
    // the update writes the bf16 shadow as well (bf16x3 over planes: both planes, once they exist)
    const bool pl = m->planes() && m->params16 && m->params16lo;
    void* p16 = (shadows_on(m) || pl) ? m->params16 : nullptr;
    ADN_TRY(adam_update(m->flat[ADN_BUF_PARAM], m->flat[ADN_BUF_GRAD], m->flat[ADN_BUF_ADAM_M], m->flat[ADN_BUF_ADAM_V],
                        (int64_t)m->flat_floats, a_t, kBeta1, kBeta2, kEps, m->stream, p16, m->poison_word(), m->poison_sticky,
                        pl ? m->params16lo : nullptr));
    m->grads_valid = false;
    m->mark_params_dirty();
    m->params16_values_fresh = p16 != nullptr;
    return ADN_OK;
}

// One Adam step applied range by range (data parallel: each gradient bucket is updated as soon as its reduction has
// landed, while later buckets are still on the wire).  begin .. range* .. end == adn_apply_adam bit for bit when the ranges
// cover [0, parameter floats) once: the update is element-wise and every range starts on an 8-float boundary.
int adn_adam_begin(adn_model* m, float learning_rate) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    ADN_CHECK(m->grads_valid, ADN_ERR_STATE, "adn_adam_begin called without gradients (call adn_compute_grads first)");
    ADN_CHECK(!m->adam_open, ADN_ERR_STATE, "adn_adam_begin: the previous ranged step was not closed with adn_adam_end");
    m->adam_t += 1;
    const float t = (float)m->adam_t;
    m->adam_a_t = learning_rate * sqrtf(1.f - powf(kBeta2, t)) / (1.f - powf(kBeta1, t));
    m->adam_open = true;
    return ADN_OK;
}
int adn_adam_range(adn_model* m, int64_t begin_floats, int64_t end_floats) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    ADN_CHECK(m->adam_open, ADN_ERR_STATE, "adn_adam_range outside adn_adam_begin / adn_adam_end");
    ADN_CHECK(begin_floats >= 0 && begin_floats <= end_floats && begin_floats % 8 == 0, ADN_ERR_INVALID, "bad range");
    const int64_t e = std::min<int64_t>(end_floats, (int64_t)m->flat_floats);      // (the tail slots behind the parameters)
    if (e <= begin_floats) return ADN_OK;
    const size_t b = (size_t)begin_floats;
    const bool pl = m->planes() && m->params16 && m->params16lo;
    void* p16 = ((shadows_on(m) || pl) && m->params16) ? static_cast<void*>(m->params16 + b * 2) : nullptr;   // (char*: bf16 elements)
    return adam_update(m->flat[ADN_BUF_PARAM] + b, m->flat[ADN_BUF_GRAD] + b, m->flat[ADN_BUF_ADAM_M] + b, m->flat[ADN_BUF_ADAM_V] + b,
                       e - begin_floats, m->adam_a_t, kBeta1, kBeta2, kEps, m->stream, p16,
                       m->poison_word(), m->poison_sticky, pl ? static_cast<void*>(m->params16lo + b * 2) : nullptr);
}
int adn_adam_ranges(adn_model* m, const int64_t* begin_floats, const int64_t* end_floats, int n) {
    ADN_CHECK(m && begin_floats && end_floats && n >= 0, ADN_ERR_INVALID, "bad argument");
    ADN_CHECK(m->adam_open, ADN_ERR_STATE, "adn_adam_ranges outside adn_adam_begin / adn_adam_end");
    std::vector<int64_t> b, e;
    for (int k = 0; k < n; ++k) {
        ADN_CHECK(begin_floats[k] >= 0 && begin_floats[k] <= end_floats[k] && begin_floats[k] % 8 == 0, ADN_ERR_INVALID, "bad range");
        const int64_t ee = std::min<int64_t>(end_floats[k], (int64_t)m->flat_floats);
        if (ee > begin_floats[k]) { b.push_back(begin_floats[k]); e.push_back(ee); }
    }
    if (b.empty()) return ADN_OK;
    const bool pl = m->planes() && m->params16 && m->params16lo;
    void* p16 = ((shadows_on(m) || pl) && m->params16) ? static_cast<void*>(m->params16) : nullptr;
    return adam_update_ranges(m->flat[ADN_BUF_PARAM], m->flat[ADN_BUF_GRAD], m->flat[ADN_BUF_ADAM_M], m->flat[ADN_BUF_ADAM_V], b.data(), e.data(),
                              (int)b.size(), m->adam_a_t, kBeta1, kBeta2, kEps, m->stream, p16, m->poison_word(), m->poison_sticky,
                              pl ? static_cast<void*>(m->params16lo) : nullptr);
}
int adn_adam_end(adn_model* m) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    ADN_CHECK(m->adam_open, ADN_ERR_STATE, "adn_adam_end without adn_adam_begin");
    m->adam_open = false;
    m->grads_valid = false;
    m->mark_params_dirty();
    m->params16_values_fresh = (shadows_on(m) || (m->planes() && m->params16lo)) && m->params16;
    return ADN_OK;
}

int adn_set_dropout_state(adn_model* m, uint32_t seed, uint32_t counter) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    m->drop_seed = seed; m->drop_counter = counter;
    return ADN_OK;
}

int adn_apply_sgd(adn_model* m, float learning_rate, float momentum, int nesterov) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    ADN_CHECK(m->grads_valid, ADN_ERR_STATE, "adn_apply_sgd called without gradients (call adn_compute_grads first)");
    ADN_CHECK(momentum >= 0.f && momentum < 1.f, ADN_ERR_INVALID, "momentum must be in [0, 1)");
    ADN_TRY(sgd_update(m->flat[ADN_BUF_PARAM], m->flat[ADN_BUF_GRAD], m->flat[ADN_BUF_ADAM_M], (int64_t)m->flat_floats,
                       learning_rate, momentum, nesterov, m->stream, m->poison_word(), m->poison_sticky));
    m->grads_valid = false;
    m->mark_params_dirty();
    return ADN_OK;
}

int adn_apply_adadelta(adn_model* m, float learning_rate, float rho, float epsilon) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    ADN_CHECK(m->grads_valid, ADN_ERR_STATE, "adn_apply_adadelta called without gradients (call adn_compute_grads first)");
    ADN_TRY(adadelta_update(m->flat[ADN_BUF_PARAM], m->flat[ADN_BUF_GRAD], m->flat[ADN_BUF_ADAM_M], m->flat[ADN_BUF_ADAM_V],
                            (int64_t)m->flat_floats, learning_rate, rho, epsilon, m->stream, m->poison_word(), m->poison_sticky));
    m->grads_valid = false;
    m->mark_params_dirty();
    return ADN_OK;
}

int adn_apply_adam_vlr(adn_model* m, const float* lr_by_param, int n) {
    ADN_CHECK(m && lr_by_param, ADN_ERR_INVALID, "null argument");
    ADN_CHECK(n == (int)m->params.size(), ADN_ERR_INVALID, "one learning rate per parameter tensor is required");
    ADN_CHECK(m->grads_valid, ADN_ERR_STATE, "adn_apply_adam_vlr called without gradients");
    // views that share a physical tensor (the per-gate matrices of one LSTM) must share their learning rate --
    // generate_lr_map keys on the LAYER name (custom/updates.py:26-32), so they always do
    std::vector<std::pair<std::pair<size_t, size_t>, float>> ranges;
    for (int i = 0; i < n; ++i) {
        const ParamDesc& p = m->params[i];
        bool found = false;
        for (auto& r : ranges)
            if (r.first.first == p.phys_begin) {
                ADN_CHECK(r.second == lr_by_param[i], ADN_ERR_INVALID,
                          "parameters stored in one tensor need one learning rate: " + p.name);
                found = true;
            }
        if (!found) ranges.push_back({{p.phys_begin, p.phys_end}, lr_by_param[i]});
    }
    m->adam_t += 1;
    const float t = (float)m->adam_t;
    const float scale = sqrtf(1.f - powf(kBeta2, t)) / (1.f - powf(kBeta1, t));
    for (auto& r : ranges) {
        const size_t b = r.first.first, e = r.first.second;
        ADN_TRY(adam_update(m->flat[ADN_BUF_PARAM] + b, m->flat[ADN_BUF_GRAD] + b, m->flat[ADN_BUF_ADAM_M] + b,
                            m->flat[ADN_BUF_ADAM_V] + b, (int64_t)(e - b), r.second * scale, kBeta1, kBeta2, kEps,
                            m->stream, nullptr, m->poison_word(), m->poison_sticky));
    }
    m->grads_valid = false;
    m->mark_params_dirty();
    return ADN_OK;
}

int adn_adam_step_count(const adn_model* m) { return m ? m->adam_t : 0; }
int adn_set_adam_step_count(adn_model* m, int t) {
    ADN_CHECK(m && t >= 0, ADN_ERR_INVALID, "bad argument");
    m->adam_t = t;
    return ADN_OK;
}

int adn_train_step(adn_model* m, const void* const* inputs, const int32_t* targets, const uint8_t* mask, int B, int T,
                   int theta, int flags, float learning_rate, float* loss) {
    // the cost is read back AFTER the update is enqueued so that the host wait overlaps nothing useful less
    float* loss_dev_or_host = loss;
    ADN_TRY(adn_compute_grads(m, inputs, targets, mask, B, T, theta, flags, 0.0, nullptr));
    ADN_TRY(adn_apply_adam(m, learning_rate));
    if (loss_dev_or_host) return fetch(m, loss_dev_or_host, m->loss, sizeof(float), flags & ADN_FLAG_DEVICE_OUTPUTS);
    return ADN_OK;
}

int adn_read_encoder_activation(adn_model* m, int stream, int layer, float* host_dst) {
    ADN_CHECK(m && host_dst, ADN_ERR_INVALID, "null argument");
    ADN_CHECK(stream >= 0 && stream < m->S, ADN_ERR_INVALID, "stream index out of range");
    StreamState& st = m->st[stream];
    ADN_CHECK(layer >= 0 && layer < st.cfg.n_enc, ADN_ERR_INVALID, "encoder layer index out of range");
    ADN_CHECK(m->lastB > 0, ADN_ERR_STATE, "no forward pass has been run yet");
    const int u = st.cfg.enc_units[layer];
    ADN_HIP_CHECK(hipStreamSynchronize(m->stream));
    const size_t rows = (size_t)m->lastB * m->lastT;
    if (m->compact && m->h_comp_of_full.size() == rows) {      // compact.hip: the matrix holds Nc rows; row b T + t of the answer is row comp_of_full
        std::vector<float> tmp((size_t)m->Nc * u);
        const int saved = m->lastB; const bool was = m->compact;
        m->compact = false; m->lastB = m->Nc; const int savedT = m->lastT; m->lastT = 1;      // (read the Nc rows through the code below)
        const int rc = adn_read_encoder_activation(m, stream, layer, tmp.data());
        m->compact = was; m->lastB = saved; m->lastT = savedT;
        if (rc != ADN_OK) return rc;
        for (size_t r = 0; r < rows; ++r) memcpy(host_dst + r * u, tmp.data() + (size_t)m->h_comp_of_full[r] * u, (size_t)u * 4);
        return ADN_OK;
    }
    if (m->planes()) {                       // bf16x3: an activation kept as its two planes only gets hi + lo written back first
        ADN_TRY(restore_fp32(m, st.act[layer]));
        ADN_HIP_CHECK(hipStreamSynchronize(m->stream));
    }
    if (shadows_on(m) && layer + 1 < st.cfg.n_enc && m->shadow_of(st.act[layer])) {
        // bf16 mode keeps only the bf16 copy of intermediate activations: widen it on the host
        std::vector<uint16_t> tmp(rows * ld_of(u));
        ADN_HIP_CHECK(hipMemcpy(tmp.data(), m->shadow_of(st.act[layer]), tmp.size() * 2, hipMemcpyDeviceToHost));
        for (size_t r = 0; r < rows; ++r)
            for (int c = 0; c < u; ++c) {
                const uint32_t bits = (uint32_t)tmp[r * ld_of(u) + c] << 16;
                memcpy(&host_dst[r * u + c], &bits, 4);
            }
        return ADN_OK;
    }
    ADN_HIP_CHECK(hipMemcpy2D(host_dst, (size_t)u * 4, st.act[layer], (size_t)ld_of(u) * 4, (size_t)u * 4, rows,
                              hipMemcpyDeviceToHost));
    return ADN_OK;
}

int adn_profile_enable(adn_model* m, int on) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    ADN_HIP_CHECK(hipStreamSynchronize(m->stream));
    if (on) { m->prof.reset(); m->prof.enabled = true; g_prof = &m->prof; }
    else { m->prof.drain(); m->prof.enabled = false; if (g_prof == &m->prof) g_prof = nullptr; }
    return ADN_OK;
}

int adn_profile_read(adn_model* m, adn_profile_entry* out, int max_entries, int* n_out) {
    ADN_CHECK(m && out && n_out, ADN_ERR_INVALID, "null argument");
    ADN_HIP_CHECK(hipStreamSynchronize(m->stream));
    m->prof.drain();
    int n = 0;
    for (int k = 0; k < PROF_COUNT && n < max_entries; ++k) {
        if (!m->prof.launches[k]) continue;
        memset(&out[n], 0, sizeof(out[n]));
        strncpy(out[n].name, kProfNames[k], sizeof(out[n].name) - 1);
        out[n].launches = m->prof.launches[k]; out[n].ms = m->prof.ms[k];
        out[n].flops = m->prof.flops[k]; out[n].bytes = m->prof.bytes[k];
        ++n;
    }
    *n_out = n;
    return ADN_OK;
}

int adn_set_deterministic(int on) { set_deterministic(on != 0); return ADN_OK; }
int adn_get_deterministic(void) { return deterministic() ? 1 : 0; }

int adn_debug_raise_exchange_error(int value) {
    int* word = nullptr;
    ADN_TRY(lstm_cluster_error_word(&word));
    ADN_HIP_CHECK(hipMemcpy(word, &value, sizeof(int), hipMemcpyHostToDevice));
    return ADN_OK;
}

int adn_synchronize(adn_model* m) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    ADN_HIP_CHECK(hipStreamSynchronize(m->stream));
    return check_device_errors(m);
}

// ---- operator-level entry points ---------------------------------------------------------------
int adn_op_gemm(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                const float* bias, int act, int accumulate, void* hip_stream) {
    GemmArgs g;
    g.layout = layout; g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.bias = bias; g.act = act; g.accumulate = accumulate;
    return gemm(g, static_cast<hipStream_t>(hip_stream));
}

int adn_op_gemm_ex(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                   const float* bias, int act, int accumulate, int precision, void* hip_stream) {
    GemmArgs g;
    g.layout = layout; g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.bias = bias; g.act = act; g.accumulate = accumulate; g.precision = precision;
    return gemm(g, static_cast<hipStream_t>(hip_stream));
}

int adn_op_gemm_shadow(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                       int ldc, const void* A16, const void* B16, void* C16, int accumulate, void* hip_stream) {
    GemmArgs g;
    g.layout = layout; g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.accumulate = accumulate; g.precision = ADN_PRECISION_BF16; g.A16 = A16; g.B16 = B16; g.C16 = C16;
    return gemm(g, static_cast<hipStream_t>(hip_stream));
}

int adn_op_to_bf16(const float* src, void* dst, int64_t n, void* hip_stream) {
    return to_bf16(src, dst, (size_t)n, static_cast<hipStream_t>(hip_stream));
}

int adn_op_delta_forward(const float* in, int ld_in, float* out, int ld_out, int B, int T, int F, int theta,
                         void* hip_stream) {
    return delta_forward(in, ld_in, out, ld_out, B, T, F, theta, 1, static_cast<hipStream_t>(hip_stream));
}

int adn_op_delta_backward(const float* dout, int ld_out, float* din, int ld_in, int B, int T, int F, int theta,
                          void* hip_stream) {
    return delta_backward(dout, ld_out, din, ld_in, B, T, F, theta, 1, static_cast<hipStream_t>(hip_stream));
}

int adn_op_adam(float* p, const float* g, float* m, float* v, int64_t n, float a_t, void* hip_stream) {
    return adam_update(p, g, m, v, n, a_t, kBeta1, kBeta2, kEps, static_cast<hipStream_t>(hip_stream));
}

int adn_op_copy_bench(const float* src, float* dst, int64_t n, int repeats, void* hip_stream, float* ms) {
    return copy_bench(src, dst, n, repeats, static_cast<hipStream_t>(hip_stream), ms);
}

// ---- feature front-end (prep.hip) ----------------------------------------------------------------
int adn_prep_seq_deltas(const float* in, int ld_in, float* out, int ld_out, const int32_t* first, const int32_t* last,
                        int n_frames, int F, int w, void* hip_stream) {
    return prep_seq_deltas(in, ld_in, out, ld_out, first, last, n_frames, F, w, static_cast<hipStream_t>(hip_stream));
}
int adn_prep_diff_images(const float* in, float* out, int ld, const int32_t* first, const int32_t* last, int n_frames, int D,
                         void* hip_stream) {
    return prep_diff_images(in, out, ld, first, last, n_frames, D, static_cast<hipStream_t>(hip_stream));
}
int adn_prep_mean_image_subtraction(const float* in, float* out, int ld, const int32_t* starts, const int32_t* lens, int n_utt,
                                    int D, void* hip_stream) {
    return prep_mean_image_subtraction(in, out, ld, starts, lens, n_utt, D, static_cast<hipStream_t>(hip_stream));
}
int adn_prep_normalize_rows(float* x, int ld, int rows, int cols, void* hip_stream) {
    return prep_normalize_rows(x, ld, rows, cols, static_cast<hipStream_t>(hip_stream));
}
int adn_prep_column_stats(const float* x, int ld, int rows, int cols, double* workspace, float* mean, float* std,
                          void* hip_stream) {
    return prep_column_stats(x, ld, rows, cols, workspace, mean, std, static_cast<hipStream_t>(hip_stream));
}
int adn_prep_apply_column_norm(const float* x, float* out, int ld, int rows, int cols, const float* mean, const float* std,
                               void* hip_stream) {
    return prep_apply_column_norm(x, out, ld, rows, cols, mean, std, static_cast<hipStream_t>(hip_stream));
}
int adn_prep_gather_columns(const float* in, int ld_in, float* out, int ld_out, const int32_t* perm, int rows, int cols,
                            void* hip_stream) {
    return prep_gather_columns(in, ld_in, out, ld_out, perm, rows, cols, static_cast<hipStream_t>(hip_stream));
}
int adn_prep_lcn(const float* x, float* y, int n_images, int H, int W, const float* filter_host, int ksize, float threshold,
                 void* hip_stream) {
    return prep_lcn(x, y, n_images, H, W, filter_host, ksize, threshold, static_cast<hipStream_t>(hip_stream));
}

}  // extern "C"
