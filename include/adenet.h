/* adenet.h -- C ABI of libadenet_hip.so: the MI355X (gfx950) implementation of the AdeNet /
 * DeltaNet training path of lzuwei/ip-avsr.
 *
 * The reference has no FFI: its boundary for this path is the Python surface its scripts use
 *  -- a `create_model(...)` graph factory plus four compiled callables
 * (`train`, `compute_train_cost`, `compute_test_cost`, `val_fn`) and Lasagne's parameter
 * accessors.  Each entry point below names the reference interface it stands in for
 * (paths relative to the reference repository).  The Python mirror of that surface lives in
 * ip_avsr_amd/ (modelzoo/, model.py) and binds these symbols with ctypes; see INTEGRATION.md.
 *
 * Conventions
 *  - plain C types only; every function returns a status code of enum adn_status, 0 = ok, and records a message
 *    retrievable with adn_last_error() (thread-local);
 *  - stream inputs are float32 (B,T,D_s) C-contiguous batch-major, mask is uint8 (B,T),
 *    targets are int32 (B,T) -- exactly what the reference's theano functions receive
 *    (runners/3stream.py:276-281,309-320); pointers may be host or device memory
 *    (ADN_FLAG_DEVICE_INPUTS / ADN_FLAG_DEVICE_OUTPUTS);
 *  - all work is enqueued on the stream given to adn_set_stream (default: the null stream);
 *    calls that return host data synchronise that stream before returning;
 *  - a model is not thread-safe; use one model per GPU / per host thread (the reference's
 *    caller is single-threaded).
 */
#ifndef ADENET_H_
#define ADENET_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADN_MAX_STREAMS 8
#define ADN_MAX_ENC_LAYERS 8
#define ADN_MAX_CLASSES 64

typedef enum {
    ADN_OK = 0,
    ADN_ERR_INVALID = 1,   /* bad argument / unsupported configuration */
    ADN_ERR_HIP = 2,       /* a HIP runtime call failed */
    ADN_ERR_NO_DEVICE = 3, /* no gfx950 device visible */
    ADN_ERR_STATE = 4      /* call sequence error (e.g. apply_adam without gradients) */
} adn_status;

/* custom/nonlinearities.py:4-16 (the subset the DBN encoders use) */
typedef enum {
    ADN_ACT_LINEAR = 0,
    ADN_ACT_RECTIFY = 1,
    ADN_ACT_SIGMOID = 2,
    ADN_ACT_TANH = 3,
    ADN_ACT_LEAKY_RECTIFY = 4,     /* slope 0.01 */
    ADN_ACT_VERY_LEAKY_RECTIFY = 5, /* slope 1/3  */
    ADN_ACT_SCALED_TANH = 6,        /* 2.4 tanh(0.5 x): ScaledTanh(0.5, 2.4) of modelzoo/avletters_convae.py:7-26 */
    ADN_ACT_SCALED_TANH_LECUN = 7   /* 1.7159 tanh(2/3 x): ScaledTanh(2./3, 1.7159) of modelzoo/avletters_convae_bndrop.py:8 */
} adn_act;

/* fusiontype of modelzoo/adenet_v2.py:68-75 and friends */
typedef enum { ADN_FUSE_NONE = 0, ADN_FUSE_SUM = 1, ADN_FUSE_ADASUM = 2, ADN_FUSE_CONCAT = 3 } adn_fusion;

typedef enum {
    ADN_PRECISION_F32 = 0, /* exact fp32 on the f32 MFMA pipe (parity-grade; all parity tests run in it) */
    ADN_PRECISION_BF16 = 1, /* GEMM operands rounded to bf16 (RNE) in flight, fp32 accumulate, fp32 master
                              weights / activations / recurrence / optimiser (BASELINE configs[1]: bf16) */
    ADN_PRECISION_BF16X3 = 2, /* fp32 everywhere like ADN_PRECISION_F32, but every large GEMM runs as three bf16 MFMA
                              products of the operands' bf16 hi / lo parts (a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32
                              accumulate: ~4e-6 relative, inside the 1e-4 parity gate) at the bf16 matrix rate; the
                              recurrent products likewise, in weight-stationary kernels, for LSTMs of <= 256 units
                              (wider ones run the fp32 per-step kernels) */
    ADN_PRECISION_MIXED = 3 /* ADN_PRECISION_BF16X3 for everything the forward pass computes (activations, probabilities,
                              votes: the same bits), recurrent kernels included; back-propagation runs ONE bf16 product per
                              GEMM over the operands' hi planes and per recurrent step over the hi image of W_hid (gradients
                              of bf16 grade, as in ADN_PRECISION_BF16; back-propagated tensors are kept as their hi plane).
                              Not a parity mode for gradients: reported under its own name, never as bf16x3 */
} adn_precision;

enum {
    ADN_FLAG_DEVICE_INPUTS = 1,
    ADN_FLAG_DEVICE_OUTPUTS = 2,
    ADN_FLAG_STOCHASTIC = 4,     /* adn_loss: dropout layers active (compute_train_cost, runners/3stream.py:372) */
    ADN_FLAG_DETERMINISTIC = 8,  /* adn_compute_grads / adn_train_step: dropout layers off (they are on by default) */
    ADN_FLAG_BF16_INPUTS = 16,   /* the stream (and auxiliary) inputs are bfloat16 arrays instead of float32 -- what a bf16
                                    feature front-end leaves in HBM.  In ADN_PRECISION_BF16 a device array of an encoder stream
                                    is then read in place by the first encoder GEMM (no staging copy, no conversion pass);
                                    everywhere else it is widened to float32 (exact) first.  The reference's theano functions
                                    take float32 (allow_input_downcast) only: this is an addition, not a replacement. */
    ADN_FLAG_PLANE_INPUTS = 32   /* ADN_PRECISION_BF16X3 / _MIXED, device inputs only: every stream input arrives as its two bfloat16
                                    planes hi = bf16(x), lo = bf16(x - hi) -- what the GEMMs of that mode read (the library
                                    otherwise makes them from the float32 array with a split pass per call).  `inputs` then holds
                                    2 S pointers: the S hi planes, then the S lo planes (dense (B, T, D) bfloat16 each).  Every
                                    stream needs an encoder and D % 8 == 0; no auxiliary inputs.  Results equal those of passing
                                    the float32 values hi + lo to the last bit or two (the library's own split of hi + lo re-rounds
                                    the hi plane where lo is exactly half a unit of it). */
};

/* what the classifier sees and how it is trained */
typedef enum {
    ADN_HEAD_FRAMES = 0, /* softmax on every frame, temporal_softmax_loss, majority vote (adenet_v2.py:77-92) */
    ADN_HEAD_LAST = 1    /* SliceLayer(-1) + softmax on the last time step, categorical cross-entropy
                            (modelzoo/adenet_v3.py:180-186, deltanet.py:48-56, avletters/trimodal.py:327-328) */
} adn_head;

/* which flat buffer a tensor accessor addresses */
typedef enum { ADN_BUF_PARAM = 0, ADN_BUF_GRAD = 1, ADN_BUF_ADAM_M = 2, ADN_BUF_ADAM_V = 3 } adn_buffer;

/* one input stream: [dense encoder] -> [delta layer] -> LSTM or summed BLSTM
 * (modelzoo/pretrained_encoder.py:4-9, custom/layers.py:105-121, modelzoo/adenet_3stream.py:166-238) */
typedef struct {
    int32_t input_dim;
    int32_t n_enc;                           /* 0: no encoder (e.g. a DCT stream) */
    int32_t enc_units[ADN_MAX_ENC_LAYERS];
    int32_t enc_act[ADN_MAX_ENC_LAYERS];     /* adn_act */
    int32_t use_delta;                       /* DeltaLayer present */
    int32_t bidirectional;                   /* 0: LSTMLayer, 1: forward+backward LSTMLayer summed */
    int32_t peepholes;
    float dropout_p;                         /* DropoutLayer ahead of the stream's LSTM (adenet_v3.py:112,123,134); 0: none */
    int32_t batchnorm;                       /* BatchNormLayer on the encoder output, ahead of the delta layer (adenet_v1.py:82) */
    int32_t aux_dim;                         /* width of an auxiliary input (B,T,aux_dim) concatenated behind the delta features
                                                (ConcatLayer([l_delta, l_dct], axis=2), adenet_v1.py:87); its array follows the
                                                n_streams stream inputs in `inputs`, in stream order; 0: none */
} adn_stream_config;

/* the whole graph: S streams -> fusion -> aggregation (B)LSTM -> per-timestep softmax
 * (modelzoo/adenet_v2.py:12-94, adenet_v2_2.py:40-132, adenet_2stream.py:116-210,
 *  adenet_3stream.py:145-264, adenet_4stream.py:12-159, avnet.py:30-114,
 *  deltanet_majority_vote.py:14-66, deltanet_v1.py:8-42, lstm_classifier_majority_vote.py:10-43) */
typedef struct {
    int32_t n_streams;
    adn_stream_config streams[ADN_MAX_STREAMS];
    int32_t fusion;          /* adn_fusion; ADN_FUSE_NONE requires n_streams == 1 */
    int32_t agg;             /* 0: none (classifier on the stream output), 1: LSTM, 2: summed BLSTM */
    int32_t agg_peepholes;
    int32_t lstm_size;       /* H, all LSTMs */
    int32_t classes;         /* C <= ADN_MAX_CLASSES */
    int32_t precision;       /* adn_precision */
    int32_t head;            /* adn_head */
    float agg_dropout_p;     /* DropoutLayer on the fused tensor (adenet_v3.py:154); 0: none */
    int32_t stream_lstm_units; /* 0 (= lstm_size) or the smaller width of the STREAM LSTMs under wider aggregation LSTMs
                                  (adenet_v1.py:89,95: lstm_size and 2 * lstm_size).  They run inside lstm_size-wide kernels
                                  with the surplus units pinned at zero: zero weights give zero state and zero gradient, so
                                  the padding never trains; parameter views have the narrow shapes.  Not with concat fusion. */
    int32_t reserved[5];
} adn_config;

typedef struct adn_model adn_model;

/* description of one trainable tensor, in Lasagne get_all_params order */
typedef struct {
    char name[96];      /* e.g. "stream0.enc1.W", "stream2.lstm0.W_hid_to_forgetgate", "agg1.cell_init",
                           "fuse.adacoeff0", "softmax.b" */
    int32_t ndim;       /* 0, 1 or 2 */
    int64_t dims[2];
    int64_t numel;
} adn_param_info_t;

const char* adn_version(void);
/* sizeof(adn_stream_config), sizeof(adn_config), sizeof(adn_param_info_t), sizeof(adn_profile_entry): lets a binding
 * check its struct declarations against the library it loaded */
void adn_abi_sizes(int32_t out[4]);
const char* adn_last_error(void);
/* number of visible HIP devices whose arch is gfx950 (0 if none / no driver) */
int adn_device_count(void);

/* <- modelzoo/<model>.create_model(...) : builds the graph, allocates parameters (zero-initialised:
 * the caller injects values, like the reference injects DBN weights / Lasagne initialisers),
 * gradients, Adam state.  Uses the current HIP device. */
int adn_create(const adn_config* cfg, adn_model** out);
void adn_destroy(adn_model* m);
int adn_set_stream(adn_model* m, void* hip_stream);
/* switch the GEMM arithmetic (adn_precision) of an existing model; parameters are fp32 either way */
int adn_set_precision(adn_model* m, int precision);

/* <- lasagne.layers.get_all_params / get_all_param_values / set_all_param_values
 *    (runners/3stream.py:305,393,425; utils/io.py:40-48) */
int adn_num_params(const adn_model* m);
int adn_param_info(const adn_model* m, int index, adn_param_info_t* info);
int adn_read_tensor(adn_model* m, int buffer /*adn_buffer*/, int index, float* host_dst);
int adn_write_tensor(adn_model* m, int buffer /*adn_buffer*/, int index, const float* host_src);
int64_t adn_total_param_count(const adn_model* m); /* logical elements (17 999 676 for 3-stream concat) */

/* flat device buffers (physical layout incl. alignment padding, plus a tail of 8 floats); the gradient
 * buffer is what a data-parallel caller all-reduces.  After adn_compute_grads the first
 * tail float of the gradient buffer holds this call's share of the cost, so ONE all-reduce sums the
 * gradients and the cost together. */
int adn_flat_buffer(adn_model* m, int buffer /*adn_buffer*/, void** device_ptr, size_t* bytes);
/* The same address for a caller that only READS the buffer (a device-side snapshot of the parameters: what the epoch drivers
 * keep as "best parameters so far", runners/3stream.py:393): unlike adn_flat_buffer(ADN_BUF_PARAM, ...) it does not mark the
 * parameters as written, so the bf16 copies / planes / MFMA-fragment images derived from them are not re-made. */
int adn_flat_buffer_const(const adn_model* m, int buffer /*adn_buffer*/, const void** device_ptr, size_t* bytes);

/* Gradient buckets for overlapping the data-parallel all-reduce with back-propagation (new; the reference is
 * single-device).  Listed in the order they become final: bucket 0 = [fusion | aggregation LSTMs | classifier |
 * cost tail] (released behind the stream LSTMs' backward launch); one bucket per stream for everything behind its
 * encoder ([BatchNorm | LSTMs]; the whole stream when it has no encoder), final before the encoder's backward starts;
 * one bucket [W_l | b_l] per stream and encoder layer, final behind that layer's weight-gradient GEMM and released
 * ahead of its input-gradient GEMM.  Order: layer-major (all streams' tops, then depth by depth) by default,
 * stream-major (a stream's top, then its layers, stream after stream) when back-propagation runs stream-major
 * (ADN_DP_STREAM_MAJOR / ADN_NO_GROUPED_BACKWARD / ADN_STREAMS); decided at the first of adn_grad_buckets /
 * adn_set_bucket_events / adn_compute_grads and kept for the model's lifetime.
 * adn_compute_grads records the caller's HIP events (one per bucket, on the model's stream) at
 * those points; a caller makes its communication stream wait on event k and reduces range k while the rest of
 * the backward pass still runs.  Ranges are in floats inside adn_flat_buffer(ADN_BUF_GRAD); together they cover
 * the buffer exactly once. */
int adn_grad_buckets(const adn_model* m, int max_buckets, int64_t* begin_floats, int64_t* end_floats, int* n_out);
/* group_of_bucket[k]: buckets with the same number are released at the same point of back-propagation (the streams' ranges
 * behind one grouped launch); they are consecutive in the list, and a caller may reduce them with ONE grouped collective
 * behind the last one's event. */
int adn_grad_bucket_groups(const adn_model* m, int max_buckets, int* group_of_bucket, int* n_out);
int adn_set_bucket_events(adn_model* m, void* const* hip_events, int n); /* n = 0 clears */

/* <- val_fn(inputs..., mask, window) -> probabilities (B,T,C)  (runners/3stream.py:320) */
int adn_forward(adn_model* m, const void* const* inputs, const uint8_t* mask, int B, int T, int theta,
                int flags, float* probs);

/* <- compute_train_cost / compute_test_cost(inputs..., targets, mask, window) -> cost
 *    (runners/3stream.py:311-318; identical here: no stochastic layers on this path) */
int adn_loss(adn_model* m, const void* const* inputs, const int32_t* targets, const uint8_t* mask, int B,
             int T, int theta, int flags, float* loss);

/* The probabilities of the LAST forward pass (adn_forward / adn_loss / adn_compute_grads on a (B, T) batch), in adn_forward's
 * layout.  The reference's epoch loop runs compute_test_cost and val_fn back to back on the same held-out split with the same
 * parameters (runners/3stream.py:373,383): two compiled functions there, one forward pass here -- adn_loss without
 * ADN_FLAG_STOCHASTIC leaves exactly what adn_forward would return. */
int adn_read_probs(adn_model* m, int B, int T, int flags, float* probs);

/* forward + temporal_softmax_loss (custom/objectives.py:4-39) + back-propagation into the gradient
 * buffer.  total_frames <= 0: normalise by this batch's valid frames (single-GPU semantics);
 * > 0: normalise by that number (data parallel: the global count, so that the SUM of the ranks'
 * gradients / losses equals the single-GPU result).  *loss is this call's share of the cost. */
int adn_compute_grads(adn_model* m, const void* const* inputs, const int32_t* targets, const uint8_t* mask,
                      int B, int T, int theta, int flags, double total_frames, float* loss);

/* <- lasagne.updates.adam (runners/3stream.py:307; formula custom/updates.py:73-99): one step on
 * the current gradient buffer; beta1=.9 beta2=.999 eps=1e-8 */
/* Data parallel: a rank whose shard of a minibatch is EMPTY (fewer utterances than ranks in the short last minibatch of
 * gen_lstm_batch_random, reference utils/datagen.py) still has to join every gradient all-reduce: this zeroes the
 * gradient buffer (cost share included), records the bucket events of adn_set_bucket_events and marks the gradients
 * valid, i.e. it is adn_compute_grads of nothing. */
int adn_zero_grads(adn_model* m);
int adn_apply_adam(adn_model* m, float learning_rate);
/* The same step applied range by range of the flat buffers (floats, begin a multiple of 8; the ranges of
 * adn_grad_buckets qualify): begin, any number of ranges, end.  Covering every parameter once equals
 * adn_apply_adam bit for bit.  Data parallel (no reference counterpart): a bucket is updated as soon as
 * its all-reduce has landed, while later buckets are still being reduced. */
int adn_adam_begin(adn_model* m, float learning_rate);
int adn_adam_range(adn_model* m, int64_t begin_floats, int64_t end_floats);
int adn_adam_ranges(adn_model* m, const int64_t* begin_floats, const int64_t* end_floats, int n);   /* n ranges, one launch per 16 */
int adn_adam_end(adn_model* m);
/* <- custom/updates.py:35-99 adam_vlr: Adam with one learning rate per parameter tensor (index order of
 * adn_param_info); tensors of one layer must share theirs, as generate_lr_map (custom/updates.py:10-32) gives */
int adn_apply_adam_vlr(adn_model* m, const float* lr_by_param, int n);
int adn_adam_step_count(const adn_model* m);
int adn_set_adam_step_count(adn_model* m, int t);

/* <- train(inputs..., targets, mask, window) -> cost, parameters updated (runners/3stream.py:309-310,370) */
int adn_train_step(adn_model* m, const void* const* inputs, const int32_t* targets, const uint8_t* mask,
                   int B, int T, int theta, int flags, float learning_rate, float* loss);

/* activations of the last forward pass, for parity checks: encoder layer `layer` (0-based) of
 * stream `stream` as (B*T, units) batch-major rows (B,T order) */
int adn_read_encoder_activation(adn_model* m, int stream, int layer, float* host_dst);

int adn_synchronize(adn_model* m);
/* Deterministic mode (process-wide; also ADN_DETERMINISTIC=1 in the environment): every reduction whose order otherwise follows
 * the arrival order of float atomics -- bias / initial-state / peephole gradients of the LSTM kernels, column and scalar sums, the
 * register-staged GEMMs' split-K -- runs in a fixed order, so that two runs from the same parameters, batches and dropout state
 * give the same bits (the reference itself is not reproducible: it never seeds).  Slower; the
 * arithmetic is otherwise the selected adn_precision's. */
int adn_set_deterministic(int on);
int adn_get_deterministic(void);
/* Frame compaction (round 5; csrc/compact.hip).  The reference pads every utterance of a minibatch to the longest with ZERO frames
 * (utils/datagen.py:104,129-142) and sends all B T frames through the dense encoders.  With the lengths of a call's batch at hand the
 * encoders run over the sum(len) valid frames plus ONE zero-input row that stands for every padding frame -- forward its output
 * enc(0) is what the delta layer sees at the padding frames, backward it carries the sum of their gradients, which is all the
 * parameter gradients ever see of them (the encoder is row-wise).  Same results as the padded computation up to summation order;
 * behind the delta layer a train step runs over length buckets (adn_set_length_buckets, below), every other call over B T rows.
 * Applies in the 16-bit arithmetics (bf16 / bf16x3 / mixed) whose encoders end in
 * a linear layer without BatchNorm, when the batch has at least 8192 rows (B T; smaller ones are latency-bound and gain nothing;
 * ADN_COMPACT_MIN_ROWS overrides) of which at least 10 % are padding; otherwise the call runs padded as before.  ADN_NO_COMPACT=1
 * in the environment switches it off.
 * Where the lengths come from (round 6):
 *  - HOST arrays (the reference's call: train(*inputs, targets, mask, window) carries no lengths, runners/3stream.py:309-320,370):
 *    read off the mask when it is a prefix mask; the device then looks at the padding frames it was just sent, and a batch with a
 *    non-zero padding frame simply runs padded.  Nothing to announce, nothing promised (adn_set_auto_compaction(m, 0) turns this off).
 *  - DEVICE arrays (ADN_FLAG_DEVICE_INPUTS): the host must say how many rows there are -- adn_set_batch_lengths() announces the
 *    lengths of the NEXT call's batch (host array of B ints, 1 <= len <= T; used up by that call, whatever becomes of it) and
 *    promises that the padding frames of the stream inputs are zero.
 * What is checked: an announcement whose B is not the call's, or with a length outside [1, T]: ADN_ERR_INVALID from that call, before
 * anything runs.  An announcement against a host mask: compared on the host, ADN_ERR_INVALID.  Against a device mask: compared by the
 * kernel that walks the mask anyway; a mismatch raises a device word that the next synchronising call (any call returning host
 * results) reports as ADN_ERR_INVALID.  The zero padding frames behind an announcement: scanned for host arrays always
 * (ADN_ERR_INVALID from the call), for device arrays under ADN_CHECK_PADDING=1 in the environment (which also makes both device checks
 * synchronous: the failing call itself returns ADN_ERR_INVALID) -- the tests run with it. */
int adn_set_batch_lengths(adn_model* m, const int32_t* lengths, int B);
/* rows of the encoder matrices in the last call: sum(len) + 1 when it ran compacted, 0 when it ran padded */
int adn_get_compact_rows(const adn_model* m);
/* The derivative the encoders' rectifiers take AT a pre-activation of exactly zero: 0 (default) or 0.5.  Lasagne's rectify is
 * Theano's 0.5 (x + |x|), whose gradient at x == 0 is 0.5 (d|x|/dx = sgn(x), sgn(0) = 0).  That is not a measure-zero case for this
 * model: the reference pads with ZERO frames (utils/datagen.py:129-142), so with zero encoder biases (SURVEY 8d's synthetic
 * parameters; not DBN-pretrained ones) every padding row has pre-activation exactly 0 in every rectifier layer and the reference
 * back-propagates half of the delta layer's leakage into b1..b3 there.  0.5 reproduces that: the forward pass marks an exactly-zero
 * pre-activation (the activation leaves as -0.0f, equal to 0 in all arithmetic) and the masks read the mark.  The specialised
 * kernels (persistent ping-pong, skinny) decline the marked rectifier: the encoders then run on the register-staged / fp32 kernels
 * -- a parity switch, not a throughput mode. */
int adn_set_relu_grad_at_zero(adn_model* m, float value);
/* lengths read off a host mask (see above): on by default */
int adn_set_auto_compaction(adn_model* m, int on);
/* Length buckets (round 6; on by default).  A COMPACTED train step (adn_compute_grads on a batch whose lengths are known, above) also
 * drops most padding frames behind the delta layer: the utterances, sorted by length inside the library, are cut into 2 - 4
 * equal buckets, each as long as its longest utterance, laid one behind the other along the time axis of every time-major tensor
 * -- the recurrent side's GEMMs, sums and loss walk ~Sum(len) / 0.85 rows instead of B T, the weight-stationary LSTM kernels run
 * one launch entry per (LSTM, bucket).  Same loss and gradients as the padded step (a padding frame contributes zero to both;
 * the sums run in another order: tests/test_gpu_buckets.py); taken when it saves >= 10 % of the rows and the model has nothing that
 * walks the time-major tensors frame by frame outside the kernels that know the layout (no dropout, no last-timestep head, no
 * adaptive fusion, no auxiliary inputs; H <= 256; not in deterministic mode).  Forward-only calls never bucket (the reference's
 * val_fn returns probabilities at padding frames too); adn_read_probs after a bucketed train step is refused for the same reason.
 * 0 keeps every train step on the B x T layout; ADN_NO_LENGTH_BUCKETS=1 does the same for the process. */
int adn_set_length_buckets(adn_model* m, int on);
/* rows of the time-major tensors in the last call when it ran over length buckets (Bb x Tt: the buckets' lengths + one spare block
 * between two of them, times the utterances per bucket), else 0 */
int adn_get_bucket_rows(const adn_model* m);
/* test hook: writes `value` into the current device's LSTM-exchange error word (what a weight-stationary LSTM kernel raises
 * when a workgroup gave up waiting for its partners; 0 clears it).  Lets a test follow the word's way through the gradient
 * tail, the data-parallel all-reduce and the optimiser's skip without provoking a real time-out. */
int adn_debug_raise_exchange_error(int value);
/* test hook: launches n_workgroups workgroups on hip_stream that each hold lds_bytes of LDS (<= 160 KB: at that size nothing
 * else fits on their CU) for `ms` milliseconds -- a stand-in for a foreign tenant (another process' kernel, a collective
 * waiting for a slow rank) beside the weight-stationary LSTM launches, whose workgroups wait for their partners
 * (tests/test_gpu_residency.py: such a tenant delays a pass, it does not break it) */
int adn_debug_occupy_cus(int n_workgroups, int lds_bytes, double ms, void* hip_stream);
/* test hook: LSTM forward passes dispatched so far per kernel family: [0] one launch per time step, [1] one-workgroup
 * persistent kernels, [2] weight-stationary kernels, [3] weight-stationary bf16x3 kernels */
int adn_debug_lstm_family_counts(int64_t out[4]);
/* ... and LSTM backward passes, same families */
int adn_debug_lstm_backward_family_counts(int64_t out[4]);
/* test hook, host logic only: the launches a set of n_lstm LSTMs of `groups` utterance groups (wg_per_group workgroups each) is
 * dealt over on a device of `cus` compute units -- ranges of (LSTM, group) pairs, pair = lstm * groups + group; returns the
 * number of launches (the first max_launches of them written), < 0 on bad arguments */
int adn_debug_plan_lstm_launches(int n_lstm, int groups, int wg_per_group, int cus, int32_t* pair0, int32_t* count, int max_launches);

/* per-kernel-class timing with HIP events recorded on the model's stream around every launch of the
 * class (bench.py's live roofline measurement).  flops / bytes are the ALGORITHMIC work of the launches
 * (DESIGN.md "Kernels"), ms their summed durations. */
typedef struct {
    char name[32];
    int64_t launches;
    double ms;
    double flops;
    double bytes;
} adn_profile_entry;
int adn_profile_enable(adn_model* m, int on);  /* on != 0: clear the counters and start recording */
int adn_profile_read(adn_model* m, adn_profile_entry* out, int max_entries, int* n_out); /* synchronises */

/* ---- operator-level entry points (used by the parity tests and micro-benchmarks) -------------- */
/* C (+)= op(A)*op(B) on device pointers; layout 0 = NN, 1 = NT (B given as [N][K]), 2 = TN (A as [K][M]) */
/* dropout masks are a counter-based hash of (seed, counter, layer, element): set both to reproduce a draw; the counter
 * advances by one after every stochastic forward pass */
int adn_set_dropout_state(adn_model* m, uint32_t seed, uint32_t counter);
/* <- lasagne.updates.sgd / momentum / nesterov_momentum (avletters/bimodal.py:446-455): momentum 0 = plain sgd; the
 * velocity lives in the ADN_BUF_ADAM_M buffer */
int adn_apply_sgd(adn_model* m, float learning_rate, float momentum, int nesterov);
/* <- lasagne.updates.adadelta (avletters/avletters_convae.py:230); accumulators in ADN_BUF_ADAM_M / ADN_BUF_ADAM_V */
int adn_apply_adadelta(adn_model* m, float learning_rate, float rho, float epsilon);

int adn_op_gemm(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                int ldc, const float* bias, int act, int accumulate, void* hip_stream);
int adn_op_gemm_ex(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                   int ldc, const float* bias, int act, int accumulate, int precision, void* hip_stream);
/* bf16 mode with pre-converted operands: A16/B16 (and optionally C16) are bf16 copies with the same leading
 * dimensions as A/B/C (adn_op_to_bf16 makes them); null pointers fall back to conversion in flight */
int adn_op_gemm_shadow(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                       int ldc, const void* A16, const void* B16, void* C16, int accumulate, void* hip_stream);
int adn_op_to_bf16(const float* src, void* dst, int64_t n, void* hip_stream);
/* utils/signal.py:59-80 on device: in (B,T,F) batch-major -> out (T,B,3F) time-major */
int adn_op_delta_forward(const float* in, int ld_in, float* out, int ld_out, int B, int T, int F, int theta,
                         void* hip_stream);
int adn_op_delta_backward(const float* dout, int ld_out, float* din, int ld_in, int B, int T, int F, int theta,
                          void* hip_stream);
int adn_op_adam(float* p, const float* g, float* m, float* v, int64_t n, float a_t, void* hip_stream);
/* dst[i] = src[i], 16 bytes per lane, `repeats` back-to-back launches; *ms = their HIP-event time on that stream.  The
 * achievable-HBM-bandwidth yardstick of bench.py (2 n floats move per launch): MI355X_MICROARCH.md quotes 6.29 TB/s for
 * this copy against the 8 TB/s datasheet figure. */
int adn_op_copy_bench(const float* src, float* dst, int64_t n, int repeats, void* hip_stream, float* ms);

/* ---- feature front-end on the GPU (reference utils/preprocessing.py) -------------------------
 * Device pointers, fp32 row-major frame matrices [sum of lengths][ld], work enqueued on hip_stream.  Utterance
 * structure is passed as int32 device vectors built from the length vector: first[t] / last[t] = index of the first /
 * last frame of the utterance frame t belongs to; starts[u] / lens[u] per utterance. */
/* one level of linear-slope deltas along time, per utterance (utils/preprocessing.py:17-51 as used by :465-489): taps
 * m = h..-h (h = w/2), positions before the utterance read its SECOND frame, positions after it its last frame; call
 * twice (out of the first call as input of the second) for delta-deltas */
int adn_prep_seq_deltas(const float* in, int ld_in, float* out, int ld_out, const int32_t* first, const int32_t* last,
                        int n_frames, int F, int w, void* hip_stream);
/* utils/preprocessing.py:506-517 compute_diff_images: out[t] = x[t] - x[t-1], frame 0 of an utterance = x[1] - x[0] */
int adn_prep_diff_images(const float* in, float* out, int ld, const int32_t* first, const int32_t* last, int n_frames, int D,
                         void* hip_stream);
/* utils/preprocessing.py:260-277 sequencewise_mean_image_subtraction */
int adn_prep_mean_image_subtraction(const float* in, float* out, int ld, const int32_t* starts, const int32_t* lens, int n_utt,
                                    int D, void* hip_stream);
/* utils/preprocessing.py:218-242 normalize_input(centralize=True): per-frame z-normalisation, population std, in place */
int adn_prep_normalize_rows(float* x, int ld, int rows, int cols, void* hip_stream);
/* utils/preprocessing.py:245-257 featurewise_normalize_sequence: column mean and population std of the centred data
 * (fp64 accumulation; workspace: 2*cols doubles) ... */
int adn_prep_column_stats(const float* x, int ld, int rows, int cols, double* workspace, float* mean, float* std,
                          void* hip_stream);
/* ... and their application, also to other splits (runners/3stream.py:102-108) */
int adn_prep_apply_column_norm(const float* x, float* out, int ld, int rows, int cols, const float* mean, const float* std,
                               void* hip_stream);
/* out[r][j] = in[r][perm[j]]: utils/preprocessing.py:492-503 reorder_data as a pixel permutation; coefficient selection */
int adn_prep_gather_columns(const float* in, int ld_in, float* out, int ld_out, const int32_t* perm, int rows, int cols,
                            void* hip_stream);
/* LeCun local contrast normalisation of n_images single-channel H x W images, [n][H][W] contiguous, device pointers
 * (utils/lcn.py:24-61 lecun_lcn, :64-104 make_lecun_lcn): X - blur(X) divided by max(column mean of the local norm, the
 * local norm, threshold).  filter_host: the normalised ksize x ksize filter in HOST memory (utils/lcn.py:9-21
 * gaussian_filter), ksize odd and <= 15; H * W <= 8192. */
int adn_prep_lcn(const float* x, float* y, int n_images, int H, int W, const float* filter_host, int ksize, float threshold,
                 void* hip_stream);

/* ---- minibatch assembly on the GPU (reference utils/datagen.py:92-153,219-229) --------------
 * The splits stay resident in HBM (one frame matrix [sum of lengths][width] per stream, float32 or bfloat16, row-major,
 * contiguous); ONE launch builds what gen_lstm_batch_random + gen_seq_batch_from_idx + the runner's label repeat
 * (runners/3stream.py:360-361) build on the host for one minibatch:
 *   out_s[i, :l]  = frames_s[offsets[u] : offsets[u] + l],  out_s[i, l:] = 0     (u = idxs[i], l = lens[u])
 *   mask[i, :l]   = 1, mask[i, l:] = 0                                           (uint8)
 *   y[i]          = uint8(frame_labels[offsets[u]])                              (datagen.py:130,142: a uint8 array)
 *   targets[i, :] = y[i]                                                         (int32, repeated over all T frames)
 * T is the caller's padding length (the split-wide maximum in the reference, datagen.py:104).  Which utterances form the
 * batch -- the np.random permutation stream, the short last batch and the reshuffle of datagen.py:117-152 -- stays on the
 * host (ip_avsr_amd/utils/datagen_gpu.py); a data-parallel rank passes only its own idxs[rank::world].  All pointers but
 * `streams` are device memory; mask / targets / y may be null.  An index outside [0, n_utt) yields an empty row. */
typedef struct {
    const void* frames;  /* device: the split's frame matrix of this stream */
    int32_t width;       /* features per frame */
    int32_t elem_bytes;  /* 4 = float32, 2 = bfloat16 (what ADN_FLAG_BF16_INPUTS takes) */
    void* out;           /* device: (B, T, width), same element type */
} adn_batch_stream;
int adn_batch_gather(const adn_batch_stream* streams, int n_streams, const int64_t* offsets, const int32_t* lens,
                     const int32_t* frame_labels, int n_utt, const int32_t* idxs, int B, int T, uint8_t* mask, int32_t* targets,
                     uint8_t* y, void* hip_stream);

/* ---- convolutional auto-encoder (reference modelzoo/avletters_convae.py:33-69) ---------------------
 * conv 5x5 (100) - maxpool 2 - conv 5x5 (150) - maxpool 2 pad (1,0) - conv 3x3 (200) - dense - bottleneck, and the
 * tied-weight decoder (transposed dense layers, Deconv2DLayer on the encoder's filters, Upscale2DLayer); ScaledTanh
 * everywhere but the bottleneck / dense8; trained on the mean squared reconstruction error
 * (avletters/avletters_convae.py:254-262).  Images are rows of image_h*image_w floats; parameters are read / written
 * in Lasagne's layouts ((out, in, kh, kw) filters, (c*h*w, units) dense7) in get_all_params order.
 * `variant` selects the --model of avletters/avletters_convae.py:245-252:
 *   ADN_CAE_BATCHNORM  modelzoo/avletters_convae_bn.py:33-74     BatchNormLayers behind both poolings, on the flattened
 *                      conv output (per feature) and behind the dense layer; layer names conv2d4 / conv2d7 / dense10 ...
 *   ADN_CAE_DROPOUT    modelzoo/avletters_convae_drop.py:33-75   DropoutLayers (0.2 on the input, 0.5 behind the poolings,
 *                      the flatten and the dense layer); 125 / 300 / 400 filters; `dense` / `bottleneck` are the widths AS
 *                      BUILT (the reference doubles options['DENSE'] / ['BOTTLENECK'] -- the caller does)
 *   ADN_CAE_BNDROP     modelzoo/avletters_convae_bndrop.py:33-77 both; BatchNorm behind every convolution's nonlinearity
 *                      (per channel) and the dense layer; ScaledTanh(2/3, 1.7159)
 * BatchNorm / dropout follow the model's conventions: non-deterministic passes (adn_cae_compute_grads, adn_cae_loss with
 * ADN_FLAG_STOCHASTIC) use batch statistics, update the running averages and draw masks; deterministic passes
 * (adn_cae_forward, adn_cae_loss, adn_cae_compute_grads with ADN_FLAG_DETERMINISTIC) use the running averages, no masks. */
typedef enum { ADN_CAE_NORMAL = 0, ADN_CAE_BATCHNORM = 1, ADN_CAE_DROPOUT = 2, ADN_CAE_BNDROP = 3 } adn_cae_variant;
typedef struct {
    int32_t image_h, image_w;   /* 30 x 40 in the reference */
    int32_t dense;              /* width of the dense layer (options['DENSE'] = 500) */
    int32_t bottleneck;         /* width of the code (options['BOTTLENECK'] = 50) */
    int32_t precision;          /* adn_precision of the GEMMs */
    int32_t variant;            /* adn_cae_variant */
    int32_t reserved[2];
} adn_cae_config;
typedef struct adn_cae adn_cae;

int adn_cae_create(const adn_cae_config* cfg, adn_cae** out);
void adn_cae_destroy(adn_cae* m);
int adn_cae_set_stream(adn_cae* m, void* hip_stream);
int adn_cae_num_params(const adn_cae* m);
int adn_cae_param_info(const adn_cae* m, int index, adn_param_info_t* info);   /* filters reported as (out, in*kh*kw) */
int adn_cae_read_tensor(adn_cae* m, int buffer /*adn_buffer*/, int index, float* host_dst);
int adn_cae_write_tensor(adn_cae* m, int buffer /*adn_buffer*/, int index, const float* host_src);
int adn_cae_flat_buffer(adn_cae* m, int buffer /*adn_buffer*/, float** ptr, int64_t* floats);
/* recon_fn / the encoder alone: recon (B x image_h*image_w) and / or code (B x bottleneck); either may be null (a null
 * recon skips the decoder) */
int adn_cae_forward(adn_cae* m, const float* x, int B, int flags, float* recon, float* code);
/* mean((recon - target)^2); target null = x (plain auto-encoding) */
int adn_cae_loss(adn_cae* m, const float* x, const float* target, int B, int flags, float* loss);
int adn_cae_compute_grads(adn_cae* m, const float* x, const float* target, int B, int flags, float* loss);
int adn_cae_apply_adadelta(adn_cae* m, float learning_rate, float rho, float epsilon);   /* lr 0.8 in the reference */
int adn_cae_apply_adam(adn_cae* m, float learning_rate);
int adn_cae_synchronize(adn_cae* m);
/* masks of the DropoutLayers = hash(seed, counter, layer 0..4, element index in the layer's (B, C, H, W) tensor); the counter
 * advances by one after every non-deterministic pass (same convention as adn_set_dropout_state) */
int adn_cae_set_dropout_state(adn_cae* m, uint32_t seed, uint32_t counter);

/* ---- RBM / DBN pre-trainer (optional; reference dbn/trainRBM.m, RBMup.m, RBMdown.m, computeStates.m) ----
 * Contrastive divergence CD-1 on one restricted Boltzmann machine: the offline producer of the w1..wN / b1..bN files the
 * encoders start from, without MATLAB.  Layer types are adn_act codes: ADN_ACT_SIGMOID ('sigm'), ADN_ACT_LINEAR ('linear'),
 * ADN_ACT_RECTIFY ('ReLu'), ADN_ACT_TANH, ADN_ACT_LEAKY_RECTIFY (visible layer only: dbn/computeStates.m samples sigm, linear
 * and ReLu units).  The caller picks the learning rates (dbn/trainRBM.m:47-51: the *_linear set as soon as a linear or ReLu
 * layer is involved) and the momentum of each minibatch (0.5, then 0.9 after rbmParams.momentumEpochThres epochs). */
typedef struct {
    int32_t num_vis, num_hid;
    int32_t vis_type, hid_type;   /* adn_act */
    int32_t cd_type;              /* rbmParams.type: 1 = probabilities in the statistics (Hinton's guide), 2 = sampled states */
    int32_t batchsize;            /* rbmParams.batchsize: the statistics are divided by it, also for a short last batch */
    float lr_w, lr_vb, lr_hb;     /* rbmParams.lrW / lrVb / lrHb (or their _linear versions) */
    float weight_penalty;         /* rbmParams.weightPenaltyL2 */
    int32_t reserved[2];
} adn_rbm_config;
typedef struct adn_rbm adn_rbm;
int adn_rbm_create(const adn_rbm_config* cfg, adn_rbm** out);          /* weights and biases start at zero: write W */
void adn_rbm_destroy(adn_rbm* m);
int adn_rbm_set_stream(adn_rbm* m, void* hip_stream);
/* which: 0 = W (num_vis x num_hid, row-major), 1 = hidbiases, 2 = visbiases, 3..5 = their momentum terms */
int adn_rbm_read(adn_rbm* m, int which, float* host_dst);
int adn_rbm_write(adn_rbm* m, int which, const float* host_src);
/* RBMup's activations of n rows (dbn/trainDBN.m:46-47: the next layer's training data) */
int adn_rbm_up(adn_rbm* m, const float* data, int n, int flags, float* probs);
/* one minibatch of dbn/trainRBM.m:98-160; noise = hash(seed, counter, stream, element) as in oracle/rbm_oracle.py; err (may be
 * null) receives sum((data - reconstruction)^2) */
int adn_rbm_train_batch(adn_rbm* m, const float* data, int n, int flags, float momentum, uint32_t seed, uint32_t counter, float* err);

#ifdef __cplusplus
}
#endif
#endif /* ADENET_H_ */
