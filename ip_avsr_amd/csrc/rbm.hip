// RBM / DBN pre-trainer (optional): contrastive divergence CD-1 as the reference's MATLAB code does it
// (dbn/trainRBM.m:98-160, RBMup.m, RBMdown.m, computeActivations.m, computeStates.m) -- the offline producer of the
// w1..wN / b1..bN files the dense encoders are initialised from, so that the pipeline needs no MATLAB.
//
// One minibatch = five GEMMs on the fp32 MFMA kernels (up, down, up; the two correlation products data^T h and v'^T h'),
// column sums for the bias statistics, and two elementwise kernels (activation + sampled state from the pre-activation;
// the momentum / weight-decay update).  The noise is the model's counter-based hash (uniform from 24 bits, normal =
// Box-Muller of two of them, indexed by the element's position in the (batch, units) matrix), the same function as
// oracle/rbm_oracle.py, so that both draw identical samples: parity is exact up to the rounding of exp / log / cos.
#include "adn_common.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

using namespace adn;

namespace {

constexpr int kLayerSigm = ADN_ACT_SIGMOID, kLayerLinear = ADN_ACT_LINEAR, kLayerRelu = ADN_ACT_RECTIFY,
              kLayerTanh = ADN_ACT_TANH, kLayerLeaky = ADN_ACT_LEAKY_RECTIFY;

__device__ __forceinline__ uint32_t rbm_hash24(uint32_t key, uint32_t idx) {       // = elementwise.hip::dropout_keep's mixer
    uint32_t x = idx * 0x9E3779B1u + key;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x >> 8;
}
__host__ __device__ __forceinline__ uint32_t rbm_key(uint32_t seed, uint32_t counter, uint32_t stream) {
    return seed ^ (stream * 0x85EBCA77u) ^ (counter * 0xC2B2AE3Du);
}

// probs = act(x), states = the layer type's sampling rule (dbn/computeStates.m); x [n][ld] in place -> probs, states (optional)
__global__ __launch_bounds__(256) void rbm_act_state_kernel(float* __restrict__ x, float* __restrict__ states, int ld, int n, int units,
                                                            int type, uint32_t key1, uint32_t key2) {
    const int64_t total = (int64_t)n * units;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % units);
        const size_t off = (size_t)(e / units) * ld + c;
        const float v = x[off];
        float p;
        switch (type) {
            case kLayerSigm: p = 1.f / (1.f + expf(-v)); break;
            case kLayerTanh: p = 2.f * (1.f / (1.f + expf(-2.f * v))) - 1.f; break;
            case kLayerRelu: p = fmaxf(0.f, v); break;
            case kLayerLeaky: p = fmaxf(0.01f * v, v); break;
            default: p = v; break;
        }
        x[off] = p;
        if (!states) continue;
        const uint32_t idx = (uint32_t)e;
        if (type == kLayerSigm) {
            const float u = (float)rbm_hash24(key1, idx) * (1.0f / 16777216.0f);
            states[off] = p > u ? 1.f : 0.f;
        } else {
            const float u1 = ((float)rbm_hash24(key1, idx) + 0.5f) * (1.0f / 16777216.0f);
            const float u2 = (float)rbm_hash24(key2, idx) * (1.0f / 16777216.0f);
            const float z = sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
            states[off] = type == kLayerRelu ? fmaxf(0.f, v + z / (1.f + expf(-v))) : p + z;
        }
    }
}

// delta = momentum delta + lr ((pos - neg) / batchsize - l2 w);  w += delta      (dbn/trainRBM.m:147-158)
__global__ __launch_bounds__(256) void rbm_update_kernel(float* __restrict__ w, float* __restrict__ delta, const float* __restrict__ pos,
                                                         const float* __restrict__ neg, int64_t n, float inv_bs, float lr, float momentum, float l2) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = momentum * delta[i] + lr * ((pos[i] - neg[i]) * inv_bs - l2 * w[i]);
        delta[i] = d;
        w[i] += d;
    }
}

// acc[0] += sum (a - b)^2 over an [n][cols] block
__global__ __launch_bounds__(256) void rbm_sqerr_kernel(const float* __restrict__ a, const float* __restrict__ b, int ld, int n, int cols,
                                                        float* __restrict__ acc) {
    __shared__ float red[256];
    float s = 0.f;
    const int64_t total = (int64_t)n * cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const size_t off = (size_t)(e / cols) * ld + (e % cols);
        const float d = a[off] - b[off];
        s += d * d;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) atomicAdd(acc, red[0]);
}

int grid_of(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096)); }

bool known_type(int t) { return t == kLayerSigm || t == kLayerLinear || t == kLayerRelu || t == kLayerTanh || t == kLayerLeaky; }
bool can_sample(int t) { return t == kLayerSigm || t == kLayerLinear || t == kLayerRelu; }

}  // namespace

struct adn_rbm {
    adn_rbm_config cfg;
    hipStream_t stream = nullptr;
    int V = 0, H = 0, ldv = 0, ldh = 0;
    float *W = nullptr, *dW = nullptr, *hb = nullptr, *vb = nullptr, *dhb = nullptr, *dvb = nullptr;      // W [V][ldh]
    // per-batch work space (grown on demand): data, hidden probabilities / states, reconstruction, its hidden probabilities,
    // the two correlation matrices, the four activity vectors, the error accumulator
    char* slab = nullptr; size_t slab_bytes = 0; int cap = 0;
    float *data = nullptr, *hp = nullptr, *hs = nullptr, *nv = nullptr, *nvs = nullptr, *nh = nullptr, *pos = nullptr, *neg = nullptr,
          *act = nullptr, *err = nullptr;
};

namespace {

int ensure(adn_rbm* m, int n) {
    if (n <= m->cap) return ADN_OK;
    const size_t V = m->V, H = m->H;
    size_t need = 0;
    auto take = [&](size_t floats) { const size_t o = need; need += (size_t)round_up((int64_t)floats * 4, 256); return o; };
    const size_t o_data = take((size_t)n * m->ldv), o_hp = take((size_t)n * m->ldh), o_hs = take((size_t)n * m->ldh),
                 o_nv = take((size_t)n * m->ldv), o_nvs = take((size_t)n * m->ldv), o_nh = take((size_t)n * m->ldh),
                 o_pos = take(V * m->ldh), o_neg = take(V * m->ldh), o_act = take(2 * (size_t)(m->ldv + m->ldh)), o_err = take(8);
    if (m->slab) { ADN_HIP_CHECK(hipStreamSynchronize(m->stream)); ADN_HIP_CHECK(hipFree(m->slab)); m->slab = nullptr; }
    ADN_HIP_CHECK(hipMalloc((void**)&m->slab, need));
    ADN_HIP_CHECK(hipMemsetAsync(m->slab, 0, need, m->stream));
    auto at = [&](size_t o) { return reinterpret_cast<float*>(m->slab + o); };
    m->data = at(o_data); m->hp = at(o_hp); m->hs = at(o_hs); m->nv = at(o_nv); m->nvs = at(o_nvs); m->nh = at(o_nh);
    m->pos = at(o_pos); m->neg = at(o_neg); m->act = at(o_act); m->err = at(o_err);
    m->slab_bytes = need; m->cap = n;
    (void)H;
    return ADN_OK;
}

int mm(adn_rbm* m, int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
       const float* bias) {
    GemmArgs g;
    g.layout = layout; g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.bias = bias;
    g.no_split = 1;
    return gemm(g, m->stream);
}

// pre-activation in `x` -> probabilities in place (+ sampled states)
int act_state(adn_rbm* m, float* x, float* states, int ld, int n, int units, int type, uint32_t seed, uint32_t counter, uint32_t stream_id) {
    hipLaunchKernelGGL(rbm_act_state_kernel, dim3(grid_of((int64_t)n * units)), dim3(256), 0, m->stream, x, states, ld, n, units, type,
                       rbm_key(seed, counter, stream_id), rbm_key(seed, counter, stream_id + 1));
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}

int stage(adn_rbm* m, const float* data, int n, int flags) {
    ADN_TRY(ensure(m, n));
    const hipMemcpyKind kind = (flags & ADN_FLAG_DEVICE_INPUTS) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    ADN_HIP_CHECK(hipMemcpy2DAsync(m->data, (size_t)m->ldv * 4, data, (size_t)m->V * 4, (size_t)m->V * 4, n, kind, m->stream));
    return ADN_OK;
}

}  // namespace

extern "C" {

int adn_rbm_create(const adn_rbm_config* cfg, adn_rbm** out) {
    ADN_CHECK(cfg && out, ADN_ERR_INVALID, "null argument");
    ADN_CHECK(cfg->num_vis >= 1 && cfg->num_hid >= 1 && cfg->num_vis <= (1 << 20) && cfg->num_hid <= (1 << 20), ADN_ERR_INVALID,
              "layer sizes out of range");
    ADN_CHECK(known_type(cfg->vis_type) && known_type(cfg->hid_type), ADN_ERR_INVALID, "unknown layer type");
    ADN_CHECK(can_sample(cfg->hid_type), ADN_ERR_INVALID,
              "the hidden layer must be sigm, linear or ReLu (dbn/computeStates.m has no sampling rule for the others)");
    ADN_CHECK(cfg->cd_type == 1 || (cfg->cd_type == 2 && can_sample(cfg->vis_type)), ADN_ERR_INVALID,
              "rbmParams.type must be 1 or 2 (2 samples the visible layer: sigm, linear or ReLu)");
    ADN_CHECK(cfg->batchsize >= 1, ADN_ERR_INVALID, "batch size must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device visible"); return ADN_ERR_NO_DEVICE; }
    adn_rbm* m = new adn_rbm();
    m->cfg = *cfg; m->V = cfg->num_vis; m->H = cfg->num_hid;
    m->ldv = (int)round_up(m->V, 64); m->ldh = (int)round_up(m->H, 64);
    const size_t wf = (size_t)m->V * m->ldh;
    float** bufs[6] = {&m->W, &m->dW, &m->hb, &m->dhb, &m->vb, &m->dvb};
    const size_t sizes[6] = {wf, wf, (size_t)m->ldh, (size_t)m->ldh, (size_t)m->ldv, (size_t)m->ldv};
    for (int k = 0; k < 6; ++k)
        if (hipMalloc((void**)bufs[k], sizes[k] * 4) != hipSuccess || hipMemset(*bufs[k], 0, sizes[k] * 4) != hipSuccess) {
            set_error("allocating the RBM buffers failed"); adn_rbm_destroy(m); return ADN_ERR_HIP;
        }
    *out = m;
    return ADN_OK;
}

void adn_rbm_destroy(adn_rbm* m) {
    if (!m) return;
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    for (float* p : {m->W, m->dW, m->hb, m->dhb, m->vb, m->dvb}) if (p) (void)hipFree(p);
    if (m->slab) (void)hipFree(m->slab);
    delete m;
}

int adn_rbm_set_stream(adn_rbm* m, void* hip_stream) {
    ADN_CHECK(m, ADN_ERR_INVALID, "null model");
    m->stream = static_cast<hipStream_t>(hip_stream);
    return ADN_OK;
}

// which: 0 = W (num_vis x num_hid), 1 = hidbiases, 2 = visbiases, 3..5 = their momentum terms (deltaW, deltaHidbias, deltaVisbias)
static int rbm_tensor(adn_rbm* m, int which, float** ptr, int* rows, int* cols, int* ld) {
    ADN_CHECK(m && which >= 0 && which < 6, ADN_ERR_INVALID, "bad tensor id");
    float* p[6] = {m->W, m->hb, m->vb, m->dW, m->dhb, m->dvb};
    *ptr = p[which];
    const int k = which % 3;
    *rows = k == 0 ? m->V : 1; *cols = k == 2 ? m->V : m->H; *ld = k == 2 ? m->ldv : m->ldh;
    return ADN_OK;
}

int adn_rbm_read(adn_rbm* m, int which, float* host_dst) {
    ADN_CHECK(host_dst, ADN_ERR_INVALID, "null argument");
    float* p; int rows, cols, ld;
    ADN_TRY(rbm_tensor(m, which, &p, &rows, &cols, &ld));
    ADN_HIP_CHECK(hipStreamSynchronize(m->stream));
    ADN_HIP_CHECK(hipMemcpy2D(host_dst, (size_t)cols * 4, p, (size_t)ld * 4, (size_t)cols * 4, rows, hipMemcpyDeviceToHost));
    return ADN_OK;
}

int adn_rbm_write(adn_rbm* m, int which, const float* host_src) {
    ADN_CHECK(host_src, ADN_ERR_INVALID, "null argument");
    float* p; int rows, cols, ld;
    ADN_TRY(rbm_tensor(m, which, &p, &rows, &cols, &ld));
    ADN_HIP_CHECK(hipStreamSynchronize(m->stream));
    ADN_HIP_CHECK(hipMemcpy2D(p, (size_t)ld * 4, host_src, (size_t)cols * 4, (size_t)cols * 4, rows, hipMemcpyHostToDevice));
    return ADN_OK;
}

int adn_rbm_up(adn_rbm* m, const float* data, int n, int flags, float* probs) {
    ADN_CHECK(m && data && probs && n >= 1, ADN_ERR_INVALID, "bad argument");
    ADN_TRY(stage(m, data, n, flags));
    ADN_TRY(mm(m, GEMM_NN, n, m->H, m->V, m->data, m->ldv, m->W, m->ldh, m->hp, m->ldh, m->hb));
    ADN_TRY(act_state(m, m->hp, nullptr, m->ldh, n, m->H, m->cfg.hid_type, 0, 0, 0));
    const hipMemcpyKind kind = (flags & ADN_FLAG_DEVICE_OUTPUTS) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    ADN_HIP_CHECK(hipMemcpy2DAsync(probs, (size_t)m->H * 4, m->hp, (size_t)m->ldh * 4, (size_t)m->H * 4, n, kind, m->stream));
    if (!(flags & ADN_FLAG_DEVICE_OUTPUTS)) ADN_HIP_CHECK(hipStreamSynchronize(m->stream));
    return ADN_OK;
}

int adn_rbm_train_batch(adn_rbm* m, const float* data, int n, int flags, float momentum, uint32_t seed, uint32_t counter, float* err) {
    ADN_CHECK(m && data && n >= 1, ADN_ERR_INVALID, "bad argument");
    const adn_rbm_config& c = m->cfg;
    hipStream_t s = m->stream;
    ADN_TRY(stage(m, data, n, flags));
    const int V = m->V, H = m->H, ldv = m->ldv, ldh = m->ldh;
    float *posvis = m->act, *negvis = m->act + ldv, *poshid = m->act + 2 * ldv, *neghid = m->act + 2 * ldv + ldh;
    // positive phase (dbn/trainRBM.m:98-117)
    ADN_TRY(mm(m, GEMM_NN, n, H, V, m->data, ldv, m->W, ldh, m->hp, ldh, m->hb));
    ADN_TRY(act_state(m, m->hp, m->hs, ldh, n, H, c.hid_type, seed, counter, 0));
    const float* posh = c.cd_type == 1 ? m->hp : m->hs;
    ADN_TRY(mm(m, GEMM_TN, V, H, n, m->data, ldv, posh, ldh, m->pos, ldh, nullptr));
    ADN_TRY(col_sum(posh, ldh, n, H, poshid, 0, s));
    ADN_TRY(col_sum(m->data, ldv, n, V, posvis, 0, s));
    // negative phase (:120-143): down from the hidden STATES, up again from the reconstruction
    ADN_TRY(mm(m, GEMM_NT, n, V, H, m->hs, ldh, m->W, ldh, m->nv, ldv, m->vb));
    ADN_TRY(act_state(m, m->nv, c.cd_type == 2 ? m->nvs : nullptr, ldv, n, V, c.vis_type, seed, counter, 2));
    const float* negv = c.cd_type == 1 ? m->nv : m->nvs;
    ADN_TRY(mm(m, GEMM_NN, n, H, V, negv, ldv, m->W, ldh, m->nh, ldh, m->hb));
    ADN_TRY(act_state(m, m->nh, nullptr, ldh, n, H, c.hid_type, 0, 0, 0));
    ADN_TRY(mm(m, GEMM_TN, V, H, n, negv, ldv, m->nh, ldh, m->neg, ldh, nullptr));
    ADN_TRY(col_sum(negv, ldv, n, V, negvis, 0, s));
    ADN_TRY(col_sum(m->nh, ldh, n, H, neghid, 0, s));
    if (err) {
        ADN_HIP_CHECK(hipMemsetAsync(m->err, 0, sizeof(float), s));
        hipLaunchKernelGGL(rbm_sqerr_kernel, dim3(grid_of((int64_t)n * V)), dim3(256), 0, s, m->data, negv, ldv, n, V, m->err);
    }
    // updates (:147-158): the reference divides by the NOMINAL batch size, also for a short last batch
    const float inv_bs = 1.f / (float)c.batchsize;
    hipLaunchKernelGGL(rbm_update_kernel, dim3(grid_of((int64_t)V * ldh)), dim3(256), 0, s, m->W, m->dW, m->pos, m->neg, (int64_t)V * ldh,
                       inv_bs, c.lr_w, momentum, c.weight_penalty);
    hipLaunchKernelGGL(rbm_update_kernel, dim3(grid_of(V)), dim3(256), 0, s, m->vb, m->dvb, posvis, negvis, (int64_t)V, inv_bs, c.lr_vb,
                       momentum, 0.f);
    hipLaunchKernelGGL(rbm_update_kernel, dim3(grid_of(H)), dim3(256), 0, s, m->hb, m->dhb, poshid, neghid, (int64_t)H, inv_bs, c.lr_hb,
                       momentum, 0.f);
    ADN_HIP_CHECK(hipGetLastError());
    if (err) {
        ADN_HIP_CHECK(hipMemcpyAsync(err, m->err, sizeof(float), hipMemcpyDeviceToHost, s));
        ADN_HIP_CHECK(hipStreamSynchronize(s));
    }
    return ADN_OK;
}

}  // extern "C"
