// Minibatch assembly on the GPU: the padded (B, T, D) tensors, the mask and the per-frame
// targets of ONE minibatch gathered from splits that stay resident in HBM, for every stream in one launch.
//
//   reference utils/datagen.py:92-153  gen_lstm_batch_random   X_batch[i] = [X[start:start+l] ; zeros(Tmax - l)],
//                                                              y_batch[i] = y[start] (uint8), mask[i, :l] = 1
//   reference utils/datagen.py:219-229 gen_seq_batch_from_idx  the same gather for the other streams
//   reference runners/3stream.py:360-361                       y.reshape((-1, 1)).repeat(Tmax, axis=-1)
//
// Byte work, HBM-bound: per frame row D_s * e bytes read (valid frames only) and written (all B * T rows).  One
// workgroup per (8 consecutive frames of an utterance, stream); a row moves in the widest unit its byte length and the buffers' alignment allow
// (16 B per lane for the 1200-wide image streams in either element type).  Which utterances go into the batch is the
// HOST's decision (the permutation stream of np.random stays where the reference has it); the kernel sees an index list.
#include "adn_common.h"
#include <algorithm>

namespace adn {

struct BatchStreams {
    const char* src[ADN_MAX_STREAMS];
    char* dst[ADN_MAX_STREAMS];
    int row_bytes[ADN_MAX_STREAMS];
    int unit[ADN_MAX_STREAMS];          // bytes moved per lane and access: 16, 8, 4 or 2 (what the stream's rows are aligned to)
    int n;
};

constexpr int kRowsPerBlock = 8;        // consecutive frames of one utterance per workgroup: the index -> length -> offset chain of
                                        // three dependent loads is paid once per 8 rows, and a lane keeps several 16-byte loads in flight

// rows [t0, t1) of utterance slot b of one stream: units of sizeof(U) bytes, `upr` per row; valid rows copy, the others are zero
template <typename U>
__device__ __forceinline__ void move_rows(char* __restrict__ dst, const char* __restrict__ src, int row_bytes, int t0, int t1, int L) {
    const int upr = row_bytes / (int)sizeof(U);
    const int total = (t1 - t0) * upr;
    U z;
    __builtin_memset(&z, 0, sizeof(U));
    U* __restrict__ d = reinterpret_cast<U*>(dst);
    const U* __restrict__ p = reinterpret_cast<const U*>(src);
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int r = i / upr;                      // (rows of the batch tensor and of the split are both contiguous: unit i of the
        d[i] = (t0 + r < L) ? p[i] : z;             //  block's first row + i is unit i of the run in either)
    }
}

__global__ __launch_bounds__(256) void batch_gather_kernel(BatchStreams st, const int64_t* __restrict__ offsets,
                                                           const int* __restrict__ lens, const int* __restrict__ frame_labels,
                                                           const int* __restrict__ idxs, int n_utt, int T, int chunks,
                                                           uint8_t* __restrict__ mask, int* __restrict__ targets,
                                                           uint8_t* __restrict__ y) {
    const int b = blockIdx.x / chunks, t0 = (blockIdx.x - b * chunks) * kRowsPerBlock, t1 = min(T, t0 + kRowsPerBlock);
    const int s = blockIdx.y;
    const int u = idxs[b];
    const bool known = (unsigned)u < (unsigned)n_utt;       // an index outside the split gives an empty row, never a wild read
    const int L = known ? lens[u] : 0;
    const int64_t off = known ? offsets[u] : 0;
    const int rb = st.row_bytes[s];
    char* d = st.dst[s] + ((size_t)b * T + t0) * rb;
    const char* p = st.src[s] + (size_t)(off + t0) * rb;    // (only dereferenced for rows t < L)
    switch (st.unit[s]) {                       // uniform over the workgroup
        case 16: move_rows<uint4>(d, p, rb, t0, t1, L); break;
        case 8: move_rows<uint2>(d, p, rb, t0, t1, L); break;
        case 4: move_rows<uint32_t>(d, p, rb, t0, t1, L); break;
        default: move_rows<uint16_t>(d, p, rb, t0, t1, L); break;
    }
    if (s == 0 && (int)threadIdx.x < t1 - t0) {
        const int t = t0 + (int)threadIdx.x, row = b * T + t;
        if (mask) mask[row] = t < L ? 1 : 0;
        if (targets || (y && t == 0)) {
            // y_batch is a uint8 array in the reference (utils/datagen.py:130): labels wrap at 256
            const int lab = (frame_labels && known) ? (frame_labels[off] & 0xFF) : 0;
            if (targets) targets[row] = lab;
            if (y && t == 0) y[b] = (uint8_t)lab;
        }
    }
}

}  // namespace adn

using namespace adn;

extern "C" int adn_batch_gather(const adn_batch_stream* streams, int n_streams, const int64_t* offsets, const int32_t* lens,
                                const int32_t* frame_labels, int n_utt, const int32_t* idxs, int B, int T, uint8_t* mask,
                                int32_t* targets, uint8_t* y, void* hip_stream) {
    ADN_CHECK(streams && n_streams >= 1 && n_streams <= ADN_MAX_STREAMS, ADN_ERR_INVALID, "adn_batch_gather: 1..8 streams");
    ADN_CHECK(offsets && lens && idxs, ADN_ERR_INVALID, "adn_batch_gather: null index vectors");
    ADN_CHECK(n_utt >= 0 && B >= 0 && T >= 1 && (int64_t)B * T < (int64_t)1 << 31, ADN_ERR_INVALID, "adn_batch_gather: bad (B, T)");
    ADN_CHECK(!(targets || y) || frame_labels, ADN_ERR_INVALID, "adn_batch_gather: targets / y asked for without frame_labels");
    if (B == 0) return ADN_OK;
    BatchStreams st{};
    st.n = n_streams;
    int max_units = 0;
    for (int k = 0; k < n_streams; ++k) {
        const adn_batch_stream& q = streams[k];
        ADN_CHECK(q.frames && q.out && q.width >= 1, ADN_ERR_INVALID, "adn_batch_gather: null stream buffer / bad width");
        ADN_CHECK(q.elem_bytes == 4 || q.elem_bytes == 2, ADN_ERR_INVALID, "adn_batch_gather: elem_bytes is 4 (float32) or 2 (bfloat16)");
        ADN_CHECK((int64_t)q.width * q.elem_bytes < (int64_t)1 << 30, ADN_ERR_INVALID, "adn_batch_gather: row too long");
        st.src[k] = static_cast<const char*>(q.frames);
        st.dst[k] = static_cast<char*>(q.out);
        st.row_bytes[k] = q.width * q.elem_bytes;
        // the widest unit every row start of this stream is aligned to, in the split and in the batch
        const uintptr_t bits = reinterpret_cast<uintptr_t>(q.frames) | reinterpret_cast<uintptr_t>(q.out) | (uintptr_t)st.row_bytes[k];
        uintptr_t align = 16;
        while (align > 2 && (bits & (align - 1))) align >>= 1;
        ADN_CHECK(!(bits & (align - 1)), ADN_ERR_INVALID, "adn_batch_gather: buffers must be aligned to their element size");
        st.unit[k] = (int)align;
        max_units = std::max(max_units, st.row_bytes[k] / (int)align);
    }
    const int chunks = cdiv(T, kRowsPerBlock);
    const int threads = max_units * kRowsPerBlock >= 192 ? 256 : (max_units * kRowsPerBlock > 64 ? 128 : 64);
    ADN_CHECK((int64_t)B * chunks < (int64_t)1 << 31, ADN_ERR_INVALID, "adn_batch_gather: batch too large");
    hipLaunchKernelGGL(batch_gather_kernel, dim3(B * chunks, n_streams), dim3(threads), 0, static_cast<hipStream_t>(hip_stream), st, offsets,
                       lens, frame_labels, idxs, n_utt, T, chunks, mask, targets, y);
    ADN_HIP_CHECK(hipGetLastError());
    return ADN_OK;
}
