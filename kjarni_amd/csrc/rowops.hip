// HBM-bound row kernels of the encoder path: embedding gather + LayerNorm,
// LayerNorm, pooling (+L2), the small classifier GEMV and the probability
// softmax / sigmoid.  One wave64 per row, 16-byte loads, wave-shuffle
// reductions; rows stay in registers between the statistics and the affine
// pass so every byte is read once.
#include <algorithm>
#include <cstdint>

#include "device_utils.h"
#include "kernels.h"

namespace kjarni {

namespace {

constexpr int MAX_V4_PER_LANE = 4;  // hidden <= 1024 on the float4 path

// ---------------------------------------------------------------------------
// LayerNorm core on a row held as float4 registers.
// (x-mean)/sqrt(var+eps)*gamma+beta with the population variance and eps inside
// the sqrt: crates/kjarni-transformers/src/cpu/normalization/layer_norm.rs
// :96-131 (scalar), :37-93 (AVX2), :203-215 (ndarray alloc path).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void ln_row_v4(f32x4 (&x)[MAX_V4_PER_LANE], int nv4, int lane,
                                          const float* __restrict__ gamma,
                                          const float* __restrict__ beta, float eps, int hidden,
                                          float* __restrict__ out_row)
{
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < MAX_V4_PER_LANE; ++i)
        if (lane + i * 64 < nv4) s += (x[i][0] + x[i][1]) + (x[i][2] + x[i][3]);
    const float mean = wave_sum(s) / (float)hidden;
    float v = 0.0f;
#pragma unroll
    for (int i = 0; i < MAX_V4_PER_LANE; ++i)
        if (lane + i * 64 < nv4) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float d = x[i][c] - mean;
                v = fmaf(d, d, v);
            }
        }
    const float var = wave_sum(v) / (float)hidden;
    const float inv_std = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < MAX_V4_PER_LANE; ++i) {
        const int c4 = lane + i * 64;
        if (c4 < nv4) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c4 * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c4 * 4);
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = (x[i][c] - mean) * inv_std * g[c] + b[c];
            *reinterpret_cast<f32x4*>(out_row + c4 * 4) = o;
        }
    }
}

// Embeddings::forward (cpu/embeddings/mod.rs:181-225) fused with embed_norm
// (cpu/encoder/transformer_encoder.rs:303-305):
//   h = W_word[id] (zeros when id >= vocab, mod.rs:232-236) [* sqrt(H)]
//       + P[offset+s] (when offset+s < max_pos, mod.rs:198-212)
//       + T[type] (row 0 when type_ids is null, mod.rs:214-223); then LayerNorm.
// A type id >= type_vocab panics in the reference; here it is clamped (the
// host validates type ids before launch).
__global__ __launch_bounds__(256) void embed_layernorm_kernel(
    const uint32_t* __restrict__ ids, const uint32_t* __restrict__ type_ids,
    const float* __restrict__ word, const float* __restrict__ pos, const float* __restrict__ type,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int64_t tokens,
    int seq, int hidden, int vocab, int max_pos, int type_vocab, int pos_offset, int scale_embeddings,
    const int32_t* __restrict__ tok_src, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)xcd_contiguous(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);  // (the XCDs' runs of rows as the projections deal them)
    if (t >= tokens) return;
    const int nv4 = hidden >> 2;
    // packed rows (ragged batches): output row t is the token at index tok_src[t] of the padded [batch, seq] arrays
    const int64_t src = tok_src ? (int64_t)tok_src[t] : t;
    const int s = (int)(src % seq);
    const uint32_t id = ids[src];
    const bool has_word = id < (uint32_t)vocab;
    const bool has_pos = pos != nullptr && (pos_offset + s) < max_pos;
    const bool has_type = type != nullptr && type_vocab > 0;
    uint32_t ty = (has_type && type_ids) ? type_ids[src] : 0u;
    if (has_type && ty >= (uint32_t)type_vocab) ty = (uint32_t)type_vocab - 1;
    const float* wrow = word + (int64_t)id * hidden;
    const float* prow = pos + (int64_t)(pos_offset + s) * hidden;
    const float* trow = type + (int64_t)ty * hidden;
    const float sc = sqrtf((float)hidden);

    f32x4 x[MAX_V4_PER_LANE];
#pragma unroll
    for (int i = 0; i < MAX_V4_PER_LANE; ++i) {
        const int c4 = lane + i * 64;
        x[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c4 < nv4) {
            if (has_word) x[i] = *reinterpret_cast<const f32x4*>(wrow + c4 * 4);
            if (scale_embeddings) x[i] *= sc;
            if (has_pos) x[i] += *reinterpret_cast<const f32x4*>(prow + c4 * 4);
            if (has_type) x[i] += *reinterpret_cast<const f32x4*>(trow + c4 * 4);
        }
    }
    float* orow = out + t * hidden;
    if (gamma != nullptr) {
        ln_row_v4(x, nv4, lane, gamma, beta, eps, hidden, orow);
    } else {
#pragma unroll
        for (int i = 0; i < MAX_V4_PER_LANE; ++i) {
            const int c4 = lane + i * 64;
            if (c4 < nv4) *reinterpret_cast<f32x4*>(orow + c4 * 4) = x[i];
        }
    }
}

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ in,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        int64_t rows, int hidden,
                                                        float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)xcd_contiguous(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);  // (the XCDs' runs of rows as the projections deal them)
    if (t >= rows) return;
    const int nv4 = hidden >> 2;
    const float* irow = in + t * hidden;
    f32x4 x[MAX_V4_PER_LANE];
#pragma unroll
    for (int i = 0; i < MAX_V4_PER_LANE; ++i) {
        const int c4 = lane + i * 64;
        x[i] = (c4 < nv4) ? *reinterpret_cast<const f32x4*>(irow + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    ln_row_v4(x, nv4, lane, gamma, beta, eps, hidden, out + t * hidden);
}

// Any-width fallback (hidden not a multiple of 4 or > 1024): one wave per row,
// three passes over global memory.  Not on the BERT-family hot path.
__global__ __launch_bounds__(256) void layernorm_generic_kernel(
    const float* __restrict__ in, const float* __restrict__ gamma, const float* __restrict__ beta,
    float eps, int64_t rows, int hidden, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= rows) return;
    const float* irow = in + t * hidden;
    float s = 0.f;
    for (int i = lane; i < hidden; i += 64) s += irow[i];
    const float mean = wave_sum(s) / (float)hidden;
    float v = 0.f;
    for (int i = lane; i < hidden; i += 64) {
        const float d = irow[i] - mean;
        v = fmaf(d, d, v);
    }
    const float inv_std = 1.0f / sqrtf(wave_sum(v) / (float)hidden + eps);
    float* orow = out + t * hidden;
    for (int i = lane; i < hidden; i += 64) {
        const float xv = irow[i];
        orow[i] = (xv - mean) * inv_std * gamma[i] + beta[i];
    }
}

__global__ __launch_bounds__(256) void embed_generic_kernel(
    const uint32_t* __restrict__ ids, const uint32_t* __restrict__ type_ids,
    const float* __restrict__ word, const float* __restrict__ pos, const float* __restrict__ type,
    int64_t tokens, int seq, int hidden, int vocab, int max_pos, int type_vocab, int pos_offset,
    int scale_embeddings, const int32_t* __restrict__ tok_src, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= tokens) return;
    const int64_t src = tok_src ? (int64_t)tok_src[t] : t;
    const int s = (int)(src % seq);
    const uint32_t id = ids[src];
    const bool has_word = id < (uint32_t)vocab;
    const bool has_pos = pos != nullptr && (pos_offset + s) < max_pos;
    const bool has_type = type != nullptr && type_vocab > 0;
    uint32_t ty = (has_type && type_ids) ? type_ids[src] : 0u;
    if (has_type && ty >= (uint32_t)type_vocab) ty = (uint32_t)type_vocab - 1;
    const float sc = sqrtf((float)hidden);
    for (int i = lane; i < hidden; i += 64) {
        float x = has_word ? word[(int64_t)id * hidden + i] : 0.0f;
        if (scale_embeddings) x *= sc;
        if (has_pos) x += pos[(int64_t)(pos_offset + s) * hidden + i];
        if (has_type) x += type[(int64_t)ty * hidden + i];
        out[t * hidden + i] = x;
    }
}

// ---------------------------------------------------------------------------
// Pooling + L2: pooling/mod.rs:11-68 and cpu/encoder/traits.rs:529-536.
// One block per sentence; thread t owns columns t, t+256, ... so every load of
// a hidden row is coalesced.  mask is the u32 attention mask (traits.rs:71
// converts it with `as f32`).
// ---------------------------------------------------------------------------
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void pool_kernel(const float* __restrict__ hs,
                                                   const uint32_t* __restrict__ mask, int seq,
                                                   int hidden, int normalize, const int32_t* __restrict__ cu,
                                                   float* __restrict__ out)
{
    constexpr int NW = THREADS / 64;
    __shared__ float red[NW];
    __shared__ int red_last[NW];
    __shared__ f32x4 part[THREADS];
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    // packed rows: sentence b is rows cu[b] .. cu[b+1] of hs, every one of them a kept token
    if (cu) seq = cu[b + 1] - cu[b];
    const float* base = hs + (cu ? (int64_t)cu[b] : b * seq) * (int64_t)hidden;
    const uint32_t* mrow = (mask && !cu) ? mask + b * seq : nullptr;

    // count of kept tokens and the last kept position (pooling/mod.rs:61-62 rposition(x > 0), else 0), all threads
    // at once (mask values are 0 / 1: the float sum is exact in any order)
    float cnt = 0.0f;
    int last = 0;
    for (int s = tid; s < seq; s += THREADS) {
        const float mv = mrow ? (float)mrow[s] : 1.0f;
        cnt += mv;
        if (mv > 0.0f) last = s;
    }
    cnt = wave_sum(cnt);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) last = max(last, __shfl_xor(last, off, kWave));
    if ((tid & 63) == 0) {
        red[tid >> 6] = cnt;
        red_last[tid >> 6] = last;
    }
    __syncthreads();
    cnt = 0.0f;
    last = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {   // (mask values are 0 / 1: exact in any order)
        cnt += red[w];
        last = max(last, red_last[w]);
    }
    __syncthreads();

    const int nv4 = hidden >> 2;
    if (MODE == POOL_MEAN && (hidden & 3) == 0 && nv4 <= 256) {
        // 16-byte columns; with hidden = 384 a 256-thread block holds two row groups (192 lanes busy), a 1024-thread block
        // ten, each summing every groups-th row; the groups' partial sums meet in LDS and are added in group order.
        const int groups = THREADS / nv4;
        const int c4 = tid % nv4, g = tid / nv4;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (g < groups) {
            if (cnt == 0.0f) {
                if (g == 0) a = *reinterpret_cast<const f32x4*>(base + c4 * 4);  // pooling/mod.rs:25-27: token 0
            } else {
                // eight rows requested before the first is added (the adds keep their order): a lone sentence is a chain
                // of memory round trips otherwise (31 us for 128 tokens)
                int s = g;
                for (; s + 7 * groups < seq; s += 8 * groups) {
                    f32x4 v[8];
                    float mv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        v[j] = *reinterpret_cast<const f32x4*>(base + (int64_t)(s + j * groups) * hidden + c4 * 4);
                        mv[j] = mrow ? (float)mrow[s + j * groups] : 1.0f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) a += v[j] * mv[j];
                }
                for (; s < seq; s += groups) {
                    const float mv = mrow ? (float)mrow[s] : 1.0f;
                    a += *reinterpret_cast<const f32x4*>(base + (int64_t)s * hidden + c4 * 4) * mv;
                }
            }
            part[tid] = a;
        }
        __syncthreads();
        float sq = 0.0f;
        if (g == 0) {
            for (int k = 1; k < groups; ++k) a += part[k * nv4 + c4];
            if (cnt != 0.0f) a = a / cnt;
            sq = (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
        }
        if (normalize) {
            sq = wave_sum(sq);
            if ((tid & 63) == 0) red[tid >> 6] = sq;
            __syncthreads();
            // (only group 0 holds non-zero squares: waves 0 .. ceil(nv4 / 64) - 1; the first four cover hidden <= 1024)
            const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
            if (norm > 0.0f) a = a / norm;  // traits.rs:529-536: divide only when the norm is > 0
        }
        if (g == 0) *reinterpret_cast<f32x4*>(out + b * hidden + c4 * 4) = a;
        return;
    }

    constexpr int MAXC = 1024 / THREADS;  // hidden <= 1024
    float acc[MAXC];
    float sq = 0.0f;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
        const int c = tid + j * THREADS;
        acc[j] = 0.0f;
        if (c >= hidden) continue;
        if (MODE == POOL_MEAN) {
            if (cnt == 0.0f) {
                acc[j] = base[c];  // pooling/mod.rs:25-27: all-masked row -> token 0
            } else {
                float a = 0.0f;
                for (int s = 0; s < seq; ++s) {
                    const float mv = mrow ? (float)mrow[s] : 1.0f;
                    a += base[(int64_t)s * hidden + c] * mv;
                }
                acc[j] = a / cnt;
            }
        } else if (MODE == POOL_CLS) {
            acc[j] = base[c];
        } else if (MODE == POOL_MAX) {
            float a = -1e9f;  // MASK_VALUE, pooling/mod.rs:41-52
            for (int s = 0; s < seq; ++s) {
                const bool masked = mrow && mrow[s] == 0u;
                const float xv = masked ? -1e9f : base[(int64_t)s * hidden + c];
                a = fmaxf(a, xv);
            }
            acc[j] = a;
        } else {
            acc[j] = base[(int64_t)last * hidden + c];
        }
        sq = fmaf(acc[j], acc[j], sq);
    }
    if (normalize) {
        sq = wave_sum(sq);
        if ((tid & 63) == 0) red[tid >> 6] = sq;
        __syncthreads();
        float n2 = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; w += 4) n2 += (red[w] + red[w + 1]) + (red[w + 2] + red[w + 3]);
        const float norm = sqrtf(n2);
        // traits.rs:529-536: divide only when the norm is > 0
        if (norm > 0.0f) {
#pragma unroll
            for (int j = 0; j < MAXC; ++j) acc[j] = acc[j] / norm;
        }
    }
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
        const int c = tid + j * THREADS;
        if (c < hidden) out[b * hidden + c] = acc[j];
    }
}

// logits[r, n] = feat[r,:] . w[n,:] + bias[n]: one wave per output element
// (num_labels is 1 or 2 for the rerank / sentiment heads:
// cpu/encoder/classifier.rs:258).
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ feat, int64_t ld,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           int64_t rows, int k, int n,
                                                           float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= rows * n) return;
    const int64_t r = o / n;
    const int j = (int)(o % n);
    const float* x = feat + r * ld;
    const float* wr = w + (int64_t)j * k;
    float s = 0.0f;
    for (int i = lane; i < k; i += 64) s = fmaf(x[i], wr[i], s);
    s = wave_sum(s);
    if (lane == 0) out[o] = s + (bias ? bias[j] : 0.0f);
}

// Classifier probabilities: softmax_inplace (activations.rs:223-242) per row, or
// sigmoid for multi-label (crates/kjarni/src/classifier/model.rs:529).
__global__ __launch_bounds__(256) void row_softmax_kernel(const float* __restrict__ in, int64_t rows,
                                                          int n, int sigmoid,
                                                          float* __restrict__ out)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float* x = in + r * n;
    float* o = out + r * n;
    if (sigmoid) {
        for (int i = 0; i < n; ++i) o[i] = 1.0f / (1.0f + expf(-x[i]));
        return;
    }
    float mx = -INFINITY;
    for (int i = 0; i < n; ++i) mx = fmaxf(mx, x[i]);
    float sum = 0.0f;
    for (int i = 0; i < n; ++i) {
        const float e = expf(x[i] - mx);
        o[i] = e;
        sum += e;
    }
    if (sum > 0.0f) {
        const float scale = 1.0f / sum;
        for (int i = 0; i < n; ++i) o[i] *= scale;
    }
}

// RoPE over the Q and K thirds of [tokens, 3*hidden] (rope/mod.rs:148-170): V = 4 rotates four neighbouring pairs per
// thread with 16-byte accesses, V = 1 is the any-shape form.
template <int V>
__global__ __launch_bounds__(256) void rope_qk_kernel(float* __restrict__ qkv, const float* __restrict__ cos_t,
                                                      const float* __restrict__ sin_t, int64_t tokens, int seq, int heads,
                                                      int head_dim, const int32_t* __restrict__ tok_src)
{
    const int half = head_dim >> 1, hv = half / V, hidden = heads * head_dim;
    const int64_t per_tok = (int64_t)2 * heads * hv, total = tokens * per_tok;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t t = idx / per_tok;
        int item = (int)(idx - t * per_tok);
        const int part = item / (heads * hv);
        item -= part * heads * hv;
        const int h = item / hv, i = (item - h * hv) * V;
        const int s = (int)((tok_src ? (int64_t)tok_src[t] : t) % seq);
        float* r = qkv + t * 3 * hidden + part * hidden + h * head_dim + i;
        const float* c = cos_t + (int64_t)s * head_dim + i;
        const float* sn = sin_t + (int64_t)s * head_dim + i;
        if (V == 4) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(r), x1 = *reinterpret_cast<const f32x4*>(r + half);
            const f32x4 cv = *reinterpret_cast<const f32x4*>(c), sv = *reinterpret_cast<const f32x4*>(sn);
            f32x4 y0, y1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // two roundings per product-difference, as the scalar reference
                y0[k] = __fsub_rn(__fmul_rn(x0[k], cv[k]), __fmul_rn(x1[k], sv[k]));
                y1[k] = __fadd_rn(__fmul_rn(x0[k], sv[k]), __fmul_rn(x1[k], cv[k]));
            }
            *reinterpret_cast<f32x4*>(r) = y0;
            *reinterpret_cast<f32x4*>(r + half) = y1;
        } else {
            const float x0 = r[0], x1 = r[half];
            r[0] = __fsub_rn(__fmul_rn(x0, c[0]), __fmul_rn(x1, sn[0]));
            r[half] = __fadd_rn(__fmul_rn(x0, sn[0]), __fmul_rn(x1, c[0]));
        }
    }
}

// ---------------------------------------------------------------------------
// Ragged batches (BatchLongest padding, pipeline/encoder/loader.rs:98-115): the forward pass can run over the KEPT
// tokens only -- a padded key contributes exactly 0 to every softmax row (utils/masks.rs:4-36 overwrites its score with
// -1e9 / -inf) and pooling skips padded rows (pooling/mod.rs:11-33) -- so the projections never see [PAD] rows.
//
// mask_lengths_kernel: lens[b] = number of kept tokens of sentence b; bit 31 set when the packed layout cannot stand
// in for the padded one: mask[b, 0] == 0 (CLS pooling and the classification head read token 0; an all-masked sentence
// pools token 0 and, with the -1e9 fill, attends uniformly to its padding) or a mask value other than 0 / 1 (the mean
// pool multiplies by the mask value, traits.rs:71 `as f32`).
// pack_index_kernel: tok_src[cu[b] + r] = b * seq + s for the r-th kept token (ascending s) of sentence b.
// One wave per sentence, ballots over 64 tokens at a time.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_lengths_kernel(const uint32_t* __restrict__ mask, int64_t batch, int seq,
                                                           uint32_t* __restrict__ lens)
{
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= batch) return;
    const uint32_t* row = mask + b * seq;
    int count = 0;
    bool bad = false;
    for (int s0 = 0; s0 < seq; s0 += 64) {
        const int s = s0 + lane;
        const uint32_t mv = s < seq ? row[s] : 0u;
        bad = bad || mv > 1u || (s == 0 && mv == 0u);
        count += __popcll(__ballot(mv != 0u));
    }
    bad = __ballot(bad) != 0ull;
    if (lane == 0) lens[b] = (uint32_t)count | (bad ? 0x80000000u : 0u);
}

__global__ __launch_bounds__(256) void pack_index_kernel(const uint32_t* __restrict__ mask, const int32_t* __restrict__ cu,
                                                         int64_t batch, int seq, int32_t* __restrict__ tok_src)
{
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= batch) return;
    const uint32_t* row = mask + b * seq;
    int32_t* dst = tok_src + cu[b];
    int done = 0;
    for (int s0 = 0; s0 < seq; s0 += 64) {
        const int s = s0 + lane;
        const bool keep = s < seq && row[s] != 0u;
        const unsigned long long bits = __ballot(keep);
        if (keep) dst[done + __popcll(bits & ((1ull << lane) - 1ull))] = (int32_t)(b * seq + s);
        done += __popcll(bits);
    }
}

inline unsigned rows_to_blocks(int64_t rows) { return (unsigned)((rows + 3) / 4); }

}  // namespace

hipError_t launch_mask_lengths(const uint32_t* mask, int64_t batch, int seq, uint32_t* lens, hipStream_t stream)
{
    if (batch <= 0) return hipSuccess;
    hipLaunchKernelGGL(mask_lengths_kernel, dim3(rows_to_blocks(batch)), dim3(256), 0, stream, mask, batch, seq, lens);
    return hipGetLastError();
}

hipError_t launch_pack_index(const uint32_t* mask, const int32_t* cu, int64_t batch, int seq, int32_t* tok_src,
                             hipStream_t stream)
{
    if (batch <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_index_kernel, dim3(rows_to_blocks(batch)), dim3(256), 0, stream, mask, cu, batch, seq, tok_src);
    return hipGetLastError();
}

hipError_t launch_rope_qk(float* qkv, const float* cos_t, const float* sin_t, int64_t tokens, int seq, int heads,
                          int head_dim, hipStream_t stream, const int32_t* tok_src)
{
    if (tokens <= 0) return hipSuccess;
    if (head_dim < 2 || (head_dim & 1) || seq <= 0) return hipErrorInvalidValue;
    const int half = head_dim / 2;
    const bool vec = (half % 4 == 0) && ((reinterpret_cast<uintptr_t>(qkv) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(cos_t) & 15) == 0) && ((reinterpret_cast<uintptr_t>(sin_t) & 15) == 0);
    const int64_t total = tokens * 2 * heads * (vec ? half / 4 : half);
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32);
    if (vec)
        hipLaunchKernelGGL(rope_qk_kernel<4>, dim3(grid), dim3(256), 0, stream, qkv, cos_t, sin_t, tokens, seq, heads, head_dim, tok_src);
    else
        hipLaunchKernelGGL(rope_qk_kernel<1>, dim3(grid), dim3(256), 0, stream, qkv, cos_t, sin_t, tokens, seq, heads, head_dim, tok_src);
    return hipGetLastError();
}

hipError_t launch_embed_layernorm(const uint32_t* ids, const uint32_t* type_ids, const float* word,
                                  const float* pos, const float* type, const float* gamma,
                                  const float* beta, float eps, int64_t tokens, int seq, int hidden,
                                  int vocab, int max_pos, int type_vocab, int pos_offset,
                                  int scale_embeddings, float* out, hipStream_t stream, const int32_t* tok_src)
{
    if (tokens <= 0) return hipSuccess;
    if (hidden % 4 == 0 && hidden <= 256 * MAX_V4_PER_LANE) {
        hipLaunchKernelGGL(embed_layernorm_kernel, dim3(rows_to_blocks(tokens)), dim3(256), 0, stream,
                           ids, type_ids, word, pos, type, gamma, beta, eps, tokens, seq, hidden,
                           vocab, max_pos, type_vocab, pos_offset, scale_embeddings, tok_src, out);
    } else {
        hipLaunchKernelGGL(embed_generic_kernel, dim3(rows_to_blocks(tokens)), dim3(256), 0, stream,
                           ids, type_ids, word, pos, type, tokens, seq, hidden, vocab, max_pos,
                           type_vocab, pos_offset, scale_embeddings, tok_src, out);
        if (gamma != nullptr)
            hipLaunchKernelGGL(layernorm_generic_kernel, dim3(rows_to_blocks(tokens)), dim3(256), 0,
                               stream, out, gamma, beta, eps, tokens, hidden, out);
    }
    return hipGetLastError();
}

hipError_t launch_layernorm(const float* in, const float* gamma, const float* beta, float eps,
                            int64_t rows, int hidden, float* out, hipStream_t stream)
{
    if (rows <= 0) return hipSuccess;
    if (hidden % 4 == 0 && hidden <= 256 * MAX_V4_PER_LANE)
        hipLaunchKernelGGL(layernorm_kernel, dim3(rows_to_blocks(rows)), dim3(256), 0, stream, in,
                           gamma, beta, eps, rows, hidden, out);
    else
        hipLaunchKernelGGL(layernorm_generic_kernel, dim3(rows_to_blocks(rows)), dim3(256), 0, stream,
                           in, gamma, beta, eps, rows, hidden, out);
    return hipGetLastError();
}

hipError_t launch_pool(const float* hidden_states, const uint32_t* mask, int64_t batch, int seq,
                       int hidden, PoolMode mode, int normalize, float* out, hipStream_t stream, const int32_t* cu)
{
    if (batch <= 0) return hipSuccess;
    if (hidden > 1024 || seq <= 0) return hipErrorInvalidValue;
    // A handful of sentences is a latency chain per workgroup (one sentence each): 1024 threads then split a sentence's
    // rows ten ways instead of two (20 -> 7 us for a call of 1 .. 32 sentences); large batches are HBM-bound either way.
    const bool wide = batch < 2048;
    dim3 grid((unsigned)batch);
#define KJ_POOL(MODE_)                                                                                                       \
    if (wide)                                                                                                                \
        hipLaunchKernelGGL((pool_kernel<MODE_, 1024>), grid, dim3(1024), 0, stream, hidden_states, mask, seq, hidden, normalize, cu, out); \
    else                                                                                                                     \
        hipLaunchKernelGGL((pool_kernel<MODE_, 256>), grid, dim3(256), 0, stream, hidden_states, mask, seq, hidden, normalize, cu, out)
    switch (mode) {
    case POOL_MEAN: KJ_POOL(POOL_MEAN); break;
    case POOL_CLS: KJ_POOL(POOL_CLS); break;
    case POOL_MAX: KJ_POOL(POOL_MAX); break;
    case POOL_LAST: KJ_POOL(POOL_LAST); break;
    default: return hipErrorInvalidValue;
    }
#undef KJ_POOL
    return hipGetLastError();
}

hipError_t launch_small_linear(const float* feat, int64_t ld, const float* w, const float* bias,
                               int64_t rows, int k, int n, float* out, hipStream_t stream)
{
    if (rows <= 0 || n <= 0) return hipSuccess;
    hipLaunchKernelGGL(small_linear_kernel, dim3(rows_to_blocks(rows * n)), dim3(256), 0, stream, feat,
                       ld, w, bias, rows, k, n, out);
    return hipGetLastError();
}

hipError_t launch_row_softmax(const float* in, int64_t rows, int n, int sigmoid, float* out,
                              hipStream_t stream)
{
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(row_softmax_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream,
                       in, rows, n, sigmoid, out);
    return hipGetLastError();
}

// Measurement aid (bench.py): ONE wave reads the shader-cycle counter (s_memtime) and the constant 100 MHz counter
// (s_memrealtime) either side of a spin of `spin_ticks` 10 ns ticks: cycles / ticks x 0.1 = the shader clock in GHz that the
// chip holds at that moment (MI355X_MICROARCH.md, "DVFS give-back" item 6).  Enqueued between steps of a timed region it
// costs its spin (tens of microseconds) and no synchronisation; out[0] = cycles, out[1] = ticks.
__global__ void clock_probe_kernel(unsigned long long* out, unsigned spin_ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1 = r0;
    // (bounded: 512 shader cycles per round, so even at 100 MHz a round is <= 512 ticks -- a counter that did not advance could
    // not hang the launch)
    for (unsigned round = 0; r1 - r0 < spin_ticks && round < spin_ticks + 4096u; ++round) {
        __builtin_amdgcn_s_sleep(8);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    out[0] = c1 - c0;
    out[1] = r1 - r0;
}

// The same reading repeated: `samples` consecutive windows of `window_ticks` (10 ns each), one (cycles, ticks) pair per window.
// Launched on a stream of its own beside the work being measured, it is the clock the chip holds UNDER that work (one wave,
// asleep between its reads; a probe between two kernels of the busy stream reads the clock of an idle chip instead).
__global__ void clock_trace_kernel(unsigned long long* out, unsigned samples, unsigned window_ticks)
{
    if (threadIdx.x != 0) return;
    for (unsigned s = 0; s < samples; ++s) {
        const unsigned long long c0 = __builtin_amdgcn_s_memtime();
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long r1 = r0;
        for (unsigned round = 0; r1 - r0 < window_ticks && round < window_ticks + 4096u; ++round) {  // (bounded as above)
            __builtin_amdgcn_s_sleep(64);
            r1 = __builtin_amdgcn_s_memrealtime();
        }
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
        r1 = __builtin_amdgcn_s_memrealtime();
        out[2 * s] = c1 - c0;
        out[2 * s + 1] = r1 - r0;
    }
}

hipError_t launch_clock_trace(uint64_t* out, unsigned samples, unsigned window_ticks, hipStream_t stream)
{
    hipLaunchKernelGGL(clock_trace_kernel, dim3(1), dim3(64), 0, stream, (unsigned long long*)out, samples, window_ticks);
    return hipGetLastError();
}

// Self-test of device_utils.h's reductions without the LDS crossbar: every form against the `__shfl_xor` butterfly it replaces,
// bit for bit, on pseudo-random values (one wave per workgroup, `waves` workgroups); out[0] += lanes that differ.
__global__ __launch_bounds__(64) void reduction_selftest_kernel(unsigned* __restrict__ out, unsigned seed)
{
    const unsigned i = blockIdx.x * 64u + threadIdx.x;
    unsigned h = (i + seed) * 2654435761u;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    // magnitudes over ~12 binades and both signs (sums that cancel), every 97th value exactly zero
    float v = ((int)(h & 0xFFFFFFu) - 0x800000) * (1.0f / 8388608.0f) * __uint_as_float(((h >> 24 & 15u) + 120u) << 23);
    if (i % 97u == 0u) v = 0.0f;
    unsigned bad = 0;
    auto same = [&](float a, float b) { bad += __float_as_uint(a) != __float_as_uint(b); };
    {   // wave-wide, butterfly 32, 16, 8, 4, 2, 1
        float s = v, m = v;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s += __shfl_xor(s, off, kWave);
            m = fmaxf(m, __shfl_xor(m, off, kWave));
        }
        same(wave_sum(v), s);
        same(wave_max(v), m);
    }
    {   // the 16-lane row, butterfly 8, 4, 2, 1
        float s = v, m = v;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            s += __shfl_xor(s, off, kWave);
            m = fmaxf(m, __shfl_xor(m, off, kWave));
        }
        same(row16_sum_desc(v), s);
        same(row16_max_desc(v), m);
    }
    {   // aligned groups of 8 lanes, butterfly 1, 2, 4
        float s = v, m = v;
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            s += __shfl_xor(s, off, kWave);
            m = fmaxf(m, __shfl_xor(m, off, kWave));
        }
        same(group8_sum_asc(v), s);
        same(group8_max_asc(v), m);
    }
    same(sum_xor32(v), v + __shfl_xor(v, 32, kWave));
    same(sum_xor16(v), v + __shfl_xor(v, 16, kWave));
    same(max_xor32(v), fmaxf(v, __shfl_xor(v, 32, kWave)));
    same(max_xor16(v), fmaxf(v, __shfl_xor(v, 16, kWave)));
    same(sum_xor32(sum_xor16(v)), [&] { float t = v + __shfl_xor(v, 16, kWave); return t + __shfl_xor(t, 32, kWave); }());
    if (bad) atomicAdd(out, bad);
}

hipError_t launch_reduction_selftest(unsigned* mismatches, unsigned waves, unsigned seed, hipStream_t stream)
{
    hipLaunchKernelGGL(reduction_selftest_kernel, dim3(waves), dim3(64), 0, stream, mismatches, seed);
    return hipGetLastError();
}

hipError_t launch_clock_probe(uint64_t* out, unsigned spin_ticks, hipStream_t stream)
{
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, stream, (unsigned long long*)out, spin_ticks);
    return hipGetLastError();
}

}  // namespace kjarni
