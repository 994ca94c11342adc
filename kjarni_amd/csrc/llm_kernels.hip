// Kernels of the decoder-only (Llama / Qwen2) path: a weight-streaming GEMV for 1-8 rows over f32 or bf16
// weights with RMSNorm folded in and the Q|K|V, SwiGLU and residual epilogues; RoPE; embedding gather;
// a two-stage argmax.  Attention over the KV cache reuses decode_attention_* (whisper_kernels.hip) with a
// grouped-query head mapping.
//
//   RMSNorm   crates/kjarni-transformers/src/cpu/normalization/rms_norm.rs:19-27
//   RoPE      cpu/rope/mod.rs:107-176
//   layer     cpu/decoder/rope_decoder_layer.rs:18-41, cpu/decoder/decoder_attention.rs:44-170,
//             cpu/feedforward/swiglu.rs:32-57
//   greedy    common/sampling.rs:83-88
#include "device_utils.h"
#include "llm_kernels.h"

namespace kjarni {

namespace {

constexpr int LLM_MAX_ROWS = 8;


struct F8 {
    float v[8];
};

// Eight consecutive weights starting at element 8*i of a row.
__device__ __forceinline__ F8 load8(const float* row, int i)
{
    const f32x4 a = *reinterpret_cast<const f32x4*>(row + i * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(row + i * 8 + 4);
    return F8{{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}};
}
__device__ __forceinline__ F8 load8(const uint16_t* row, int i)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 p = *reinterpret_cast<const u32x4*>(row + i * 8);
    F8 r;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        r.v[2 * c] = __uint_as_float(p[c] << 16);
        r.v[2 * c + 1] = __uint_as_float(p[c] & 0xFFFF0000u);
    }
    return r;
}

enum : int { LE_NONE = 0, LE_RESIDUAL = 1, LE_SWIGLU = 2 };

// Y[r, n] = epi(norm?(X[r, :]) . W[n, :] + bias[n]) for up to 8 rows; one wave per output column.
//  NORM: rows are RMS-normalised on the fly: (x / sqrt(mean(x^2) + eps)) * gamma, statistics recomputed per wave.
//  LE_SWIGLU: W2 is the `up` matrix; the output is silu(x.W[n]) * (x.W2[n]).
//  LE_RESIDUAL: + R[r, n].
//  seg > 0: columns [0,seg_q) -> Y0, then two segments of seg_kv columns -> Y1 / Y2 at row (*row_off_ptr | row_off) + r
//  (Q to scratch, K and V straight into the cache).
template <typename WT, int EPI, bool NORM>
__global__ __launch_bounds__(256) void llm_gemv_kernel(const float* __restrict__ X, int64_t ldx, int rows,
                                                       const float* __restrict__ gamma, float eps, const WT* __restrict__ W,
                                                       const WT* __restrict__ W2, const float* __restrict__ bias,
                                                       const float* __restrict__ R, int64_t ldr, int n_out, int k, int seg_q,
                                                       int seg_kv, float* __restrict__ Y0, int64_t ldy0, float* __restrict__ Y1,
                                                       float* __restrict__ Y2, int64_t ldy12, int row_off,
                                                       const int* __restrict__ row_off_ptr)
{
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_out) return;
    const int k8 = k >> 3;
    float scale[LLM_MAX_ROWS];
#pragma unroll
    for (int r = 0; r < LLM_MAX_ROWS; ++r) scale[r] = 1.0f;
    if (NORM) {
#pragma unroll
        for (int r = 0; r < LLM_MAX_ROWS; ++r) {
            if (r < rows) {
                float s = 0.0f;
                for (int i = lane; i < k8; i += 64) {
                    const F8 x = load8(X + r * ldx, i);
#pragma unroll
                    for (int c = 0; c < 8; ++c) s = fmaf(x.v[c], x.v[c], s);
                }
                scale[r] = sqrtf(wave_sum(s) / (float)k + eps);  // the rms; the reference divides by it
            }
        }
    }
    const WT* w_row = W + n * (int64_t)k;
    const WT* w2_row = EPI == LE_SWIGLU ? W2 + n * (int64_t)k : nullptr;
    float acc[LLM_MAX_ROWS], acc2[LLM_MAX_ROWS];
#pragma unroll
    for (int r = 0; r < LLM_MAX_ROWS; ++r) acc[r] = acc2[r] = 0.0f;
    for (int i = lane; i < k8; i += 64) {
        const F8 w = load8(w_row, i);
        F8 w2;
        if (EPI == LE_SWIGLU) w2 = load8(w2_row, i);
        F8 g;
        if (NORM) g = load8(gamma, i);
#pragma unroll
        for (int r = 0; r < LLM_MAX_ROWS; ++r) {
            if (r < rows) {
                F8 x = load8(X + r * ldx, i);
                if (NORM) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) x.v[c] = (x.v[c] / scale[r]) * g.v[c];
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    acc[r] = fmaf(x.v[c], w.v[c], acc[r]);
                    if (EPI == LE_SWIGLU) acc2[r] = fmaf(x.v[c], w2.v[c], acc2[r]);
                }
            }
        }
    }
    const float b = bias ? bias[n] : 0.0f;
    int which = 0;
    int64_t col = n;
    if (seg_q > 0 && n >= seg_q) {
        which = 1 + (int)((n - seg_q) / seg_kv);
        col = (n - seg_q) - (int64_t)(which - 1) * seg_kv;
    }
    float* Y = which == 0 ? Y0 : (which == 1 ? Y1 : Y2);
    const int64_t ldy = which == 0 ? ldy0 : ldy12;
    const int64_t r0 = which == 0 ? 0 : (row_off_ptr ? *row_off_ptr : row_off);
#pragma unroll
    for (int r = 0; r < LLM_MAX_ROWS; ++r) {
        if (r < rows) {
            float v = wave_sum(acc[r]) + b;
            if (EPI == LE_SWIGLU) {
                const float u = wave_sum(acc2[r]);
                v = (v / (1.0f + expf(-v))) * u;  // silu(gate) * up
            }
            if (EPI == LE_RESIDUAL) v += R[r * ldr + n];
            if (lane == 0) Y[(r0 + r) * ldy + col] = v;
        }
    }
}

// In-place rotation of `rows` rows of [n_heads * d] (rope/mod.rs:156-176): pairs (i, i + d/2), position =
// (*pos_ptr | pos) + row; x may start at row (*row_off_ptr | row_off) of a cache.
__global__ __launch_bounds__(256) void rope_kernel(float* __restrict__ x, int64_t ldx, int rows, int n_heads, int head_dim,
                                                   const float* __restrict__ cos_t, const float* __restrict__ sin_t, int pos,
                                                   const int* __restrict__ pos_ptr, int at_cache_row)
{
    const int half = head_dim >> 1;
    const int base = pos_ptr ? *pos_ptr : pos;
    const int total = rows * n_heads * half;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int i = idx % half;
        const int h = (idx / half) % n_heads;
        const int r = idx / (half * n_heads);
        const int p = base + r;
        float* row = x + (int64_t)(at_cache_row ? p : r) * ldx + h * head_dim;
        const float c = cos_t[(int64_t)p * half + i], s = sin_t[(int64_t)p * half + i];
        const float x0 = row[i], x1 = row[i + half];
        row[i] = x0 * c - x1 * s;
        row[i + half] = x0 * s + x1 * c;
    }
}

// out[r, :] = (x[r, :] / sqrt(mean(x^2) + eps)) * gamma   (rms_norm.rs:19-27); one wave per row.
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma, float eps,
                                                      int rows, int hidden, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* row = x + (int64_t)r * hidden;
    float s = 0.0f;
    for (int i = lane; i < hidden; i += 64) s = fmaf(row[i], row[i], s);
    const float rms = sqrtf(wave_sum(s) / (float)hidden + eps);
    for (int i = lane; i < hidden; i += 64) out[(int64_t)r * hidden + i] = (row[i] / rms) * gamma[i];
}

template <typename WT>
__global__ __launch_bounds__(256) void llm_embed_kernel(const uint32_t* __restrict__ ids, int hidden, int vocab,
                                                        const WT* __restrict__ table, float* __restrict__ out)
{
    const int s = blockIdx.x;
    const uint32_t id = ids[s];
    for (int i = threadIdx.x; i < hidden / 8; i += 256) {
        F8 v;
#pragma unroll
        for (int c = 0; c < 8; ++c) v.v[c] = 0.0f;
        if (id < (uint32_t)vocab) v = load8(table + (int64_t)id * hidden, i);
#pragma unroll
        for (int c = 0; c < 8; ++c) out[(int64_t)s * hidden + i * 8 + c] = v.v[c];
    }
}

__device__ __forceinline__ unsigned long long argmax_key(float v, int idx)
{
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // larger float -> larger uint
    if (v != v) u = 0;                               // NaN lowest
    return ((unsigned long long)u << 32) | (uint32_t)idx;  // equal values: the LAST index wins (Iterator::max_by)
}

__global__ __launch_bounds__(256) void argmax_partial_kernel(const float* __restrict__ logits, int vocab,
                                                             unsigned long long* __restrict__ best)
{
    __shared__ unsigned long long red[4];
    unsigned long long key = 0ull;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < vocab; i += gridDim.x * 256) {
        const unsigned long long k = argmax_key(logits[i], i);
        key = k > key ? k : key;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(key, off, kWave);
        key = o > key ? o : key;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) key = red[w] > key ? red[w] : key;
        atomicMax(best, key);
    }
}

// Publishes the winner, resets the accumulator, and (graph replay) appends the token and advances the counters.
__global__ void argmax_finalize_kernel(unsigned long long* __restrict__ best, int32_t* __restrict__ out,
                                       int32_t* __restrict__ history, int* __restrict__ count, int* __restrict__ pos)
{
    const int tok = (int)(uint32_t)(*best & 0xFFFFFFFFull);
    *best = 0ull;
    *out = tok;
    if (history) {
        history[*count] = tok;
        *count += 1;
        *pos += 1;
    }
}

}  // namespace

template <typename WT>
static hipError_t launch_gemv_t(const LlmGemvArgs& a, hipStream_t stream)
{
    const dim3 grid((unsigned)((a.n_out + 3) / 4));
    const WT* W = static_cast<const WT*>(a.W);
    const WT* W2 = static_cast<const WT*>(a.W2);
#define KJ_LLM(EPI, NORM)                                                                                                       \
    hipLaunchKernelGGL((llm_gemv_kernel<WT, EPI, NORM>), grid, dim3(256), 0, stream, a.X, a.ldx, a.rows, a.gamma, a.eps, W, W2,  \
                       a.bias, a.R, a.ldr, a.n_out, a.k, a.seg_q, a.seg_kv, a.Y0, a.ldy0, a.Y1, a.Y2, a.ldy12, a.row_off,       \
                       a.row_off_ptr)
    const bool norm = a.gamma != nullptr;
    if (a.swiglu) {
        if (norm) KJ_LLM(LE_SWIGLU, true);
        else KJ_LLM(LE_SWIGLU, false);
    } else if (a.R) {
        if (norm) KJ_LLM(LE_RESIDUAL, true);
        else KJ_LLM(LE_RESIDUAL, false);
    } else {
        if (norm) KJ_LLM(LE_NONE, true);
        else KJ_LLM(LE_NONE, false);
    }
#undef KJ_LLM
    return hipGetLastError();
}

hipError_t launch_llm_gemv(const LlmGemvArgs& a, hipStream_t stream)
{
    if (a.rows <= 0 || a.n_out <= 0) return hipSuccess;
    if (a.rows > LLM_MAX_ROWS || (a.k & 7) || (a.ldx & 3) || (reinterpret_cast<uintptr_t>(a.X) & 15) ||
        (reinterpret_cast<uintptr_t>(a.W) & 15))
        return hipErrorInvalidValue;
    return a.bf16 ? launch_gemv_t<uint16_t>(a, stream) : launch_gemv_t<float>(a, stream);
}

hipError_t launch_rope(float* x, int64_t ldx, int rows, int n_heads, int head_dim, const float* cos_t, const float* sin_t, int pos,
                       const int* pos_ptr, int at_cache_row, hipStream_t stream)
{
    const int total = rows * n_heads * (head_dim / 2);
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, ldx, rows, n_heads, head_dim, cos_t,
                       sin_t, pos, pos_ptr, at_cache_row);
    return hipGetLastError();
}

hipError_t launch_rmsnorm(const float* x, const float* gamma, float eps, int rows, int hidden, float* out, hipStream_t stream)
{
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(rmsnorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, gamma, eps, rows, hidden, out);
    return hipGetLastError();
}

hipError_t launch_llm_embed(const uint32_t* ids, int n, int hidden, int vocab, const void* table, int bf16, float* out,
                            hipStream_t stream)
{
    if (hidden & 7) return hipErrorInvalidValue;
    if (bf16)
        hipLaunchKernelGGL(llm_embed_kernel<uint16_t>, dim3((unsigned)n), dim3(256), 0, stream, ids, hidden, vocab,
                           static_cast<const uint16_t*>(table), out);
    else
        hipLaunchKernelGGL(llm_embed_kernel<float>, dim3((unsigned)n), dim3(256), 0, stream, ids, hidden, vocab,
                           static_cast<const float*>(table), out);
    return hipGetLastError();
}

hipError_t launch_argmax(const float* logits, int vocab, unsigned long long* best_scratch, int32_t* out, int32_t* history, int* count,
                         int* pos, hipStream_t stream)
{
    int blocks = (vocab + 2047) / 2048;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(argmax_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, logits, vocab, best_scratch);
    hipLaunchKernelGGL(argmax_finalize_kernel, dim3(1), dim3(1), 0, stream, best_scratch, out, history, count, pos);
    return hipGetLastError();
}

}  // namespace kjarni
