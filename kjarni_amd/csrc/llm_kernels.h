// Launchers of llm_kernels.hip (conventions of kernels.h: enqueue on `stream`, no allocation, no sync).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kjarni {

// Y = epi(rmsnorm?(X) W^T + b) for up to 8 rows; W (and W2) are f32 or bf16 [n_out, k] row-major, k % 8 == 0.
struct LlmGemvArgs {
    const float* X = nullptr;
    int64_t ldx = 0;
    int rows = 0;
    const float* gamma = nullptr;  // non-null: RMS-normalise the rows first
    float eps = 0.0f;
    const void* W = nullptr;
    const void* W2 = nullptr;      // SwiGLU `up` matrix
    int bf16 = 0;
    int swiglu = 0;
    const float* bias = nullptr;
    const float* R = nullptr;      // non-null: + residual
    int64_t ldr = 0;
    int n_out = 0, k = 0;
    int seg_q = 0, seg_kv = 0;     // seg_q > 0: [0,seg_q) -> Y0, then two seg_kv-wide segments -> Y1, Y2 at row row_off + r
    float* Y0 = nullptr;
    int64_t ldy0 = 0;
    float *Y1 = nullptr, *Y2 = nullptr;
    int64_t ldy12 = 0;
    int row_off = 0;
    const int* row_off_ptr = nullptr;
    // att_splits > 0 (only where llm_gemv_merges_attention() says so): X is not the context row but the decode attention's
    // slabs [k / att_head_dim heads][att_splits][att_head_dim + 4] (whisper_kernels.hip), merged while the weights stream in
    int att_splits = 0, att_head_dim = 0;
    float* norm_out = nullptr;     // rows == 1 with gamma (only where llm_gemv_streams()): the normalised row is also stored here
};
hipError_t launch_llm_gemv(const LlmGemvArgs& args, hipStream_t stream);
// One row, + residual, no norm: can the projection merge `splits` attention slabs of `head_dim`-wide heads itself?
bool llm_gemv_merges_attention(int k, int splits, int head_dim);
// Does a one-row projection with these sizes take the weight-streaming kernel (which honours norm_out)?
bool llm_gemv_streams(int k, const void* W, const void* W2);
#ifdef KJARNI_TUNING
void set_llm_gemv_variant(int variant);  // 0 = default (single-row kernel for rows == 1), 1 = always the multi-row kernel
#endif

// One new token: RMSNorm + Q|K|V projection + RoPE in one launch; Q -> Q[n_heads*head_dim], K / V -> row `pos`
// (or *pos_ptr) of the caches [*, n_kv_heads*head_dim].  W is the fused [Q;K;V] matrix.
// embed_ids != null (only where llm_qkv_rope_embeds() says so): the input row is gathered from the embedding `table` (row
// *embed_ids, the weights' dtype; an id >= vocab leaves zeros) instead of read from X, and stored to x_raw_out (the residual stream).
hipError_t launch_llm_qkv_rope(const float* X, const float* gamma, float eps, const void* W, int bf16, const float* bias, int k,
                               int n_heads, int n_kv_heads, int head_dim, const float* cos_t, const float* sin_t, float* Q, float* Kc,
                               float* Vc, int pos, const int* pos_ptr, hipStream_t stream, const uint32_t* embed_ids = nullptr,
                               const void* table = nullptr, int vocab = 0, float* x_raw_out = nullptr);
bool llm_qkv_rope_embeds(int k, const float* gamma, const void* W, const void* table);

// Prefill: Y[M, N] = A[M, K] . W[N, K]^T + bias (+ R) on the fp32 matrix cores, W bf16 or f32; K % 32 == 0.  R may alias Y.
// split_scratch (prefill_gemm_scratch_floats() floats, or null): short prompts split K over up to 8 workgroups per tile.
hipError_t launch_prefill_gemm(const float* A, int64_t lda, const void* W, int bf16, const float* bias, const float* R, int64_t ldr, float* Y,
                               int64_t ldy, int M, int N, int K, hipStream_t stream, float* split_scratch = nullptr,
                               float* silu_gate = nullptr);  // silu_gate: [M, N] gate activations, overwritten with silu(gate) * (this product)
size_t prefill_gemm_scratch_floats(int max_rows, int max_n);
// bf16 -> f32 copy of n values (n % 8 == 0): long prompts run the f32 tile GEMM (gemm.hip) on a widened weight matrix.
hipError_t launch_widen_bf16(const void* src, float* dst, size_t n, hipStream_t stream);

// Prefill: causal grouped-query attention of `rows` new rows (positions base .. base + rows - 1) over the cache rows
// 0 .. base + rows - 1; head_dim in {16, 32, 64, 128}.
bool prefill_attention_supported(int head_dim);
hipError_t launch_prefill_attention(const float* q, int64_t ldq, int rows, const float* K, int64_t ldk, const float* V, int64_t ldv, int base,
                                    int heads, int head_dim, int kv_group, float* ctx, int64_t ldc, hipStream_t stream);

// Prefill: gate = silu(gate) * up (n multiple of 4).
hipError_t launch_swiglu_mul(float* gate, const float* up, size_t n, hipStream_t stream);

// In-place RoPE on `rows` rows of [n_heads * head_dim]; cos/sin tables are [max_pos, head_dim/2].
// at_cache_row: row r of the call lives at row (pos + r) of x (the KV cache), else at row r.
hipError_t launch_rope(float* x, int64_t ldx, int rows, int n_heads, int head_dim, const float* cos_t, const float* sin_t, int pos,
                       const int* pos_ptr, int at_cache_row, hipStream_t stream);
hipError_t launch_rmsnorm(const float* x, const float* gamma, float eps, int rows, int hidden, float* out, hipStream_t stream);
#ifdef KJARNI_TUNING
hipError_t launch_touch(const void* p, size_t bytes, unsigned* sink, hipStream_t stream);  // measurements: warm the memory-side cache
#endif
hipError_t launch_llm_embed(const uint32_t* ids, int n, int hidden, int vocab, const void* table, int bf16, float* out,
                            hipStream_t stream);
// argmax (last maximum wins); best_scratch: one zero-initialised u64 (re-zeroed by the call); history/count/pos may be null.
hipError_t launch_argmax(const float* logits, int vocab, unsigned long long* best_scratch, int32_t* out, int32_t* history, int* count,
                         int* pos, hipStream_t stream);

// ---- sampled decoding: the O(vocab) part on the device (llm_kernels.hip) ----------------------------------------------
struct SampleHeader {   // 32 bytes, device memory mirrored to the host per sampled token
    float mx;           // maximum logit
    float sum;          // sum of exp(logit - mx) over the vocabulary (deterministic, not in index order)
    float floor;        // every token with logit >= floor is in the candidate list
    uint32_t count;     // candidates appended (may exceed the capacity: then overflow is set)
    uint32_t overflow;  // 1: the list is not usable (too long, or no cut could be placed) -- fetch the logits
    uint32_t pad[3];
};
struct SampleCandidate {
    uint32_t token;
    float logit;
};
size_t sample_scratch_bytes();   // zero-initialised device scratch of launch_sample_candidates
// top_k < 0 / top_p < 0 / min_p < 0: that filter is off (as SamplingParams).
hipError_t launch_sample_candidates(const float* logits, int vocab, int64_t top_k, float top_p, float min_p, void* scratch,
                                    SampleHeader* header, SampleCandidate* candidates, int capacity, hipStream_t stream);
// Logits processors on the device.  State: counts[vocab] (occurrences of each token in the history), distinct[] (the tokens
// with a non-zero count) and *n_distinct, all zero-initialised and advanced by launch_token_counts for every token that
// joins the history; tokens[len] is the history itself, in order.
hipError_t launch_token_counts(const int32_t* tokens, int n, int vocab, int* counts, int32_t* distinct, int* n_distinct, hipStream_t stream);
hipError_t launch_logits_processors(float* logits, int vocab, const int32_t* tokens, int len, const int* counts, const int32_t* distinct,
                                    const int* n_distinct, float repetition_penalty, int no_repeat_ngram, hipStream_t stream);

}  // namespace kjarni
