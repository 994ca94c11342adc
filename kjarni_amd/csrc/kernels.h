// Launchers for the hand-written gfx950 kernels of the encoder hot path.
// Every launcher enqueues on `stream` and returns the launch status; nothing
// here allocates, synchronises or touches the host (graph-capturable).
//
// Reference rows each kernel replaces are cited at the
// kernel definitions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kjarni {

// Epilogues of the projection GEMM  Y = X * W^T + b  (W is [N,K] row-major).
enum GemmEpilogue : int {
    EPI_BIAS = 0,           // y = acc + b
    EPI_BIAS_GELU = 1,      // erf GELU   (activations.rs:56-59)
    EPI_BIAS_GELU_NEW = 2,  // tanh GELU  (activations.rs:62-66)
    EPI_BIAS_RELU = 3,
    EPI_BIAS_TANH = 4,
    EPI_BIAS_RESIDUAL = 5,  // y = acc + b + R   (encoder_layer.rs:129-136, 155-163)
    EPI_BIAS_MUL_SILU = 6,  // y = silu(R) * (acc + b): the up projection of a SwiGLU FFN over the gate's output R
                            // (cpu/feedforward/swiglu.rs:40-50, activations.rs:74-82)
};

enum PoolMode : int { POOL_MEAN = 0, POOL_CLS = 1, POOL_MAX = 2, POOL_LAST = 3 };

// R2 + R3: word/pos/type gather-add fused with the embedding LayerNorm.
hipError_t launch_embed_layernorm(const uint32_t* ids, const uint32_t* type_ids, const float* word,
                                  const float* pos, const float* type, const float* gamma,
                                  const float* beta, float eps, int64_t tokens, int seq, int hidden,
                                  int vocab, int max_pos, int type_vocab, int pos_offset,
                                  int scale_embeddings, float* out, hipStream_t stream, const int32_t* tok_src = nullptr);

// Ragged batches run over the kept tokens only ("packed rows", rowops.hip): sentence b of a chunk is rows
// cu[b] .. cu[b+1] of every activation buffer, tok_src[row] is that token's index in the padded [batch, seq] arrays.
//   launch_mask_lengths: lens[b] = kept tokens of sentence b, bit 31 = the sentence needs the padded layout
//   launch_pack_index:   tok_src from mask + cu
// The kernels below that take `cu` / `tok_src` read the packed layout when it is non-null.
hipError_t launch_mask_lengths(const uint32_t* mask, int64_t batch, int seq, uint32_t* lens, hipStream_t stream);
hipError_t launch_pack_index(const uint32_t* mask, const int32_t* cu, int64_t batch, int seq, int32_t* tok_src,
                             hipStream_t stream);

// R3: row LayerNorm, in == out allowed.
hipError_t launch_layernorm(const float* in, const float* gamma, const float* beta, float eps,
                            int64_t rows, int hidden, float* out, hipStream_t stream);

// R4/R5/R9: fp32 MFMA GEMM with fused epilogue.  A rows are `lda` floats apart
// (lets the head GEMM read CLS rows in place), R/Y rows ldr/ldy apart.
// scratch (optional): a slab the call may use for partial tiles -- calls of 257 .. 8192 rows then take quarter-size tiles
// with K slices (gemm.hip, "calls of a few hundred to a few thousand rows") instead of the large-batch tiles.
struct GemmScratch {
    float* p = nullptr;
    size_t floats = 0;
};
hipError_t launch_gemm(const float* A, int64_t lda, const float* W, const float* bias,
                       const float* R, int64_t ldr, float* Y, int64_t ldy, int64_t M, int N, int K,
                       GemmEpilogue epi, hipStream_t stream, GemmScratch scratch = GemmScratch());

// R5 + R3 of the post-norm layer in one launch: Y = LayerNorm(A W^T + bias + R) * gamma + beta, R == Y allowed
// (cpu/encoder/encoder_layer.rs:129-147, 155-176).  Row widths the kernel covers: see ..._supported();
// anything else returns hipErrorInvalidValue and the caller runs launch_gemm + launch_layernorm.
bool gemm_residual_layernorm_supported(int N, int K);
// With a scratch slab of gemm_scratch_floats(rows, N) floats, calls of 257 .. 8192 rows (and up to 256 with a long K) also take any N <= 1024:
bool gemm_mid_layernorm_supported(int64_t M, int N, int K);
// Calls of up to this many rows take the few-rows kernel (K over the waves of a workgroup; LayerNorm a launch of its own
// unless the K-sliced route above applies): 256 for models up to 512 wide, 128 up to 1 024, 64 beyond.
int64_t gemm_few_rows_max(int hidden);
// Calls of up to this many rows run every projection on the mid-size route (gemm.hip): what the lanes of encoder.cpp rely on.
int64_t gemm_mid_route_max_rows();
size_t gemm_scratch_floats(int64_t max_rows, int max_narrow_n);
hipError_t launch_gemm_residual_layernorm(const float* A, int64_t lda, const float* W, const float* bias,
                                          const float* R, int64_t ldr, const float* gamma, const float* beta, float eps,
                                          float* Y, int64_t ldy, int64_t M, int N, int K, hipStream_t stream,
                                          GemmScratch scratch = GemmScratch());

// The mid-size projections' kernel (gemm_flex.hip; gemm.hip's route for calls of a few hundred to a few thousand rows calls it):
// one workgroup per CU, tile shape chosen per call.  ksplit = physical K slices (1 or logical_slices), logical_slices = slices in
// the result's summation order; partials != null: raw sums to slabs [ksplit][M][N] for the reduce kernels, else Y with the epilogue.
bool gemm_flex_shape_ok(int64_t M, int N, int K, int64_t lda, int64_t ldy, int64_t ldr, const float* A, const float* W, const float* Y,
                        const float* bias, const float* R);
double gemm_flex_cost(int M, int N, int K, int ksplit, int logical_slices);  // estimated shader cycles of the call
hipError_t launch_gemm_flex(const float* A, int64_t lda, const float* W, const float* bias, const float* R, int64_t ldr, float* Y,
                            int64_t ldy, int M, int N, int K, GemmEpilogue epi, int ksplit, int logical_slices, float* partials,
                            hipStream_t stream);

// Opt-in (process-wide; default off, or KJARNI_HIP_F32_ON_BF16=1): the large-batch projections compute their f32 products on the
// bf16 matrix cores -- every operand split exactly into three bf16 pieces, six of the nine cross products (gemm.hip, "fp32
// products on the bf16 matrix cores").  f32 in, f32 out, f32-level error; 2-3x the f32 MFMA rate.
void set_f32_on_bf16(bool on);
bool get_f32_on_bf16();
// (gemm_split.hip; N % 128 == 0, K % 64 == 0; the residual / gate operand R as launch_gemm)
hipError_t launch_gemm_split(const float* A, int64_t lda, const float* W, const float* bias, const float* R, int64_t ldr, float* Y,
                             int64_t ldy, int64_t M, int N, int K, GemmEpilogue epi, hipStream_t stream);

// f32 activations x bf16 weights [N, K] on the bf16 matrix cores: every activation split exactly into three bf16 pieces, the
// weights taken as they are -- the same products as an f32 GEMM on widened weights (gemm_split.hip).  N % 128 == 0, K % 64 == 0,
// 16-byte aligned rows; epilogues EPI_BIAS, EPI_BIAS_RESIDUAL (R may alias Y), EPI_BIAS_MUL_SILU (R = the gate, may alias Y).
hipError_t launch_gemm_bf16_weights(const float* A, int64_t lda, const void* W_bf16, const float* bias, const float* R, int64_t ldr,
                                    float* Y, int64_t ldy, int64_t M, int N, int K, GemmEpilogue epi, hipStream_t stream);

// The opt-in mode's 64 x 64-tile projections (gemm.hip's mid-size route calls this when the mode is on): partials != null -> the
// K slices' partial tiles [ksplit][M][N] for mid_reduce_*; else ksplit == 1 and the epilogue is applied.  K / ksplit % 32 == 0.
hipError_t launch_gemm_mid_split(const float* A, int64_t lda, const float* W, const float* bias, const float* R, int64_t ldr, float* Y,
                                 int64_t ldy, int M, int N, int K, int ksplit, float* partials, GemmEpilogue epi, hipStream_t stream);

// The 64 x 64-tile form for the decoder's short prompt blocks: the grid, tile order, K slices and partial layout of
// llm_kernels.hip's prefill_gemm_kernel (whose launcher calls this for bf16 weights); K / ksplit a multiple of 32, K % 8 == 0.
hipError_t launch_prefill_tiles_bf16w(unsigned grid, const float* A, int64_t lda, const void* W_bf16, const float* bias, const float* R,
                                      int64_t ldr, float* Y, int64_t ldy, int M, int N, int K, int m_tiles, int ksplit, float* partials,
                                      hipStream_t stream);

// (the kernel A/B switches of the tuning build live in tuning.h)

// R6/R7/R8: fused QK^T -> scale -> mask -> softmax -> PV for all heads.
// qkv is [tokens, 3*hidden] (Q | K | V), mask is u32 [batch, seq], ctx is
// [tokens, hidden] with heads merged.
// Packed rows: cu != null, mask is ignored (every row is a kept token), seq = the longest sentence of the call.
hipError_t launch_attention(const float* qkv, const uint32_t* mask, int64_t batch, int seq,
                            int heads, int head_dim, float mask_value, float* ctx,
                            hipStream_t stream, const int32_t* cu = nullptr);

// The reference's full operator signature (EncoderSelfAttention::forward / forward_noalloc, encoder_self_attention.rs:57-140,
// 143-307): an optional ADDITIVE position bias [heads, bias_seq, bias_seq] (broadcast over sentences; added after the scale and
// before the padding mask) and the scale_qk switch.  None of the registry's encoders passes a bias, so
// the model path never comes here: this is the any-shape kernel (one wave per query row), kept for the reference's own layer
// goldens (encoder_layer.rs:349-448), which use one.
hipError_t launch_attention_biased(const float* qkv, const uint32_t* mask, const float* pos_bias, int bias_seq, int64_t batch, int seq,
                                   int heads, int head_dim, bool scale_qk, float mask_value, float* ctx, hipStream_t stream);

int attention_small_call_items();  // calls of up to this many (sentence, head) items take the small-call attention kernel

// RoPE on the Q and K thirds of qkv [tokens, 3*hidden] in place, position = token index within its sentence
// (RoPE::apply_3d with offset 0, cpu/rope/mod.rs:118-170, 210-245; encoder_self_attention.rs:81-85).
// cos / sin are the reference's caches [>= seq, head_dim].
hipError_t launch_rope_qk(float* qkv, const float* cos_t, const float* sin_t, int64_t tokens, int seq, int heads,
                          int head_dim, hipStream_t stream, const int32_t* tok_src = nullptr);

// R11: pooling (+ optional L2 normalisation) of [batch, seq, hidden].
hipError_t launch_pool(const float* hidden_states, const uint32_t* mask, int64_t batch, int seq,
                       int hidden, PoolMode mode, int normalize, float* out, hipStream_t stream, const int32_t* cu = nullptr);

// R12 tail: logits[b, n] = feat[b,:] . Wc[n,:] + bc[n]  for small num_labels.
hipError_t launch_small_linear(const float* feat, int64_t ld, const float* w, const float* bias,
                               int64_t rows, int k, int n, float* out, hipStream_t stream);

// Row softmax / sigmoid over [rows, n] logits (classifier probabilities).
hipError_t launch_row_softmax(const float* in, int64_t rows, int n, int sigmoid, float* out,
                              hipStream_t stream);

// R14: cosine scores of every corpus row against `nq` queries.
// mode 0 = VectorStore (kjarni-search/src/vector.rs:131-148),
// mode 1 = Segment (kjarni-rag/src/segment.rs:355-371).
hipError_t launch_cosine_scores(const float* queries, int nq, const float* corpus, int64_t n_docs,
                                int dim, int mode, float* scores, hipStream_t stream);

// R14 selection: per-query top-k of scores [nq, n_docs], score descending,
// ties by ascending index.  workspace must hold cosine_topk_workspace_bytes().
size_t cosine_topk_workspace_bytes(int nq, int64_t n_docs, int k);
hipError_t launch_cosine_topk(const float* scores, int nq, int64_t n_docs, int k, void* workspace,
                              int64_t* out_idx, float* out_score, hipStream_t stream);

// R14, scan + selection in one call: per-query top-k without a caller-visible score array.  One query over a width the
// streaming kernel is specialised for (128 .. 1024) and k <= 256 runs ONE fused pass -- every wave keeps its best keys in
// registers, the scores never exist in memory -- plus one small merge launch; everything else runs launch_cosine_scores +
// launch_cosine_topk inside the workspace.  out_idx / out_score: [nq, k].
size_t cosine_search_workspace_bytes(int nq, int64_t n_docs, int dim, int k);
// nq = 1 with 16-byte aligned query / corpus pointers (the caller's promise): the fused pass's 4 MB where it applies.
size_t cosine_search_one_query_workspace_bytes(int64_t n_docs, int dim, int k);
hipError_t launch_cosine_search(const float* queries, int nq, const float* corpus, int64_t n_docs, int dim, int mode, int k,
                                void* workspace, int64_t* out_idx, float* out_score, hipStream_t stream);

// Measurement aid: one wave stamps the shader-cycle counter and the 100 MHz counter around a spin of spin_ticks x 10 ns;
// out[0] = shader cycles, out[1] = 10 ns ticks (clock in GHz = out[0] / out[1] / 10).  rowops.hip.
hipError_t launch_clock_probe(uint64_t* out, unsigned spin_ticks, hipStream_t stream);
hipError_t launch_clock_trace(uint64_t* out, unsigned samples, unsigned window_ticks, hipStream_t stream);
// Self-test: device_utils.h's DPP / permlane-swap reductions against the __shfl_xor butterflies they replace, bit for bit, on
// `waves` x 64 pseudo-random values; *mismatches (device, zeroed by the caller) += lanes that differ.  rowops.hip.
hipError_t launch_reduction_selftest(unsigned* mismatches, unsigned waves, unsigned seed, hipStream_t stream);

}  // namespace kjarni
