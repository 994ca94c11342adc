/*
 * kjarni_hip.h -- token-level C ABI of the MI355X encoder core.
 *
 * These entry points sit at the reference's token-level boundary
 *   EncoderLanguageModel::get_hidden_states_batch_from_ids
 *     (crates/kjarni-transformers/src/cpu/encoder/traits.rs:66-139)
 *   CpuEncoderOps::forward_tokens (traits.rs:295-313)
 *   SentenceEncoder::encode_batch_flat
 *     (crates/kjarni-models/src/models/sentence_encoder/model.rs:201-218)
 *   CrossEncoder::predict_pairs / SequenceClassifier::predict_logits
 *     (models/cross_encoder/model.rs:170-240, models/sequence_classifier/mod.rs:265-346)
 *   VectorStore::search / Segment::search_vectors
 *     (crates/kjarni-search/src/vector.rs:150-166, crates/kjarni-rag/src/segment.rs:307-337)
 * i.e. the place where the reference's wgpu backend plugs in
 * (GpuEncoderOps, traits.rs:443-527).  The string-level kjarni-ffi surface
 * (kjarni.h) is built on top of them.
 *
 * Plain C: pointers and sizes only.  "_dev" arguments are device pointers on
 * the encoder's HIP device; work is enqueued on `stream` (a hipStream_t passed
 * as void*, NULL = the default stream) and NOT synchronised unless stated.
 * ids / attention mask / token type ids are uint32 [batch, seq] row-major, as
 * in the reference (Array2<u32>).
 */
#ifndef KJARNI_HIP_H
#define KJARNI_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "kjarni.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct KjarniHipEncoder KjarniHipEncoder;

/* Pooling strategies: crates/kjarni-transformers/src/cpu/encoder/config.rs (PoolingStrategy). */
typedef enum KjarniHipPooling {
    KJARNI_HIP_POOL_MEAN = 0,
    KJARNI_HIP_POOL_CLS = 1,
    KJARNI_HIP_POOL_MAX = 2,
    KJARNI_HIP_POOL_LAST_TOKEN = 3,
} KjarniHipPooling;

/* Padding-mask fill value.  The reference overwrites masked scores with -1e9 on
 * its alloc path (utils/masks.rs:4-36) and with -inf on its no-alloc path
 * (cpu/encoder/encoder_self_attention.rs:311-325); AUTO applies its selection
 * rule (cpu/strategy.rs:43-44: no-alloc iff tokens <= 1 or tokens >= 1000) for
 * embed/hidden_states and the alloc value for logits (forward_tokens always
 * takes the alloc path). */
typedef enum KjarniHipMaskFill {
    KJARNI_HIP_MASK_AUTO = 0,
    KJARNI_HIP_MASK_NEG_1E9 = 1,
    KJARNI_HIP_MASK_NEG_INF = 2,
} KjarniHipMaskFill;

/* Number of visible HIP devices (0 when none / no runtime). */
int32_t kjarni_hip_device_count(void);

/* Load <model_dir>/{config.json, model.safetensors} onto HIP device `device`.
 * Errors: KJARNI_ERROR_GPU_UNAVAILABLE (no device), KJARNI_ERROR_MODEL_NOT_FOUND
 * (files missing), KJARNI_ERROR_LOAD_FAILED (bad files). */
KjarniErrorCode kjarni_hip_encoder_load(const char* model_dir, int32_t device, KjarniHipEncoder** out);
void kjarni_hip_encoder_free(KjarniHipEncoder* enc);

int32_t kjarni_hip_encoder_hidden_size(const KjarniHipEncoder* enc);
int32_t kjarni_hip_encoder_num_layers(const KjarniHipEncoder* enc);
int32_t kjarni_hip_encoder_max_seq_len(const KjarniHipEncoder* enc);
int32_t kjarni_hip_encoder_vocab_size(const KjarniHipEncoder* enc);
int32_t kjarni_hip_encoder_num_labels(const KjarniHipEncoder* enc); /* 0: no classification head */
int32_t kjarni_hip_encoder_device(const KjarniHipEncoder* enc);
/* Tokens processed per internal chunk (workspace size); 0 keeps the default. */
KjarniErrorCode kjarni_hip_encoder_set_chunk_tokens(KjarniHipEncoder* enc, int64_t tokens);
/* Ragged batches: embed / logits run the layers over the kept tokens only (mask != 0) when a call has padding, every
 * sentence keeps its token 0 and the mask holds only 0 / 1 -- a padded token is observable through neither output
 * (pooling skips it, pooling/mod.rs:11-33; as a key its score is overwritten, utils/masks.rs:4-36).  Results agree with the
 * padded layout to rounding (<= 1e-6).  `on`:
 *   0  never: every call takes the padded layout (what hidden_states always does);
 *   1  (default) the host-pointer entry points (..._host, and everything built on them: the string-level handles, groups),
 *      whose mask is already on the host -- the device-pointer entry points stay "enqueued on `stream`, never synchronised",
 *      graph-capturable, on the padded layout;
 *   2  the device-pointer embed / logits too: a call of more than one sentence then reads 4 bytes per sentence back and
 *      SYNCHRONISES `stream` once before the layers are enqueued (it blocks the calling thread behind whatever the stream
 *      holds; on a capturing stream the call falls back to the padded layout).  Opt in when ragged device-resident batches
 *      matter more than asynchrony (1.76 x on lengths U{16..128}). */
KjarniErrorCode kjarni_hip_encoder_set_packing(KjarniHipEncoder* enc, int32_t on);

/* Concurrent small calls on one handle -- OPT-IN, default off.  The reference serialises nothing on a handle
 * (kjarni-ffi/src/lib.rs:25-32) and gives every call its own deterministic result; so does this library by default: N threads
 * that each embed or classify one sentence run N independent forwards (on up to four workspaces / streams at once), each
 * bit-identical to the same call made alone.  A single-sentence forward is ~45 dependent launches on a fraction of the chip,
 * so a service that prefers throughput over bit-reproducibility can turn COMBINING on (1 here, or KJARNI_HIP_COMBINE=1 in the
 * environment for every handle the process loads, the string-level handles and groups included): small host-pointer calls
 * (<= 8 rows, <= 1 024 tokens) that arrive while another small call of the same kind is on the device then ride along with the
 * next leader as ONE packed batch (16 threads: 9.7 k -> 28 k encodes/s).  A call that finds the handle idle still runs at once.
 * What it costs: a combined call's result equals the solo call's to rounding (<= 1e-6: the rows take the packed layout and
 * possibly another tile route), not bit for bit, and WHICH calls share a forward depends on arrival times.  If a shared
 * forward fails, each of its calls is run again alone, so only the call that fails by itself reports an error. */
KjarniErrorCode kjarni_hip_encoder_set_combining(KjarniHipEncoder* enc, int32_t on);

/* Mid-size host-pointer calls (2 304 .. 24 576 kept tokens: the reference's default batch of 32 sentences and a few of them)
 * run as two or three parts on as many workspaces / streams, the other parts enqueued by helper threads of the handle: one
 * part's launch gaps, prologues and output bursts fall under the others' matrix work (32 / 48 / 72 / 96 / 160 x 128 tokens: 1.02 /
 * 1.57 / 2.35 / 2.68 / 4.54 -> 0.89 / 1.36 / 1.99 / 2.47 / 3.93 ms).  Sentences are independent.  Up to 8 192 tokens the parts take the
 * projection route the whole call would, so results are bit-identical to the unsplit call (and a call that finds the helpers
 * busy runs unsplit); above that the unsplit call would take the large-batch kernels, the two forms agree to rounding
 * (<= 1e-6), and the split form is what the call computes whenever this is on (deterministically: a part whose helper is busy
 * runs on the caller's thread).  On by default; 0 (or environment KJARNI_HIP_TWO_LANES=0) runs every call as one launch
 * sequence. */
KjarniErrorCode kjarni_hip_encoder_set_two_lanes(KjarniHipEncoder* enc, int32_t on);

/* Opt-in, process-wide, default OFF (or environment KJARNI_HIP_F32_ON_BF16=1, read once at the first projection): the
 * projections of calls above the few-rows range (more than 256 token rows; 128 for models wider than 512) compute their f32
 * products on the bf16 matrix cores.  Every
 * f32 operand is split EXACTLY into three bf16 pieces (8 + 8 + 8 significand bits) on its way into LDS and six of the nine
 * cross products are accumulated in f32 -- the three dropped ones are below 2^-24 of a product, f32's own rounding --
 * so inputs, outputs and the error level are those of f32 arithmetic (measured against float64: the same 2-4e-6 as the f32
 * MFMA kernels, tests/test_gpu_split.py), at 1.1-1.3x their speed (the reference's default batch of 32 sentences x 128
 * tokens: 1.12 -> 0.94 ms per call).  Differences: sums run in another
 * order (results equal to the default path to rounding, not bit for bit), and NON-FINITE inputs: an element that is +-inf or NaN
 * turns every product it takes part in into NaN (the split computes x - bf16(x), and inf - inf is NaN) where the default
 * path keeps an infinity an infinity (inf * finite = inf); finite inputs whose products overflow behave the same in both.  The reference computes in f32
 * (kjarni-transformers/src/linear_layer/linear_layer.rs:160-282); whether f32 results assembled from bf16 pieces meet a
 * deployment's precision policy is the integrator's decision, hence off by default.  Returns the previous setting. */
int32_t kjarni_hip_set_f32_on_bf16(int32_t on);
int32_t kjarni_hip_get_f32_on_bf16(void);

/* Self-test (no counterpart in the reference): the kernels' cross-lane sums and maxima avoid the LDS crossbar (csrc/device_utils.h:
 * v_permlane32_swap / v_permlane16_swap and DPP row rotations in place of ds_bpermute_b32) and claim the `__shfl_xor` butterfly's
 * values bit for bit; this runs every such form against the butterfly it replaces on `waves` x 64 pseudo-random values of
 * seed `seed` (1 .. 4 194 304 waves) and writes the number of lanes that differ to *mismatches_out (0 = the claim holds). */
KjarniErrorCode kjarni_hip_selftest_reductions(int32_t device, uint32_t waves, uint32_t seed, uint32_t* mismatches_out);

/* Measurement aid (bench.py's `clock_ghz` fields; no counterpart in the reference): enqueues ONE wave on `stream` that reads the
 * shader-cycle counter and the constant 100 MHz counter either side of a spin of `spin_us` microseconds (0 = 20; at most
 * 10 000) and writes out_dev[0] = shader cycles, out_dev[1] = 10 ns ticks.  out_dev[0] / out_dev[1] / 10 is the shader clock in
 * GHz the chip holds at that point of the stream -- what a power-limited run lowers while a short one does not.  Asynchronous;
 * out_dev is a device pointer to two uint64. */
KjarniErrorCode kjarni_hip_clock_probe(uint64_t* out_dev, uint32_t spin_us, void* stream);
/* The same reading over `samples` consecutive windows of `window_us` microseconds (1 .. 4096 windows of 10 .. 1 000 000 us, at
 * most ten seconds per launch): out_dev[2 s] = shader cycles, out_dev[2 s + 1] = 10 ns ticks of window s.  Enqueued on a stream
 * of its OWN beside the work being measured (one wave, asleep between its reads) it is the clock the chip holds UNDER that work:
 * f32 matrix-core kernels run the board at its power cap and the clock it settles at there -- not the 2.4 GHz of the data sheet
 * -- is what their rate is a fraction of.  A probe between two kernels of the busy stream reads an idle chip's clock instead. */
KjarniErrorCode kjarni_hip_clock_trace(uint64_t* out_dev, uint32_t samples, uint32_t window_us, void* stream);
/* A non-blocking stream of the library's own for such readings (one per process, made on first use on the current device, never
 * destroyed; NULL if it cannot be made).  hipStreamSynchronize / a later blocking copy from out_dev orders the caller behind it. */
void* kjarni_hip_measurement_stream(void);
/* Waits for it and destroys it (the next kjarni_hip_measurement_stream() makes a new one).  A process has a handful of hardware
 * queues and HIP deals its streams over them: release the stream once the readings are in, or one of an encoder's own streams may
 * share a queue with it. */
void kjarni_hip_measurement_stream_release(void);

/* Thread safety (every entry point of a KjarniHipEncoder / KjarniHipEncoderGroup, device- and host-pointer forms):
 * calls may be made concurrently from any number of host threads and on any streams, as on the reference's
 * handles (its model types are Send + Sync, crates/kjarni-ffi/src/lib.rs:25-32).  The weights are immutable; each
 * call leases one of up to 4 activation workspaces of the handle (a fifth concurrent call waits for a lease),
 * and a workspace that moves from one stream to another is ordered by an event.  Device-pointer forms only
 * ENQUEUE on `stream` (NULL = the legacy default stream); host-pointer forms run on a stream of their own and
 * return when the result is in the caller's buffer.  The profiler (profile_begin / _end) is single-caller.
 *
 * What a sentence's result depends on.  Sentences never influence each other's VALUES beyond rounding, and within one kernel
 * route a row's result is bit-identical whatever shares its call (tests/test_gpu_large_batch.py, test_gpu_encoder.py).  The
 * route, however, follows the call's size -- projections: up to 256 rows / 257 .. 8 192 / more; attention: up to 128
 * (sentence, head) items / more; pooling: calls below 2 048 sentences / more; packed or padded layout -- and every route sums
 * in its own order, so the same sentence in calls of different sizes (or combined with other callers' sentences, below)
 * agrees to rounding (measured <= 1e-6 on unit-norm embeddings), not bit for bit.  The reference makes the same kind of
 * choice by size (linear_layer.rs:164-173: row-vector kernel below 1 000 rows, 4 x 3 block kernel above).
 *
 * hidden_out_dev: f32 [batch, seq, hidden].  type_ids_dev may be NULL (token
 * type row 0 is added to every token, cpu/embeddings/mod.rs:216-223). */
KjarniErrorCode kjarni_hip_encoder_hidden_states(KjarniHipEncoder* enc, const uint32_t* ids_dev,
                                                 const uint32_t* mask_dev, const uint32_t* type_ids_dev,
                                                 int64_t batch, int32_t seq, KjarniHipMaskFill fill,
                                                 float* hidden_out_dev, void* stream);

/* out_dev: f32 [batch, hidden].  mean-pool + normalize = encode_batch_flat. */
KjarniErrorCode kjarni_hip_encoder_embed(KjarniHipEncoder* enc, const uint32_t* ids_dev,
                                         const uint32_t* mask_dev, const uint32_t* type_ids_dev,
                                         int64_t batch, int32_t seq, KjarniHipPooling pooling,
                                         int32_t normalize, KjarniHipMaskFill fill, float* out_dev,
                                         void* stream);

/* logits_out_dev: f32 [batch, num_labels] (rerank score = column 0). */
KjarniErrorCode kjarni_hip_encoder_logits(KjarniHipEncoder* enc, const uint32_t* ids_dev,
                                          const uint32_t* mask_dev, const uint32_t* type_ids_dev,
                                          int64_t batch, int32_t seq, KjarniHipMaskFill fill,
                                          float* logits_out_dev, void* stream);

/* Host-pointer variants: copy in, run, copy out, synchronise. */
KjarniErrorCode kjarni_hip_encoder_hidden_states_host(KjarniHipEncoder* enc, const uint32_t* ids,
                                                      const uint32_t* mask, const uint32_t* type_ids,
                                                      int64_t batch, int32_t seq, KjarniHipMaskFill fill,
                                                      float* hidden_out);
KjarniErrorCode kjarni_hip_encoder_embed_host(KjarniHipEncoder* enc, const uint32_t* ids,
                                              const uint32_t* mask, const uint32_t* type_ids,
                                              int64_t batch, int32_t seq, KjarniHipPooling pooling,
                                              int32_t normalize, KjarniHipMaskFill fill, float* out);
KjarniErrorCode kjarni_hip_encoder_logits_host(KjarniHipEncoder* enc, const uint32_t* ids,
                                               const uint32_t* mask, const uint32_t* type_ids,
                                               int64_t batch, int32_t seq, KjarniHipMaskFill fill,
                                               float* logits_out);

/* ---- several devices in one process ---------------------------------------------------
 * A group holds one replica of the model per listed device (weights replicated, ~90 MB for MiniLM).  Rows are
 * independent on this path, so a batch is cut into balanced contiguous row blocks -- floor(rows / n) each, the first
 * rows % n one more (kjarni_hip_group_shard) -- one host thread + one stream per device, no exchange inside the
 * layer loop.  This is what the string-level handles (kjarni_embedder_*, kjarni_reranker_*, kjarni_classifier_*)
 * use internally with the devices of KJARNI_HIP_DEVICES="0,1,..." (default: every visible device -- or, in a process
 * whose launcher exported LOCAL_RANK (one process per GPU), that one device; a device may be
 * listed twice); n_devices = 0 here means the same list.  A batch smaller than 8 rows per device uses fewer
 * devices, chosen in rotation. */
typedef struct KjarniHipEncoderGroup KjarniHipEncoderGroup;
KjarniErrorCode kjarni_hip_group_load(const char* model_dir, const int32_t* devices, size_t n_devices,
                                      KjarniHipEncoderGroup** out);
void kjarni_hip_group_free(KjarniHipEncoderGroup* group);
size_t kjarni_hip_group_size(const KjarniHipEncoderGroup* group);
int32_t kjarni_hip_group_device(const KjarniHipEncoderGroup* group, size_t i);
int32_t kjarni_hip_group_hidden_size(const KjarniHipEncoderGroup* group);
int32_t kjarni_hip_group_num_labels(const KjarniHipEncoderGroup* group);
/* Row block [*start_out, *start_out + *count_out) of `rows` rows that replica i owns. */
KjarniErrorCode kjarni_hip_group_shard(const KjarniHipEncoderGroup* group, int64_t rows, size_t i, int64_t* start_out,
                                       int64_t* count_out);
/* The collective that kjarni_hip_group_*_allgather issues for `rows` rows of `width` floats over n devices, as a list (host
 * arithmetic only: no device, no group needed): rank `rank` calls ncclAllGather in place (root = -1; equal blocks) or one
 * ncclBroadcast per non-empty block (root >= 0; blocks that differ by a row) on `floats` values at float offset `offset` of
 * its full output buffer.  Writes up to `cap` operations to ops_out, the total count to *count_out (n for equal blocks,
 * n x non-empty blocks otherwise). */
typedef struct KjarniHipGatherOp {
    int32_t rank;
    int32_t root;
    int64_t offset;
    int64_t floats;
} KjarniHipGatherOp;
KjarniErrorCode kjarni_hip_group_gather_plan(int64_t rows, size_t n, int64_t width, KjarniHipGatherOp* ops_out, size_t cap,
                                             size_t* count_out);
/* Host pointers: every device stages, encodes and returns its block straight into `out` ([batch, hidden] /
 * [batch, num_labels]); the mask fill follows the size of the whole call. */
KjarniErrorCode kjarni_hip_group_embed_host(KjarniHipEncoderGroup* group, const uint32_t* ids, const uint32_t* mask,
                                            const uint32_t* type_ids, int64_t batch, int32_t seq, KjarniHipPooling pooling,
                                            int32_t normalize, KjarniHipMaskFill fill, float* out);
KjarniErrorCode kjarni_hip_group_logits_host(KjarniHipEncoderGroup* group, const uint32_t* ids, const uint32_t* mask,
                                             const uint32_t* type_ids, int64_t batch, int32_t seq, KjarniHipMaskFill fill,
                                             float* logits_out);
/* Device-resident: ids_dev[i] / mask_dev[i] / type_ids_dev[i] point at replica i's row block ON device i, out_dev[i]
 * at a full [batch_total, hidden] (or [batch_total, num_labels]) buffer on device i.  Every replica computes its block
 * in place, then ONE collective leaves the whole output in every buffer: ncclAllGather over xGMI (grouped
 * ncclBroadcast when the blocks differ by a row) when the devices are distinct, peer copies otherwise.  Returns when
 * the collective has completed.  type_ids_dev may be NULL.  kjarni_hip_group_transport: "rccl" or "memcpy"
 * ("memcpy (ncclCommInitAll failed: ...)" when RCCL is present but could not build the communicator). */
KjarniErrorCode kjarni_hip_group_embed_allgather(KjarniHipEncoderGroup* group, const uint32_t* const* ids_dev,
                                                 const uint32_t* const* mask_dev, const uint32_t* const* type_ids_dev,
                                                 int64_t batch_total, int32_t seq, KjarniHipPooling pooling,
                                                 int32_t normalize, KjarniHipMaskFill fill, float* const* out_dev);
KjarniErrorCode kjarni_hip_group_logits_allgather(KjarniHipEncoderGroup* group, const uint32_t* const* ids_dev,
                                                  const uint32_t* const* mask_dev, const uint32_t* const* type_ids_dev,
                                                  int64_t batch_total, int32_t seq, KjarniHipMaskFill fill,
                                                  float* const* logits_out_dev);
const char* kjarni_hip_group_transport(KjarniHipEncoderGroup* group);

/* ---- single operators (host pointers) ---------------------------------------------
 * The kernels of the forward pass, one at a time, at the granularity of the
 * reference's own operator types: LinearLayer::matmul (+ fused epilogue),
 * EncoderSelfAttention after the QKV projection, LayerNorm::forward.  Used by the
 * single-op parity tests (tolerance 1e-5) and the kernel micro-benchmarks: when
 * iters > 0 the launch is repeated `iters` times between two HIP events and the
 * average milliseconds per launch are written to *ms_out (ms_out may be NULL). */
typedef enum KjarniHipEpilogue {
    KJARNI_HIP_EPI_BIAS = 0,
    KJARNI_HIP_EPI_BIAS_GELU = 1,
    KJARNI_HIP_EPI_BIAS_GELU_NEW = 2,
    KJARNI_HIP_EPI_BIAS_RELU = 3,
    KJARNI_HIP_EPI_BIAS_TANH = 4,
    KJARNI_HIP_EPI_BIAS_RESIDUAL = 5,
    KJARNI_HIP_EPI_BIAS_MUL_SILU = 6, /* y = silu(residual) * (x . w^T + bias): the up projection of SwiGluFeedForward over the
                                       * gate projection's output (cpu/feedforward/swiglu.rs:40-50) */
} KjarniHipEpilogue;

/* y[m,n] = epilogue(x[m,k] . w[n,k]^T + bias[n] (+ residual[m,n])); bias / residual may be NULL. */
KjarniErrorCode kjarni_hip_op_linear(int32_t device, const float* x, const float* w, const float* bias,
                                     const float* residual, int64_t m, int32_t k, int32_t n,
                                     KjarniHipEpilogue epilogue, float* y, int32_t iters, float* ms_out);
/* The same with bf16 weights (w: n x k bf16 values, row-major): the decoder's prompt projections on the bf16 matrix cores --
 * every activation split exactly into three bf16 pieces, the weights taken as they are, f32 accumulation: the products of an
 * f32 GEMM on the widened weights.  n % 128 == 0, k % 64 == 0; epilogues BIAS, BIAS_RESIDUAL, BIAS_MUL_SILU. */
KjarniErrorCode kjarni_hip_op_linear_bf16_weights(int32_t device, const float* x, const uint16_t* w_bf16, const float* bias,
                                                  const float* residual, int64_t m, int32_t k, int32_t n,
                                                  KjarniHipEpilogue epilogue, float* y, int32_t iters, float* ms_out);
/* qkv: [batch*seq, 3*heads*head_dim] (Q | K | V); mask: u32 [batch, seq] or NULL; ctx: [batch*seq, heads*head_dim]. */
KjarniErrorCode kjarni_hip_op_attention(int32_t device, const float* qkv, const uint32_t* mask, int64_t batch,
                                        int32_t seq, int32_t heads, int32_t head_dim, float mask_value,
                                        float* ctx, int32_t iters, float* ms_out);
/* The operator with the reference's full argument list (EncoderSelfAttention::forward / forward_noalloc,
 * cpu/encoder/encoder_self_attention.rs:57-140, 143-307): position_bias = NULL or an additive f32 [heads, bias_seq, bias_seq]
 * (bias_seq >= seq; broadcast over sentences; added after the 1/sqrt(head_dim) scale and before the padding mask), scale_qk = 0
 * skips the scale.  No encoder of the registry passes a bias, so models never take this entry: it runs the any-shape kernel and
 * exists so that the reference's own layer goldens (encoder_layer.rs:349-448: hidden 4, 2 heads, a position bias) run on the GPU. */
KjarniErrorCode kjarni_hip_op_attention_biased(int32_t device, const float* qkv, const uint32_t* mask, const float* position_bias,
                                               int32_t bias_seq, int64_t batch, int32_t seq, int32_t heads, int32_t head_dim,
                                               int32_t scale_qk, float mask_value, float* ctx);
/* Pooling of hidden states [batch, seq, hidden] under a u32 mask [batch, seq] (NULL = all ones), optionally followed by L2
 * normalisation: mean / cls / max / last-token pooling as kjarni-transformers/src/pooling/mod.rs:11-68 and
 * cpu/encoder/traits.rs:66-139 (l2_normalize_inplace).  out: [batch, hidden]; hidden <= 1024. */
KjarniErrorCode kjarni_hip_op_pool(int32_t device, const float* hidden_states, const uint32_t* mask, int64_t batch, int32_t seq,
                                   int32_t hidden, KjarniHipPooling pooling, int32_t normalize, float* out);
KjarniErrorCode kjarni_hip_op_layer_norm(int32_t device, const float* x, const float* gamma, const float* beta,
                                         float eps, int64_t rows, int32_t hidden, float* y, int32_t iters,
                                         float* ms_out);
/* y = LayerNorm(x . w^T + bias + residual) * gamma + beta -- the post-norm layer's residual projection with the
 * LayerNorm folded into the GEMM epilogue where the kernel covers the row width n (384, 256), otherwise GEMM + LayerNorm
 * (encoder_layer.rs:129-147, 155-176; layer_norm.rs:37-131).  bias may be NULL. */
KjarniErrorCode kjarni_hip_op_linear_layer_norm(int32_t device, const float* x, const float* w, const float* bias,
                                                const float* residual, const float* gamma, const float* beta, float eps,
                                                int64_t m, int32_t k, int32_t n, float* y, int32_t iters, float* ms_out);

/* ---- per-kernel timing (HIP events on the launch stream) -------------------------
 * profile_begin() switches the encoder into timed mode: every kernel launch of the
 * forward pass is bracketed by two hipEvents on the stream it is launched on.
 * profile_end() synchronises, resolves the events and fills up to `capacity`
 * entries (one per kernel kind of the forward pass); *count_out = entries written.
 * flops / bytes are ALGORITHMIC (2*M*N*K per GEMM, QK^T + PV for attention; every
 * operand read once and every output written once), summed over the launches. */
typedef struct KjarniHipKernelStat {
    const char* kind;   /* static string, e.g. "gemm_fc1" */
    const char* symbol; /* static string: the kernel function launched */
    uint64_t launches;
    double total_ms;
    double flops;
    double bytes;
} KjarniHipKernelStat;

KjarniErrorCode kjarni_hip_encoder_profile_begin(KjarniHipEncoder* enc);
/* Same, timing only the kernel kinds whose bit is set (bit k = k-th entry profile_end() returns):
 * fewer events on the stream when only some kernels are of interest. */
KjarniErrorCode kjarni_hip_encoder_profile_begin_kinds(KjarniHipEncoder* enc, uint32_t kinds_mask);
KjarniErrorCode kjarni_hip_encoder_profile_end(KjarniHipEncoder* enc, KjarniHipKernelStat* stats_out,
                                               size_t capacity, size_t* count_out);

/* ---- cosine scan ------------------------------------------------------------
 * corpus: f32 [n_docs, dim] row-major (the layout of vectors.bin,
 * crates/kjarni-rag/src/segment.rs:240-262).  queries: f32 [n_queries, dim].
 * mode 0 = VectorStore semantics (vector.rs:131-148: dot / max(|q||d|, 1e-9)),
 * mode 1 = Segment semantics (segment.rs:355-371: 0 when |d| < 1e-9). */
typedef enum KjarniHipCosineMode {
    KJARNI_HIP_COSINE_VECTOR_STORE = 0,
    KJARNI_HIP_COSINE_SEGMENT = 1,
} KjarniHipCosineMode;

/* scores_out_dev: f32 [n_queries, n_docs]. */
KjarniErrorCode kjarni_hip_cosine_scores(int32_t device, const float* queries_dev, int32_t n_queries,
                                         const float* corpus_dev, int64_t n_docs, int32_t dim,
                                         KjarniHipCosineMode mode, float* scores_out_dev, void* stream);

/* Top-k of a score matrix [n_queries, n_docs]: score descending, equal scores by
 * ascending document index (stable sort of an index-ordered list, vector.rs:162).
 * idx_out_dev: int64 [n_queries, k], score_out_dev: f32 [n_queries, k]; entries
 * past n_docs are (-1, -inf).  workspace_dev must hold
 * kjarni_hip_cosine_topk_workspace_bytes(n_queries, n_docs, k) bytes. */
size_t kjarni_hip_cosine_topk_workspace_bytes(int32_t n_queries, int64_t n_docs, int32_t k);
KjarniErrorCode kjarni_hip_cosine_topk(int32_t device, const float* scores_dev, int32_t n_queries,
                                       int64_t n_docs, int32_t k, void* workspace_dev,
                                       int64_t* idx_out_dev, float* score_out_dev, void* stream);

/* Scan + selection in one call on device pointers, enqueued on `stream`: per-query top-k with no caller-visible score
 * array.  One query over dim 128 / 256 / 384 / 512 / 768 / 1024 with k <= 256 runs ONE fused pass (every wave keeps its
 * best keys in registers; the scores never exist in memory) and one small merge launch.  2 .. 1 024 queries over >= 20 000
 * documents of those widths, k <= 1 024: a sample bounds every query's k-th best score, ONE bf16 pass over the corpus per
 * 64 queries keeps the documents that can reach it (the pass's rounding error is bounded and the bound relaxed by it), and
 * the survivors' cosines are taken exactly, with the f32 matrix-core scan's arithmetic; other widths (multiples of 16) from
 * 20 queries and 400 000 documents on: the f32 matrix-core scan itself selects.  Anything else runs the two calls above
 * inside the workspace.  Same order and the same scores as kjarni_hip_cosine_scores + kjarni_hip_cosine_topk, bit for bit --
 * except that with 2 .. 19 queries kjarni_hip_cosine_scores streams the corpus and sums in another order than the
 * matrix-core scan whose arithmetic the search then uses (scores within 2e-6).
 * workspace_dev: kjarni_hip_cosine_search_workspace_bytes(n_queries, n_docs, dim, k) bytes. */
size_t kjarni_hip_cosine_search_workspace_bytes(int32_t n_queries, int64_t n_docs, int32_t dim, int32_t k);
KjarniErrorCode kjarni_hip_cosine_search(int32_t device, const float* queries_dev, int32_t n_queries,
                                         const float* corpus_dev, int64_t n_docs, int32_t dim,
                                         KjarniHipCosineMode mode, int32_t k, void* workspace_dev,
                                         int64_t* idx_out_dev, float* score_out_dev, void* stream);

/* Whole search with host buffers (scan + top-k + copies + synchronise):
 * idx_out int64 [n_queries, k], score_out f32 [n_queries, k]; *n_hits_out =
 * min(k, n_docs) (0 for a Segment-mode query whose norm is < 1e-9 is reported
 * per query through idx -1). */
KjarniErrorCode kjarni_hip_cosine_search_host(int32_t device, const float* queries, int32_t n_queries,
                                              const float* corpus, int64_t n_docs, int32_t dim,
                                              KjarniHipCosineMode mode, int32_t k, int64_t* idx_out,
                                              float* score_out, int64_t* n_hits_out);

/* Where the calling thread's last kjarni_searcher_search* / kjarni_hip_index_search call spent its time, in microseconds:
 * out[0] re-opening the index, out[1] the scan over the index's device image (lookup, copies, two launches, wait), out[2] of it
 * between the query's H2D copy and the hits' arrival, out[3] the whole call; the rest is host work (BM25, rank fusion, reading
 * the hits' documents and metadata, for a Searcher the query's tokenisation and embedding).  Up to n <= 4 values. */
void kjarni_hip_search_breakdown(double* out, size_t n);

/* Keyword search (BM25, host side) walks the segments of an index of at least this many documents on a few host threads
 * (contiguous runs of segments, hits merged in segment order: the result does not depend on it).  Default 50 000; 0 = always
 * when there are four segments or more, SIZE_MAX = never.  Process-wide. */
void kjarni_hip_set_keyword_parallel_min_docs(size_t docs);

/* ---- tokenizer -----------------------------------------------------------------
 * The reference tokenises on the host with the HF `tokenizers` crate configured at
 * load time (crates/kjarni-transformers/src/pipeline/encoder/loader.rs:98-115:
 * truncation max_length = max_seq_len, padding BatchLongest, pad id 0).  This handle
 * exposes the same step (BERT WordPiece pipelines) so callers can pre-tokenise and
 * use the token-level entry points above.  Token ids are bit-exact with that crate. */
typedef struct KjarniTokenizer KjarniTokenizer;

typedef struct KjarniTokenBatch {
    uint32_t* ids;            /* [batch, seq] */
    uint32_t* attention_mask; /* [batch, seq] */
    uint32_t* type_ids;       /* [batch, seq] */
    size_t batch;
    size_t seq; /* longest sequence in the batch */
} KjarniTokenBatch;

/* max_length 0 keeps the default (512). */
KjarniErrorCode kjarni_tokenizer_load(const char* tokenizer_json_path, size_t max_length, KjarniTokenizer** out);
void kjarni_tokenizer_free(KjarniTokenizer* tok);
/* texts_b NULL: single sequences ([CLS] a [SEP]); otherwise pairs ([CLS] a [SEP] b [SEP], type ids 0/1).
 * The library allocates *out; free with kjarni_token_batch_free. */
KjarniErrorCode kjarni_tokenizer_encode_batch(const KjarniTokenizer* tok, const char* const* texts_a,
                                              const char* const* texts_b, size_t n, KjarniTokenBatch* out);
void kjarni_token_batch_free(const KjarniTokenBatch* batch);

/* ---- host-side search logic, one function at a time (parity tests) -----------------
 * BM25 tokenizer (crates/kjarni-search/src/bm25.rs:191-197), glob-match as used by
 * MetadataFilter (crates/kjarni-rag/src/index_reader.rs:54-76) and reciprocal-rank fusion
 * (crates/kjarni-search/src/hybrid.rs:3-31; ids ranked best-first, outputs hold up to
 * n_keyword + n_semantic entries). */
KjarniErrorCode kjarni_bm25_tokenize(const char* text, KjarniStringArray* out);
int32_t kjarni_glob_match(const char* pattern, const char* path);
KjarniErrorCode kjarni_rrf_fuse(const size_t* keyword_ids, size_t n_keyword, const size_t* semantic_ids,
                                size_t n_semantic, size_t limit, size_t* ids_out, float* scores_out,
                                size_t* n_out);

/* Retrieval over an on-disk index (the reference's segmented layout) with a caller-supplied query
 * embedding: Searcher::search_with_options (crates/kjarni/src/searcher/model.rs:120-187) minus the
 * encoder and the reranker.  options->mode -1 = hybrid, top_k 0 = 10; use_reranker is ignored.
 * text_query may be NULL in semantic mode, query_emb in keyword mode (which needs no GPU). */
KjarniErrorCode kjarni_hip_index_search(const char* index_path, const char* text_query, const float* query_emb,
                                        size_t dim, const KjarniSearchOptions* options, KjarniSearchResults* out);

/* ---- indexing pipeline, one host-side piece at a time (parity tests) ----------------
 * TextSplitter::split (crates/kjarni-rag/src/splitter.rs:68-120; separator NULL = "\n\n"),
 * Indexer::collect_files (crates/kjarni/src/indexer/model.rs:727-810; reads recursive, include_hidden,
 * extensions, exclude_patterns and max_file_size of the config) and IndexWriter
 * (crates/kjarni-rag/src/index_writer.rs:12-191) driven with caller-supplied embeddings:
 * texts[i], embeddings[i*dimension ..], metadata_json[i] (flat JSON object of strings, or NULL);
 * append 0 = IndexWriter::open, 1 = open_existing; max_docs_per_segment 0 = 10 000. */
KjarniErrorCode kjarni_text_split(const char* text, size_t chunk_size, size_t chunk_overlap, const char* separator,
                                  KjarniStringArray* out);
KjarniErrorCode kjarni_collect_files(const KjarniIndexerConfig* config, const char* const* inputs, size_t num_inputs,
                                     KjarniStringArray* out);
KjarniErrorCode kjarni_index_write(const char* index_path, size_t dimension, size_t max_docs_per_segment,
                                   const char* embedding_model, const char* const* texts,
                                   const char* const* metadata_json, const float* embeddings, size_t n, int32_t append);

/* ---- Whisper, one stage at a time (parity tests, benchmarks) ----------------------------
 * model_dir holds config.json (WhisperConfig, crates/kjarni-models/src/models/whisper/config.rs:11-37),
 * tokenizer.json and model.safetensors with the HF tensor names of config.rs:81-190. */
typedef struct KjarniHipWhisper KjarniHipWhisper;
KjarniErrorCode kjarni_hip_whisper_load(const char* model_dir, int32_t device, KjarniHipWhisper** out);
void kjarni_hip_whisper_free(KjarniHipWhisper* whisper);
KjarniErrorCode kjarni_hip_whisper_dims(const KjarniHipWhisper* whisper, int32_t* d_model, int32_t* n_mels, int32_t* vocab,
                                        int32_t* encoder_frames);
/* compute_mel_spectrogram, MelConfig::whisper() (crates/kjarni-transformers/src/audio/mel.rs:44-136):
 * mel_out f32 [n_mels, 3000]. */
KjarniErrorCode kjarni_hip_whisper_log_mel(KjarniHipWhisper* whisper, const float* samples, size_t num_samples, float* mel_out);
/* WhisperModel::encode_mel (whisper/transcriber.rs:122-141): mel f32 [n_mels, frames] -> hidden_out f32
 * [frames/2, d_model] (may be NULL: the result stays on the device for the decoder). */
KjarniErrorCode kjarni_hip_whisper_encode_mel(KjarniHipWhisper* whisper, const float* mel, int32_t frames, float* hidden_out);
/* log-mel + encode without leaving the device. */
KjarniErrorCode kjarni_hip_whisper_encode_audio(KjarniHipWhisper* whisper, const float* samples, size_t num_samples,
                                                float* hidden_out);
/* Decoder over the current encoder output (cpu_decoder.rs:399-516): begin = cross K/V + empty cache; forward runs
 * n <= 8 new tokens; hidden_out f32 [n, d_model] (final-normed), logits_out f32 [vocab] of the last row. */
KjarniErrorCode kjarni_hip_whisper_decode_begin(KjarniHipWhisper* whisper);
KjarniErrorCode kjarni_hip_whisper_decode_forward(KjarniHipWhisper* whisper, const uint32_t* ids, int32_t n, float* hidden_out,
                                                  float* logits_out);
/* decode_chunk's loop (transcriber.rs:144-240): generated ids, *n_out = how many (may exceed capacity). */
KjarniErrorCode kjarni_hip_whisper_greedy(KjarniHipWhisper* whisper, const uint32_t* prompt, int32_t n_prompt, int32_t timestamps,
                                          size_t max_tokens, uint32_t* ids_out, size_t capacity, size_t* n_out);
KjarniErrorCode kjarni_hip_whisper_decode_text(const KjarniHipWhisper* whisper, const uint32_t* ids, size_t n, int32_t skip_special,
                                               char** out);
/* Host-only: load_audio with the Transcriber's loader config (audio/loader.rs:125-208: mono, 16 kHz, linear
 * resampling) and the ByteLevel decode of `tokenizers` (Tokenizer::decode). */
KjarniErrorCode kjarni_audio_load_wav(const char* path, KjarniFloatArray* out, uint32_t* original_sample_rate);
KjarniErrorCode kjarni_bytelevel_decode(const char* tokenizer_json_path, const uint32_t* ids, size_t n, int32_t skip_special,
                                        char** out);

/* ---- decoder-only generation (Llama / Qwen2 layouts), token level ----------------------------
 * model_dir: config.json (crates/kjarni-models/src/models/llama/config.rs:98-158, qwen/config.rs:80-125) and
 * model.safetensors with the HF tensor names of llama/config.rs:283-330.  weights_dtype: 0 = as stored (BF16 stays
 * bf16 in HBM, other dtypes are widened to f32), 1 = f32, 2 = bf16 (f32 rounded to nearest even).  Arithmetic,
 * activations and the KV cache are f32 either way, as in the reference's bf16 LinearLayer.  max_context <= 0 =
 * max_position_embeddings.  No tokenizer is involved: prompts and results are token ids. */
typedef struct KjarniHipDecoder KjarniHipDecoder;
KjarniErrorCode kjarni_hip_decoder_load(const char* model_dir, int32_t device, int32_t weights_dtype, int32_t max_context,
                                        KjarniHipDecoder** out);
void kjarni_hip_decoder_free(KjarniHipDecoder* decoder);
KjarniErrorCode kjarni_hip_decoder_dims(const KjarniHipDecoder* decoder, int32_t* hidden, int32_t* layers, int32_t* vocab,
                                        int32_t* context, int32_t* weights_bf16, uint64_t* weight_bytes);
KjarniErrorCode kjarni_hip_decoder_reset(KjarniHipDecoder* decoder); /* empty KV cache */
/* Prompt projections that took the 128 x 128-tile GEMM route since load (0 on NULL): a route counter for tests. */
uint64_t kjarni_hip_decoder_tile_gemm_calls(const KjarniHipDecoder* decoder);
/* kjarni_hip_decoder_generate with a repetition penalty / n-gram ban: processors on the device (default) or on a host copy
 * of the logits (0), as kjarni_hip_chat_set_device_sampling. */
void kjarni_hip_decoder_set_device_sampling(KjarniHipDecoder* decoder, int32_t on);
/* CpuDecoder::forward + final norm + lm head (llama/cpu_decoder.rs:196-219): appends n tokens to the cache;
 * the prompt is processed 8 rows at a time and hidden_out receives the final-normed rows of the LAST block,
 * f32 [((n-1) mod 8) + 1, hidden]; logits_out f32 [vocab] of the last position.  Either may be NULL. */
KjarniErrorCode kjarni_hip_decoder_forward(KjarniHipDecoder* decoder, const uint32_t* ids, int32_t n, float* hidden_out,
                                           float* logits_out);
/* run_generation_loop with DecodingStrategy::Greedy (crates/kjarni-transformers/src/decoder/generator.rs:228-381):
 * prefill, then up to max_new_tokens tokens; stops at an eos id of config.json (not emitted) or the context limit.
 * repetition_penalty 1.0 and no_repeat_ngram_size 0 switch those processors off (common/sampling.rs:207-235).
 * on_token (may be NULL; text is NULL here) returning false stops.  *n_out = tokens generated (may exceed capacity). */
KjarniErrorCode kjarni_hip_decoder_generate(KjarniHipDecoder* decoder, const uint32_t* prompt, size_t n_prompt, size_t max_new_tokens,
                                            float repetition_penalty, int32_t no_repeat_ngram_size, KjarniTokenCallbackFn on_token,
                                            void* user_data, uint32_t* ids_out, size_t capacity, size_t* n_out);

/* ---- device memory helpers for callers without a HIP runtime binding --------- */
KjarniErrorCode kjarni_hip_malloc(int32_t device, size_t bytes, void** out_dev);
KjarniErrorCode kjarni_hip_free(int32_t device, void* ptr_dev);
KjarniErrorCode kjarni_hip_memcpy_h2d(int32_t device, void* dst_dev, const void* src, size_t bytes);
KjarniErrorCode kjarni_hip_memcpy_d2h(int32_t device, void* dst, const void* src_dev, size_t bytes);
KjarniErrorCode kjarni_hip_synchronize(int32_t device);

/* ---- chat, one stage at a time (host side; parity tests) ---------------------------------------
 * Byte-level BPE tokenizer: the `tokenizers` 0.22.1 pipeline for Llama 3 / Qwen 2 / GPT-2 style tokenizer.json
 * (added-token extraction, NFC, Split + ByteLevel, BPE with merge ranks and ignore_merges), as the reference calls
 * it: encode(text, add_special_tokens = false) (crates/kjarni-transformers/src/decoder/generator.rs:141-146),
 * right-truncated to max_length when non-zero (pipeline/decoder/loader.rs:115-120), decode(ids, skip_special). */
typedef struct KjarniBpeTokenizer KjarniBpeTokenizer;
KjarniErrorCode kjarni_bpe_tokenizer_load(const char* tokenizer_json_path, KjarniBpeTokenizer** out);
void kjarni_bpe_tokenizer_free(KjarniBpeTokenizer* tokenizer);
KjarniErrorCode kjarni_bpe_tokenizer_encode(const KjarniBpeTokenizer* tokenizer, const char* text, size_t max_length, uint32_t* ids_out,
                                            size_t capacity, size_t* n_out);
KjarniErrorCode kjarni_bpe_tokenizer_decode(const KjarniBpeTokenizer* tokenizer, const uint32_t* ids, size_t n, int32_t skip_special,
                                            char** out);
KjarniErrorCode kjarni_bpe_tokenizer_pre_tokenize(const KjarniBpeTokenizer* tokenizer, const char* text, KjarniStringArray* out);

/* ChatTemplate::apply (crates/kjarni-transformers/src/chat/{llama3,chatml,mistral}.rs): template_kind 0 = Llama 3
 * (for_generation), 1 = ChatML, 2 = Mistral; roles 0 system / 1 user / 2 assistant. */
KjarniErrorCode kjarni_chat_template_apply(int32_t template_kind, const int32_t* roles, const char* const* contents, size_t n, char** out);

/* sample_token's distribution (crates/kjarni-transformers/src/common/sampling.rs:89-108): top-k, top-p, min-p
 * (negative = not set), temperature, softmax; probs_out[vocab] holds 0 for filtered tokens.  kjarni_sample_from_probs
 * is sample_from_probs (:173-184) for a given draw; kjarni_logits_process applies the repetition penalty (:8-27) and
 * the no-repeat-n-gram ban (:29-57, 0 = off) in place. */
KjarniErrorCode kjarni_sampling_distribution(const float* logits, size_t vocab, float temperature, int64_t top_k, float top_p, float min_p,
                                             float* probs_out);
/* The same distribution decided from CANDIDATES only -- every token whose logit is within `tau` of the maximum, the
 * maximum, and the sum of exp(logit - max) over the vocabulary in a non-index order -- which is what the decode loop's
 * device kernels hand to the host per sampled token instead of 4 x vocab bytes of logits.  *decided = 0 when the
 * candidates do not decide it (a filter reaches past them, or a top-p crossing lies within the rounding of the sum):
 * the loop then fetches the logits and runs kjarni_sampling_distribution's path.  Host-side emulation for parity tests. */
KjarniErrorCode kjarni_sampling_distribution_candidates(const float* logits, size_t vocab, float tau, float temperature, int64_t top_k,
                                                        float top_p, float min_p, float* probs_out, int32_t* decided,
                                                        size_t* n_candidates);
uint32_t kjarni_sample_from_probs(const float* probs, size_t vocab, float uniform);
KjarniErrorCode kjarni_logits_process(float* logits, size_t vocab, const uint32_t* tokens, size_t n_tokens, float repetition_penalty,
                                      size_t no_repeat_ngram);

/* resolve_generation_config (crates/kjarni/src/generation/resolution.rs:8-85) over the model defaults
 * (llama/model.rs:373-396, qwen/model.rs:261-282, or generation_config_json when it deserializes as
 * HFGenerationDefaults) and the chat mode (mode < 0: no mode defaults, the bare Generator). */
typedef struct KjarniResolvedGeneration {
    int32_t strategy;          /* 0 greedy, 1 sample, 2 beam search */
    float temperature;
    int64_t top_k;             /* -1 = None */
    float top_p;               /* < 0 = None */
    float min_p;               /* < 0 = None */
    float repetition_penalty;
    size_t no_repeat_ngram_size;
    int64_t max_new_tokens;    /* -1 = None */
    size_t max_length;
    int32_t add_bos_token;
} KjarniResolvedGeneration;
KjarniErrorCode kjarni_generation_resolve(const char* model_type, size_t max_position_embeddings, const char* generation_config_json,
                                          int32_t mode, const KjarniGenerationConfig* runtime, KjarniResolvedGeneration* out);

/* On a live chat handle: the resolved config of a call, the prompt the template produces (roles == NULL and n == 0:
 * Chat::create_conversation; otherwise Chat::history_to_conversation; message appended as the user turn when non-NULL),
 * the token ids DecoderGenerator::encode produces for a prompt (BOS rule included), and the seed of the sampler's
 * generator (the reference draws from rand::thread_rng()). */
KjarniErrorCode kjarni_hip_chat_resolve(const KjarniChat* chat, const KjarniGenerationConfig* runtime, KjarniResolvedGeneration* out);
KjarniErrorCode kjarni_hip_chat_format_prompt(const KjarniChat* chat, const int32_t* roles, const char* const* contents, size_t n,
                                              const char* message, char** out);
KjarniErrorCode kjarni_hip_chat_encode(const KjarniChat* chat, const char* prompt, const KjarniGenerationConfig* runtime, uint32_t* ids_out,
                                       size_t capacity, size_t* n_out);
void kjarni_hip_chat_seed(KjarniChat* chat, uint64_t seed);
/* Sampled tokens and the logits processors: by default everything that is O(vocab) runs on the device and the host
 * decides on a candidate list (a 32-byte header + a few hundred (token, logit) pairs per token); 0 = the logits travel
 * to the host for every token, which is the checker the tests compare with -- same tokens for the same seed.  The
 * counters: tokens decided from candidates / tokens that needed the logits after all. */
void kjarni_hip_chat_set_device_sampling(KjarniChat* chat, int32_t on);
void kjarni_hip_chat_sampling_counters(KjarniChat* chat, uint64_t* from_candidates, uint64_t* from_logits);

#ifdef __cplusplus
}
#endif

#endif /* KJARNI_HIP_H */
