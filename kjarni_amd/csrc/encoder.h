// EncoderModel: a BERT-family encoder resident in HBM + the launch sequence of
// its forward pass.  The HIP analogue of the reference's CPU engine
// (CpuTransformerEncoder, crates/kjarni-transformers/src/cpu/encoder/
// transformer_encoder.rs:30-368) and of its wgpu plug-in point
// (GpuTransformerEncoder, cpu/encoder/gpu.rs:34-383).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "kernels.h"

namespace kjarni {

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct GpuUnavailable : std::runtime_error {
    using std::runtime_error::runtime_error;
};

void hip_check(hipError_t e, const char* what);
int visible_device_count();  // 0 when no HIP device / runtime is usable

struct EncoderConfig {
    std::string model_type;  // "bert" | "distilbert" | "roberta" | "distilroberta" | "mpnet" | "nomic_bert"
    int hidden = 0, layers = 0, heads = 0, inter = 0, vocab = 0, max_pos = 0, type_vocab = 0;
    int pos_offset = 0;
    float eps = 1e-12f;
    GemmEpilogue ffn_act = EPI_BIAS_GELU;
    bool gated_ffn = false;   // SwiGLU: down(silu(gate(x)) * up(x)) (transformer_encoder.rs:163-172)
    float rope_theta = 0.0f;  // > 0: no position table, Q / K rotated instead (transformer_encoder.rs:222-227)
    // classification head (cpu/encoder/classifier.rs:103-202), 0 = none
    int num_labels = 0;
    int head_kind = 0;  // 0 none, 1 dense+tanh (bert.pooler / classifier.dense), 2 dense+relu (pre_classifier), 3 classifier only
    std::vector<std::string> labels;  // id2label order
    std::string config_json;
};

struct DeviceLayer {
    float *wqkv = nullptr, *bqkv = nullptr, *wo = nullptr, *bo = nullptr, *ln1_g = nullptr,
          *ln1_b = nullptr, *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr,
          *ln2_g = nullptr, *ln2_b = nullptr;
    float* wg = nullptr;  // SwiGLU gate [inter, hidden]; w1 is then the up projection
};

// Per-kernel HIP-event timing on the launch stream (bench.py's roofline leg).
// Off by default; when on, every launch of the forward pass is bracketed by two
// events that are resolved at profile_end().
enum KernelKind : int {
    KK_EMBED_LN = 0, KK_GEMM_QKV, KK_ATTENTION, KK_GEMM_OUT, KK_LAYERNORM, KK_GEMM_FC1, KK_GEMM_FC2,
    KK_POOL, KK_HEAD, KK_ROPE, KK_COUNT
};

struct KernelStat {
    const char* kind;    // e.g. "gemm_fc1"
    const char* symbol;  // kernel function the launches ran
    uint64_t launches = 0;
    double total_ms = 0.0;
    double flops = 0.0;  // algorithmic FLOPs summed over the launches
    double bytes = 0.0;  // algorithmic bytes (operands read once + outputs written once)
};

class EncoderModel {
public:
    // Loads <dir>/config.json + <dir>/model.safetensors onto `device`.
    static std::unique_ptr<EncoderModel> load(const std::string& dir, int device);
    ~EncoderModel();
    EncoderModel(const EncoderModel&) = delete;
    EncoderModel& operator=(const EncoderModel&) = delete;

    const EncoderConfig& config() const { return cfg_; }
    int device() const { return device_; }
    size_t weight_bytes() const { return weight_bytes_; }
    void set_chunk_tokens(int64_t t) { chunk_tokens_ = t > 0 ? t : chunk_tokens_; }
    int64_t chunk_tokens() const { return chunk_tokens_; }

    // All pointers are DEVICE pointers on this model's device; work is enqueued
    // on `stream` and not synchronised.  ids/mask/type_ids: u32 [batch, seq].
    // R1: get_hidden_states_batch_from_ids (cpu/encoder/traits.rs:66-139).
    void hidden_states(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids,
                       int64_t batch, int seq, float mask_value, float* out, hipStream_t stream);
    // R1 + R11: hidden states -> pool -> optional L2 (traits.rs:203-225,
    // sentence_encoder/model.rs:201-218).  out: [batch, hidden].
    void embed(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch,
               int seq, PoolMode pool, bool normalize, float mask_value, float* out,
               hipStream_t stream);
    // R12: forward_tokens + classification head -> logits [batch, num_labels]
    // (cross_encoder/model.rs:170-240, sequence_classifier/mod.rs:265-346).
    void logits(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch,
                int seq, float mask_value, float* out, hipStream_t stream);

    // kinds_mask: bit k enables KernelKind k (all ones = every launch of the forward pass).
    void profile_begin(uint32_t kinds_mask = 0xFFFFFFFFu);
    // Synchronises the device, resolves the events; returns KK_COUNT entries.
    std::vector<KernelStat> profile_end();

    // Scratch on this model's device (grown on demand, reused between calls).
    void* scratch(size_t bytes);
    void* scratch2(size_t bytes);

private:
    EncoderModel() = default;
    float* upload(const std::vector<float>& host);
    void ensure_workspace(int64_t tokens, int64_t sentences);
    // Runs embeddings + all layers for `batch` sentences into hidden (device, [batch*seq, H]).
    void forward_chunk(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids,
                       int64_t batch, int seq, float mask_value, float* hidden, hipStream_t stream);
    int64_t sentences_per_chunk(int seq) const;

    struct PendingEvent {
        int kind;
        hipEvent_t start, stop;
    };
    void prof_start(int kind, hipStream_t stream, double flops, double bytes);
    void prof_stop(hipStream_t stream);
    bool prof_on_ = false;
    uint32_t prof_mask_ = 0xFFFFFFFFu;
    bool prof_cur_active_ = false;
    std::vector<PendingEvent> prof_pending_;
    std::vector<hipEvent_t> prof_pool_;
    KernelStat prof_stats_[KK_COUNT];
    hipEvent_t prof_cur_stop_ = nullptr;

    EncoderConfig cfg_;
    int device_ = 0;
    size_t weight_bytes_ = 0;
    int64_t chunk_tokens_ = 131072;
    std::vector<void*> allocs_;

    float *word_ = nullptr, *pos_ = nullptr, *type_ = nullptr, *emb_ln_g_ = nullptr,
          *emb_ln_b_ = nullptr;
    float *rope_cos_ = nullptr, *rope_sin_ = nullptr;  // [max_pos, head_dim]
    std::vector<DeviceLayer> layers_;
    float *head_dense_w_ = nullptr, *head_dense_b_ = nullptr, *head_cls_w_ = nullptr,
          *head_cls_b_ = nullptr;

    // workspace
    int64_t ws_tokens_ = 0, ws_sentences_ = 0;
    float *ws_hidden_ = nullptr, *ws_qkv_ = nullptr, *ws_ctx_ = nullptr, *ws_mid_ = nullptr,
          *ws_feat_ = nullptr;
    void* scratch_ = nullptr;
    size_t scratch_bytes_ = 0;
    void* scratch2_ = nullptr;
    size_t scratch2_bytes_ = 0;
};

}  // namespace kjarni
