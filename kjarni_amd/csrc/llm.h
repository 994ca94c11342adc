// LlmModel: a decoder-only transformer (Llama / Qwen2 layouts) resident in HBM with an f32 KV cache, and the
// greedy generation loop.
//
//   config + tensor names   crates/kjarni-models/src/models/llama/config.rs:98-330, qwen/config.rs:80-275
//   layer                   crates/kjarni-transformers/src/cpu/decoder/rope_decoder_layer.rs:18-41
//   model forward           crates/kjarni-models/src/models/llama/cpu_decoder.rs:142-219
//   generation loop         crates/kjarni-transformers/src/decoder/generator.rs:228-381 (DecodingStrategy::Greedy)
#pragma once
#include "device_arena.h"
#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "encoder.h"
#include "sampling.h"

namespace kjarni {

struct LlmConfig {
    std::string model_type;
    int hidden = 0, layers = 0, heads = 0, kv_heads = 0, head_dim = 0, inter = 0, vocab = 0, max_pos = 0;
    float eps = 1e-5f, rope_theta = 500000.0f;
    bool tie_embeddings = true;
    bool has_rope_scaling = false;
    std::string rope_type;
    float rope_factor = 1.0f, rope_low = 1.0f, rope_high = 4.0f;
    int rope_original_max = 8192;
    std::vector<uint32_t> eos_ids;
    bool has_bos = false;
    uint32_t bos_id = 0;
    static LlmConfig from_json(const std::string& text);
};

// One run of run_generation_loop (generator.rs:228-381).
struct GenerateOptions {
    size_t max_new_tokens = 0;
    size_t max_len = 0;  // prompt + generated cap (generator.rs:243-246); 0 = prompt + max_new_tokens
    float repetition_penalty = 1.0f;
    int no_repeat_ngram = 0;
    bool sample = false;  // DecodingStrategy::Sample(params) instead of Greedy
    SamplingParams sampling;
    std::vector<uint32_t> stop_ids;  // empty: every eos_token_id of config.json
    std::function<float()> uniform;  // the draw in [0, 1) for each sampled token
};

class LlmModel {
public:
    // weights: 0 = as stored (BF16 stays bf16, everything else f32), 1 = f32, 2 = bf16 (f32 rounded to nearest even).
    static std::unique_ptr<LlmModel> load(const std::string& dir, int device, int weights, int max_context);
    ~LlmModel();
    LlmModel(const LlmModel&) = delete;
    LlmModel& operator=(const LlmModel&) = delete;

    const LlmConfig& config() const { return cfg_; }
    bool bf16() const { return bf16_; }
    size_t weight_bytes() const { return weight_bytes_; }
    int context() const { return cache_cap_; }
    int cache_len() const { return cache_len_; }
    // Prompt projections that ran the encoder's 128 x 128-tile GEMM since load (the other route is the 64 x 64 prompt kernel):
    // lets a test assert which route a geometry took.
    uint64_t tile_gemm_calls() const { return tile_gemm_calls_; }
    // Sampled decoding / logits processors: the O(vocab) work runs on the device and the host decides on a candidate list
    // (default); off = the logits travel to the host every token (the checker in tests).  The counters say how many tokens
    // were decided from candidates and how many needed the logits after all.
    void set_device_sampling(bool on) { device_sampling_ = on; }
    uint64_t tokens_from_candidates() const { return tokens_from_candidates_; }
    uint64_t tokens_from_logits() const { return tokens_from_logits_; }

    void reset();  // empty KV cache
    // Appends n tokens (any n: processed 8 rows at a time); the logits of the last position stay on the device.
    void forward(const uint32_t* ids, int n);
    void last_hidden(float* out, int rows) const;  // final-normed hidden states of the last (<= 8-row) pass
    void logits_to_host(float* out) const;
    uint32_t argmax();

    // run_generation_loop with the Greedy strategy: returns the generated ids (stop token excluded).
    std::vector<uint32_t> generate(const std::vector<uint32_t>& prompt, size_t max_new_tokens, float repetition_penalty,
                                   int no_repeat_ngram, const std::function<bool(uint32_t)>& on_token);
    // The same loop with a sampling strategy, explicit stop tokens and the max_length cap.
    std::vector<uint32_t> generate(const std::vector<uint32_t>& prompt, const GenerateOptions& options,
                                   const std::function<bool(uint32_t)>& on_token);

private:
    LlmModel() = default;
    void* upload_weight(const std::vector<float>& host);  // f32 or bf16 according to bf16_
    float* upload_f32(const std::vector<float>& host);
    float* dalloc(size_t floats);
    void pass(const uint32_t* ids_dev, int n, bool device_pos);
    void prefill_rows(const uint32_t* ids_host, int n);  // n new tokens through the matrix-core GEMMs
    void enqueue_argmax(bool record);
    hipGraphExec_t step_graph();

    struct Layer {
        void *wqkv, *wo, *gate, *up, *down;
        float *bqkv, *ln1, *ln2;
        float *k_cache, *v_cache;
    };
    LlmConfig cfg_;
    int device_ = 0;
    bool bf16_ = false;
    size_t weight_bytes_ = 0;
    DeviceArena arena_;   // every device buffer of the model
    std::vector<Layer> layers_;
    void *embed_ = nullptr, *lm_head_ = nullptr;
    float *final_norm_ = nullptr, *cos_ = nullptr, *sin_ = nullptr;
    // workspace
    float *h_ = nullptr, *q_ = nullptr, *ctx_ = nullptr, *mid_ = nullptr, *last_ = nullptr, *logits_ = nullptr, *att_scratch_ = nullptr;
    uint32_t* ids_ = nullptr;
    int32_t *token_ = nullptr, *hist_ = nullptr;
    int *pos_ = nullptr, *count_ = nullptr;
    unsigned long long* best_ = nullptr;
    int cache_len_ = 0, cache_cap_ = 0, last_rows_ = 0, hist_cap_ = 0, splits_ = 16;
    // prefill workspace (allocated on first use)
    float *ph_ = nullptr, *pn_ = nullptr, *pq_ = nullptr, *pctx_ = nullptr, *pg_ = nullptr, *pu_ = nullptr;
    uint32_t* pids_ = nullptr;
    float* pw32_ = nullptr;    // f32 copy of the weight matrix a long prompt's GEMM is working on (bf16 checkpoints)
    float* psplit_ = nullptr;  // K-slice partial tiles of the prompt GEMMs (short prompts)
    float* host_logits_ = nullptr;  // pinned
    // sampled decoding on the device (allocated on first use): candidate header + list and its pinned mirror, the token
    // history with per-token counts and the list of distinct tokens (logits processors)
    static constexpr int kCandCap = 4096, kCandFirst = 512;
    void* samp_scratch_ = nullptr;
    uint8_t *samp_out_ = nullptr, *samp_host_ = nullptr;
    int32_t *samp_tokens_ = nullptr, *samp_distinct_ = nullptr;
    int *samp_counts_ = nullptr, *samp_ndistinct_ = nullptr;
    bool device_sampling_ = true;
    uint64_t tokens_from_candidates_ = 0, tokens_from_logits_ = 0;
    int prefill_cap_ = 0;
    uint64_t tile_gemm_calls_ = 0;
    hipStream_t stream_ = nullptr;
    hipGraphExec_t graph_ = nullptr;
};

}  // namespace kjarni
