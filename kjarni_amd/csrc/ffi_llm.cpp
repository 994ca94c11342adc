// Token-level C ABI of the decoder-only path (kjarni_hip.h).  The reference's string-level chat group
// (crates/kjarni-ffi/src/chat.rs) sits on top of the same loop (crates/kjarni-transformers/src/decoder/
// generator.rs:228-381) plus a BPE tokenizer and chat templates, which are not built here.
#include <cstring>
#include <mutex>

#include <cstdlib>
#include "llm_kernels.h"
#include "../../include/kjarni_hip.h"
#include "ffi_common.h"
#include "llm.h"

using namespace kjarni;

struct KjarniHipDecoder {
    std::unique_ptr<LlmModel> model;
    std::mutex mu;
};

KJARNI_EXPORT KjarniErrorCode kjarni_hip_decoder_load(const char* model_dir, int32_t device, int32_t weights_dtype, int32_t max_context,
                                                      KjarniHipDecoder** out)
{
    if (!model_dir || !out) return KJARNI_ERROR_NULL_POINTER;
    if (weights_dtype < 0 || weights_dtype > 2) return KJARNI_ERROR_INVALID_CONFIG;
    return guarded(KJARNI_ERROR_LOAD_FAILED, [&] {
#ifdef KJARNI_TUNING
        if (const char* v = std::getenv("KJARNI_HIP_LLM_GEMV")) set_llm_gemv_variant(std::atoi(v));  // kernel A/B measurements
#endif
        auto h = std::make_unique<KjarniHipDecoder>();
        h->model = LlmModel::load(model_dir, device, weights_dtype, max_context);
        *out = h.release();
    });
}

KJARNI_EXPORT void kjarni_hip_decoder_free(KjarniHipDecoder* d) { delete d; }

KJARNI_EXPORT KjarniErrorCode kjarni_hip_decoder_dims(const KjarniHipDecoder* d, int32_t* hidden, int32_t* layers, int32_t* vocab,
                                                      int32_t* context, int32_t* weights_bf16, uint64_t* weight_bytes)
{
    if (!d) return KJARNI_ERROR_NULL_POINTER;
    if (hidden) *hidden = d->model->config().hidden;
    if (layers) *layers = d->model->config().layers;
    if (vocab) *vocab = d->model->config().vocab;
    if (context) *context = d->model->context();
    if (weights_bf16) *weights_bf16 = d->model->bf16() ? 1 : 0;
    if (weight_bytes) *weight_bytes = (uint64_t)d->model->weight_bytes();
    return KJARNI_OK;
}

KJARNI_EXPORT uint64_t kjarni_hip_decoder_tile_gemm_calls(const KjarniHipDecoder* d) { return d ? d->model->tile_gemm_calls() : 0; }

KJARNI_EXPORT void kjarni_hip_decoder_set_device_sampling(KjarniHipDecoder* d, int32_t on)
{
    if (d) d->model->set_device_sampling(on != 0);
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_decoder_reset(KjarniHipDecoder* d)
{
    if (!d) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        std::lock_guard<std::mutex> lock(d->mu);
        d->model->reset();
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_decoder_forward(KjarniHipDecoder* d, const uint32_t* ids, int32_t n, float* hidden_out,
                                                         float* logits_out)
{
    if (!d || !ids) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        std::lock_guard<std::mutex> lock(d->mu);
        d->model->forward(ids, n);
        if (hidden_out) d->model->last_hidden(hidden_out, n < 8 ? n : 8);
        if (logits_out) d->model->logits_to_host(logits_out);
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_decoder_generate(KjarniHipDecoder* d, const uint32_t* prompt, size_t n_prompt,
                                                          size_t max_new_tokens, float repetition_penalty, int32_t no_repeat_ngram_size,
                                                          KjarniTokenCallbackFn on_token, void* user_data, uint32_t* ids_out,
                                                          size_t capacity, size_t* n_out)
{
    if (!d || !prompt || !n_out || (capacity && !ids_out)) return KJARNI_ERROR_NULL_POINTER;
    *n_out = 0;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        std::lock_guard<std::mutex> lock(d->mu);
        std::function<bool(uint32_t)> cb;
        if (on_token)
            cb = [&](uint32_t id) {
                KjarniToken t;
                t.text = nullptr;  // token-level API: no tokenizer behind it
                t.token_id = id;
                t.is_special = false;
                return on_token(t, user_data);
            };
        const std::vector<uint32_t> ids = d->model->generate(std::vector<uint32_t>(prompt, prompt + n_prompt), max_new_tokens,
                                                             repetition_penalty, no_repeat_ngram_size, cb);
        *n_out = ids.size();
        if (capacity) std::memcpy(ids_out, ids.data(), std::min(capacity, ids.size()) * sizeof(uint32_t));
    });
}
