// String-level kjarni-ffi surface (include/kjarni.h): Embedder, Reranker,
// Classifier, array frees, cosine similarity, plus the tokenizer handle of
// kjarni_hip.h.  Host logic only: tokenise on the CPU (as the reference does),
// run the encoder on the GPU through EncoderModel, shape results.
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <numeric>

#include "host_util.h"
#include "../../include/kjarni_hip.h"
#include "ffi_common.h"
#include "json.h"
#include "pipeline.h"
#include "registry.h"
#include "unicode.h"
#include "wordpiece.h"

using namespace kjarni;

namespace {

constexpr float kNegInf = -std::numeric_limits<float>::infinity();

bool is_dir(const std::string& p)
{
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

}  // namespace

namespace kjarni {

// Builder semantics of crates/kjarni/src/{embedder,reranker,classifier}/model.rs:
// model_path wins; otherwise resolve the registry name, validate architecture/task,
// require the files on disk (no download here), load.
std::unique_ptr<Pipeline> load_pipeline(const char* cache_dir, const char* model_name, const char* model_path,
                                        const char* default_name, Want want)
{
    std::string dir, name;
    if (model_path) {
        dir = model_path;
        name = dir;
        if (!is_dir(dir)) throw std::runtime_error("Model path does not exist: " + dir);
        if (!model_files_present(dir))
            throw std::runtime_error("Model path '" + dir +
                                     "' must contain config.json, tokenizer.json and model.safetensors");
    } else {
        name = model_name ? model_name : default_name;
        std::string err;
        const RegistryEntry* e = resolve_model(name, err);
        if (!e) throw ModelNotFound(err);
        // validate_for_{embedding,reranking,classification}: encoder architectures only, matching task
        // (crates/kjarni/src/embedder/validation.rs:8-40, reranker/validation.rs:8-45).
        const bool task_ok =
            (want == Want::Embedding)
                ? (e->task == ModelTask::Embedding || e->task == ModelTask::ReRanking || e->task == ModelTask::Classification)
                : (e->task == ModelTask::ReRanking || e->task == ModelTask::Classification);
        if (e->arch != ModelArch::Bert || !task_ok)
            throw std::runtime_error(std::string("Model '") + e->cli_name +
                                     "' is not compatible with this component (the HIP encoder serves bert, "
                                     "distilbert, roberta and mpnet checkpoints of the matching task)");
        const std::string cache = cache_dir ? std::string(cache_dir) : default_cache_dir();
        dir = model_dir_for(*e, cache);
        if (!model_files_present(dir))
            throw ModelNotFound(std::string("Model '") + e->cli_name + "' is not downloaded (expected config.json, "
                                "tokenizer.json and model.safetensors in " + dir +
                                "); this build never downloads models");
        name = e->cli_name;
    }
    auto p = std::make_unique<Pipeline>();
    p->model_name = name;
    // One replica per device of KJARNI_HIP_DEVICES (default: every visible GPU); batches are cut into row blocks.
    p->group = EncoderGroup::load(dir, devices_from_env());
    p->tokenizer = BertTokenizer::from_file(dir + "/tokenizer.json");
    // loader.rs:108-111: truncation max_length = max_seq_len
    p->tokenizer.set_max_length((size_t)p->config().max_pos);
    if (want != Want::Embedding && p->config().head_kind == 0)
        throw std::runtime_error("Model '" + name + "' has no classification head");
    return p;
}

}  // namespace kjarni

namespace {

// cpu/strategy.rs:43-44
float embed_mask_value(size_t tokens) { return (tokens <= 1 || tokens >= 1000) ? kNegInf : -1e9f; }

}  // namespace

namespace kjarni {

// texts -> [n, H] embeddings.  Token-type ids are not passed on this path
// (get_hidden_states_batch_from_ids: embed_tokens(ids, None, 0), traits.rs:79).
std::vector<float> embed_texts(Pipeline& p, const std::vector<std::string>& texts, PoolMode pool, bool normalize)
{
    return embed_encoding(p, p.tokenizer.encode_batch(texts), pool, normalize);
}

std::vector<float> embed_encoding(Pipeline& p, const BatchEncoding& be, PoolMode pool, bool normalize)
{
    const size_t H = (size_t)p.config().hidden;
    std::vector<float> out(be.batch * H);
    if (be.batch == 0 || be.seq == 0) return out;
    // The mask fill follows the size of the WHOLE call, as in the reference, however the rows are spread.
    p.group->embed_host(be.ids.data(), be.attention_mask.data(), nullptr, (int64_t)be.batch, (int)be.seq, pool, normalize,
                        embed_mask_value(be.batch * be.seq), out.data());
    return out;
}

// pairs -> logits [n, num_labels]; forward_tokens always takes the alloc path (mask -1e9).
std::vector<float> pair_logits(Pipeline& p, const BatchEncoding& be)
{
    const size_t L = (size_t)p.config().num_labels;
    std::vector<float> out(be.batch * L);
    if (be.batch == 0 || be.seq == 0) return out;
    p.group->logits_host(be.ids.data(), be.attention_mask.data(), be.type_ids.data(), (int64_t)be.batch, (int)be.seq, -1e9f,
                         out.data());
    return out;
}

}  // namespace kjarni

namespace {

// crates/kjarni/src/embedder/model.rs:247-257
float cosine_k(const float* a, const float* b, size_t n)
{
    float dot = 0.0f, na = 0.0f, nb = 0.0f;
    for (size_t i = 0; i < n; ++i) dot += a[i] * b[i];
    for (size_t i = 0; i < n; ++i) na += a[i] * a[i];
    for (size_t i = 0; i < n; ++i) nb += b[i] * b[i];
    na = std::sqrt(na);
    nb = std::sqrt(nb);
    if (na == 0.0f || nb == 0.0f) return 0.0f;
    return dot / (na * nb);
}

}  // namespace

// ---- arrays ------------------------------------------------------------------

KJARNI_EXPORT void kjarni_float_array_free(const KjarniFloatArray* arr)
{
    if (!arr) return;
    if (arr->data && arr->len > 0) std::free(arr->data);
}

KJARNI_EXPORT void kjarni_float_2d_array_free(const KjarniFloat2DArray* arr)
{
    if (!arr) return;
    if (arr->data && arr->rows > 0 && arr->cols > 0) std::free(arr->data);
}

KJARNI_EXPORT void kjarni_string_free(char* s) { std::free(s); }

KJARNI_EXPORT void kjarni_string_array_free(const KjarniStringArray* arr)
{
    if (!arr) return;
    if (arr->strings && arr->len > 0) {
        for (size_t i = 0; i < arr->len; ++i) std::free(arr->strings[i]);
        std::free(arr->strings);
    }
}

KJARNI_EXPORT float kjarni_cosine_similarity(const float* a, const float* b, size_t len)
{
    if (!a || !b || len == 0) return 0.0f;
    return cosine_k(a, b, len);
}

// ---- Embedder ----------------------------------------------------------------

struct KjarniEmbedder {
    std::unique_ptr<Pipeline> p;
    bool normalize = true;
};

KJARNI_EXPORT KjarniEmbedderConfig kjarni_embedder_config_default(void)
{
    KjarniEmbedderConfig c;
    c.device = KJARNI_DEVICE_CPU;
    c.cache_dir = nullptr;
    c.model_name = nullptr;
    c.model_path = nullptr;
    c.normalize = 1;
    c.quiet = 0;
    return c;
}

KJARNI_EXPORT KjarniErrorCode kjarni_embedder_new(const KjarniEmbedderConfig* config, KjarniEmbedder** out)
{
    if (!out) return KJARNI_ERROR_NULL_POINTER;
    const KjarniEmbedderConfig dflt = kjarni_embedder_config_default();
    const KjarniEmbedderConfig& c = config ? *config : dflt;
    for (const char* s : {c.cache_dir, c.model_name, c.model_path})
        if (s && !valid_utf8(s)) return KJARNI_ERROR_INVALID_UTF8;
    return guarded(KJARNI_ERROR_LOAD_FAILED, [&] {
        auto h = std::make_unique<KjarniEmbedder>();
        h->p = load_pipeline(c.cache_dir, c.model_name, c.model_path, "minilm-l6-v2", Want::Embedding);
        h->normalize = c.normalize != 0;
        *out = h.release();
    });
}

KJARNI_EXPORT void kjarni_embedder_free(KjarniEmbedder* e) { delete e; }

KJARNI_EXPORT KjarniErrorCode kjarni_embedder_encode(KjarniEmbedder* e, const char* text, KjarniFloatArray* out)
{
    if (!e || !text || !out) return KJARNI_ERROR_NULL_POINTER;
    if (!valid_utf8(text)) return KJARNI_ERROR_INVALID_UTF8;
    out->data = nullptr;
    out->len = 0;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        // Embedder::embed -> encode_with(text, "mean", normalize) (embedder/model.rs:118-140)
        std::vector<float> v = embed_texts(*e->p, {std::string(text)}, POOL_MEAN, e->normalize);
        float* d = static_cast<float*>(std::malloc(std::max<size_t>(v.size(), 1) * sizeof(float)));
        if (!d) throw std::bad_alloc();
        std::memcpy(d, v.data(), v.size() * sizeof(float));
        out->data = d;
        out->len = v.size();
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_embedder_encode_batch(KjarniEmbedder* e, const char* const* texts,
                                                           size_t num_texts, KjarniFloat2DArray* out)
{
    if (!e || !texts || !out) return KJARNI_ERROR_NULL_POINTER;
    out->data = nullptr;
    out->rows = 0;
    out->cols = 0;
    if (num_texts == 0) return KJARNI_OK;
    std::vector<std::string> v;
    v.reserve(num_texts);
    for (size_t i = 0; i < num_texts; ++i) {
        if (!texts[i]) return KJARNI_ERROR_NULL_POINTER;
        if (!valid_utf8(texts[i])) return KJARNI_ERROR_INVALID_UTF8;
        v.emplace_back(texts[i]);
    }
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        // encode_batch_flat: mean pool + L2 normalise ALWAYS (sentence_encoder/model.rs:201-218)
        std::vector<float> r = embed_texts(*e->p, v, POOL_MEAN, true);
        float* d = static_cast<float*>(std::malloc(std::max<size_t>(r.size(), 1) * sizeof(float)));
        if (!d) throw std::bad_alloc();
        std::memcpy(d, r.data(), r.size() * sizeof(float));
        out->data = d;
        out->rows = num_texts;
        out->cols = (size_t)e->p->config().hidden;
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_embedder_similarity(KjarniEmbedder* e, const char* t1, const char* t2, float* out)
{
    if (!e || !t1 || !t2 || !out) return KJARNI_ERROR_NULL_POINTER;
    if (!valid_utf8(t1) || !valid_utf8(t2)) return KJARNI_ERROR_INVALID_UTF8;
    *out = 0.0f;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        // Embedder::similarity: embed_batch([t1,t2]) (mean, config normalize) -> cosine (model.rs:178-181)
        std::vector<float> r = embed_texts(*e->p, {std::string(t1), std::string(t2)}, POOL_MEAN, e->normalize);
        const size_t H = (size_t)e->p->config().hidden;
        *out = cosine_k(r.data(), r.data() + H, H);
    });
}

KJARNI_EXPORT size_t kjarni_embedder_dim(const KjarniEmbedder* e) { return e ? (size_t)e->p->config().hidden : 0; }

// ---- Reranker ----------------------------------------------------------------

struct KjarniReranker {
    std::unique_ptr<Pipeline> p;
};

KJARNI_EXPORT void kjarni_rerank_results_free(const KjarniRerankResults* r)
{
    if (!r) return;
    if (r->results && r->len > 0) std::free(r->results);
}

KJARNI_EXPORT KjarniRerankerConfig kjarni_reranker_config_default(void)
{
    KjarniRerankerConfig c;
    c.device = KJARNI_DEVICE_CPU;
    c.cache_dir = nullptr;
    c.model_name = nullptr;
    c.model_path = nullptr;
    c.quiet = 0;
    return c;
}

KJARNI_EXPORT KjarniErrorCode kjarni_reranker_new(const KjarniRerankerConfig* config, KjarniReranker** out)
{
    if (!out) return KJARNI_ERROR_NULL_POINTER;
    const KjarniRerankerConfig dflt = kjarni_reranker_config_default();
    const KjarniRerankerConfig& c = config ? *config : dflt;
    for (const char* s : {c.cache_dir, c.model_name, c.model_path})
        if (s && !valid_utf8(s)) return KJARNI_ERROR_INVALID_UTF8;
    return guarded(KJARNI_ERROR_LOAD_FAILED, [&] {
        auto h = std::make_unique<KjarniReranker>();
        h->p = load_pipeline(c.cache_dir, c.model_name, c.model_path, "minilm-l6-v2-cross-encoder", Want::Reranking);
        *out = h.release();
    });
}

KJARNI_EXPORT void kjarni_reranker_free(KjarniReranker* r) { delete r; }

namespace kjarni {

// CrossEncoder::predict_pairs (cross_encoder/model.rs:170-240): logits column 0.
std::vector<float> rerank_scores(Pipeline& p, const std::string& query, const std::vector<std::string>& docs)
{
    std::vector<std::pair<std::string, std::string>> pairs;
    pairs.reserve(docs.size());
    for (const std::string& d : docs) pairs.emplace_back(query, d);
    const BatchEncoding be = p.tokenizer.encode_batch_pairs(pairs);
    const std::vector<float> logits = pair_logits(p, be);
    const size_t L = (size_t)p.config().num_labels;
    std::vector<float> scores(docs.size());
    for (size_t i = 0; i < docs.size(); ++i) scores[i] = logits[i * L];
    return scores;
}

}  // namespace kjarni

namespace {

KjarniErrorCode rerank_impl(KjarniReranker* r, const char* query, const char* const* documents, size_t num_docs,
                            bool have_k, size_t top_k, KjarniRerankResults* out)
{
    if (!r || !query || !documents || !out) return KJARNI_ERROR_NULL_POINTER;
    out->results = nullptr;
    out->len = 0;
    if (num_docs == 0) return KJARNI_OK;
    if (!valid_utf8(query)) return KJARNI_ERROR_INVALID_UTF8;
    std::vector<std::string> docs;
    docs.reserve(num_docs);
    for (size_t i = 0; i < num_docs; ++i) {
        if (!documents[i]) return KJARNI_ERROR_NULL_POINTER;
        if (!valid_utf8(documents[i])) return KJARNI_ERROR_INVALID_UTF8;
        docs.emplace_back(documents[i]);
    }
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        const std::vector<float> scores = rerank_scores(*r->p, query, docs);
        // CrossEncoder::rerank: stable sort by score descending, partial_cmp (NaN == Equal)
        // (cross_encoder/model.rs:251-252); Reranker::rerank_with_config truncates to top_k
        // (crates/kjarni/src/reranker/model.rs:270-273).
        std::vector<size_t> order(num_docs);
        std::iota(order.begin(), order.end(), (size_t)0);
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return scores[a] > scores[b]; });
        size_t n = num_docs;
        if (have_k && top_k < n) n = top_k;
        if (n == 0) return;
        auto* res = static_cast<KjarniRerankResult*>(std::malloc(n * sizeof(KjarniRerankResult)));
        if (!res) throw std::bad_alloc();
        for (size_t i = 0; i < n; ++i) {
            res[i].index = order[i];
            res[i].score = scores[order[i]];
        }
        out->results = res;
        out->len = n;
    });
}

}  // namespace

KJARNI_EXPORT KjarniErrorCode kjarni_reranker_score(KjarniReranker* r, const char* query, const char* document, float* out)
{
    if (!r || !query || !document || !out) return KJARNI_ERROR_NULL_POINTER;
    if (!valid_utf8(query) || !valid_utf8(document)) return KJARNI_ERROR_INVALID_UTF8;
    *out = 0.0f;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        *out = rerank_scores(*r->p, query, {std::string(document)})[0];
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_reranker_rerank(KjarniReranker* r, const char* query, const char* const* documents,
                                                     size_t num_docs, KjarniRerankResults* out)
{
    return rerank_impl(r, query, documents, num_docs, false, 0, out);
}

KJARNI_EXPORT KjarniErrorCode kjarni_reranker_rerank_top_k(KjarniReranker* r, const char* query,
                                                           const char* const* documents, size_t num_docs, size_t top_k,
                                                           KjarniRerankResults* out)
{
    return rerank_impl(r, query, documents, num_docs, true, top_k, out);
}

// ---- Classifier --------------------------------------------------------------

struct KjarniClassifier {
    std::unique_ptr<Pipeline> p;
    std::vector<std::string> labels;
    bool multi_label = false;
};

KJARNI_EXPORT void kjarni_class_results_free(const KjarniClassResults* r)
{
    if (!r) return;
    if (r->results && r->len > 0) {
        for (size_t i = 0; i < r->len; ++i) std::free(r->results[i].label);
        std::free(r->results);
    }
}

KJARNI_EXPORT KjarniClassifierConfig kjarni_classifier_config_default(void)
{
    KjarniClassifierConfig c;
    c.device = KJARNI_DEVICE_CPU;
    c.cache_dir = nullptr;
    c.model_name = nullptr;
    c.model_path = nullptr;
    c.labels = nullptr;
    c.num_labels = 0;
    c.multi_label = 0;
    c.quiet = 0;
    return c;
}

KJARNI_EXPORT KjarniErrorCode kjarni_classifier_new(const KjarniClassifierConfig* config, KjarniClassifier** out)
{
    if (!out) return KJARNI_ERROR_NULL_POINTER;
    const KjarniClassifierConfig dflt = kjarni_classifier_config_default();
    const KjarniClassifierConfig& c = config ? *config : dflt;
    for (const char* s : {c.cache_dir, c.model_name, c.model_path})
        if (s && !valid_utf8(s)) return KJARNI_ERROR_INVALID_UTF8;
    std::vector<std::string> custom;
    if (c.labels && c.num_labels > 0) {
        for (size_t i = 0; i < c.num_labels; ++i) {
            if (!c.labels[i]) return KJARNI_ERROR_NULL_POINTER;
            if (!valid_utf8(c.labels[i])) return KJARNI_ERROR_INVALID_UTF8;
            custom.emplace_back(c.labels[i]);
        }
    }
    return guarded(KJARNI_ERROR_LOAD_FAILED, [&] {
        auto h = std::make_unique<KjarniClassifier>();
        // Default name "sentiment" (kjarni-ffi/src/classifier.rs:133) is not a registry name in the
        // reference either: it resolves to "Unknown model 'sentiment'. Did you mean: ...".
        h->p = load_pipeline(c.cache_dir, c.model_name, c.model_path, "sentiment", Want::Classification);
        const EncoderConfig& mc = h->p->config();
        h->labels = mc.labels;
        if (h->labels.empty())
            for (int i = 0; i < mc.num_labels; ++i) h->labels.push_back("LABEL_" + std::to_string(i));
        if (!custom.empty()) {
            // crates/kjarni/src/classifier/model.rs:123-137
            if (!mc.labels.empty() && custom.size() != mc.labels.size())
                throw std::runtime_error("Model expects " + std::to_string(mc.labels.size()) + " labels but " +
                                         std::to_string(custom.size()) + " provided");
            if (custom.size() != (size_t)mc.num_labels)
                throw std::runtime_error("Model has " + std::to_string(mc.num_labels) + " outputs but " +
                                         std::to_string(custom.size()) + " labels provided");
            h->labels = custom;
        }
        // model.rs:138-150: explicit multi_label wins, else config problem_type decides.
        bool cfg_multi = false;
        try {
            cfg_multi = Json::parse(mc.config_json).get_string("problem_type", "") == "multi_label_classification";
        } catch (...) {
        }
        h->multi_label = c.multi_label != 0 || cfg_multi;
        *out = h.release();
    });
}

KJARNI_EXPORT void kjarni_classifier_free(KjarniClassifier* c) { delete c; }

KJARNI_EXPORT KjarniErrorCode kjarni_classifier_classify(KjarniClassifier* c, const char* text, KjarniClassResults* out)
{
    if (!c || !text || !out) return KJARNI_ERROR_NULL_POINTER;
    if (!valid_utf8(text)) return KJARNI_ERROR_INVALID_UTF8;
    out->results = nullptr;
    out->len = 0;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        Pipeline& p = *c->p;
        // SequenceClassifier::predict_logits (sequence_classifier/mod.rs:265-346): single sequences,
        // type ids from the tokenizer (all 0), alloc path.
        const BatchEncoding be = p.tokenizer.encode_batch({std::string(text)});
        std::vector<float> logits = pair_logits(p, be);
        const size_t L = logits.size();
        std::vector<float> probs(L);
        if (c->multi_label) {
            for (size_t i = 0; i < L; ++i) probs[i] = 1.0f / (1.0f + std::exp(-logits[i]));  // model.rs:529
        } else {
            // softmax_inplace (activations.rs:223-242)
            float mx = kNegInf;
            for (float v : logits) mx = std::max(mx, v);
            float sum = 0.0f;
            for (size_t i = 0; i < L; ++i) {
                probs[i] = std::exp(logits[i] - mx);
                sum += probs[i];
            }
            if (sum > 0.0f) {
                const float sc = 1.0f / sum;
                for (float& v : probs) v *= sc;
            }
        }
        // ClassificationResult::from_scores_with_labels: all labels, stable sort by score descending
        // (crates/kjarni/src/classifier/types.rs:106-131).
        std::vector<size_t> order(L);
        std::iota(order.begin(), order.end(), (size_t)0);
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return probs[a] > probs[b]; });
        if (L == 0) throw std::runtime_error("No results");
        auto* res = static_cast<KjarniClassResult*>(std::calloc(L, sizeof(KjarniClassResult)));
        if (!res) throw std::bad_alloc();
        for (size_t i = 0; i < L; ++i) {
            res[i].label = dup_cstr(c->labels[order[i]]);
            res[i].score = probs[order[i]];
        }
        out->results = res;
        out->len = L;
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_classifier_labels(const KjarniClassifier* c, KjarniStringArray* out)
{
    if (!c || !out) return KJARNI_ERROR_NULL_POINTER;
    out->strings = nullptr;
    out->len = 0;
    return guarded(KJARNI_ERROR_UNKNOWN, [&] {
        const size_t n = c->labels.size();
        if (n == 0) return;
        char** arr = static_cast<char**>(std::calloc(n, sizeof(char*)));
        if (!arr) throw std::bad_alloc();
        for (size_t i = 0; i < n; ++i) arr[i] = dup_cstr(c->labels[i]);
        out->strings = arr;
        out->len = n;
    });
}

KJARNI_EXPORT size_t kjarni_classifier_num_labels(const KjarniClassifier* c) { return c ? c->labels.size() : 0; }

// ---- tokenizer handle (kjarni_hip.h) -------------------------------------------

struct KjarniTokenizer {
    BertTokenizer tok;
};

KJARNI_EXPORT KjarniErrorCode kjarni_tokenizer_load(const char* tokenizer_json_path, size_t max_length,
                                                    KjarniTokenizer** out)
{
    if (!tokenizer_json_path || !out) return KJARNI_ERROR_NULL_POINTER;
    *out = nullptr;
    return guarded(KJARNI_ERROR_LOAD_FAILED, [&] {
        auto h = std::make_unique<KjarniTokenizer>();
        h->tok = BertTokenizer::from_file(tokenizer_json_path);
        if (max_length > 0) h->tok.set_max_length(max_length);
        *out = h.release();
    });
}

KJARNI_EXPORT void kjarni_tokenizer_free(KjarniTokenizer* t) { delete t; }

KJARNI_EXPORT KjarniErrorCode kjarni_tokenizer_encode_batch(const KjarniTokenizer* t, const char* const* texts_a,
                                                            const char* const* texts_b, size_t n,
                                                            KjarniTokenBatch* out)
{
    if (!t || !out || (n > 0 && !texts_a)) return KJARNI_ERROR_NULL_POINTER;
    std::memset(out, 0, sizeof(*out));
    for (size_t i = 0; i < n; ++i) {
        if (!texts_a[i] || (texts_b && !texts_b[i])) return KJARNI_ERROR_NULL_POINTER;
        if (!valid_utf8(texts_a[i]) || (texts_b && !valid_utf8(texts_b[i]))) return KJARNI_ERROR_INVALID_UTF8;
    }
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        BatchEncoding be;
        if (texts_b) {
            std::vector<std::pair<std::string, std::string>> pairs;
            for (size_t i = 0; i < n; ++i) pairs.emplace_back(texts_a[i], texts_b[i]);
            be = t->tok.encode_batch_pairs(pairs);
        } else {
            std::vector<std::string> v;
            for (size_t i = 0; i < n; ++i) v.emplace_back(texts_a[i]);
            be = t->tok.encode_batch(v);
        }
        const size_t cnt = be.batch * be.seq;
        if (cnt == 0) {
            out->batch = be.batch;
            return;
        }
        auto* buf = static_cast<uint32_t*>(std::malloc(3 * cnt * sizeof(uint32_t)));
        if (!buf) throw std::bad_alloc();
        std::memcpy(buf, be.ids.data(), cnt * 4);
        std::memcpy(buf + cnt, be.attention_mask.data(), cnt * 4);
        std::memcpy(buf + 2 * cnt, be.type_ids.data(), cnt * 4);
        out->ids = buf;
        out->attention_mask = buf + cnt;
        out->type_ids = buf + 2 * cnt;
        out->batch = be.batch;
        out->seq = be.seq;
    });
}

KJARNI_EXPORT void kjarni_token_batch_free(const KjarniTokenBatch* b)
{
    if (b && b->ids) std::free(b->ids);
}
