// extern "C" Chat group: crates/kjarni-ffi/src/chat.rs:13-758, entry point for entry point, plus the
// one-stage-at-a-time hooks of kjarni_hip.h (tokenizer, templates, sampling, config resolution).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "host_util.h"
#include "../../include/kjarni_hip.h"
#include "bpe.h"
#include "chat.h"
#include "ffi_common.h"
#include "sampling.h"
#include "unicode.h"

using namespace kjarni;

struct KjarniChat {
    std::unique_ptr<Chat> inner;
};

// chat.rs:163-166: owns the history, points back at the parent (which must outlive it).
struct KjarniChatConversation {
    KjarniChat* chat = nullptr;
    std::vector<ChatMessage> history;
};

struct KjarniBpeTokenizer {
    BpeTokenizer tok;
};

namespace {

// CString::new(response): an interior NUL is an error here, not an empty string (chat.rs:303-313).
char* response_cstr(const std::string& s)
{
    if (s.find('\0') != std::string::npos) throw std::runtime_error("Response contained null byte");
    char* p = static_cast<char*>(std::malloc(s.size() + 1));
    if (!p) throw std::bad_alloc();
    std::memcpy(p, s.data(), s.size());
    p[s.size()] = '\0';
    return p;
}

GenerationOverrides to_overrides(const KjarniGenerationConfig& c)  // chat.rs:82-123
{
    GenerationOverrides o;
    if (c.temperature >= 0.0f) o.temperature = Opt<float>(c.temperature);
    if (c.top_k >= 0) o.top_k = Opt<size_t>((size_t)c.top_k);
    if (c.top_p >= 0.0f) o.top_p = Opt<float>(c.top_p);
    if (c.min_p >= 0.0f) o.min_p = Opt<float>(c.min_p);
    if (c.repetition_penalty >= 0.0f) o.repetition_penalty = Opt<float>(c.repetition_penalty);
    if (c.max_new_tokens >= 0) o.max_new_tokens = Opt<size_t>((size_t)c.max_new_tokens);
    if (c.do_sample == 0) o.do_sample = Opt<bool>(false);
    else if (c.do_sample == 1) o.do_sample = Opt<bool>(true);
    return o;
}

// Generation errors carry ChatError::GenerationFailed's prefix (chat/types.rs:33-34).
template <class F>
KjarniErrorCode generation_guarded(F&& fn)
{
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        try {
            fn();
        } catch (const GpuUnavailable&) {
            throw;
        } catch (const std::exception& e) {
            const std::string what = e.what();
            if (what.rfind("generation failed: ", 0) == 0 || what == "Response contained null byte") throw;
            throw std::runtime_error("generation failed: " + what);
        }
    });
}

std::vector<ChatMessage> conversation_with_user(const Chat& chat, const char* message)
{
    std::vector<ChatMessage> c = chat.create_conversation();
    c.push_back({ChatRole::User, message});
    return c;
}

// The streaming loop of chat.rs:360-391: cancellation is looked at before each token is handed over.
std::string stream_tokens(Chat& chat, const std::string& prompt, const GenerationOverrides& o, KjarniStreamCallbackFn cb, void* user_data,
                          const KjarniCancelToken* cancel)
{
    std::string full;
    chat.generate_stream(prompt, o, [&](const std::string& text) {
        if (kjarni_cancel_token_is_cancelled(cancel)) return false;
        full += text;
        if (text.find('\0') != std::string::npos) return true;  // tokens with a NUL byte are skipped, not fatal
        return cb(text.c_str(), user_data);
    });
    return full;
}

}  // namespace

KJARNI_EXPORT KjarniChatConfig kjarni_chat_config_default(void)
{
    KjarniChatConfig c;
    c.device = KJARNI_DEVICE_CPU;
    c.cache_dir = nullptr;
    c.model_name = nullptr;
    c.model_path = nullptr;
    c.system_prompt = nullptr;
    c.mode = 0;
    c.quiet = 0;
    return c;
}

KJARNI_EXPORT KjarniGenerationConfig kjarni_generation_config_default(void)
{
    KjarniGenerationConfig c;
    c.temperature = -1.0f;
    c.top_k = -1;
    c.top_p = -1.0f;
    c.min_p = -1.0f;
    c.repetition_penalty = -1.0f;
    c.max_new_tokens = -1;
    c.do_sample = -1;
    return c;
}

KJARNI_EXPORT KjarniErrorCode kjarni_chat_new(const KjarniChatConfig* config, KjarniChat** out)
{
    if (!out) return KJARNI_ERROR_NULL_POINTER;
    const KjarniChatConfig dflt = kjarni_chat_config_default();
    const KjarniChatConfig& c = config ? *config : dflt;
    if (!c.model_name) {
        set_last_error("model_name is required");
        return KJARNI_ERROR_INVALID_CONFIG;
    }
    if (!valid_utf8(c.model_name)) return KJARNI_ERROR_INVALID_UTF8;
    if (c.cache_dir && !valid_utf8(c.cache_dir)) return KJARNI_ERROR_INVALID_UTF8;
    if (c.system_prompt && !valid_utf8(c.system_prompt)) return KJARNI_ERROR_INVALID_UTF8;
    if (c.model_path && !valid_utf8(c.model_path)) return KJARNI_ERROR_INVALID_UTF8;
    return guarded(KJARNI_ERROR_LOAD_FAILED, [&] {
        const std::string system = c.system_prompt ? c.system_prompt : "";
        auto h = std::make_unique<KjarniChat>();
        h->inner = Chat::create(c.model_name, c.model_path ? c.model_path : "", c.cache_dir ? c.cache_dir : "",
                                c.system_prompt ? &system : nullptr, c.mode, c.quiet != 0);
        *out = h.release();
    });
}

KJARNI_EXPORT void kjarni_chat_free(KjarniChat* chat) { delete chat; }

KJARNI_EXPORT KjarniErrorCode kjarni_chat_send(KjarniChat* chat, const char* message, const KjarniGenerationConfig* gen_config, char** out)
{
    if (!chat || !message || !out) return KJARNI_ERROR_NULL_POINTER;
    if (!valid_utf8(message)) return KJARNI_ERROR_INVALID_UTF8;
    const KjarniErrorCode rc = generation_guarded([&] {
        Chat& c = *chat->inner;
        const std::string prompt = c.format_prompt(conversation_with_user(c, message));
        *out = response_cstr(c.generate(prompt, gen_config ? to_overrides(*gen_config) : GenerationOverrides()));
    });
    if (rc != KJARNI_OK) *out = nullptr;
    return rc;
}

KJARNI_EXPORT KjarniErrorCode kjarni_chat_stream(KjarniChat* chat, const char* message, const KjarniGenerationConfig* gen_config,
                                                 KjarniStreamCallbackFn callback, void* user_data, const KjarniCancelToken* cancel_token)
{
    if (!chat || !message) return KJARNI_ERROR_NULL_POINTER;
    if (!callback) return KJARNI_ERROR_NULL_POINTER;
    if (!valid_utf8(message)) return KJARNI_ERROR_INVALID_UTF8;
    return generation_guarded([&] {
        Chat& c = *chat->inner;
        const std::string prompt = c.format_prompt(conversation_with_user(c, message));
        stream_tokens(c, prompt, gen_config ? to_overrides(*gen_config) : GenerationOverrides(), callback, user_data, cancel_token);
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_chat_send_with_history(KjarniChat* chat, const int32_t* roles, const char* const* contents,
                                                            size_t history_len, const char* message,
                                                            const KjarniGenerationConfig* gen_config, char** out)
{
    if (!chat || !message || !out) return KJARNI_ERROR_NULL_POINTER;
    if (history_len > 0 && (!roles || !contents)) return KJARNI_ERROR_NULL_POINTER;
    if (!valid_utf8(message)) return KJARNI_ERROR_INVALID_UTF8;
    std::vector<ChatMessage> history;
    for (size_t i = 0; i < history_len; ++i) {
        if (!contents[i]) return KJARNI_ERROR_NULL_POINTER;
        if (!valid_utf8(contents[i])) return KJARNI_ERROR_INVALID_UTF8;
        switch (roles[i]) {
        case 0:  // a system entry rebuilds the history around it (chat.rs:441-444)
            history.clear();
            history.push_back({ChatRole::System, contents[i]});
            break;
        case 1: history.push_back({ChatRole::User, contents[i]}); break;
        case 2: history.push_back({ChatRole::Assistant, contents[i]}); break;
        default:
            set_last_error("Invalid role: " + std::to_string(roles[i]));
            return KJARNI_ERROR_INVALID_CONFIG;
        }
    }
    const KjarniErrorCode rc = generation_guarded([&] {
        Chat& c = *chat->inner;
        std::vector<ChatMessage> conv = c.history_to_conversation(history);
        conv.push_back({ChatRole::User, message});
        *out = response_cstr(c.generate(c.format_prompt(conv), gen_config ? to_overrides(*gen_config) : GenerationOverrides()));
    });
    if (rc != KJARNI_OK) *out = nullptr;
    return rc;
}

KJARNI_EXPORT KjarniErrorCode kjarni_chat_conversation_new(KjarniChat* chat, KjarniChatConversation** out)
{
    if (!chat || !out) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_UNKNOWN, [&] {
        auto h = std::make_unique<KjarniChatConversation>();
        h->chat = chat;
        if (const std::string* system = chat->inner->system_prompt()) h->history.push_back({ChatRole::System, *system});
        *out = h.release();
    });
}

KJARNI_EXPORT void kjarni_chat_conversation_free(KjarniChatConversation* convo) { delete convo; }

KJARNI_EXPORT KjarniErrorCode kjarni_chat_conversation_send(KjarniChatConversation* convo, const char* message,
                                                            const KjarniGenerationConfig* gen_config, char** out)
{
    if (!convo || !message || !out) return KJARNI_ERROR_NULL_POINTER;
    if (!convo->chat) {
        set_last_error("Parent chat has been freed");
        return KJARNI_ERROR_NULL_POINTER;
    }
    if (!valid_utf8(message)) return KJARNI_ERROR_INVALID_UTF8;
    convo->history.push_back({ChatRole::User, message});  // stays in the history even when generation fails
    const KjarniErrorCode rc = generation_guarded([&] {
        Chat& c = *convo->chat->inner;
        const std::string prompt = c.format_prompt(c.history_to_conversation(convo->history));
        const std::string response = c.generate(prompt, gen_config ? to_overrides(*gen_config) : GenerationOverrides());
        convo->history.push_back({ChatRole::Assistant, response});
        *out = response_cstr(response);
    });
    if (rc != KJARNI_OK) *out = nullptr;
    return rc;
}

KJARNI_EXPORT KjarniErrorCode kjarni_chat_conversation_stream(KjarniChatConversation* convo, const char* message,
                                                              const KjarniGenerationConfig* gen_config, KjarniStreamCallbackFn callback,
                                                              void* user_data, const KjarniCancelToken* cancel_token)
{
    if (!convo || !message) return KJARNI_ERROR_NULL_POINTER;
    if (!callback) return KJARNI_ERROR_NULL_POINTER;
    if (!convo->chat) {
        set_last_error("Parent chat has been freed");
        return KJARNI_ERROR_NULL_POINTER;
    }
    if (!valid_utf8(message)) return KJARNI_ERROR_INVALID_UTF8;
    convo->history.push_back({ChatRole::User, message});
    return generation_guarded([&] {
        Chat& c = *convo->chat->inner;
        const std::string prompt = c.format_prompt(c.history_to_conversation(convo->history));
        const std::string full =
            stream_tokens(c, prompt, gen_config ? to_overrides(*gen_config) : GenerationOverrides(), callback, user_data, cancel_token);
        if (!full.empty()) convo->history.push_back({ChatRole::Assistant, full});  // untrimmed, as streamed (chat.rs:683-688)
    });
}

KJARNI_EXPORT size_t kjarni_chat_conversation_len(const KjarniChatConversation* convo) { return convo ? convo->history.size() : 0; }

KJARNI_EXPORT void kjarni_chat_conversation_clear(KjarniChatConversation* convo, int32_t keep_system)
{
    if (!convo) return;
    if (keep_system) {
        std::vector<ChatMessage> kept;
        for (const ChatMessage& m : convo->history)
            if (m.role == ChatRole::System) kept.push_back(m);
        convo->history.swap(kept);
    } else {
        convo->history.clear();
    }
}

// chat.rs:719-745: without a buffer the byte length; with one, the number of bytes copied (NUL not counted).
KJARNI_EXPORT size_t kjarni_chat_model_name(const KjarniChat* chat, char* buf, size_t buf_len)
{
    if (!chat) return 0;
    const std::string& name = chat->inner->model_name();
    if (!buf || buf_len == 0) return name.size();
    const size_t n = std::min(name.size(), buf_len - 1);
    std::memcpy(buf, name.data(), n);
    buf[n] = '\0';
    return n;
}

KJARNI_EXPORT size_t kjarni_chat_context_size(const KjarniChat* chat) { return chat ? chat->inner->context_size() : 0; }

// ---- one stage at a time (kjarni_hip.h) -------------------------------------------------------------

KJARNI_EXPORT KjarniErrorCode kjarni_bpe_tokenizer_load(const char* tokenizer_json_path, KjarniBpeTokenizer** out)
{
    if (!tokenizer_json_path || !out) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_LOAD_FAILED, [&] {
        auto h = std::make_unique<KjarniBpeTokenizer>();
        h->tok.load(tokenizer_json_path);
        *out = h.release();
    });
}

KJARNI_EXPORT void kjarni_bpe_tokenizer_free(KjarniBpeTokenizer* t) { delete t; }

KJARNI_EXPORT KjarniErrorCode kjarni_bpe_tokenizer_encode(const KjarniBpeTokenizer* t, const char* text, size_t max_length, uint32_t* ids_out,
                                                          size_t capacity, size_t* n_out)
{
    if (!t || !text || !n_out) return KJARNI_ERROR_NULL_POINTER;
    if (!valid_utf8(text)) return KJARNI_ERROR_INVALID_UTF8;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        const std::vector<uint32_t> ids = t->tok.encode(text, max_length);
        *n_out = ids.size();
        if (ids_out)
            for (size_t i = 0; i < ids.size() && i < capacity; ++i) ids_out[i] = ids[i];
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_bpe_tokenizer_decode(const KjarniBpeTokenizer* t, const uint32_t* ids, size_t n, int32_t skip_special,
                                                          char** out)
{
    if (!t || !out || (n && !ids)) return KJARNI_ERROR_NULL_POINTER;
    *out = nullptr;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        std::string s = t->tok.decode(std::vector<uint32_t>(ids, ids + n), skip_special != 0);
        std::string clean;
        for (char ch : s)
            if (ch != '\0') clean.push_back(ch);  // a C string cannot carry NUL; tests avoid the byte
        *out = response_cstr(clean);
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_bpe_tokenizer_pre_tokenize(const KjarniBpeTokenizer* t, const char* text, KjarniStringArray* out)
{
    if (!t || !text || !out) return KJARNI_ERROR_NULL_POINTER;
    if (!valid_utf8(text)) return KJARNI_ERROR_INVALID_UTF8;
    out->strings = nullptr;
    out->len = 0;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        const std::vector<std::string> pieces = t->tok.pre_tokenize(text);
        if (pieces.empty()) return;
        char** arr = static_cast<char**>(std::calloc(pieces.size(), sizeof(char*)));
        if (!arr) throw std::bad_alloc();
        for (size_t i = 0; i < pieces.size(); ++i) arr[i] = response_cstr(pieces[i]);
        out->strings = arr;
        out->len = pieces.size();
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_chat_template_apply(int32_t template_kind, const int32_t* roles, const char* const* contents, size_t n,
                                                         char** out)
{
    if (!out || (n && (!roles || !contents))) return KJARNI_ERROR_NULL_POINTER;
    *out = nullptr;
    if (template_kind < 0 || template_kind > 2) return KJARNI_ERROR_INVALID_CONFIG;
    return guarded(KJARNI_ERROR_UNKNOWN, [&] {
        std::vector<ChatMessage> conv;
        for (size_t i = 0; i < n; ++i) {
            if (!contents[i]) throw std::runtime_error("null content");
            if (roles[i] < 0 || roles[i] > 2) throw InvalidConfig("Invalid role: " + std::to_string(roles[i]));
            conv.push_back({(ChatRole)roles[i], contents[i]});
        }
        *out = response_cstr(apply_chat_template((ChatTemplateKind)template_kind, conv));
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_sampling_distribution(const float* logits, size_t vocab, float temperature, int64_t top_k, float top_p,
                                                           float min_p, float* probs_out)
{
    if (!logits || !probs_out) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_UNKNOWN, [&] {
        SamplingParams p;
        p.temperature = temperature;
        p.top_k = top_k;
        p.top_p = top_p;
        p.min_p = min_p;
        std::vector<uint32_t> ids;
        std::vector<float> probs;
        sampling_distribution(std::vector<float>(logits, logits + vocab), p, ids, probs);
        std::fill(probs_out, probs_out + vocab, 0.0f);
        for (size_t i = 0; i < ids.size(); ++i) probs_out[ids[i]] = probs[i];
    });
}

// The candidate form of the same distribution (what the decode loop uses: the device hands over every token within `tau`
// of the maximum + the maximum + the sum of exp over the vocabulary; sampling.h).  Here the device's part is emulated on
// the host (sum accumulated in double: a different rounding order, as on the device), so the CPU suite can hold the
// candidate form to the full one.  *decided = 0: the candidates do not decide the distribution (probs_out untouched).
KJARNI_EXPORT KjarniErrorCode kjarni_sampling_distribution_candidates(const float* logits, size_t vocab, float tau, float temperature,
                                                                      int64_t top_k, float top_p, float min_p, float* probs_out,
                                                                      int32_t* decided, size_t* n_candidates)
{
    if (!logits || !probs_out || !decided) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_UNKNOWN, [&] {
        SamplingParams p;
        p.temperature = temperature;
        p.top_k = top_k;
        p.top_p = top_p;
        p.min_p = min_p;
        float mx = -std::numeric_limits<float>::infinity();
        for (size_t i = 0; i < vocab; ++i) mx = std::max(mx, logits[i]);
        const float floor = mx - tau;
        double sum = 0.0;
        std::vector<uint32_t> cid;
        std::vector<float> cval;
        for (size_t i = vocab; i-- > 0;) {  // (descending ids: the candidate list arrives in no particular order)
            sum += (double)std::exp(logits[i] - mx);
            if (logits[i] >= floor) {
                cid.push_back((uint32_t)i);
                cval.push_back(logits[i]);
            }
        }
        if (n_candidates) *n_candidates = cid.size();
        std::vector<uint32_t> ids;
        std::vector<float> probs;
        *decided = sampling_distribution_candidates(cid.data(), cval.data(), cid.size(), mx, floor, (float)sum, vocab, p, ids, probs) ? 1 : 0;
        if (*decided) {
            std::fill(probs_out, probs_out + vocab, 0.0f);
            for (size_t i = 0; i < ids.size(); ++i) probs_out[ids[i]] = probs[i];
        }
    });
}

KJARNI_EXPORT uint32_t kjarni_sample_from_probs(const float* probs, size_t vocab, float uniform)
{
    if (!probs || vocab == 0) return 0;
    std::vector<uint32_t> ids;
    std::vector<float> p;
    for (size_t i = 0; i < vocab; ++i)
        if (probs[i] != 0.0f) {
            ids.push_back((uint32_t)i);
            p.push_back(probs[i]);
        }
    return sample_from_distribution(ids, p, uniform, vocab);
}

KJARNI_EXPORT KjarniErrorCode kjarni_logits_process(float* logits, size_t vocab, const uint32_t* tokens, size_t n_tokens,
                                                    float repetition_penalty, size_t no_repeat_ngram)
{
    if (!logits || (n_tokens && !tokens)) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_UNKNOWN, [&] {
        std::vector<float> lg(logits, logits + vocab);
        const std::vector<uint32_t> toks(tokens, tokens + n_tokens);
        apply_repetition_penalty(lg, toks, repetition_penalty);
        if (no_repeat_ngram > 0) apply_no_repeat_ngram(lg, toks, no_repeat_ngram);
        std::copy(lg.begin(), lg.end(), logits);
    });
}

namespace {
void fill_resolved(const GenerationConfig& c, KjarniResolvedGeneration* out)
{
    out->strategy = (int32_t)c.strategy;
    out->temperature = c.temperature;
    out->top_k = c.top_k.has ? (int64_t)c.top_k.value : -1;
    out->top_p = c.top_p.has ? c.top_p.value : -1.0f;
    out->min_p = c.min_p.has ? c.min_p.value : -1.0f;
    out->repetition_penalty = c.repetition_penalty;
    out->no_repeat_ngram_size = c.no_repeat_ngram_size;
    out->max_new_tokens = c.max_new_tokens.has ? (int64_t)c.max_new_tokens.value : -1;
    out->max_length = c.max_length;
    out->add_bos_token = c.add_bos_token ? 1 : 0;
}
}  // namespace

KJARNI_EXPORT KjarniErrorCode kjarni_generation_resolve(const char* model_type, size_t max_position_embeddings,
                                                        const char* generation_config_json, int32_t mode,
                                                        const KjarniGenerationConfig* runtime, KjarniResolvedGeneration* out)
{
    if (!model_type || !out) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_UNKNOWN, [&] {
        const std::string hf = generation_config_json ? generation_config_json : "";
        const GenerationConfig defaults = model_default_generation_config(model_type, max_position_embeddings, generation_config_json ? &hf : nullptr);
        static const float kModeTemperature[3] = {0.7f, 0.9f, 0.3f};
        static const size_t kModeMaxTokens[3] = {512, 1024, 2048};
        GenerationOverrides user;
        if (mode >= 0) {  // mode < 0: the bare Generator (no chat mode defaults)
            const int m = mode == 1 || mode == 2 ? mode : 0;
            user.temperature = Opt<float>(kModeTemperature[m]);
            user.max_new_tokens = Opt<size_t>(kModeMaxTokens[m]);
        }
        const GenerationConfig built = resolve_generation_config(defaults, user, GenerationOverrides());
        fill_resolved(resolve_generation_config(built, user, runtime ? to_overrides(*runtime) : GenerationOverrides()), out);
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_chat_resolve(const KjarniChat* chat, const KjarniGenerationConfig* runtime, KjarniResolvedGeneration* out)
{
    if (!chat || !out) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_UNKNOWN,
                   [&] { fill_resolved(chat->inner->resolve(runtime ? to_overrides(*runtime) : GenerationOverrides()), out); });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_chat_format_prompt(const KjarniChat* chat, const int32_t* roles, const char* const* contents, size_t n,
                                                            const char* message, char** out)
{
    if (!chat || !out || (n && (!roles || !contents))) return KJARNI_ERROR_NULL_POINTER;
    *out = nullptr;
    return guarded(KJARNI_ERROR_UNKNOWN, [&] {
        const Chat& c = *chat->inner;
        std::vector<ChatMessage> conv;
        if (n == 0 && roles == nullptr) {
            conv = c.create_conversation();
        } else {
            std::vector<ChatMessage> history;
            for (size_t i = 0; i < n; ++i) {
                if (roles[i] == 0) history.clear();
                history.push_back({(ChatRole)roles[i], contents[i]});
            }
            conv = c.history_to_conversation(history);
        }
        if (message) conv.push_back({ChatRole::User, message});
        *out = response_cstr(c.format_prompt(conv));
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_chat_encode(const KjarniChat* chat, const char* prompt, const KjarniGenerationConfig* runtime,
                                                     uint32_t* ids_out, size_t capacity, size_t* n_out)
{
    if (!chat || !prompt || !n_out) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        const Chat& c = *chat->inner;
        const std::vector<uint32_t> ids = c.encode(prompt, c.resolve(runtime ? to_overrides(*runtime) : GenerationOverrides()));
        *n_out = ids.size();
        if (ids_out)
            for (size_t i = 0; i < ids.size() && i < capacity; ++i) ids_out[i] = ids[i];
    });
}

KJARNI_EXPORT void kjarni_hip_chat_seed(KjarniChat* chat, uint64_t seed)
{
    if (chat) chat->inner->reseed(seed);
}

KJARNI_EXPORT void kjarni_hip_chat_set_device_sampling(KjarniChat* chat, int32_t on)
{
    if (chat) chat->inner->model().set_device_sampling(on != 0);
}

KJARNI_EXPORT void kjarni_hip_chat_sampling_counters(KjarniChat* chat, uint64_t* from_candidates, uint64_t* from_logits)
{
    if (from_candidates) *from_candidates = chat ? chat->inner->model().tokens_from_candidates() : 0;
    if (from_logits) *from_logits = chat ? chat->inner->model().tokens_from_logits() : 0;
}
