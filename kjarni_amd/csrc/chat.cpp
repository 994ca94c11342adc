// Chat host logic (see chat.h for the reference files each part follows).
#include "chat.h"

#include <sys/stat.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "ffi_common.h"
#include "json.h"
#include "unicode.h"

namespace kjarni {

// ---- templates ------------------------------------------------------------------------------------

namespace {
const char* role_name(ChatRole r) { return r == ChatRole::System ? "system" : (r == ChatRole::User ? "user" : "assistant"); }
}  // namespace

std::string apply_chat_template(ChatTemplateKind kind, const std::vector<ChatMessage>& conv)
{
    std::string p;
    switch (kind) {
    case ChatTemplateKind::Llama3:  // llama3.rs:57-75, for_generation(): BOS and the assistant header are added
        p += "<|begin_of_text|>";
        for (const ChatMessage& m : conv) {
            p += "<|start_header_id|>";
            p += role_name(m.role);
            p += "<|end_header_id|>\n\n";
            p += m.content;
            p += "<|eot_id|>";
        }
        p += "<|start_header_id|>assistant<|end_header_id|>\n\n";
        break;
    case ChatTemplateKind::ChatML:  // chatml.rs:15-34
        for (const ChatMessage& m : conv) {
            p += "<|im_start|>";
            p += role_name(m.role);
            p += "\n";
            p += m.content;
            p += "<|im_end|>\n";
        }
        p += "<|im_start|>assistant\n";
        break;
    case ChatTemplateKind::Mistral: {  // mistral.rs:16-75
        if (conv.empty()) return p;
        p += "<s>";
        size_t i = 0;
        const std::string* system = nullptr;
        if (conv[0].role == ChatRole::System) {
            system = &conv[0].content;
            i = 1;
        }
        bool first_user = true;
        for (; i < conv.size(); ++i) {
            const ChatMessage& m = conv[i];
            if (m.role == ChatRole::User) {
                p += "[INST] ";
                if (first_user) {
                    if (system) {
                        p += *system;
                        p += "\n\n";
                    }
                    first_user = false;
                }
                p += m.content;
                p += " [/INST]";
            } else if (m.role == ChatRole::Assistant) {
                p += ' ';
                p += m.content;
                p += "</s>";
            }
        }
        break;
    }
    }
    return p;
}

std::vector<std::string> chat_stop_sequences(ChatTemplateKind kind)
{
    switch (kind) {
    case ChatTemplateKind::Llama3: return {"<|eot_id|>", "<|end_of_text|>"};
    case ChatTemplateKind::ChatML: return {"<|im_end|>", "<|endoftext|>"};
    case ChatTemplateKind::Mistral: return {"</s>"};
    }
    return {};
}

const char* chat_default_system_prompt(ChatTemplateKind kind)
{
    switch (kind) {
    case ChatTemplateKind::Llama3: return "You are a helpful, harmless, and honest assistant.";
    case ChatTemplateKind::ChatML: return "You are a helpful assistant.";
    case ChatTemplateKind::Mistral: return nullptr;
    }
    return nullptr;
}

std::string trim_unicode(const std::string& s)
{
    std::vector<uint32_t> cps;
    if (!unicode::decode_utf8(s.data(), s.size(), cps)) return s;
    size_t a = 0, b = cps.size();
    while (a < b && unicode::is_whitespace(cps[a])) ++a;
    while (b > a && unicode::is_whitespace(cps[b - 1])) --b;
    return unicode::encode_utf8(std::vector<uint32_t>(cps.begin() + (ptrdiff_t)a, cps.begin() + (ptrdiff_t)b));
}

// ---- generation config ----------------------------------------------------------------------------

GenerationConfig resolve_generation_config(GenerationConfig config, const GenerationOverrides& user, const GenerationOverrides& runtime)
{
    const Opt<size_t> beams = runtime.num_beams.or_else(user.num_beams);
    const bool force_beams = beams.has && beams.value > 1;
    const Opt<bool> sample = runtime.do_sample.or_else(user.do_sample);
    const bool force_greedy = sample.has && !sample.value;
    const bool force_sampling = sample.has && sample.value;

    if (force_beams) {
        if (config.strategy != Strategy::BeamSearch) {  // BeamSearchParams::default()
            config.num_beams = 4;
            config.length_penalty = 1.0f;
        }
        config.strategy = Strategy::BeamSearch;
    } else if (force_greedy) {
        config.strategy = Strategy::Greedy;
    } else if (force_sampling) {
        if (config.strategy != Strategy::Sample) {  // SamplingParams::default()
            config.temperature = 0.7f;
            config.top_k = Opt<size_t>(50);
            config.top_p = Opt<float>(0.9f);
            config.min_p = Opt<float>(0.1f);
        }
        config.strategy = Strategy::Sample;
    }
    if (const auto v = runtime.max_new_tokens.or_else(user.max_new_tokens); v.has) config.max_new_tokens = v;
    if (const auto v = runtime.repetition_penalty.or_else(user.repetition_penalty); v.has) config.repetition_penalty = v.value;
    if (const auto v = runtime.no_repeat_ngram_size.or_else(user.no_repeat_ngram_size); v.has) config.no_repeat_ngram_size = v.value;
    if (config.strategy == Strategy::Sample) {
        if (const auto v = runtime.temperature.or_else(user.temperature); v.has) config.temperature = v.value;
        if (const auto v = runtime.top_k.or_else(user.top_k); v.has) config.top_k = v;
        if (const auto v = runtime.top_p.or_else(user.top_p); v.has) config.top_p = v;
        if (const auto v = runtime.min_p.or_else(user.min_p); v.has) config.min_p = v;
    } else if (config.strategy == Strategy::BeamSearch) {
        if (beams.has) config.num_beams = beams.value;
        if (const auto v = runtime.length_penalty.or_else(user.length_penalty); v.has) config.length_penalty = v.value;
    }
    return config;
}

GenerationConfig model_default_generation_config(const std::string& model_type, size_t max_pos, const std::string* hf_json)
{
    GenerationConfig c;
    if (hf_json && model_type != "mistral") {  // HFGenerationDefaults (a file that does not deserialize is ignored); Mistral never reads it
        try {
            const Json j = Json::parse(*hf_json);
            if (!j.is_object()) throw std::runtime_error("not an object");
            auto typed = [&](const char* k, Json::Type t) {  // serde: a present field of the wrong type fails the whole file
                const Json* v = j.find(k);
                if (v && !v->is_null() && v->type != t) throw std::runtime_error("type");
                return v && !v->is_null() ? v : nullptr;
            };
            const Json* do_sample = typed("do_sample", Json::Bool);
            const Json* temperature = typed("temperature", Json::Number);
            const Json* top_p = typed("top_p", Json::Number);
            const Json* top_k = typed("top_k", Json::Number);
            const Json* max_new = typed("max_new_tokens", Json::Number);
            const Json* max_len = typed("max_length", Json::Number);
            const Json* rep = typed("repetition_penalty", Json::Number);
            typed("decoder_start_token_id", Json::Number);
            if (j.find("do_sample") && j.find("do_sample")->is_null()) throw std::runtime_error("do_sample: null");
            if (j.find("temperature") && j.find("temperature")->is_null()) throw std::runtime_error("temperature: null");
            if (do_sample && do_sample->as_bool()) {
                c.strategy = Strategy::Sample;
                c.temperature = temperature ? (float)temperature->as_double() : 1.0f;
                c.top_k = top_k ? Opt<size_t>((size_t)top_k->as_int()) : Opt<size_t>();
                c.top_p = top_p ? Opt<float>((float)top_p->as_double()) : Opt<float>();
                c.min_p = Opt<float>();
            } else {
                c.strategy = Strategy::Greedy;
            }
            c.max_new_tokens = max_new ? Opt<size_t>((size_t)max_new->as_int()) : Opt<size_t>();
            c.max_length = max_len ? (size_t)max_len->as_int() : max_pos;
            c.repetition_penalty = rep ? (float)rep->as_double() : 1.0f;
            c.no_repeat_ngram_size = 0;
            c.add_bos_token = true;
            return c;
        } catch (const std::exception&) {
        }
    }
    c.max_length = max_pos;
    c.no_repeat_ngram_size = 0;
    c.strategy = Strategy::Sample;
    if (model_type == "mistral") {  // mistral/model.rs:236-252
        c.max_new_tokens = Opt<size_t>(512);
        c.repetition_penalty = 1.15f;
        c.add_bos_token = true;
        c.temperature = 0.7f;
        c.top_k = Opt<size_t>(40);
        c.top_p = Opt<float>(0.9f);
        c.min_p = Opt<float>(0.05f);
    } else if (model_type == "qwen2") {  // qwen/model.rs:267-281
        c.max_new_tokens = Opt<size_t>(512);
        c.repetition_penalty = 1.1f;
        c.add_bos_token = false;
        c.temperature = 0.7f;
        c.top_k = Opt<size_t>(40);
        c.top_p = Opt<float>(0.8f);
        c.min_p = Opt<float>(0.05f);
    } else {  // llama/model.rs:381-395
        c.max_new_tokens = Opt<size_t>(256);
        c.repetition_penalty = 1.0f;
        c.add_bos_token = true;
        c.temperature = 0.6f;
        c.top_k = Opt<size_t>();
        c.top_p = Opt<float>(0.9f);
        c.min_p = Opt<float>(0.05f);
    }
    return c;
}

// ---- Chat -----------------------------------------------------------------------------------------

namespace {

struct ChatModelInfo {
    const char* cli_name;
    const char* arch;     // ModelArchitecture::display_name
    const char* family;   // llama | qwen2 | mistral | phi3 | gpt | encoder | whisper | seq2seq
    const char* task;     // format!("{:?}", task).to_lowercase()
};
// registry.rs ModelType::info(): architecture and task of every registry entry.
const ChatModelInfo kChatModels[] = {
    {"minilm-l6-v2", "BERT", "encoder", "embedding"},
    {"minilm-l6-v2-cross-encoder", "BERT", "encoder", "reranking"},
    {"mpnet-base-v2", "Mpnet", "encoder", "embedding"},
    {"distilbert-base", "BERT", "encoder", "embedding"},
    {"nomic-embed-text", "Nomic-BERT", "encoder", "embedding"},
    {"bge-m3", "BERT", "encoder", "embedding"},
    {"distilbert-sentiment", "BERT", "encoder", "sentimentanalysis"},
    {"roberta-sentiment", "BERT", "encoder", "sentimentanalysis"},
    {"bert-sentiment-multilingual", "BERT", "encoder", "sentimentanalysis"},
    {"roberta-emotions", "BERT", "encoder", "classification"},
    {"distilroberta-emotion", "BERT", "encoder", "classification"},
    {"toxic-bert", "BERT", "encoder", "classification"},
    {"qwen2.5-0.5b-instruct", "Qwen2 (Biased)", "qwen2", "chat"},
    {"qwen2.5-1.5b", "Qwen2 (Biased)", "qwen2", "chat"},
    {"llama3.2-1b-instruct", "Llama (Standard)", "llama", "chat"},
    {"llama3.2-3b-instruct", "Llama (Standard)", "llama", "chat"},
    {"phi3.5-mini", "Phi-3 (LongRoPE)", "phi3", "reasoning"},
    {"mistral-7b", "Mistral (SWA)", "mistral", "chat"},
    {"llama3.1-8b-instruct", "Llama (Standard)", "llama", "chat"},
    {"deepseek-r1-8b", "Llama (Standard)", "llama", "reasoning"},
    {"flan-t5-base", "T5", "seq2seq", "seq2seq"},
    {"flan-t5-large", "T5", "seq2seq", "seq2seq"},
    {"bart-large-cnn", "BART", "seq2seq", "seq2seq"},
    {"distilbart-cnn", "BART", "seq2seq", "seq2seq"},
    {"whisper-small", "Whisper (ASR)", "whisper", "speechtotext"},
    {"whisper-large-v3", "Whisper (ASR)", "whisper", "speechtotext"},
    {"distilgpt2", "GPT", "gpt", "generation"},
    {"gpt2", "GPT", "gpt", "generation"},
};


bool read_file(const std::string& p, std::string& out)
{
    std::ifstream f(p, std::ios::binary);
    if (!f) return false;
    std::ostringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
}

}  // namespace

std::unique_ptr<Chat> Chat::create(const std::string& model_name, const std::string& model_dir, const std::string& cache_dir,
                                   const std::string* system_prompt, int mode, bool quiet)
{
    std::string err;
    const RegistryEntry* entry = resolve_model(model_name, err);
    if (!entry) throw ModelNotFound(err);  // ChatError::UnknownModel
    const ChatModelInfo* info = nullptr;
    for (const ChatModelInfo& m : kChatModels)
        if (std::strcmp(m.cli_name, entry->cli_name) == 0) info = &m;
    if (!info) throw ModelNotFound("Unknown model '" + model_name + "'");
    const std::string cli = entry->cli_name;
    const std::string family = info->family;
    auto incompatible = [&](const std::string& reason) { return InvalidConfig("model '" + cli + "' is incompatible with chat: " + reason); };
    // validate_for_chat (chat/validation.rs:36-116)
    if (family == "encoder")
        throw incompatible(std::string("Model architecture '") + info->arch +
                           "' is an encoder and cannot generate text. Use an Encoder for embeddings instead.");
    if (family == "whisper")
        throw incompatible(std::string("Model architecture '") + info->arch +
                           "' is designed for speech-to-text transcription. Use a SpeechToText model instead.");
    if (family == "seq2seq")
        throw incompatible(std::string("Model architecture '") + info->arch +
                           "' is a seq2seq model designed for translation/summarization. Use Translator or Summarizer instead.");
    const std::string task = info->task;
    if (task == "generation" && !quiet) std::fprintf(stderr, "Warning: [info] Model '%s' is a base model, not instruction-tuned.\n", cli.c_str());

    const std::string dir = !model_dir.empty() ? model_dir : model_dir_for(*entry, cache_dir.empty() ? default_cache_dir() : cache_dir);
    if (!model_files_present(dir))  // DownloadPolicy: this library never downloads
        throw ModelNotFound("model '" + cli + "' not downloaded. run: kjarni model download " + cli);

    auto load_failed = [&](const std::string& why) { return std::runtime_error("failed to load model '" + cli + "': " + why); };
    if (family == "phi3") throw load_failed("Phi3 model loading not yet implemented");
    if (family == "gpt")  // loads in the reference, then fails the template check (chat/model.rs:113-116)
        throw InvalidConfig("model '" + cli + "' does not have a chat template. use Generator for raw text generation.");

    std::unique_ptr<Chat> chat(new Chat());
    chat->model_name_ = cli;
    chat->template_ = family == "qwen2" ? ChatTemplateKind::ChatML : (family == "mistral" ? ChatTemplateKind::Mistral : ChatTemplateKind::Llama3);
    chat->mode_ = mode;
    if (system_prompt) {
        chat->has_system_ = true;
        chat->system_prompt_ = *system_prompt;
    }
    int device = 0;
    if (const char* dv = std::getenv("KJARNI_HIP_DEVICE")) device = std::atoi(dv);
    int context_cap = 32768;
    if (const char* cv = std::getenv("KJARNI_HIP_CHAT_CONTEXT")) context_cap = std::max(16, std::atoi(cv));
    try {
        chat->tokenizer_.load(dir + "/tokenizer.json");
        chat->model_ = LlmModel::load(dir, device, 0, context_cap);
    } catch (const GpuUnavailable&) {
        throw;
    } catch (const std::exception& e) {
        throw load_failed(e.what());
    }
    const LlmConfig& cfg = chat->model_->config();
    chat->tokenizer_.set_truncation((size_t)cfg.max_pos);  // loader.rs:115-120

    std::string hf;
    const bool has_hf = read_file(dir + "/generation_config.json", hf);
    const GenerationConfig defaults = model_default_generation_config(cfg.model_type, (size_t)cfg.max_pos, has_hf ? &hf : nullptr);
    // Chat::from_builder: the mode supplies temperature and max_new_tokens unless the builder set them (the C ABI never does).
    static const float kModeTemperature[3] = {0.7f, 0.9f, 0.3f};
    static const size_t kModeMaxTokens[3] = {512, 1024, 2048};
    const int m = mode == 1 || mode == 2 ? mode : 0;
    chat->mode_overrides_.temperature = Opt<float>(kModeTemperature[m]);
    chat->mode_overrides_.max_new_tokens = Opt<size_t>(kModeMaxTokens[m]);
    chat->generation_config_ = resolve_generation_config(defaults, chat->mode_overrides_, GenerationOverrides());

    // stop_token_ids (models/base.rs:261-271): the first eos id and <|eot_id|> when the tokenizer has it
    if (!cfg.eos_ids.empty()) chat->stop_ids_.push_back(cfg.eos_ids[0]);
    uint32_t eot = 0;
    if (chat->tokenizer_.token_to_id("<|eot_id|>", eot) && std::find(chat->stop_ids_.begin(), chat->stop_ids_.end(), eot) == chat->stop_ids_.end())
        chat->stop_ids_.push_back(eot);
    if (chat->stop_ids_.empty()) chat->stop_ids_.push_back(UINT32_MAX);  // nothing stops generation but the length
    return chat;
}

std::vector<ChatMessage> Chat::create_conversation() const
{
    std::vector<ChatMessage> c;
    if (has_system_) c.push_back({ChatRole::System, system_prompt_});
    else if (const char* d = chat_default_system_prompt(template_)) c.push_back({ChatRole::System, d});
    return c;
}

std::vector<ChatMessage> Chat::history_to_conversation(const std::vector<ChatMessage>& history) const
{
    std::vector<ChatMessage> conv;
    bool has_system = false;
    for (const ChatMessage& m : history) {
        if (m.role == ChatRole::System) {  // a system message restarts the conversation (model.rs:157-160)
            conv.clear();
            conv.push_back(m);
            has_system = true;
        } else {
            conv.push_back(m);
        }
    }
    if (!has_system && has_system_) {
        std::vector<ChatMessage> with;
        with.push_back({ChatRole::System, system_prompt_});
        for (const ChatMessage& m : conv)
            if (m.role != ChatRole::System) with.push_back(m);
        return with;
    }
    return conv;
}

GenerationConfig Chat::resolve(const GenerationOverrides& runtime) const
{
    return resolve_generation_config(generation_config_, mode_overrides_, runtime);
}

std::vector<uint32_t> Chat::encode(const std::string& prompt, const GenerationConfig& config) const
{
    std::vector<uint32_t> tokens = tokenizer_.encode(prompt);
    const LlmConfig& cfg = model_->config();
    if (config.add_bos_token && cfg.has_bos && (tokens.empty() || tokens[0] != cfg.bos_id)) tokens.insert(tokens.begin(), cfg.bos_id);
    return tokens;
}

std::string Chat::run(const std::string& prompt, const GenerationOverrides& runtime, const std::function<bool(const std::string&)>& on_text)
{
    std::lock_guard<std::mutex> lock(mutex_);
    const GenerationConfig config = resolve(runtime);
    if (config.strategy == Strategy::BeamSearch) throw std::runtime_error("generation failed: Beam search is not supported in this generator.");
    const std::vector<uint32_t> tokens = encode(prompt, config);
    if (tokens.empty()) throw std::runtime_error("generation failed: cannot generate from empty prompt");

    GenerateOptions opt;
    opt.max_new_tokens = config.max_new_tokens.has ? config.max_new_tokens.value
                                                   : (config.max_length > tokens.size() ? config.max_length - tokens.size() : 0);
    opt.max_len = config.max_new_tokens.has ? tokens.size() + config.max_new_tokens.value : config.max_length;
    opt.max_len = std::min(opt.max_len, context_size());
    opt.repetition_penalty = config.repetition_penalty;
    opt.no_repeat_ngram = (int)config.no_repeat_ngram_size;
    opt.stop_ids = stop_ids_;
    if (config.strategy == Strategy::Sample) {
        opt.sample = true;
        opt.sampling.temperature = config.temperature;
        opt.sampling.top_k = config.top_k.has ? (int64_t)config.top_k.value : -1;
        opt.sampling.top_p = config.top_p.has ? config.top_p.value : -1.0f;
        opt.sampling.min_p = config.min_p.has ? config.min_p.value : -1.0f;
        opt.uniform = [this] { return rng_.next(); };
    }
    std::string text;
    std::vector<uint32_t> prompt_tokens = tokens;
    if ((int)prompt_tokens.size() > model_->context()) prompt_tokens.resize((size_t)model_->context());
    model_->generate(prompt_tokens, opt, [&](uint32_t id) {
        const std::string piece = tokenizer_.decode({id}, false);  // one token at a time, specials kept (generator.rs:343-345)
        text += piece;
        return on_text ? on_text(piece) : true;
    });
    return text;
}

std::string Chat::generate(const std::string& prompt, const GenerationOverrides& runtime)
{
    std::string cleaned = trim_unicode(run(prompt, runtime, nullptr));
    for (const std::string& stop : chat_stop_sequences(template_))
        if (cleaned.size() >= stop.size() && cleaned.compare(cleaned.size() - stop.size(), stop.size(), stop) == 0)
            cleaned = trim_unicode(cleaned.substr(0, cleaned.size() - stop.size()));
    return cleaned;
}

std::string Chat::generate_stream(const std::string& prompt, const GenerationOverrides& runtime,
                                  const std::function<bool(const std::string&)>& on_text)
{
    return run(prompt, runtime, on_text);
}

}  // namespace kjarni
