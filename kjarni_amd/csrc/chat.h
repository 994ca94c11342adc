// Chat: prompt templating, generation-config resolution and the text side of the decoder generation loop.
//
//   templates            crates/kjarni-transformers/src/chat/{llama3.rs:38-96, chatml.rs:15-48, mistral.rs:16-80}
//   Conversation/History crates/kjarni-transformers/src/chat/templates.rs:51-131, crates/kjarni/src/chat/types.rs:197-238
//   Chat                 crates/kjarni/src/chat/model.rs:30-352 (builder defaults: chat/builder.rs, modes: chat/types.rs:122-146)
//   config resolution    crates/kjarni/src/generation/resolution.rs:8-85
//   model defaults       crates/kjarni-models/src/models/{llama/model.rs:373-396, qwen/model.rs:261-282},
//                        crates/kjarni-transformers/src/common/mod.rs:297-349 (generation_config.json)
//   encode / stop ids    crates/kjarni-transformers/src/decoder/generator.rs:141-163, models/base.rs:261-271
//   token text, cleanup  decoder/generator.rs:343-360, crates/kjarni/src/chat/model.rs:283-303
#pragma once
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "bpe.h"
#include "llm.h"
#include "registry.h"
#include "sampling.h"

namespace kjarni {

enum class ChatTemplateKind { Llama3 = 0, ChatML = 1, Mistral = 2 };
enum class ChatRole { System = 0, User = 1, Assistant = 2 };

struct ChatMessage {
    ChatRole role;
    std::string content;
};

std::string apply_chat_template(ChatTemplateKind kind, const std::vector<ChatMessage>& conversation);
std::vector<std::string> chat_stop_sequences(ChatTemplateKind kind);
const char* chat_default_system_prompt(ChatTemplateKind kind);  // nullptr when the template has none

template <class T>
struct Opt {
    bool has = false;
    T value{};
    Opt() = default;
    Opt(const T& v) : has(true), value(v) {}
    Opt or_else(const Opt& other) const { return has ? *this : other; }
};

struct GenerationOverrides {
    Opt<float> temperature, top_p, min_p, repetition_penalty, length_penalty;
    Opt<size_t> top_k, no_repeat_ngram_size, max_new_tokens, num_beams;
    Opt<bool> do_sample;
};

enum class Strategy { Greedy = 0, Sample = 1, BeamSearch = 2 };

struct GenerationConfig {
    Opt<size_t> max_new_tokens;
    size_t max_length = 0;
    float repetition_penalty = 1.0f;
    size_t no_repeat_ngram_size = 0;
    bool add_bos_token = true;
    Strategy strategy = Strategy::Greedy;
    // SamplingParams (temperature 0.7, top_k 50, top_p 0.9 are SamplingParams::default())
    float temperature = 0.7f;
    Opt<size_t> top_k;
    Opt<float> top_p, min_p;
    size_t num_beams = 4;  // BeamSearchParams::default()
    float length_penalty = 1.0f;
};

GenerationConfig resolve_generation_config(GenerationConfig model_defaults, const GenerationOverrides& user,
                                           const GenerationOverrides& runtime);
// get_default_generation_config: generation_config.json when it parses, the per-architecture fallback otherwise.
GenerationConfig model_default_generation_config(const std::string& model_type, size_t max_position_embeddings,
                                                 const std::string* generation_config_json);

class Chat {
public:
    // model_name: registry name (decides architecture, template and the on-disk directory); model_dir overrides the
    // directory.  mode: 0 default, 1 creative, 2 reasoning.
    static std::unique_ptr<Chat> create(const std::string& model_name, const std::string& model_dir, const std::string& cache_dir,
                                        const std::string* system_prompt, int mode, bool quiet);

    const std::string& model_name() const { return model_name_; }
    size_t context_size() const { return (size_t)model_->config().max_pos; }
    const std::string* system_prompt() const { return has_system_ ? &system_prompt_ : nullptr; }
    ChatTemplateKind template_kind() const { return template_; }
    const BpeTokenizer& tokenizer() const { return tokenizer_; }
    LlmModel& model() { return *model_; }

    std::string format_prompt(const std::vector<ChatMessage>& conversation) const { return apply_chat_template(template_, conversation); }
    std::vector<ChatMessage> create_conversation() const;  // Chat::create_conversation
    std::vector<ChatMessage> history_to_conversation(const std::vector<ChatMessage>& history) const;
    GenerationConfig resolve(const GenerationOverrides& runtime) const;
    std::vector<uint32_t> encode(const std::string& prompt, const GenerationConfig& config) const;  // DecoderGenerator::encode

    // Generator::generate_with_config + Chat::generate: concatenated token texts, trimmed, stop sequences stripped.
    std::string generate(const std::string& prompt, const GenerationOverrides& runtime);
    // generate_stream: on_text per generated token (return false to stop); returns the concatenation of what was emitted.
    std::string generate_stream(const std::string& prompt, const GenerationOverrides& runtime,
                                const std::function<bool(const std::string&)>& on_text);
    void reseed(uint64_t seed) { rng_.reseed(seed); }

private:
    Chat() = default;
    std::string run(const std::string& prompt, const GenerationOverrides& runtime, const std::function<bool(const std::string&)>& on_text);

    std::string model_name_;
    std::unique_ptr<LlmModel> model_;
    BpeTokenizer tokenizer_;
    ChatTemplateKind template_ = ChatTemplateKind::Llama3;
    bool has_system_ = false;
    std::string system_prompt_;
    int mode_ = 0;
    GenerationConfig generation_config_;   // Generator::generation_config (model defaults + mode)
    GenerationOverrides mode_overrides_;   // Generator::user_overrides
    std::vector<uint32_t> stop_ids_;
    UniformRng rng_;
    std::mutex mutex_;  // one generation at a time per handle (the KV cache is the handle's)
};

// str::trim (Unicode White_Space at both ends).
std::string trim_unicode(const std::string& s);

}  // namespace kjarni
