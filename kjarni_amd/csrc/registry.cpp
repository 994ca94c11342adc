#include "registry.h"

#include <sys/stat.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace kjarni {

namespace {

// registry.rs:226-275 (cli_name) with the repo ids of :318-708.
const RegistryEntry kEntries[] = {
    {"minilm-l6-v2", "sentence-transformers/all-MiniLM-L6-v2", ModelTask::Embedding, ModelArch::Bert},
    {"nomic-embed-text", "nomic-ai/nomic-embed-text-v1.5", ModelTask::Embedding, ModelArch::Bert},
    {"bge-m3", "BAAI/bge-m3", ModelTask::Embedding, ModelArch::Bert},
    {"mpnet-base-v2", "sentence-transformers/all-mpnet-base-v2", ModelTask::Embedding, ModelArch::Bert},
    {"distilbert-base", "distilbert-base-cased-distilled-squad", ModelTask::Embedding, ModelArch::Bert},
    {"minilm-l6-v2-cross-encoder", "cross-encoder/ms-marco-MiniLM-L-6-v2", ModelTask::ReRanking, ModelArch::Bert},
    {"distilbert-sentiment", "distilbert/distilbert-base-uncased-finetuned-sst-2-english", ModelTask::Classification, ModelArch::Bert},
    {"roberta-sentiment", "olafuraron/twitter-roberta-base-sentiment-latest-safetensors", ModelTask::Classification, ModelArch::Bert},
    {"bert-sentiment-multilingual", "olafuraron/bert-base-multilingual-uncased-sentiment-safetensors", ModelTask::Classification, ModelArch::Bert},
    {"roberta-emotions", "SamLowe/roberta-base-go_emotions", ModelTask::Classification, ModelArch::Bert},
    {"distilroberta-emotion", "olafuraron/emotion-english-distilroberta-base-safetensors", ModelTask::Classification, ModelArch::Bert},
    {"toxic-bert", "olafuraron/toxic-bert-safetensors", ModelTask::Classification, ModelArch::Bert},
    {"qwen2.5-0.5b-instruct", "Qwen/Qwen2.5-0.5B-Instruct", ModelTask::Other, ModelArch::Other},
    {"qwen2.5-1.5b", "Qwen/Qwen2.5-1.5B-Instruct", ModelTask::Other, ModelArch::Other},
    {"llama3.2-1b-instruct", "meta-llama/Llama-3.2-1B-Instruct", ModelTask::Other, ModelArch::Other},
    {"llama3.2-3b-instruct", "meta-llama/Llama-3.2-3B-Instruct", ModelTask::Other, ModelArch::Other},
    {"phi3.5-mini", "microsoft/Phi-3.5-mini-instruct", ModelTask::Other, ModelArch::Other},
    {"mistral-7b", "mistralai/Mistral-7B-Instruct-v0.3", ModelTask::Other, ModelArch::Other},
    {"llama3.1-8b-instruct", "meta-llama/Llama-3.1-8B-Instruct", ModelTask::Other, ModelArch::Other},
    {"deepseek-r1-8b", "deepseek-ai/DeepSeek-R1-Distill-Llama-8B", ModelTask::Other, ModelArch::Other},
    {"flan-t5-base", "google/flan-t5-base", ModelTask::Other, ModelArch::Other},
    {"flan-t5-large", "google/flan-t5-large", ModelTask::Other, ModelArch::Other},
    {"distilbart-cnn", "olafuraron/distilbart-cnn-12-6", ModelTask::Other, ModelArch::Other},
    {"bart-large-cnn", "facebook/bart-large-cnn", ModelTask::Other, ModelArch::Other},
    {"whisper-small", "openai/whisper-small", ModelTask::Other, ModelArch::Other},
    {"whisper-large-v3", "openai/whisper-large-v3", ModelTask::Other, ModelArch::Other},
    {"distilgpt2", "distilgpt2/resolve", ModelTask::Other, ModelArch::Other},
    {"gpt2", "gpt2/resolve", ModelTask::Other, ModelArch::Other},
};

struct Alias {
    const char* alias;
    const char* cli_name;
};

// registry.rs:761-797 (HF aliases), lower-cased.
const Alias kAliases[] = {
    {"all-minilm-l6-v2", "minilm-l6-v2"},
    {"sentence-transformers/all-minilm-l6-v2", "minilm-l6-v2"},
    {"all-mpnet-base-v2", "mpnet-base-v2"},
    {"sentence-transformers/all-mpnet-base-v2", "mpnet-base-v2"},
    {"ms-marco-minilm-l-6-v2", "minilm-l6-v2-cross-encoder"},
    {"cross-encoder/ms-marco-minilm-l-6-v2", "minilm-l6-v2-cross-encoder"},
    {"nomic-embed-text-v1.5", "nomic-embed-text"},
    {"nomic-ai/nomic-embed-text-v1.5", "nomic-embed-text"},
    {"baai/bge-m3", "bge-m3"},
    {"distilbert-base-uncased-finetuned-sst-2-english", "distilbert-sentiment"},
    {"twitter-roberta-base-sentiment-latest", "roberta-sentiment"},
    {"bert-base-multilingual-uncased-sentiment", "bert-sentiment-multilingual"},
    {"bert-base-multilingual-uncased-sentiment-safetensors", "bert-sentiment-multilingual"},
    {"toxic-bert-safetensors", "toxic-bert"},
    {"unitary/toxic-bert", "toxic-bert"},
    {"roberta-base-go_emotions", "roberta-emotions"},
    {"samlowe/roberta-base-go_emotions", "roberta-emotions"},
    {"emotion-english-distilroberta-base", "distilroberta-emotion"},
    {"olafuraron/distilbart-cnn-12-6", "distilbart-cnn"},
    {"distilbart-cnn-12-6", "distilbart-cnn"},
    {"facebook/bart-large-cnn", "bart-large-cnn"},
    {"openai/whisper-small", "whisper-small"},
    {"openai/whisper-large-v3", "whisper-large-v3"},
    {"distilgpt2/resolve/main/model.safetensors", "distilgpt2"},
    {"gpt2/resolve/main/model.safetensors", "gpt2"},
};

std::string to_lower(const std::string& s)
{
    std::string o = s;
    for (char& c : o)
        if (c >= 'A' && c <= 'Z') c = (char)(c + 32);
    return o;
}

const RegistryEntry* by_cli(const std::string& cli)
{
    for (const RegistryEntry& e : kEntries)
        if (cli == e.cli_name) return &e;
    return nullptr;
}

size_t levenshtein(const std::string& a, const std::string& b)
{
    std::vector<size_t> prev(b.size() + 1), cur(b.size() + 1);
    for (size_t j = 0; j <= b.size(); ++j) prev[j] = j;
    for (size_t i = 1; i <= a.size(); ++i) {
        cur[0] = i;
        for (size_t j = 1; j <= b.size(); ++j) {
            const size_t cost = a[i - 1] == b[j - 1] ? 0 : 1;
            cur[j] = std::min({prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + cost});
        }
        prev.swap(cur);
    }
    return prev[b.size()];
}

bool is_file(const std::string& p)
{
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

}  // namespace

const RegistryEntry* resolve_model(const std::string& name, std::string& error)
{
    const std::string norm = to_lower(name);
    if (const RegistryEntry* e = by_cli(norm)) return e;
    for (const Alias& a : kAliases)
        if (norm == a.alias) return by_cli(a.cli_name);

    // registry.rs:725-739: substring matches first ...
    std::string subs;
    for (const RegistryEntry& e : kEntries)
        if (std::strstr(e.cli_name, norm.c_str()) != nullptr) {
            if (!subs.empty()) subs += ", ";
            subs += e.cli_name;
        }
    if (!subs.empty()) {
        error = "Unknown model '" + name + "'. Did you mean: " + subs + "?";
        return nullptr;
    }
    // ... then up to 3 names with similarity >= 0.4 (registry.rs:741-751, 803-806).
    std::vector<std::pair<float, std::string>> sims;
    for (const RegistryEntry& e : kEntries) {
        const std::string cand = e.cli_name;
        const size_t mx = std::max(norm.size(), cand.size());
        const float sim = mx == 0 ? 1.0f : 1.0f - (float)levenshtein(norm, cand) / (float)mx;
        if (sim >= 0.4f) sims.emplace_back(sim, cand);
    }
    std::stable_sort(sims.begin(), sims.end(), [](const auto& x, const auto& y) { return x.first > y.first; });
    if (sims.empty()) {
        error = "Unknown model '" + name + "'";
    } else {
        std::string names;
        for (size_t i = 0; i < sims.size() && i < 3; ++i) names += (i ? ", " : "") + sims[i].second;
        error = "Unknown model '" + name + "'. Did you mean: " + names + "?";
    }
    return nullptr;
}

std::string default_cache_dir()
{
    // dirs::cache_dir(): $XDG_CACHE_HOME or $HOME/.cache, then /kjarni.
    const char* xdg = std::getenv("XDG_CACHE_HOME");
    if (xdg && xdg[0] == '/') return std::string(xdg) + "/kjarni";
    const char* home = std::getenv("HOME");
    return std::string(home ? home : ".") + "/.cache/kjarni";
}

std::string model_dir_for(const RegistryEntry& e, const std::string& cache_dir)
{
    std::string repo = e.repo_id;
    std::replace(repo.begin(), repo.end(), '/', '_');
    return cache_dir + "/" + repo;
}

bool model_files_present(const std::string& dir)
{
    // model_weights.rs:51-53: one file or an index over shards
    return is_file(dir + "/config.json") && is_file(dir + "/tokenizer.json") &&
           (is_file(dir + "/model.safetensors") || is_file(dir + "/model.safetensors.index.json"));
}

}  // namespace kjarni
