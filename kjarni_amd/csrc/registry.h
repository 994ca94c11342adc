// Model-name resolution and on-disk model directories.
// Names/aliases: ModelType::{cli_name, from_cli_name, resolve}
// (crates/kjarni-transformers/src/models/registry.rs:226-275, 720-800);
// directory layout <cache>/<org>_<repo>/{config.json,tokenizer.json,model.safetensors}
// (registry.rs:808-811, 851-860; pipeline/encoder/loader.rs:67);
// default cache dir dirs::cache_dir()/kjarni (crates/kjarni/src/common/download.rs:10-21).
// This library never downloads: a model that is not on disk is ModelNotFound.
#pragma once
#include <string>
#include <vector>

namespace kjarni {

enum class ModelTask { Embedding, ReRanking, Classification, Other };
enum class ModelArch { Bert, Other };  // Bert: an encoder family this library runs (BERT, DistilBERT, RoBERTa, MPNet)

struct RegistryEntry {
    const char* cli_name;
    const char* repo_id;  // org/repo as in the HF URL
    ModelTask task;
    ModelArch arch;
};

// Case-insensitive; CLI slugs then HF aliases.  Returns nullptr when unknown and
// fills `error` with the reference's "Unknown model '<name>'. Did you mean: ...?" text.
const RegistryEntry* resolve_model(const std::string& name, std::string& error);

std::string default_cache_dir();
// <cache>/<org>_<repo>
std::string model_dir_for(const RegistryEntry& e, const std::string& cache_dir);
// config.json + tokenizer.json + model.safetensors present (registry.rs:814-827).
bool model_files_present(const std::string& dir);

}  // namespace kjarni
