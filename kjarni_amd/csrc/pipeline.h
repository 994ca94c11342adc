// A loaded model + tokenizer and the string-level operations built on it, shared by the
// Embedder / Reranker / Classifier entry points (ffi_api.cpp) and the Searcher (ffi_searcher.cpp).
#pragma once
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "encoder.h"
#include "group.h"
#include "wordpiece.h"

namespace kjarni {

// The reference's model types are Send + Sync and nothing serialises calls on a handle
// (crates/kjarni-transformers/src/traits.rs:33, kjarni-ffi/src/lib.rs:25-32).  Same here: the weights are
// immutable, every call leases its own workspace + stream (encoder.h), the tokenizer is read-only.
// `group` holds one replica of the model per device of KJARNI_HIP_DEVICES (default: all visible);
// batches are cut into row blocks across them (group.h).
struct Pipeline {
    std::unique_ptr<EncoderGroup> group;
    BertTokenizer tokenizer;
    std::string model_name;
    EncoderModel& model() { return group->replica(0); }
    const EncoderConfig& config() const { return group->config(); }
};

enum class Want { Embedding, Reranking, Classification };

std::unique_ptr<Pipeline> load_pipeline(const char* cache_dir, const char* model_name, const char* model_path,
                                        const char* default_name, Want want);
// texts -> [n, H] embeddings (tokenise on the host, encode + pool on the GPU).
std::vector<float> embed_texts(Pipeline& p, const std::vector<std::string>& texts, PoolMode pool, bool normalize);
// The GPU half of embed_texts for a batch that is already tokenised (the indexer tokenises ahead of the GPU).
std::vector<float> embed_encoding(Pipeline& p, const BatchEncoding& be, PoolMode pool, bool normalize);
// (query, doc_i) pairs -> scores (logit column 0).
std::vector<float> rerank_scores(Pipeline& p, const std::string& query, const std::vector<std::string>& docs);

// The index at `root` is being deleted: drop the device copies of its segments' vectors in every live Searcher
// (ffi_searcher.cpp).
void forget_device_segments_under(const std::string& root);

}  // namespace kjarni
