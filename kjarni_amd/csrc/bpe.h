// Byte-level BPE tokenizer for the decoder-only models (Llama 3, Qwen 2, GPT-2 style tokenizer.json).
//
// The reference tokenizes prompts with the third-party `tokenizers` crate 0.22.1 (Cargo.toml:34), loaded in
// crates/kjarni-transformers/src/pipeline/decoder/loader.rs:107-121 and called as `encode(prompt, false)` in
// crates/kjarni-transformers/src/decoder/generator.rs:141-163.  That crate is not vendored, so this is a restatement
// of its published pipeline for the tokenizer.json shapes those models ship:
//   added-token extraction  -> tokenizers/src/tokenizer/added_vocabulary.rs (leftmost-longest, lstrip/rstrip/single_word)
//   normalizer              -> none | NFC
//   pre-tokenizer           -> Split(<Llama 3 | Qwen 2 regex>, Isolated) + ByteLevel(use_regex = false), or
//                              ByteLevel(use_regex = true) (the GPT-2 regex)
//   model                   -> BPE (merge ranks, ignore_merges), no dropout
//   post-processor          -> skipped (add_special_tokens = false)
// and the SentencePiece-style shape of Llama 2 / Mistral files: Prepend + Replace normalizer (or a non-splitting Metaspace
// pre-tokenizer), BPE with byte fallback and a fused <unk>, decoder Replace + ByteFallback + Fuse + Strip.
// Anything else in the file is a load error rather than a silently different tokenization.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace kjarni {

class Json;

class BpeTokenizer {
public:
    void load(const std::string& tokenizer_json_path);
    void load_json(const std::string& text, const std::string& origin);

    // Tokenizer::encode(text, add_special_tokens = false).get_ids(), right-truncated to max_length when set.
    std::vector<uint32_t> encode(const std::string& text) const { return encode(text, max_length_); }
    std::vector<uint32_t> encode(const std::string& text, size_t max_length) const;  // 0 = no truncation
    // Tokenizer::decode(ids, skip_special_tokens) with the ByteLevel decoder: String::from_utf8_lossy of the bytes.
    std::string decode(const std::vector<uint32_t>& ids, bool skip_special) const;
    bool token_to_id(const std::string& token, uint32_t& id) const;
    size_t vocab_size() const { return id_to_token_.size(); }
    void set_truncation(size_t max_length) { max_length_ = max_length; }

    // The pre-tokenizer alone (pieces as UTF-8), for tests.
    std::vector<std::string> pre_tokenize(const std::string& text) const;

private:
    enum class Pattern { Llama3, Qwen2, Gpt2 };
    struct Added {
        std::string content, match;  // match: what is searched for (normalized when the token is)
        uint32_t id = 0;
        bool special = false, normalized = false, lstrip = false, rstrip = false, single_word = false;
    };
    struct Split {
        size_t begin, end;  // byte range of the text
        int64_t token;      // added-token id, or -1 for ordinary text
    };

    void split_on_added(const std::string& text, bool normalized_set, std::vector<Split>& out) const;
    void encode_segment(const std::string& text, std::vector<uint32_t>& out) const;
    void scan_pieces(const std::vector<uint32_t>& cps, std::vector<std::pair<size_t, size_t>>& pieces) const;
    void bpe_word(const std::string& piece_utf8, std::vector<uint32_t>& out) const;
    void encode_segment_sp(const std::string& text, bool at_start, std::vector<uint32_t>& out) const;
    void merge_symbols(std::vector<uint32_t>& sym) const;  // Word::merge_all with a heap: O(n log n) on long words
    std::string decode_sp(const std::vector<uint32_t>& ids, bool skip_special) const;

    std::vector<std::string> id_to_token_;
    std::vector<uint8_t> has_token_, special_;
    std::unordered_map<std::string, uint32_t> vocab_;        // model vocabulary
    std::unordered_map<std::string, uint32_t> token_to_id_;  // vocabulary + added tokens
    std::unordered_map<uint64_t, std::pair<uint32_t, uint32_t>> merges_;  // (a << 32 | b) -> (rank, merged id)
    std::vector<Added> added_;
    std::vector<uint32_t> added_by_first_[256];
    std::string byte_chars_[256];  // byte -> UTF-8 of its ByteLevel character
    std::unordered_map<uint32_t, uint8_t> char_bytes_;
    Pattern pattern_ = Pattern::Gpt2;
    bool nfc_ = false, add_prefix_space_ = false, ignore_merges_ = false, has_unk_ = false, fuse_unk_ = false;
    uint32_t unk_id_ = 0;
    size_t max_length_ = 0;
    // SentencePiece-style mode
    bool sp_mode_ = false;
    int sp_prepend_ = 0;  // 0 = Prepend normalizer (every segment), 1 = Metaspace first, 2 = Metaspace always, 3 = never
    int32_t byte_ids_[256];
};

}  // namespace kjarni
