// SentencePiece-Unigram tokenizer.json pipeline (XLM-R / bge-m3 layout), as the `tokenizers` crate (0.22, the reference's
// dependency: Cargo.toml:34) runs it for `encode_batch_texts` (crates/kjarni-transformers/src/cpu/encoder/traits.rs:141-145):
//
//   added tokens cut out of the raw text                       tokenizer/added_vocabulary.rs (extract_and_normalize)
//   normalizer  Sequence[Precompiled, Replace, Strip]          normalizers/precompiled.rs + the spm_precompiled crate
//   pre-tokenizer Sequence[WhitespaceSplit, Metaspace]         pre_tokenizers/{whitespace,metaspace}.rs
//   model       Unigram (Viterbi best path, fused unknowns)    models/unigram/model.rs (encode_optimized, tokenize)
//
// The framing (TemplateProcessing), truncation and padding stay in BertTokenizer (wordpiece.cpp).  Anything outside this
// pipeline fails at load with a message naming the unsupported piece.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace kjarni {

class Json;

class UnigramTokenizer {
public:
    void load_json(const std::string& text, const std::string& origin);
    size_t vocab_size() const { return pieces_.size(); }
    bool token_to_id(const std::string& token, uint32_t& id) const;
    // ids of one sequence, no framing
    std::vector<uint32_t> encode(const std::string& text) const;
    // test hooks
    std::string normalize(const std::string& text) const;
    std::vector<std::string> pre_tokenize(const std::string& normalized) const;

private:
    struct AddedToken {
        std::string content;
        uint32_t id = 0;
        bool special = false, single_word = false, lstrip = false, rstrip = false, normalized = false;
    };
    enum NormKind { NORM_PRECOMPILED, NORM_REPLACE_SPACES, NORM_REPLACE_LITERAL, NORM_STRIP };
    struct NormStep {
        NormKind kind;
        std::string from, to;    // Replace
        bool left = false, right = false;  // Strip
    };

    void parse_normalizer(const Json& n);
    void parse_pre_tokenizer(const Json& p);
    void load_charsmap(const std::string& base64);
    bool charsmap_transform(const char* s, size_t len, const char*& out, size_t& out_len) const;
    void precompiled(const std::string& in, std::string& out) const;
    void encode_word(const std::string& word, std::vector<uint32_t>& ids) const;
    // longest-first is not needed: every vocabulary piece that is a prefix of s[pos..] in increasing length
    template <typename F>
    void common_prefixes(const std::string& s, size_t pos, F&& f) const;
    void encode_segment(const std::string& raw, std::vector<uint32_t>& ids) const;

    std::vector<std::pair<std::string, double>> pieces_;            // id -> (piece, score)
    std::vector<std::pair<std::string, uint32_t>> sorted_;          // piece -> id (last id wins), byte-wise sorted
    uint32_t first_[257] = {};                                       // sorted_ range of every first byte
    double min_score_ = 0.0;
    bool has_unk_ = false, byte_fallback_ = false, fuse_unk_ = true;
    uint32_t unk_id_ = 0;

    std::vector<NormStep> norm_;
    std::vector<uint32_t> trie_;   // darts-clone units of the precompiled character map
    std::string normalized_blob_;  // NUL-separated replacement strings

    bool whitespace_split_ = false, metaspace_ = false, metaspace_split_ = true;
    int prepend_scheme_ = 0;  // 0 always, 1 first, 2 never
    std::string replacement_ = "\xE2\x96\x81";

    std::vector<AddedToken> added_;
};

}  // namespace kjarni
