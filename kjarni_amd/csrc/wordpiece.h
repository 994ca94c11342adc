// Encoder-side tokenizer reading HF tokenizer.json: BERT WordPiece (BERT, DistilBERT, MPNet) or byte-level BPE
// (RoBERTa; the model itself is bpe.h), with the post-processing, truncation and padding the reference applies.
//
// The reference tokenises with the HF `tokenizers` crate (Tokenizer::from_file,
// crates/kjarni-transformers/src/pipeline/encoder/loader.rs:98-115; call sites
// cpu/encoder/traits.rs:141-145, kjarni-models/.../cross_encoder/model.rs:176-179,
// .../sequence_classifier/mod.rs:272-275).  That crate is not vendored; this is
// a restatement of its pipeline for the BERT family:
//   added-token split -> BertNormalizer -> BertPreTokenizer -> WordPiece
//   (or: added-token split -> ByteLevel + BPE, for a RoBERTa tokenizer.json)
//   -> truncation (LongestFirst, right) -> TemplateProcessing / BertProcessing / RobertaProcessing
//   -> BatchLongest right padding (pad id 0, type 0).
// Token ids are integer work: they must be bit-exact, and are pinned against the
// same Rust core through Python `tokenizers` in tests/test_tokenizer.py.
// The in-repo fallback tokenizer of the reference
// (crates/kjarni-transformers/src/tokenizer/wordpiece.rs:56-135) has the same
// greedy longest-match core; its known-answer tests are reproduced too.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

namespace kjarni {

class BpeTokenizer;
class UnigramTokenizer;

struct Encoding {
    std::vector<uint32_t> ids, type_ids, attention_mask;
};

struct BatchEncoding {
    size_t batch = 0, seq = 0;
    std::vector<uint32_t> ids, type_ids, attention_mask;  // [batch, seq] row-major
};

class BertTokenizer {
public:
    // Throws std::runtime_error when the file is missing or uses a pipeline
    // other than Bert normalizer / Bert pre-tokenizer / WordPiece.
    static BertTokenizer from_file(const std::string& path);
    static BertTokenizer from_json(const std::string& json_text);

    // Truncation: loader.rs:108-111 sets max_length = max_seq_len, other params default.
    void set_max_length(size_t n) { max_length_ = n; }
    size_t max_length() const { return max_length_; }

    // text_b == nullptr: single sequence.  add_special_tokens = true everywhere in the reference.
    Encoding encode(const std::string& text_a, const std::string* text_b) const;
    // BatchLongest padding (loader.rs:112-115).
    BatchEncoding encode_batch(const std::vector<std::string>& texts) const;
    BatchEncoding encode_batch_pairs(const std::vector<std::pair<std::string, std::string>>& pairs) const;

    size_t vocab_size() const;
    int64_t token_to_id(const std::string& tok) const;

    // Exposed for tests: the normalised string of a plain segment.
    std::string normalize(const std::string& text) const;

private:
    struct AddedToken {
        std::string content;
        uint32_t id = 0;
        bool special = false, single_word = false, lstrip = false, rstrip = false, normalized = false;
    };
    struct TemplatePiece {
        bool is_special = false;
        uint32_t id = 0;      // special token id
        int sequence = 0;     // 0 = A, 1 = B
        uint32_t type_id = 0;
    };

    void parse_post_processor(const class Json& root);
    void tokenize_sequence(const std::string& text, std::vector<uint32_t>& ids) const;
    void tokenize_segment(const std::vector<uint32_t>& cps, std::vector<uint32_t>& ids) const;
    void wordpiece(const std::vector<uint32_t>& word, std::vector<uint32_t>& ids) const;
    void normalize_cps(const std::vector<uint32_t>& in, std::vector<uint32_t>& out) const;
    static BatchEncoding pad_batch(std::vector<Encoding>& encs);

    std::shared_ptr<const BpeTokenizer> bpe_;  // set: the sequence tokenizer is byte-level BPE (RoBERTa)
    std::shared_ptr<const UnigramTokenizer> unigram_;  // set: SentencePiece Unigram (XLM-R / bge-m3)
    std::unordered_map<std::string, uint32_t> vocab_;
    std::vector<AddedToken> added_;  // matched in raw text (normalized == false) or in normalised text
    std::string unk_token_ = "[UNK]";
    uint32_t unk_id_ = 0;
    std::string prefix_ = "##";
    size_t max_chars_per_word_ = 100;
    bool clean_text_ = true, handle_chinese_ = true, strip_accents_ = true, lowercase_ = true;
    bool has_normalizer_ = true;
    std::vector<TemplatePiece> single_, pair_;
    size_t max_length_ = 512;
    uint32_t pad_id_ = 0;
};

}  // namespace kjarni
