// Unicode helpers for the BERT tokenizer.  The property tables
// (unicode_tables.inc) are probed out of the HF `tokenizers` core the reference
// links (see tools/gen_unicode_tables.py), so normalisation agrees with it code
// point by code point.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace kjarni {
namespace unicode {

// Strict UTF-8 decode (what Rust's CStr::to_str accepts).  Returns false on
// invalid input (overlongs, surrogates, > U+10FFFF, truncated sequences).
bool decode_utf8(const char* s, size_t len, std::vector<uint32_t>& out);
bool is_valid_utf8(const char* s, size_t len);
void append_utf8(std::string& out, uint32_t cp);
std::string encode_utf8(const std::vector<uint32_t>& cps);

// BertNormalizer::do_clean_text: 0 keep, 1 drop (NUL, U+FFFD, Cc/Cf/Co except \t \n \r), 2 -> ' '.
int clean_class(uint32_t cp);
bool is_cjk(uint32_t cp);                 // BertNormalizer::is_chinese_char
bool is_mark_nonspacing(uint32_t cp);     // unicode_categories Mn
bool is_whitespace(uint32_t cp);          // Rust char::is_whitespace
bool is_bert_punctuation(uint32_t cp);    // ascii punctuation or Unicode P*
bool is_alphanumeric(uint32_t cp);        // Rust char::is_alphanumeric (BM25 tokenizer)
int combining_class(uint32_t cp);         // canonical combining class
// Canonical decomposition (NFD) of a sequence, with canonical reordering.
void nfd(const std::vector<uint32_t>& in, std::vector<uint32_t>& out);
// Per-char to_lowercase (may expand, e.g. U+0130 -> "i̇").
void lowercase(const std::vector<uint32_t>& in, std::vector<uint32_t>& out);
// Rust str::to_lowercase: the same, plus the context-sensitive final sigma.
void lowercase_str(const std::vector<uint32_t>& in, std::vector<uint32_t>& out);

}  // namespace unicode
}  // namespace kjarni
