#include "unigram.h"

#include <algorithm>
#include <cstring>
#include <limits>
#include <stdexcept>

#include "json.h"
#include "unicode.h"

namespace kjarni {

namespace {

#include "unigram_unicode_tables.inc"

template <size_t N>
bool in_ranges(const uint32_t (&t)[N][2], uint32_t cp)
{
    size_t lo = 0, hi = N;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (cp < t[mid][0]) hi = mid;
        else if (cp > t[mid][1]) lo = mid + 1;
        else return true;
    }
    return false;
}

// Grapheme_Cluster_Break classes that can change which bytes the Precompiled normaliser sees as one chunk.  Hangul
// (L / V / T / LV / LVT), Regional_Indicator pairs and the Indic conjunct rule only ever join clusters of >= 6 bytes, which
// the normaliser walks char by char anyway (normalizers/precompiled.rs: `if grapheme.len() < 6`), so they are left out.
enum Gcb { G_OTHER, G_CR, G_LF, G_CONTROL, G_EXTEND, G_ZWJ, G_SPACING, G_PREPEND };

Gcb gcb(uint32_t cp)
{
    if (cp == 0x0D) return G_CR;
    if (cp == 0x0A) return G_LF;
    if (cp == 0x200D) return G_ZWJ;
    if (cp < 0x300 && cp >= 0x20 && cp != 0x7F && cp != 0xAD && !(cp >= 0x80 && cp < 0xA0)) return G_OTHER;
    if (in_ranges(kGcbControl, cp)) return G_CONTROL;
    if (in_ranges(kGcbExtend, cp)) return G_EXTEND;
    if (in_ranges(kGcbSpacingMark, cp)) return G_SPACING;
    if (in_ranges(kGcbPrepend, cp)) return G_PREPEND;
    return G_OTHER;
}

// Length of the sequence a lead byte opens; a stray continuation byte (possible only in strings out of a damaged
// tokenizer.json) counts as one.
size_t utf8_len(uint8_t lead)
{
    if (lead >= 0xC0 && lead < 0xE0) return 2;
    if (lead >= 0xE0 && lead < 0xF0) return 3;
    if (lead >= 0xF0 && lead < 0xF8) return 4;
    return 1;
}

size_t char_len(const std::string& s, size_t pos) { return std::min(utf8_len((uint8_t)s[pos]), s.size() - pos); }

uint32_t decode_one(const char* s, size_t len)
{
    const uint8_t* p = reinterpret_cast<const uint8_t*>(s);
    switch (len) {
    case 1: return p[0];
    case 2: return ((p[0] & 0x1Fu) << 6) | (p[1] & 0x3Fu);
    case 3: return ((p[0] & 0x0Fu) << 12) | ((p[1] & 0x3Fu) << 6) | (p[2] & 0x3Fu);
    default: return ((p[0] & 0x07u) << 18) | ((p[1] & 0x3Fu) << 12) | ((p[2] & 0x3Fu) << 6) | (p[3] & 0x3Fu);
    }
}

std::vector<uint8_t> base64_decode(const std::string& s)
{
    static int8_t table[256];
    static bool init = false;
    if (!init) {
        std::memset(table, -1, sizeof table);
        const char* a = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; ++i) table[(uint8_t)a[i]] = (int8_t)i;
        init = true;
    }
    std::vector<uint8_t> out;
    out.reserve(s.size() * 3 / 4);
    uint32_t acc = 0;
    int bits = 0;
    for (const char ch : s) {
        if (ch == '=') break;
        const int8_t v = table[(uint8_t)ch];
        if (v < 0) throw std::runtime_error("tokenizer.json: precompiled_charsmap is not base64");
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out.push_back((uint8_t)(acc >> bits));
        }
    }
    return out;
}

const char kMeta[] = "\xE2\x96\x81";  // U+2581

}  // namespace

// ------------------------------------------------------------------------------------------------ load

void UnigramTokenizer::load_charsmap(const std::string& base64)
{
    // sentencepiece normalizer.cc DecodePrecompiledCharsMap: u32 trie size | darts-clone units | NUL-separated strings
    const std::vector<uint8_t> blob = base64_decode(base64);
    if (blob.size() < 4) throw std::runtime_error("tokenizer.json: precompiled_charsmap is too short");
    const uint32_t trie_bytes = (uint32_t)blob[0] | ((uint32_t)blob[1] << 8) | ((uint32_t)blob[2] << 16) | ((uint32_t)blob[3] << 24);
    if (trie_bytes % 4 || (size_t)trie_bytes > blob.size() - 4) throw std::runtime_error("tokenizer.json: precompiled_charsmap is corrupt");
    trie_.resize(trie_bytes / 4);
    std::memcpy(trie_.data(), blob.data() + 4, trie_bytes);
    normalized_blob_.assign(reinterpret_cast<const char*>(blob.data()) + 4 + trie_bytes, blob.size() - 4 - trie_bytes);
    if (!unicode::is_valid_utf8(normalized_blob_.data(), normalized_blob_.size()))
        throw std::runtime_error("tokenizer.json: precompiled_charsmap strings are not UTF-8");
}

void UnigramTokenizer::parse_normalizer(const Json& n)
{
    const std::string type = n.get_string("type", "");
    if (type == "Sequence") {
        for (const Json& c : n.at("normalizers").arr) parse_normalizer(c);
    } else if (type == "Precompiled") {
        if (!trie_.empty()) throw std::runtime_error("tokenizer.json: more than one Precompiled normalizer");
        load_charsmap(n.get_string("precompiled_charsmap", ""));
        norm_.push_back({NORM_PRECOMPILED, "", "", false, false});
    } else if (type == "Replace") {
        const Json& pat = n.at("pattern");
        NormStep st{NORM_REPLACE_LITERAL, "", n.get_string("content", ""), false, false};
        if (const Json* re = pat.find("Regex")) {
            if (re->str != " {2,}") throw std::runtime_error("unsupported normalizer Replace regex '" + re->str + "' (only ' {2,}' is)");
            st.kind = NORM_REPLACE_SPACES;
        } else {
            st.from = pat.get_string("String", "");
            if (st.from.empty()) throw std::runtime_error("normalizer Replace with an empty pattern");
        }
        norm_.push_back(st);
    } else if (type == "Strip") {
        norm_.push_back({NORM_STRIP, "", "", n.get_bool("strip_left", true), n.get_bool("strip_right", true)});
    } else {
        throw std::runtime_error("unsupported normalizer '" + type + "' for a Unigram tokenizer (Precompiled, Replace, Strip and Sequence are)");
    }
}

void UnigramTokenizer::parse_pre_tokenizer(const Json& p)
{
    const std::string type = p.get_string("type", "");
    if (type == "Sequence") {
        for (const Json& c : p.at("pretokenizers").arr) parse_pre_tokenizer(c);
    } else if (type == "WhitespaceSplit") {
        if (metaspace_) throw std::runtime_error("unsupported pre-tokenizer order: WhitespaceSplit after Metaspace");
        whitespace_split_ = true;
    } else if (type == "Metaspace") {
        metaspace_ = true;
        replacement_ = p.get_string("replacement", kMeta);
        if (replacement_ != kMeta) throw std::runtime_error("unsupported Metaspace replacement (only U+2581 is)");
        const Json* scheme = p.find("prepend_scheme");
        if (scheme && scheme->is_string()) {
            prepend_scheme_ = scheme->str == "always" ? 0 : scheme->str == "first" ? 1 : 2;
            if (scheme->str != "always" && scheme->str != "first" && scheme->str != "never")
                throw std::runtime_error("unknown Metaspace prepend_scheme '" + scheme->str + "'");
        } else {
            prepend_scheme_ = p.get_bool("add_prefix_space", true) ? 0 : 2;  // the pre-0.19 spelling
        }
        metaspace_split_ = p.get_bool("split", true);
    } else {
        throw std::runtime_error("unsupported pre-tokenizer '" + type + "' for a Unigram tokenizer (WhitespaceSplit, Metaspace and Sequence are)");
    }
}

void UnigramTokenizer::load_json(const std::string& text, const std::string& origin)
{
    const Json root = Json::parse(text);
    const Json& model = root.at("model");
    if (model.get_string("type", "") != "Unigram") throw std::runtime_error(origin + ": model.type is not Unigram");
    const Json& vocab = model.at("vocab");
    if (!vocab.is_array() || vocab.arr.empty()) throw std::runtime_error(origin + ": model.vocab must be a non-empty array");
    pieces_.reserve(vocab.arr.size());
    min_score_ = std::numeric_limits<double>::infinity();
    for (const Json& e : vocab.arr) {
        if (!e.is_array() || e.arr.size() != 2 || !e.arr[0].is_string() || !e.arr[1].is_number())
            throw std::runtime_error(origin + ": model.vocab entries must be [piece, score]");
        pieces_.emplace_back(e.arr[0].str, e.arr[1].num);
        min_score_ = std::min(min_score_, e.arr[1].num);
    }
    if (const Json* u = model.find("unk_id"); u && u->is_number()) {
        has_unk_ = true;
        unk_id_ = (uint32_t)u->as_int();
        if (unk_id_ >= pieces_.size()) throw std::runtime_error(origin + ": model.unk_id is outside the vocabulary");
    }
    byte_fallback_ = model.get_bool("byte_fallback", false);
    sorted_.reserve(pieces_.size());
    for (size_t i = 0; i < pieces_.size(); ++i) sorted_.emplace_back(pieces_[i].first, (uint32_t)i);
    // token_to_ids.insert in vocabulary order: the last id of a repeated piece wins
    std::stable_sort(sorted_.begin(), sorted_.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    size_t w = 0;
    for (size_t i = 0; i < sorted_.size(); ++i) {
        if (w > 0 && sorted_[w - 1].first == sorted_[i].first) sorted_[w - 1].second = sorted_[i].second;
        else sorted_[w++] = sorted_[i];
    }
    sorted_.resize(w);
    {
        size_t i = 0;
        for (int b = 0; b < 256; ++b) {
            while (i < sorted_.size() && (sorted_[i].first.empty() || (uint8_t)sorted_[i].first[0] < b)) ++i;
            first_[b] = (uint32_t)i;
        }
        first_[256] = (uint32_t)sorted_.size();
    }

    if (const Json* n = root.find("normalizer"); n && !n->is_null()) parse_normalizer(*n);
    if (const Json* p = root.find("pre_tokenizer"); p && !p->is_null()) parse_pre_tokenizer(*p);
    if (const Json* added = root.find("added_tokens"); added && added->is_array())
        for (const Json& a : added->arr) {
            AddedToken t;
            t.content = a.get_string("content", "");
            t.id = (uint32_t)a.get_int("id", 0);
            t.special = a.get_bool("special", false);
            t.single_word = a.get_bool("single_word", false);
            t.lstrip = a.get_bool("lstrip", false);
            t.rstrip = a.get_bool("rstrip", false);
            t.normalized = a.get_bool("normalized", !t.special);
            if (!t.content.empty()) added_.push_back(t);
        }
}

bool UnigramTokenizer::token_to_id(const std::string& token, uint32_t& id) const
{
    for (const AddedToken& a : added_)
        if (a.content == token) {
            id = a.id;
            return true;
        }
    auto it = std::lower_bound(sorted_.begin(), sorted_.end(), token, [](const auto& e, const std::string& t) { return e.first < t; });
    if (it == sorted_.end() || it->first != token) return false;
    id = it->second;
    return true;
}

// ------------------------------------------------------------------------------------------------ normalizer

// spm_precompiled DoubleArray::common_prefix_search + Precompiled::transform: the replacement of the SHORTEST key that is a
// prefix of the chunk (`results[0]`), not of the whole chunk.
bool UnigramTokenizer::charsmap_transform(const char* s, size_t len, const char*& out, size_t& out_len) const
{
    if (trie_.empty()) return false;
    auto offset = [](uint32_t u) { return (size_t)((u >> 10) << ((u & (1u << 9)) >> 6)); };
    size_t node = 0;
    uint32_t unit = trie_[0];
    node ^= offset(unit);
    for (size_t i = 0; i < len; ++i) {
        const uint8_t c = (uint8_t)s[i];
        if (c == 0) break;
        node ^= c;
        if (node >= trie_.size()) return false;
        unit = trie_[node];
        if ((unit & ((1u << 31) | 0xFFu)) != c) return false;
        node ^= offset(unit);
        if (node >= trie_.size()) return false;
        if ((unit >> 8) & 1u) {
            const size_t index = trie_[node] & 0x7FFFFFFFu;
            if (index >= normalized_blob_.size()) return false;
            out = normalized_blob_.data() + index;
            out_len = std::strlen(out);  // the blob is NUL separated; std::string keeps a terminator behind the last one
            return true;
        }
    }
    return false;
}

// normalizers/precompiled.rs: walk extended grapheme clusters; a cluster shorter than 6 bytes is looked up whole, otherwise
// (or when that finds nothing) char by char.
void UnigramTokenizer::precompiled(const std::string& in, std::string& out) const
{
    out.clear();
    out.reserve(in.size());
    const size_t n = in.size();
    size_t i = 0;
    while (i < n) {
        // one extended grapheme cluster [i, j)
        size_t j = i;
        size_t cl = char_len(in, j);
        uint32_t cp = decode_one(in.data() + j, cl);
        Gcb prev = gcb(cp);
        bool pict_chain = in_ranges(kExtPict, cp);  // ExtPict Extend* so far (GB11)
        bool after_zwj_of_pict = false;
        j += cl;
        while (j < n) {
            cl = char_len(in, j);
            cp = decode_one(in.data() + j, cl);
            const Gcb cur = gcb(cp);
            bool join;
            if (prev == G_CR && cur == G_LF) join = true;                                    // GB3
            else if (prev == G_CR || prev == G_LF || prev == G_CONTROL) join = false;        // GB4
            else if (cur == G_CR || cur == G_LF || cur == G_CONTROL) join = false;           // GB5
            else if (cur == G_EXTEND || cur == G_ZWJ || cur == G_SPACING) join = true;       // GB9, GB9a
            else if (prev == G_PREPEND) join = true;                                         // GB9b
            else if (after_zwj_of_pict && in_ranges(kExtPict, cp)) join = true;              // GB11
            else join = false;
            if (!join) break;
            after_zwj_of_pict = pict_chain && cur == G_ZWJ;
            if (cur == G_EXTEND) { /* the chain continues */ }
            else if (cur == G_ZWJ) pict_chain = false;
            else pict_chain = in_ranges(kExtPict, cp);
            prev = cur;
            j += cl;
        }
        const char* rep;
        size_t rep_len;
        if (j - i < 6 && charsmap_transform(in.data() + i, j - i, rep, rep_len)) {
            out.append(rep, rep_len);
        } else {
            for (size_t k = i; k < j;) {
                const size_t l = char_len(in, k);
                if (charsmap_transform(in.data() + k, l, rep, rep_len)) out.append(rep, rep_len);
                else out.append(in, k, l);
                k += l;
            }
        }
        i = j;
    }
}

std::string UnigramTokenizer::normalize(const std::string& text) const
{
    std::string cur = text, next;
    for (const NormStep& st : norm_) {
        switch (st.kind) {
        case NORM_PRECOMPILED:
            precompiled(cur, next);
            cur.swap(next);
            break;
        case NORM_REPLACE_SPACES: {  // Regex " {2,}": every maximal run of two or more U+0020
            next.clear();
            for (size_t i = 0; i < cur.size();) {
                if (cur[i] == ' ') {
                    size_t k = i;
                    while (k < cur.size() && cur[k] == ' ') ++k;
                    if (k - i >= 2) next += st.to;
                    else next += ' ';
                    i = k;
                } else {
                    next += cur[i++];
                }
            }
            cur.swap(next);
            break;
        }
        case NORM_REPLACE_LITERAL: {
            next.clear();
            for (size_t i = 0; i < cur.size();) {
                if (cur.compare(i, st.from.size(), st.from) == 0) {
                    next += st.to;
                    i += st.from.size();
                } else {
                    next += cur[i++];
                }
            }
            cur.swap(next);
            break;
        }
        case NORM_STRIP: {  // NormalizedString::strip: char::is_whitespace on either side
            size_t b = 0, e = cur.size();
            if (st.left)
                while (b < e) {
                    const size_t l = char_len(cur, b);
                    if (!unicode::is_whitespace(decode_one(cur.data() + b, l))) break;
                    b += l;
                }
            if (st.right)
                while (e > b) {
                    size_t k = e - 1;
                    while (k > b && ((uint8_t)cur[k] & 0xC0) == 0x80) --k;
                    if (!unicode::is_whitespace(decode_one(cur.data() + k, std::min<size_t>(e - k, 4)))) break;
                    e = k;
                }
            cur = cur.substr(b, e - b);
            break;
        }
        }
    }
    return cur;
}

// ------------------------------------------------------------------------------------------------ pre-tokenizer

std::vector<std::string> UnigramTokenizer::pre_tokenize(const std::string& s) const
{
    // WhitespaceSplit: split on char::is_whitespace, separators removed, empty pieces dropped
    std::vector<std::pair<size_t, std::string>> words;  // (byte offset in s, piece)
    if (whitespace_split_) {
        size_t start = 0, i = 0;
        while (i < s.size()) {
            const size_t l = char_len(s, i);
            if (unicode::is_whitespace(decode_one(s.data() + i, l))) {
                if (i > start) words.emplace_back(start, s.substr(start, i - start));
                start = i + l;
            }
            i += l;
        }
        if (s.size() > start) words.emplace_back(start, s.substr(start));
    } else if (!s.empty()) {
        words.emplace_back(0, s);
    }
    std::vector<std::string> out;
    if (!metaspace_) {
        for (auto& w : words) out.push_back(std::move(w.second));
        return out;
    }
    const std::string& rep = replacement_;
    for (auto& [off, w0] : words) {
        // Metaspace::pre_tokenize: ' ' -> U+2581, prepend by scheme, split MergedWithNext on U+2581
        std::string w;
        w.reserve(w0.size() + 3);
        for (const char c : w0) {
            if (c == ' ') w += rep;
            else w += c;
        }
        const bool starts = w.compare(0, rep.size(), rep) == 0;
        if (!starts && (prepend_scheme_ == 0 || (prepend_scheme_ == 1 && off == 0))) w.insert(0, rep);
        if (!metaspace_split_) {
            if (!w.empty()) out.push_back(std::move(w));
            continue;
        }
        size_t start = 0, i = 0;
        while (i < w.size()) {
            if (w.compare(i, rep.size(), rep) == 0) {
                if (i > start) out.push_back(w.substr(start, i - start));
                start = i;  // the delimiter opens the next piece
                i += rep.size();
            } else {
                ++i;
            }
        }
        if (w.size() > start) out.push_back(w.substr(start));
    }
    return out;
}

// ------------------------------------------------------------------------------------------------ model

template <typename F>
void UnigramTokenizer::common_prefixes(const std::string& s, size_t pos, F&& f) const
{
    if (pos >= s.size()) return;
    // depth 0 from the first-byte table, then binary search inside the shrinking range
    size_t lo = first_[(uint8_t)s[pos]], hi = first_[(uint8_t)s[pos] + 1];
    if (lo < hi && sorted_[lo].first.size() == 1) f(1, sorted_[lo].second);
    for (size_t d = 1; pos + d < s.size() && lo < hi; ++d) {
        const uint8_t c = (uint8_t)s[pos + d];
        // entries of length <= d sort first inside the range and cannot continue
        auto key = [&](size_t idx) -> int { return sorted_[idx].first.size() > d ? (int)(uint8_t)sorted_[idx].first[d] : -1; };
        size_t a = lo, b = hi;
        while (a < b) {
            const size_t mid = (a + b) / 2;
            if (key(mid) < (int)c) a = mid + 1;
            else b = mid;
        }
        const size_t new_lo = a;
        b = hi;
        while (a < b) {
            const size_t mid = (a + b) / 2;
            if (key(mid) <= (int)c) a = mid + 1;
            else b = mid;
        }
        lo = new_lo;
        hi = a;
        if (lo < hi && sorted_[lo].first.size() == d + 1) f(d + 1, sorted_[lo].second);
    }
}

// models/unigram/model.rs: encode_optimized (Viterbi over byte positions, an unknown character costs min_score - 10, runs
// of unknowns are fused into one piece) followed by tokenize (piece -> id, byte fallback, unk_id).
void UnigramTokenizer::encode_word(const std::string& word, std::vector<uint32_t>& ids) const
{
    const size_t size = word.size();
    if (size == 0) return;
    struct Node {
        uint32_t id = 0;
        double score = 0.0;
        int64_t starts_at = -1;
    };
    std::vector<Node> best(size + 1);
    const double unk_score = min_score_ - 10.0;
    size_t at = 0;
    while (at < size) {
        const double here = best[at].score;
        bool has_single = false;
        const size_t mblen = char_len(word, at);
        common_prefixes(word, at, [&](size_t len, uint32_t id) {
            Node& target = best[at + len];
            const double cand = pieces_[id].second + here;
            if (target.starts_at < 0 || cand > target.score) {
                target.score = cand;
                target.starts_at = (int64_t)at;
                target.id = id;
            }
            if (!has_single && len == mblen) has_single = true;
        });
        if (!has_single) {
            if (!has_unk_) throw std::runtime_error("Unigram tokenizer: a character is not in the vocabulary and the model has no unk_id");
            Node& target = best[at + mblen];
            const double cand = unk_score + here;
            if (target.starts_at < 0 || cand > target.score) {
                target.score = cand;
                target.starts_at = (int64_t)at;
                target.id = unk_id_;
            }
        }
        at += mblen;
    }
    // walk back; fused unknown runs become one piece
    struct Piece {
        size_t begin, end;
        bool unk;
    };
    std::vector<Piece> rev;
    size_t end = size;
    bool open_unk = false;
    while (end > 0) {
        const Node& node = best[end];
        const size_t begin = (size_t)node.starts_at;
        const bool is_unk = fuse_unk_ && has_unk_ && node.id == unk_id_;
        if (is_unk && open_unk) {
            rev.back().begin = begin;
        } else {
            rev.push_back({begin, end, is_unk});
            open_unk = is_unk;
        }
        if (!is_unk) open_unk = false;
        end = begin;
    }
    for (size_t r = rev.size(); r-- > 0;) {
        const std::string piece = word.substr(rev[r].begin, rev[r].end - rev[r].begin);
        uint32_t id;
        auto it = std::lower_bound(sorted_.begin(), sorted_.end(), piece, [](const auto& e, const std::string& t) { return e.first < t; });
        if (it != sorted_.end() && it->first == piece) {
            ids.push_back(it->second);
            continue;
        }
        if (byte_fallback_) {
            std::vector<uint32_t> bytes;
            bool all = true;
            for (const char ch : piece) {
                static const char* hex = "0123456789ABCDEF";
                const uint8_t b = (uint8_t)ch;
                const std::string name = std::string("<0x") + hex[b >> 4] + hex[b & 15] + ">";
                auto bt = std::lower_bound(sorted_.begin(), sorted_.end(), name, [](const auto& e, const std::string& t) { return e.first < t; });
                if (bt == sorted_.end() || bt->first != name) {
                    all = false;
                    break;
                }
                bytes.push_back(bt->second);
            }
            if (all) {
                ids.insert(ids.end(), bytes.begin(), bytes.end());
                continue;
            }
        }
        if (!has_unk_) throw std::runtime_error("Unigram tokenizer: a piece is not in the vocabulary and the model has no unk_id");
        id = unk_id_;
        ids.push_back(id);
    }
}

void UnigramTokenizer::encode_segment(const std::string& raw, std::vector<uint32_t>& ids) const
{
    const std::string normalized = normalize(raw);
    for (const std::string& word : pre_tokenize(normalized)) encode_word(word, ids);
}

// AddedVocabulary::extract_and_normalize for tokens with normalized == false (every special token of these checkpoints):
// leftmost-longest matches are cut out of the raw text, lstrip / rstrip swallow neighbouring whitespace, single_word needs
// non-word neighbours; what lies between goes through the normaliser, the pre-tokenizer and the model.
std::vector<uint32_t> UnigramTokenizer::encode(const std::string& text) const
{
    std::vector<uint32_t> ids;
    const size_t n = text.size();
    if (n == 0) return ids;
    for (const AddedToken& a : added_)
        if (a.normalized) throw std::runtime_error("Unigram tokenizer: added tokens matched against the normalised text are not supported");
    auto word_char = [](uint32_t c) { return c == '_' || unicode::is_alphanumeric(c) || unicode::is_mark_nonspacing(c); };
    auto prev_cp = [&](size_t pos, size_t& start) {
        size_t k = pos;
        do --k; while (k > 0 && ((uint8_t)text[k] & 0xC0) == 0x80);
        start = k;
        return decode_one(text.data() + k, pos - k);
    };
    auto next_cp = [&](size_t pos, size_t& end) {
        const size_t l = std::min(utf8_len((uint8_t)text[pos]), n - pos);
        end = pos + l;
        return decode_one(text.data() + pos, l);
    };
    size_t start_offset = 0, pos = 0;
    while (pos < n) {
        const AddedToken* best = nullptr;
        for (const AddedToken& t : added_) {
            if (t.content.size() > n - pos || text[pos] != t.content[0]) continue;
            if (std::memcmp(text.data() + pos, t.content.data(), t.content.size()) != 0) continue;
            if (!best || t.content.size() > best->content.size()) best = &t;
        }
        if (!best) {
            ++pos;
            continue;
        }
        size_t start = pos, stop = pos + best->content.size();
        pos = stop;
        if (best->single_word) {
            size_t tmp;
            const bool start_space = start == 0 || !word_char(prev_cp(start, tmp));
            const bool stop_space = stop == n || !word_char(next_cp(stop, tmp));
            if (!start_space || !stop_space) continue;
        }
        if (best->lstrip)
            while (start > start_offset) {
                size_t k;
                if (!unicode::is_whitespace(prev_cp(start, k))) break;
                start = k;
            }
        if (best->rstrip)
            while (stop < n) {
                size_t k;
                if (!unicode::is_whitespace(next_cp(stop, k))) break;
                stop = k;
            }
        if (start_offset < start) encode_segment(text.substr(start_offset, start - start_offset), ids);
        ids.push_back(best->id);
        start_offset = stop;
        if (pos < stop) pos = stop;
    }
    if (start_offset < n) encode_segment(text.substr(start_offset), ids);
    return ids;
}

}  // namespace kjarni
