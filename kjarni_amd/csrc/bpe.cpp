// Byte-level BPE tokenizer (see bpe.h for the pipeline being restated).
#include "host_util.h"
#include "bpe.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <queue>
#include <sstream>
#include <stdexcept>

#include "json.h"
#include "unicode.h"

namespace kjarni {
namespace {

#include "bpe_unicode_tables.inc"

template <size_t N>
bool in_table(const uint32_t (&t)[N][2], uint32_t cp)
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

// \p{L}, \p{N}, \s as Oniguruma evaluates them.
inline bool is_letter(uint32_t c)
{
    if (c < 0x80) return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z');
    return in_table(kBpeLetterRanges, c);
}
inline bool is_number(uint32_t c)
{
    if (c < 0x80) return c >= '0' && c <= '9';
    return in_table(kBpeNumberRanges, c);
}
inline bool is_space(uint32_t c)
{
    if (c < 0x80) return c == ' ' || (c >= 9 && c <= 13);
    return in_table(kBpeSpaceRanges, c);
}
inline bool is_newline(uint32_t c) { return c == '\r' || c == '\n'; }
inline bool is_other(uint32_t c) { return !is_space(c) && !is_letter(c) && !is_number(c); }  // [^\s\p{L}\p{N}]

const char* const kLlama3Pattern =
    "(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\\r\\n\\p{L}\\p{N}]?\\p{L}+|\\p{N}{1,3}| ?[^\\s\\p{L}\\p{N}]+[\\r\\n]*|\\s*[\\r\\n]+|\\s+(?!\\S)|\\s+";
const char* const kQwen2Pattern =
    "(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\\r\\n\\p{L}\\p{N}]?\\p{L}+|\\p{N}| ?[^\\s\\p{L}\\p{N}]+[\\r\\n]*|\\s*[\\r\\n]+|\\s+(?!\\S)|\\s+";
const char* const kGpt2Pattern = "'s|'t|'re|'ve|'m|'ll|'d| ?\\p{L}+| ?\\p{N}+| ?[^\\s\\p{L}\\p{N}]+|\\s+(?!\\S)|\\s+";

// Length (in code points) of a contraction suffix at cps[i] == '\'' or 0.  Alternation order 's 't 're 've 'm 'll 'd;
// `fold`: (?i:...) -- ASCII case plus U+017F, the only other letter Oniguruma folds onto one of these.
size_t contraction(const std::vector<uint32_t>& cps, size_t i, bool fold)
{
    const size_t n = cps.size();
    if (i + 1 >= n) return 0;
    auto lower = [&](uint32_t c) -> uint32_t {
        if (!fold) return c;
        if (c >= 'A' && c <= 'Z') return c + 32;
        if (c == 0x17F) return 's';
        return c;
    };
    const uint32_t a = lower(cps[i + 1]);
    const uint32_t b = i + 2 < n ? lower(cps[i + 2]) : 0;
    if (a == 's' || a == 't') return 2;
    if (a == 'r' && b == 'e') return 3;
    if (a == 'v' && b == 'e') return 3;
    if (a == 'm') return 2;
    if (a == 'l' && b == 'l') return 3;
    if (a == 'd') return 2;
    return 0;
}

// String::from_utf8_lossy: every maximal invalid subpart becomes U+FFFD.
std::string from_utf8_lossy(const std::string& in)
{
    std::string out;
    out.reserve(in.size());
    const size_t n = in.size();
    size_t i = 0;
    auto cont = [&](size_t k, uint8_t lo, uint8_t hi) { return k < n && (uint8_t)in[k] >= lo && (uint8_t)in[k] <= hi; };
    while (i < n) {
        const uint8_t c = (uint8_t)in[i];
        size_t len = 0;  // valid sequence length, or 0 with `bad` = bytes to replace
        size_t bad = 1;
        if (c < 0x80) len = 1;
        else if (c >= 0xC2 && c <= 0xDF) {
            if (cont(i + 1, 0x80, 0xBF)) len = 2;
        } else if (c >= 0xE0 && c <= 0xEF) {
            const uint8_t lo = c == 0xE0 ? 0xA0 : 0x80, hi = c == 0xED ? 0x9F : 0xBF;
            if (cont(i + 1, lo, hi)) {
                if (cont(i + 2, 0x80, 0xBF)) len = 3;
                else bad = 2;
            }
        } else if (c >= 0xF0 && c <= 0xF4) {
            const uint8_t lo = c == 0xF0 ? 0x90 : 0x80, hi = c == 0xF4 ? 0x8F : 0xBF;
            if (cont(i + 1, lo, hi)) {
                if (cont(i + 2, 0x80, 0xBF)) {
                    if (cont(i + 3, 0x80, 0xBF)) len = 4;
                    else bad = 3;
                } else bad = 2;
            }
        }
        if (len) {
            out.append(in, i, len);
            i += len;
        } else {
            out += "\xEF\xBF\xBD";
            i += bad;
        }
    }
    return out;
}

// NFC: canonical decomposition + ordering (unicode::nfd), then canonical composition.
void nfc(const std::vector<uint32_t>& in, std::vector<uint32_t>& out)
{
    bool plain = true;
    for (uint32_t c : in)
        if (c >= 0x300) {
            plain = false;
            break;
        }
    if (plain) {  // nothing below U+0300 decomposes or combines under NFC
        out = in;
        return;
    }
    std::vector<uint32_t> d;
    unicode::nfd(in, d);
    auto compose = [](uint32_t a, uint32_t b) -> uint32_t {
        constexpr uint32_t S = 0xAC00, L = 0x1100, V = 0x1161, T = 0x11A7;
        if (a >= L && a < L + 19 && b >= V && b < V + 21) return S + ((a - L) * 21 + (b - V)) * 28;
        if (a >= S && a < S + 11172 && (a - S) % 28 == 0 && b > T && b < T + 28) return a + (b - T);
        size_t lo = 0, hi = sizeof(kNfcPairs) / sizeof(kNfcPairs[0]);
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            const uint32_t* e = kNfcPairs[mid];
            if (e[0] < a || (e[0] == a && e[1] < b)) lo = mid + 1;
            else hi = mid;
        }
        if (lo < sizeof(kNfcPairs) / sizeof(kNfcPairs[0]) && kNfcPairs[lo][0] == a && kNfcPairs[lo][1] == b) return kNfcPairs[lo][2];
        return 0;
    };
    out.clear();
    out.reserve(d.size());
    ptrdiff_t starter = -1;  // index in `out` of the last starter
    int last_ccc = -1;       // class of the last character kept after the starter (-1: none)
    for (uint32_t c : d) {
        const int cc = unicode::combining_class(c);
        if (starter >= 0) {
            // Not blocked: nothing between, or everything between has a lower non-zero class.
            const bool blocked = last_ccc != -1 && (last_ccc == 0 || last_ccc >= cc);
            if (!blocked) {
                if (const uint32_t m = compose(out[(size_t)starter], c)) {
                    out[(size_t)starter] = m;
                    continue;
                }
            }
        }
        if (cc == 0) {
            starter = (ptrdiff_t)out.size();
            last_ccc = -1;
        } else {
            last_ccc = cc;
        }
        out.push_back(c);
    }
}

}  // namespace

void BpeTokenizer::load(const std::string& path) { load_json(slurp(path), path); }

void BpeTokenizer::load_json(const std::string& text, const std::string& origin)
{
    const Json j = Json::parse(text);
    auto fail = [&](const std::string& what) -> void { throw std::runtime_error(origin + ": " + what); };

    // bytes <-> ByteLevel characters (tokenizers/src/pre_tokenizers/byte_level.rs bytes_char)
    {
        std::vector<int> bs;
        for (int b = '!'; b <= '~'; ++b) bs.push_back(b);
        for (int b = 0xA1; b <= 0xAC; ++b) bs.push_back(b);
        for (int b = 0xAE; b <= 0xFF; ++b) bs.push_back(b);
        std::vector<uint32_t> cs(bs.begin(), bs.end());
        uint32_t n = 0;
        for (int b = 0; b < 256; ++b)
            if (std::find(bs.begin(), bs.end(), b) == bs.end()) {
                bs.push_back(b);
                cs.push_back(256 + n++);
            }
        char_bytes_.clear();
        for (size_t i = 0; i < bs.size(); ++i) {
            std::string s;
            unicode::append_utf8(s, cs[i]);
            byte_chars_[bs[i]] = s;
            char_bytes_[cs[i]] = (uint8_t)bs[i];
        }
    }

    // SentencePiece-style files (Llama 2 / Mistral): no ByteLevel anywhere, byte fallback in the model.
    sp_mode_ = false;
    {
        const Json* model0 = j.find("model");
        const Json* pt0 = j.find("pre_tokenizer");
        const bool no_pt = !pt0 || pt0->is_null();
        const bool metaspace = pt0 && !pt0->is_null() && pt0->get_string("type", "") == "Metaspace";
        if (model0 && model0->get_bool("byte_fallback", false) && (no_pt || metaspace)) sp_mode_ = true;
    }
    if (sp_mode_) {
        const std::string sp = "\xE2\x96\x81";  // U+2581
        const Json* nz = j.find("normalizer");
        const Json* pt0 = j.find("pre_tokenizer");
        const bool has_nz = nz && !nz->is_null();
        const bool has_pt = pt0 && !pt0->is_null();
        if (has_nz) {
            const Json* list = nz->find("normalizers");
            if (nz->get_string("type", "") != "Sequence" || !list || list->arr.size() != 2 || list->arr[0].get_string("type", "") != "Prepend" ||
                list->arr[0].get_string("prepend", "") != sp || list->arr[1].get_string("type", "") != "Replace")
                fail("unsupported normalizer for a byte-fallback BPE (expected Prepend + Replace)");
            const Json* pat = list->arr[1].find("pattern");
            if (!pat || pat->get_string("String", "") != " " || list->arr[1].get_string("content", "") != sp)
                fail("unsupported Replace normalizer (expected \" \" -> U+2581)");
            if (has_pt) fail("Prepend normalizer together with a pre_tokenizer is not supported");
            sp_prepend_ = 0;
        } else if (has_pt) {
            if (pt0->get_string("replacement", sp) != sp || pt0->get_bool("split", true)) fail("unsupported Metaspace pre_tokenizer (expected split = false)");
            const std::string scheme = pt0->get_string("prepend_scheme", "always");
            sp_prepend_ = scheme == "first" ? 1 : (scheme == "always" ? 2 : 3);
        } else {
            sp_prepend_ = 3;
        }
        const Json* dec = j.find("decoder");
        const Json* dl = dec && !dec->is_null() ? dec->find("decoders") : nullptr;
        const bool dec_ok = dl && dl->is_array() && dl->arr.size() == 4 && dl->arr[0].get_string("type", "") == "Replace" &&
                            dl->arr[1].get_string("type", "") == "ByteFallback" && dl->arr[2].get_string("type", "") == "Fuse" &&
                            dl->arr[3].get_string("type", "") == "Strip" && dl->arr[3].get_string("content", "") == " " &&
                            dl->arr[3].get_int("start", 0) == 1 && dl->arr[3].get_int("stop", 0) == 0;
        if (!dec_ok) fail("unsupported decoder for a byte-fallback BPE (expected Replace + ByteFallback + Fuse + Strip(\" \", 1, 0))");
    }

    // normalizer
    nfc_ = false;
    if (!sp_mode_)
    if (const Json* nz = j.find("normalizer"); nz && !nz->is_null()) {
        const std::string t = nz->get_string("type", "");
        if (t == "NFC") nfc_ = true;
        else if (t == "Sequence") {
            const Json* list = nz->find("normalizers");
            if (!list || !list->is_array()) fail("normalizer Sequence without a list");
            for (const Json& e : list->arr) {
                if (e.get_string("type", "") == "NFC") nfc_ = true;
                else fail("unsupported normalizer '" + e.get_string("type", "") + "'");
            }
        } else fail("unsupported normalizer '" + t + "'");
    }

    // pre-tokenizer
    auto byte_level = [&](const Json& e, bool& use_regex, bool& prefix) {
        use_regex = e.get_bool("use_regex", true);
        prefix = e.get_bool("add_prefix_space", true);
    };
    auto pattern_of = [&](const Json& split) -> Pattern {
        const Json* p = split.find("pattern");
        const Json* re = p ? p->find("Regex") : nullptr;
        if (!re || !re->is_string()) fail("Split pre-tokenizer without a Regex pattern");
        if (split.get_string("behavior", "") != "Isolated" || split.get_bool("invert", false))
            fail("Split pre-tokenizer must be Isolated / not inverted");
        if (re->as_string() == kLlama3Pattern) return Pattern::Llama3;
        if (re->as_string() == kQwen2Pattern) return Pattern::Qwen2;
        if (re->as_string() == kGpt2Pattern) return Pattern::Gpt2;
        fail("unsupported pre-tokenizer regex: " + re->as_string());
        return Pattern::Gpt2;
    };
    const Json* pt = j.find("pre_tokenizer");
    if (!sp_mode_ && (!pt || pt->is_null())) fail("no pre_tokenizer (expected ByteLevel)");
    if (!sp_mode_) {
        const std::string t = pt->get_string("type", "");
        bool use_regex = true, prefix = true;
        if (t == "ByteLevel") {
            byte_level(*pt, use_regex, prefix);
            if (!use_regex) fail("ByteLevel pre-tokenizer without a regex and without a Split");
            pattern_ = Pattern::Gpt2;
            add_prefix_space_ = prefix;
        } else if (t == "Sequence") {
            const Json* list = pt->find("pretokenizers");
            if (!list || !list->is_array() || list->arr.size() != 2 || list->arr[0].get_string("type", "") != "Split" ||
                list->arr[1].get_string("type", "") != "ByteLevel")
                fail("unsupported pre_tokenizer Sequence (expected [Split, ByteLevel])");
            pattern_ = pattern_of(list->arr[0]);
            byte_level(list->arr[1], use_regex, prefix);
            if (use_regex || prefix) fail("ByteLevel after Split must have use_regex = false and add_prefix_space = false");
            add_prefix_space_ = false;
        } else fail("unsupported pre_tokenizer '" + t + "'");
    }

    // model
    const Json* model = j.find("model");
    if (!model || model->get_string("type", "BPE") != "BPE") fail("model is not BPE");
    if (const Json* d = model->find("dropout"); d && !d->is_null() && d->as_double() != 0.0) fail("BPE dropout is not supported");
    if (model->get_bool("byte_fallback", false) && !sp_mode_) fail("byte_fallback is not supported with a ByteLevel pre-tokenizer");
    for (const char* key : {"continuing_subword_prefix", "end_of_word_suffix"})
        if (const Json* v = model->find(key); v && v->is_string() && !v->as_string().empty()) fail(std::string(key) + " is not supported");
    ignore_merges_ = model->get_bool("ignore_merges", false);
    fuse_unk_ = model->get_bool("fuse_unk", false);
    const Json* vocab = model->find("vocab");
    if (!vocab || !vocab->is_object()) fail("no model.vocab");
    id_to_token_.clear();
    has_token_.clear();
    special_.clear();
    vocab_.clear();
    token_to_id_.clear();
    auto put = [&](uint32_t id, const std::string& tok, bool special) {
        if (id > (1u << 24)) fail("token id " + std::to_string(id) + " is out of range");  // a corrupt file must not size the tables
        if (id >= id_to_token_.size()) {
            id_to_token_.resize(id + 1);
            has_token_.resize(id + 1, 0);
            special_.resize(id + 1, 0);
        }
        id_to_token_[id] = tok;
        has_token_[id] = 1;
        special_[id] = special ? 1 : 0;
        token_to_id_[tok] = id;
    };
    vocab_.reserve(vocab->obj.size() * 2);
    for (const auto& kv : vocab->obj) {
        const uint32_t id = (uint32_t)kv.second.as_int();
        vocab_[kv.first] = id;
        put(id, kv.first, false);
    }
    has_unk_ = false;
    if (const Json* u = model->find("unk_token"); u && u->is_string()) {
        auto it = vocab_.find(u->as_string());
        if (it != vocab_.end()) {
            has_unk_ = true;
            unk_id_ = it->second;
        }
    }
    for (int b = 0; b < 256; ++b) {
        char name[8];
        std::snprintf(name, sizeof(name), "<0x%02X>", b);
        auto it = vocab_.find(name);
        byte_ids_[b] = it == vocab_.end() ? -1 : (int32_t)it->second;
    }
    merges_.clear();
    if (const Json* merges = model->find("merges"); merges && merges->is_array()) {
        merges_.reserve(merges->arr.size() * 2);
        uint32_t rank = 0;
        for (const Json& m : merges->arr) {
            std::string a, b;
            if (m.is_string()) {
                const std::string& s = m.as_string();
                const size_t sp = s.find(' ');
                if (sp == std::string::npos) fail("malformed merge '" + s + "'");
                a = s.substr(0, sp);
                b = s.substr(sp + 1);
            } else if (m.is_array() && m.arr.size() == 2) {
                a = m.arr[0].as_string();
                b = m.arr[1].as_string();
            } else fail("malformed merge entry");
            auto ia = vocab_.find(a), ib = vocab_.find(b), ic = vocab_.find(a + b);
            if (ia == vocab_.end() || ib == vocab_.end() || ic == vocab_.end()) fail("merge token out of vocabulary: " + a + " " + b);
            merges_[((uint64_t)ia->second << 32) | ib->second] = {rank, ic->second};
            ++rank;
        }
    }

    // added tokens
    added_.clear();
    if (const Json* added = j.find("added_tokens"); added && added->is_array())
        for (const Json& a : added->arr) {
            const Json* id = a.find("id");
            const Json* content = a.find("content");
            if (!id || !content) continue;
            Added t;
            t.content = content->as_string();
            t.id = (uint32_t)id->as_int();
            t.special = a.get_bool("special", false);
            t.normalized = a.get_bool("normalized", !t.special);
            t.lstrip = a.get_bool("lstrip", false);
            t.rstrip = a.get_bool("rstrip", false);
            t.single_word = a.get_bool("single_word", false);
            put(t.id, t.content, t.special);
            t.match = t.content;
            if (t.normalized && nfc_) {  // normalized tokens are matched against normalized text
                std::vector<uint32_t> in, o;
                unicode::decode_utf8(t.content.data(), t.content.size(), in);
                nfc(in, o);
                t.match = unicode::encode_utf8(o);
            }
            if (!t.match.empty()) added_.push_back(std::move(t));
        }
    for (auto& v : added_by_first_) v.clear();
    for (size_t i = 0; i < added_.size(); ++i) added_by_first_[(uint8_t)added_[i].match[0]].push_back((uint32_t)i);
}

bool BpeTokenizer::token_to_id(const std::string& token, uint32_t& id) const
{
    auto it = token_to_id_.find(token);
    if (it == token_to_id_.end()) return false;
    id = it->second;
    return true;
}

std::string BpeTokenizer::decode(const std::vector<uint32_t>& ids, bool skip_special) const
{
    if (sp_mode_) return decode_sp(ids, skip_special);
    std::string bytes;
    for (uint32_t id : ids) {
        if (id >= id_to_token_.size() || !has_token_[id]) continue;
        if (skip_special && special_[id]) continue;
        const std::string& tok = id_to_token_[id];
        std::vector<uint32_t> cps;
        std::string mapped;
        bool ok = unicode::decode_utf8(tok.data(), tok.size(), cps);
        if (ok)
            for (uint32_t cp : cps) {
                auto it = char_bytes_.find(cp);
                if (it == char_bytes_.end()) {
                    ok = false;
                    break;
                }
                mapped.push_back((char)it->second);
            }
        bytes += ok ? mapped : tok;  // a token with a character outside the alphabet keeps its own bytes
    }
    return from_utf8_lossy(bytes);
}

// AddedVocabulary::find_matches over one of the two token sets (normalized = false: raw text; true: normalized text).
void BpeTokenizer::split_on_added(const std::string& text, bool normalized_set, std::vector<Split>& out) const
{
    out.clear();
    const size_t n = text.size();
    if (n == 0) return;
    auto word_char = [](uint32_t c) { return c == '_' || unicode::is_alphanumeric(c) || unicode::is_mark_nonspacing(c); };
    auto prev_cp = [&](size_t pos, size_t& start_of_prev) -> uint32_t {  // code point ending at byte `pos`
        size_t k = pos;
        do --k; while (k > 0 && ((uint8_t)text[k] & 0xC0) == 0x80);
        start_of_prev = k;
        std::vector<uint32_t> one;
        unicode::decode_utf8(text.data() + k, pos - k, one);
        return one.empty() ? 0xFFFD : one[0];
    };
    auto next_cp = [&](size_t pos, size_t& end_of_next) -> uint32_t {
        size_t k = pos + 1;
        while (k < n && ((uint8_t)text[k] & 0xC0) == 0x80) ++k;
        end_of_next = k;
        std::vector<uint32_t> one;
        unicode::decode_utf8(text.data() + pos, k - pos, one);
        return one.empty() ? 0xFFFD : one[0];
    };
    size_t start_offset = 0, pos = 0;
    while (pos < n) {
        // leftmost-longest: the longest token starting at the first position where any token starts
        const Added* best = nullptr;
        for (const uint32_t idx : added_by_first_[(uint8_t)text[pos]]) {
            const Added& t = added_[idx];
            if (t.normalized != normalized_set) continue;
            const std::string& content = t.match;
            if (content.size() > n - pos) continue;
            if (std::memcmp(text.data() + pos, content.data(), content.size()) != 0) continue;
            if (!best || content.size() > best->match.size()) best = &t;
        }
        if (!best) {
            ++pos;
            continue;
        }
        size_t start = pos, stop = pos + best->match.size();
        pos = stop;
        if (best->single_word) {
            size_t tmp;
            const bool start_space = start == 0 || !word_char(prev_cp(start, tmp));
            const bool stop_space = stop == n || !word_char(next_cp(stop, tmp));
            if (!start_space || !stop_space) continue;
        }
        if (best->lstrip)
            while (start > 0) {
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
        if (start_offset < start) out.push_back({start_offset, start, -1});
        out.push_back({start, stop, (int64_t)best->id});
        start_offset = stop;
        if (pos < stop) pos = stop;
    }
    if (start_offset != n) out.push_back({start_offset, n, -1});
}

void BpeTokenizer::scan_pieces(const std::vector<uint32_t>& cps, std::vector<std::pair<size_t, size_t>>& pieces) const
{
    const size_t n = cps.size();
    size_t i = 0;
    const bool gpt2 = pattern_ == Pattern::Gpt2;
    while (i < n) {
        const uint32_t c = cps[i];
        size_t end = 0;
        // 1. contractions
        if (c == '\'') {
            const size_t len = contraction(cps, i, !gpt2);
            if (len) end = i + len;
        }
        if (!end && gpt2) {
            // ` ?\p{L}+` | ` ?\p{N}+` | ` ?[^\s\p{L}\p{N}]+`
            const size_t s = (c == ' ' && i + 1 < n) ? i + 1 : i;
            auto run = [&](bool (*pred)(uint32_t)) -> size_t {
                size_t k = s;
                while (k < n && pred(cps[k])) ++k;
                return k > s ? k : 0;
            };
            end = run(is_letter);
            if (!end) end = run(is_number);
            if (!end) end = run(is_other);
            if (!end && s != i) {  // the optional space did not help; c == ' ' is whitespace, handled below
            }
        }
        if (!end && !gpt2) {
            // 2. `[^\r\n\p{L}\p{N}]?\p{L}+`
            if (is_letter(c)) {
                size_t k = i;
                while (k < n && is_letter(cps[k])) ++k;
                end = k;
            } else if (!is_newline(c) && !is_number(c) && i + 1 < n && is_letter(cps[i + 1])) {
                size_t k = i + 1;
                while (k < n && is_letter(cps[k])) ++k;
                end = k;
            }
            // 3. `\p{N}{1,3}` (Llama 3) | `\p{N}` (Qwen 2)
            if (!end && is_number(c)) {
                const size_t cap = pattern_ == Pattern::Llama3 ? 3 : 1;
                size_t k = i;
                while (k < n && k - i < cap && is_number(cps[k])) ++k;
                end = k;
            }
            // 4. ` ?[^\s\p{L}\p{N}]+[\r\n]*`
            if (!end) {
                const size_t s = (c == ' ' && i + 1 < n && is_other(cps[i + 1])) ? i + 1 : i;
                if (is_other(cps[s])) {
                    size_t k = s;
                    while (k < n && is_other(cps[k])) ++k;
                    while (k < n && is_newline(cps[k])) ++k;
                    end = k;
                }
            }
            // 5. `\s*[\r\n]+`: the whitespace run up to and including its last newline
            if (!end && is_space(c)) {
                size_t j = i;
                while (j < n && is_space(cps[j])) ++j;
                size_t last = 0;
                bool found = false;
                for (size_t k = i; k < j; ++k)
                    if (is_newline(cps[k])) {
                        last = k;
                        found = true;
                    }
                if (found) end = last + 1;
            }
        }
        // `\s+(?!\S)` then `\s+`
        if (!end && is_space(c)) {
            size_t j = i;
            while (j < n && is_space(cps[j])) ++j;
            if (j == n || j - i == 1) end = j;  // to the end of text, or a single space before a non-space (`\s+`)
            else end = j - 1;                   // leave the last space to the next piece
        }
        if (!end) end = i + 1;  // unreachable for these patterns; keeps the scan total
        pieces.emplace_back(i, end);
        i = end;
    }
}

void BpeTokenizer::bpe_word(const std::string& piece, std::vector<uint32_t>& out) const
{
    // ByteLevel: every byte becomes its visible character; the BPE word is that string.
    std::string mapped;
    mapped.reserve(piece.size() * 2);
    for (unsigned char b : piece) mapped += byte_chars_[b];
    if (ignore_merges_) {
        auto it = vocab_.find(mapped);
        if (it != vocab_.end()) {
            out.push_back(it->second);
            return;
        }
    }
    std::vector<uint32_t> sym;
    sym.reserve(piece.size());
    bool last_unk = false;
    for (unsigned char b : piece) {
        auto it = vocab_.find(byte_chars_[b]);
        if (it != vocab_.end()) {
            sym.push_back(it->second);
            last_unk = false;
        } else if (has_unk_) {
            if (!(fuse_unk_ && last_unk)) sym.push_back(unk_id_);
            last_unk = true;
        }  // no unk token: the character is dropped, as in BPE::merge_word
    }
    // Word::merge_all: repeatedly apply the lowest-ranked merge, leftmost first.
    while (sym.size() > 1) {
        uint32_t best_rank = UINT32_MAX, best_id = 0;
        size_t best_pos = 0;
        for (size_t k = 0; k + 1 < sym.size(); ++k) {
            auto it = merges_.find(((uint64_t)sym[k] << 32) | sym[k + 1]);
            if (it != merges_.end() && it->second.first < best_rank) {
                best_rank = it->second.first;
                best_id = it->second.second;
                best_pos = k;
            }
        }
        if (best_rank == UINT32_MAX) break;
        sym[best_pos] = best_id;
        sym.erase(sym.begin() + (ptrdiff_t)best_pos + 1);
    }
    out.insert(out.end(), sym.begin(), sym.end());
}

void BpeTokenizer::encode_segment(const std::string& text, std::vector<uint32_t>& out) const
{
    if (text.empty()) return;
    std::vector<uint32_t> cps;
    if (!unicode::decode_utf8(text.data(), text.size(), cps)) throw std::runtime_error("tokenizer input is not valid UTF-8");
    if (add_prefix_space_ && (cps.empty() || cps[0] != ' ')) cps.insert(cps.begin(), ' ');
    std::vector<std::pair<size_t, size_t>> pieces;
    scan_pieces(cps, pieces);
    std::string piece;
    for (const auto& p : pieces) {
        piece.clear();
        for (size_t k = p.first; k < p.second; ++k) unicode::append_utf8(piece, cps[k]);
        bpe_word(piece, out);
    }
}

// Word::merge_all (tokenizers/src/models/bpe/word.rs): a heap of candidate merges ordered by (rank, position); stale
// entries are dropped when popped.  Without a pre-tokenizer a whole prompt is one word, so this has to be n log n.
void BpeTokenizer::merge_symbols(std::vector<uint32_t>& sym) const
{
    const int n = (int)sym.size();
    if (n < 2) return;
    std::vector<int> prev((size_t)n), next((size_t)n);
    std::vector<uint8_t> alive((size_t)n, 1);
    for (int i = 0; i < n; ++i) {
        prev[(size_t)i] = i - 1;
        next[(size_t)i] = i + 1 < n ? i + 1 : -1;
    }
    struct Cand {
        uint32_t rank;
        int pos;
        uint32_t a, b, merged;
    };
    auto worse = [](const Cand& x, const Cand& y) { return x.rank != y.rank ? x.rank > y.rank : x.pos > y.pos; };
    std::priority_queue<Cand, std::vector<Cand>, decltype(worse)> heap(worse);
    auto push = [&](int pos) {
        const int nx = next[(size_t)pos];
        if (nx < 0) return;
        auto it = merges_.find(((uint64_t)sym[(size_t)pos] << 32) | sym[(size_t)nx]);
        if (it != merges_.end()) heap.push({it->second.first, pos, sym[(size_t)pos], sym[(size_t)nx], it->second.second});
    };
    for (int i = 0; i + 1 < n; ++i) push(i);
    while (!heap.empty()) {
        const Cand c = heap.top();
        heap.pop();
        if (!alive[(size_t)c.pos] || sym[(size_t)c.pos] != c.a) continue;
        const int nx = next[(size_t)c.pos];
        if (nx < 0 || sym[(size_t)nx] != c.b) continue;
        sym[(size_t)c.pos] = c.merged;
        alive[(size_t)nx] = 0;
        next[(size_t)c.pos] = next[(size_t)nx];
        if (next[(size_t)nx] >= 0) prev[(size_t)next[(size_t)nx]] = c.pos;
        if (prev[(size_t)c.pos] >= 0) push(prev[(size_t)c.pos]);
        push(c.pos);
    }
    std::vector<uint32_t> out;
    out.reserve((size_t)n);
    for (int i = 0; i < n; ++i)
        if (alive[(size_t)i]) out.push_back(sym[(size_t)i]);
    sym.swap(out);
}

// One segment between added tokens in the SentencePiece-style pipeline: spaces become U+2581, the prefix rule of the
// normalizer / Metaspace applies, every character is a symbol (its bytes as <0xNN> tokens when it is not in the
// vocabulary, a fused <unk> when those are missing too), then the merges.
void BpeTokenizer::encode_segment_sp(const std::string& text, bool at_start, std::vector<uint32_t>& out) const
{
    if (text.empty()) return;
    std::vector<uint32_t> cps;
    if (!unicode::decode_utf8(text.data(), text.size(), cps)) throw std::runtime_error("tokenizer input is not valid UTF-8");
    constexpr uint32_t kSp = 0x2581;
    for (uint32_t& c : cps)
        if (c == ' ') c = kSp;
    bool prepend = false;
    switch (sp_prepend_) {
    case 0: prepend = true; break;                           // Prepend normalizer: every non-empty segment
    case 1: prepend = at_start && cps[0] != kSp; break;      // Metaspace, prepend_scheme = first
    case 2: prepend = cps[0] != kSp; break;                  // Metaspace, always
    default: break;
    }
    if (prepend) cps.insert(cps.begin(), kSp);
    std::vector<uint32_t> sym;
    sym.reserve(cps.size());
    bool last_unk = false;
    std::string ch;
    for (uint32_t c : cps) {
        ch.clear();
        unicode::append_utf8(ch, c);
        auto it = vocab_.find(ch);
        if (it != vocab_.end()) {
            sym.push_back(it->second);
            last_unk = false;
            continue;
        }
        bool bytes_ok = true;
        for (unsigned char b : ch)
            if (byte_ids_[b] < 0) bytes_ok = false;
        if (bytes_ok) {
            for (unsigned char b : ch) sym.push_back((uint32_t)byte_ids_[b]);
            last_unk = false;
        } else if (has_unk_) {
            if (!(fuse_unk_ && last_unk)) sym.push_back(unk_id_);
            last_unk = true;
        }
    }
    merge_symbols(sym);
    out.insert(out.end(), sym.begin(), sym.end());
}

// decoders::Sequence[Replace(U+2581 -> " "), ByteFallback, Fuse, Strip(" ", 1, 0)].
std::string BpeTokenizer::decode_sp(const std::vector<uint32_t>& ids, bool skip_special) const
{
    std::string fused, run;
    auto flush = [&] {
        if (run.empty()) return;
        if (unicode::is_valid_utf8(run.data(), run.size())) fused += run;
        else
            for (size_t i = 0; i < run.size(); ++i) fused += "\xEF\xBF\xBD";  // one replacement character per byte
        run.clear();
    };
    for (uint32_t id : ids) {
        if (id >= id_to_token_.size() || !has_token_[id]) continue;
        if (skip_special && special_[id]) continue;
        const std::string& tok = id_to_token_[id];
        if (tok.size() == 6 && tok.compare(0, 3, "<0x") == 0 && tok[5] == '>') {
            const std::string hex = tok.substr(3, 2);
            char* end = nullptr;
            const long v = std::strtol(hex.c_str(), &end, 16);
            if (end == hex.c_str() + 2) {
                run.push_back((char)v);
                continue;
            }
        }
        flush();
        for (size_t i = 0; i < tok.size();) {  // U+2581 -> ' '
            if (tok.compare(i, 3, "\xE2\x96\x81") == 0) {
                fused += ' ';
                i += 3;
            } else fused += tok[i++];
        }
    }
    flush();
    if (!fused.empty() && fused[0] == ' ') fused.erase(0, 1);
    return fused;
}

std::vector<uint32_t> BpeTokenizer::encode(const std::string& text, size_t max_length) const
{
    std::vector<uint32_t> ids;
    std::vector<Split> raw, inner;
    split_on_added(text, false, raw);
    for (const Split& s : raw) {
        if (s.token >= 0) {
            ids.push_back((uint32_t)s.token);
            continue;
        }
        std::string seg = text.substr(s.begin, s.end - s.begin);
        if (nfc_) {
            std::vector<uint32_t> in, o;
            if (!unicode::decode_utf8(seg.data(), seg.size(), in)) throw std::runtime_error("tokenizer input is not valid UTF-8");
            nfc(in, o);
            seg = unicode::encode_utf8(o);
        }
        if (sp_mode_) {
            encode_segment_sp(seg, s.begin == 0, ids);
            continue;
        }
        split_on_added(seg, true, inner);
        for (const Split& t : inner) {
            if (t.token >= 0) ids.push_back((uint32_t)t.token);
            else encode_segment(seg.substr(t.begin, t.end - t.begin), ids);
        }
    }
    if (max_length && ids.size() > max_length) ids.resize(max_length);
    return ids;
}

std::vector<std::string> BpeTokenizer::pre_tokenize(const std::string& text) const
{
    std::vector<uint32_t> cps;
    if (!unicode::decode_utf8(text.data(), text.size(), cps)) throw std::runtime_error("tokenizer input is not valid UTF-8");
    if (nfc_) {
        std::vector<uint32_t> o;
        nfc(cps, o);
        cps.swap(o);
    }
    std::vector<std::pair<size_t, size_t>> pieces;
    scan_pieces(cps, pieces);
    std::vector<std::string> out;
    for (const auto& p : pieces) {
        std::string s;
        for (size_t k = p.first; k < p.second; ++k) unicode::append_utf8(s, cps[k]);
        out.push_back(std::move(s));
    }
    return out;
}

}  // namespace kjarni
