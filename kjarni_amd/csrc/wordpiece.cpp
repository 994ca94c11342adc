#include "host_util.h"
#include "wordpiece.h"

#include <cstdint>

#include "bpe.h"
#include "unigram.h"

#include <exception>
#include <thread>

#include <algorithm>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "json.h"
#include "unicode.h"

namespace kjarni {

namespace {

bool is_word_char(uint32_t cp)
{
    if (cp < 0x80) return (cp >= '0' && cp <= '9') || (cp >= 'a' && cp <= 'z') || (cp >= 'A' && cp <= 'Z') || cp == '_';
    return !unicode::is_whitespace(cp) && !unicode::is_bert_punctuation(cp);
}

}  // namespace

void BertTokenizer::parse_post_processor(const Json& root)
{
    if (const Json* pp = root.find("post_processor")) {
        if (!pp->is_null()) {
            const std::string ppt = pp->get_string("type", "");
            auto special_id = [&](const Json& specials, const std::string& name) -> uint32_t {
                if (const Json* s = specials.find(name))
                    if (const Json* ids = s->find("ids"))
                        if (ids->is_array() && !ids->arr.empty()) return (uint32_t)ids->arr[0].as_int();
                const int64_t id = token_to_id(name);
                if (id < 0) throw std::runtime_error("post_processor token not in vocab: " + name);
                return (uint32_t)id;
            };
            if (ppt == "TemplateProcessing") {
                const Json empty;
                const Json* specials = pp->find("special_tokens");
                auto parse = [&](const Json& arr, std::vector<TemplatePiece>& out) {
                    for (const Json& piece : arr.arr) {
                        TemplatePiece tp;
                        if (const Json* st = piece.find("SpecialToken")) {
                            tp.is_special = true;
                            tp.id = special_id(specials ? *specials : empty, st->get_string("id", ""));
                            tp.type_id = (uint32_t)st->get_int("type_id", 0);
                        } else if (const Json* sq = piece.find("Sequence")) {
                            tp.sequence = sq->get_string("id", "A") == "B" ? 1 : 0;
                            tp.type_id = (uint32_t)sq->get_int("type_id", 0);
                        } else {
                            throw std::runtime_error("unknown TemplateProcessing piece");
                        }
                        out.push_back(tp);
                    }
                };
                parse(pp->at("single"), single_);
                parse(pp->at("pair"), pair_);
            } else if (ppt == "BertProcessing") {
                const uint32_t sep = (uint32_t)pp->at("sep").arr.at(1).as_int();
                const uint32_t cls = (uint32_t)pp->at("cls").arr.at(1).as_int();
                single_ = {{true, cls, 0, 0}, {false, 0, 0, 0}, {true, sep, 0, 0}};
                pair_ = {{true, cls, 0, 0}, {false, 0, 0, 0}, {true, sep, 0, 0}, {false, 0, 1, 1}, {true, sep, 0, 1}};
            } else if (ppt == "RobertaProcessing") {  // <s> A </s>  |  <s> A </s></s> B </s>, every type id 0
                const uint32_t sep = (uint32_t)pp->at("sep").arr.at(1).as_int();
                const uint32_t cls = (uint32_t)pp->at("cls").arr.at(1).as_int();
                single_ = {{true, cls, 0, 0}, {false, 0, 0, 0}, {true, sep, 0, 0}};
                pair_ = {{true, cls, 0, 0}, {false, 0, 0, 0}, {true, sep, 0, 0}, {true, sep, 0, 0}, {false, 0, 1, 0}, {true, sep, 0, 0}};
            } else {
                throw std::runtime_error("post_processor '" + ppt + "' is not supported");
            }
        }
    }
    if (single_.empty()) single_ = {{false, 0, 0, 0}};
    if (pair_.empty()) pair_ = {{false, 0, 0, 0}, {false, 0, 1, 1}};
}

BertTokenizer BertTokenizer::from_file(const std::string& path) { return from_json(slurp(path)); }

BertTokenizer BertTokenizer::from_json(const std::string& text)
{
    const Json root = Json::parse(text);
    BertTokenizer t;

    const Json& model = root.at("model");
    const std::string mtype = model.get_string("type", "WordPiece");
    if (mtype == "BPE") {  // RoBERTa: byte-level BPE for the sequences, the framing below stays the same
        auto bpe = std::make_shared<BpeTokenizer>();
        bpe->load_json(text, "tokenizer.json");
        t.bpe_ = bpe;
        t.has_normalizer_ = false;
        t.parse_post_processor(root);
        t.pad_id_ = 0;
        return t;
    }
    if (mtype == "Unigram") {  // XLM-R (bge-m3): SentencePiece Unigram for the sequences, TemplateProcessing framing
        auto uni = std::make_shared<UnigramTokenizer>();
        uni->load_json(text, "tokenizer.json");
        t.unigram_ = uni;
        t.has_normalizer_ = false;
        t.parse_post_processor(root);
        t.pad_id_ = 0;  // PaddingParams::default() (loader.rs:112-115), not the checkpoint's <pad>
        return t;
    }
    if (mtype != "WordPiece")
        throw std::runtime_error("tokenizer model '" + mtype + "' is not supported (WordPiece, byte-level BPE and Unigram are)");
    const Json& vocab = model.at("vocab");
    if (!vocab.is_object()) throw std::runtime_error("tokenizer.json: model.vocab must be an object");
    t.vocab_.reserve(vocab.obj.size() * 2);
    for (const auto& kv : vocab.obj) t.vocab_[kv.first] = (uint32_t)kv.second.as_int();
    t.unk_token_ = model.get_string("unk_token", "[UNK]");
    t.prefix_ = model.get_string("continuing_subword_prefix", "##");
    t.max_chars_per_word_ = (size_t)model.get_int("max_input_chars_per_word", 100);
    auto unk = t.vocab_.find(t.unk_token_);
    if (unk == t.vocab_.end()) throw std::runtime_error("tokenizer.json: unk_token missing from the vocabulary");
    t.unk_id_ = unk->second;

    // normalizer
    const Json* norm = root.find("normalizer");
    if (!norm || norm->is_null()) {
        t.has_normalizer_ = false;
    } else {
        const std::string nt = norm->get_string("type", "");
        if (nt != "BertNormalizer")
            throw std::runtime_error("tokenizer normalizer '" + nt + "' is not supported (BertNormalizer only)");
        t.clean_text_ = norm->get_bool("clean_text", true);
        t.handle_chinese_ = norm->get_bool("handle_chinese_chars", true);
        t.lowercase_ = norm->get_bool("lowercase", true);
        const Json* sa = norm->find("strip_accents");
        t.strip_accents_ = (sa && sa->is_bool()) ? sa->b : t.lowercase_;  // None -> follows lowercase
    }
    // pre-tokenizer
    if (const Json* pt = root.find("pre_tokenizer")) {
        if (!pt->is_null()) {
            const std::string ptt = pt->get_string("type", "");
            if (ptt != "BertPreTokenizer")
                throw std::runtime_error("pre_tokenizer '" + ptt + "' is not supported (BertPreTokenizer only)");
        }
    }
    // added tokens
    if (const Json* added = root.find("added_tokens")) {
        if (added->is_array())
            for (const Json& a : added->arr) {
                AddedToken at;
                at.content = a.get_string("content", "");
                at.id = (uint32_t)a.get_int("id", 0);
                at.special = a.get_bool("special", false);
                at.single_word = a.get_bool("single_word", false);
                at.lstrip = a.get_bool("lstrip", false);
                at.rstrip = a.get_bool("rstrip", false);
                at.normalized = a.get_bool("normalized", !at.special);
                if (!at.content.empty()) t.added_.push_back(at);
            }
    }
    t.parse_post_processor(root);
    // PaddingParams::default(): pad_id 0, pad_type_id 0 (loader.rs:112-115)
    t.pad_id_ = 0;
    return t;
}

size_t BertTokenizer::vocab_size() const
{
    return bpe_ ? bpe_->vocab_size() : unigram_ ? unigram_->vocab_size() : vocab_.size();
}

int64_t BertTokenizer::token_to_id(const std::string& tok) const
{
    if (bpe_) {
        uint32_t id = 0;
        return bpe_->token_to_id(tok, id) ? (int64_t)id : -1;
    }
    if (unigram_) {
        uint32_t id = 0;
        return unigram_->token_to_id(tok, id) ? (int64_t)id : -1;
    }
    auto it = vocab_.find(tok);
    return it == vocab_.end() ? -1 : (int64_t)it->second;
}

// tokenizers/src/normalizers/bert.rs: clean_text -> handle_chinese_chars ->
// strip_accents (NFD + drop Mn) -> lowercase (per char).
void BertTokenizer::normalize_cps(const std::vector<uint32_t>& in, std::vector<uint32_t>& out) const
{
    if (!has_normalizer_) {
        out = in;
        return;
    }
    std::vector<uint32_t> a, b;
    a.reserve(in.size() + 8);
    if (clean_text_) {
        for (uint32_t cp : in) {
            const int c = unicode::clean_class(cp);
            if (c == 1) continue;
            a.push_back(c == 2 ? (uint32_t)' ' : cp);
        }
    } else {
        a = in;
    }
    if (handle_chinese_) {
        b.clear();
        b.reserve(a.size() + 8);
        for (uint32_t cp : a) {
            if (cp >= 0x3400 && unicode::is_cjk(cp)) {
                b.push_back(' ');
                b.push_back(cp);
                b.push_back(' ');
            } else {
                b.push_back(cp);
            }
        }
        a.swap(b);
    }
    if (strip_accents_) {
        unicode::nfd(a, b);
        a.clear();
        for (uint32_t cp : b)
            if (cp < 0x300 || !unicode::is_mark_nonspacing(cp)) a.push_back(cp);
    }
    if (lowercase_) {
        unicode::lowercase(a, b);
        a.swap(b);
    }
    out.swap(a);
}

std::string BertTokenizer::normalize(const std::string& text) const
{
    std::vector<uint32_t> cps, out;
    if (!unicode::decode_utf8(text.data(), text.size(), cps)) throw std::runtime_error("invalid UTF-8");
    normalize_cps(cps, out);
    return unicode::encode_utf8(out);
}

// tokenizers/src/models/wordpiece/mod.rs tokenize(): greedy longest-match-first;
// a word longer than max_input_chars_per_word, or with any unmatched remainder,
// becomes a single [UNK].
void BertTokenizer::wordpiece(const std::vector<uint32_t>& word, std::vector<uint32_t>& ids) const
{
    if (word.size() > max_chars_per_word_) {
        ids.push_back(unk_id_);
        return;
    }
    const size_t first = ids.size();
    size_t start = 0;
    std::string sub;
    while (start < word.size()) {
        size_t end = word.size();
        bool found = false;
        while (start < end) {
            sub.clear();
            if (start > 0) sub = prefix_;
            for (size_t i = start; i < end; ++i) unicode::append_utf8(sub, word[i]);
            auto it = vocab_.find(sub);
            if (it != vocab_.end()) {
                ids.push_back(it->second);
                found = true;
                break;
            }
            --end;
        }
        if (!found) {
            ids.resize(first);
            ids.push_back(unk_id_);
            return;
        }
        start = end;
    }
}

// BertPreTokenizer: split on whitespace (removed), then isolate punctuation.
void BertTokenizer::tokenize_segment(const std::vector<uint32_t>& cps, std::vector<uint32_t>& ids) const
{
    std::vector<uint32_t> word;
    auto flush = [&] {
        if (!word.empty()) {
            wordpiece(word, ids);
            word.clear();
        }
    };
    for (uint32_t cp : cps) {
        if (unicode::is_whitespace(cp)) {
            flush();
        } else if (unicode::is_bert_punctuation(cp)) {
            flush();
            word.push_back(cp);
            flush();
        } else {
            word.push_back(cp);
        }
    }
    flush();
}

// AddedVocabulary::extract_and_normalize: added tokens with normalized == false
// are cut out of the RAW text (leftmost-longest), the rest is normalised, then
// added tokens with normalized == true are cut out of the normalised text.
void BertTokenizer::tokenize_sequence(const std::string& text, std::vector<uint32_t>& ids) const
{
    if (bpe_) {
        const std::vector<uint32_t> got = bpe_->encode(text, 0);
        ids.insert(ids.end(), got.begin(), got.end());
        return;
    }
    if (unigram_) {
        const std::vector<uint32_t> got = unigram_->encode(text);
        ids.insert(ids.end(), got.begin(), got.end());
        return;
    }
    std::vector<uint32_t> cps;
    if (!unicode::decode_utf8(text.data(), text.size(), cps)) throw std::runtime_error("invalid UTF-8 in input text");

    struct Tok {
        std::vector<uint32_t> cps;
        uint32_t id;
        const AddedToken* at;
    };
    std::vector<Tok> raw_toks, norm_toks;
    for (const AddedToken& a : added_) {
        Tok t;
        unicode::decode_utf8(a.content.data(), a.content.size(), t.cps);
        t.id = a.id;
        t.at = &a;
        (a.normalized ? norm_toks : raw_toks).push_back(std::move(t));
    }

    // Splits `s` on the given tokens; calls on_text(segment) / on_token(id).
    auto split = [&](const std::vector<uint32_t>& s, const std::vector<Tok>& toks, auto&& on_text,
                     auto&& on_token) {
        if (toks.empty()) {
            if (!s.empty()) on_text(s);
            return;
        }
        std::vector<uint32_t> seg;
        size_t i = 0;
        while (i < s.size()) {
            const Tok* best = nullptr;
            for (const Tok& t : toks) {
                const size_t n = t.cps.size();
                if (n == 0 || i + n > s.size()) continue;
                if (!std::equal(t.cps.begin(), t.cps.end(), s.begin() + (ptrdiff_t)i)) continue;
                if (t.at->single_word) {
                    const bool left_ok = (i == 0) || !is_word_char(s[i - 1]);
                    const bool right_ok = (i + n == s.size()) || !is_word_char(s[i + n]);
                    if (!left_ok || !right_ok) continue;
                }
                if (!best || n > best->cps.size()) best = &t;
            }
            if (!best) {
                seg.push_back(s[i]);
                ++i;
                continue;
            }
            if (best->at->lstrip)
                while (!seg.empty() && unicode::is_whitespace(seg.back())) seg.pop_back();
            if (!seg.empty()) on_text(seg);
            seg.clear();
            on_token(best->id);
            i += best->cps.size();
            if (best->at->rstrip)
                while (i < s.size() && unicode::is_whitespace(s[i])) ++i;
        }
        if (!seg.empty()) on_text(seg);
    };

    split(
        cps, raw_toks,
        [&](const std::vector<uint32_t>& seg) {
            std::vector<uint32_t> normed;
            normalize_cps(seg, normed);
            split(
                normed, norm_toks, [&](const std::vector<uint32_t>& s2) { tokenize_segment(s2, ids); },
                [&](uint32_t id) { ids.push_back(id); });
        },
        [&](uint32_t id) { ids.push_back(id); });
}

Encoding BertTokenizer::encode(const std::string& text_a, const std::string* text_b) const
{
    std::vector<uint32_t> a, b;
    tokenize_sequence(text_a, a);
    if (text_b) tokenize_sequence(*text_b, b);
    const std::vector<TemplatePiece>& tmpl = text_b ? pair_ : single_;
    size_t n_added = 0;
    for (const TemplatePiece& p : tmpl) n_added += p.is_special ? 1 : 0;

    // tokenizers/src/utils/truncation.rs truncate_encodings, strategy LongestFirst,
    // direction Right, stride 0; max_length is reduced by the special tokens first
    // (tokenizers/src/tokenizer/mod.rs post_process).
    // max_length - n_added is a usize subtraction in the crate: when the frame alone exceeds max_length it wraps in a
    // release build and nothing is truncated.
    const size_t max_len = max_length_ >= n_added ? max_length_ - n_added : SIZE_MAX;
    const size_t total = a.size() + b.size();
    if (max_len == 0) {
        a.clear();
        b.clear();
    } else if (total > max_len) {
        if (text_b) {
            size_t n1 = a.size(), n2 = b.size();
            bool swap = false;
            if (n1 > n2) {
                swap = true;
                std::swap(n1, n2);
            }
            if (n1 > max_len) n2 = n1;
            else n2 = std::max(n1, max_len - n1);
            if (n1 + n2 > max_len) {
                n1 = max_len / 2;
                n2 = n1 + max_len % 2;
            }
            if (swap) std::swap(n1, n2);
            if (a.size() > n1) a.resize(n1);
            if (b.size() > n2) b.resize(n2);
        } else {
            a.resize(max_len);
        }
    }

    Encoding e;
    for (const TemplatePiece& p : tmpl) {
        if (p.is_special) {
            e.ids.push_back(p.id);
            e.type_ids.push_back(p.type_id);
        } else {
            const std::vector<uint32_t>& src = p.sequence == 0 ? a : b;
            e.ids.insert(e.ids.end(), src.begin(), src.end());
            e.type_ids.insert(e.type_ids.end(), src.size(), p.type_id);
        }
    }
    e.attention_mask.assign(e.ids.size(), 1u);
    return e;
}

BatchEncoding BertTokenizer::pad_batch(std::vector<Encoding>& encs)
{
    BatchEncoding out;
    out.batch = encs.size();
    for (const Encoding& e : encs) out.seq = std::max(out.seq, e.ids.size());
    out.ids.assign(out.batch * out.seq, 0u);  // pad id 0
    out.type_ids.assign(out.batch * out.seq, 0u);
    out.attention_mask.assign(out.batch * out.seq, 0u);
    for (size_t i = 0; i < encs.size(); ++i) {
        const Encoding& e = encs[i];
        std::copy(e.ids.begin(), e.ids.end(), out.ids.begin() + (ptrdiff_t)(i * out.seq));
        std::copy(e.type_ids.begin(), e.type_ids.end(), out.type_ids.begin() + (ptrdiff_t)(i * out.seq));
        std::copy(e.attention_mask.begin(), e.attention_mask.end(),
                  out.attention_mask.begin() + (ptrdiff_t)(i * out.seq));
    }
    return out;
}

namespace {

// The `tokenizers` crate encodes a batch on its rayon pool; here a batch large enough to matter is
// split over a few host threads (encode() is const and touches no shared mutable state).
template <class F>
void parallel_rows(size_t n, F&& fn)
{
    size_t workers = std::thread::hardware_concurrency();
    if (workers > 16) workers = 16;
    if (workers > n / 64) workers = n / 64;
    if (workers < 2) {
        for (size_t i = 0; i < n; ++i) fn(i);
        return;
    }
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> errs(workers);
    for (size_t w = 0; w < workers; ++w)
        pool.emplace_back([&, w] {
            try {
                for (size_t i = n * w / workers, e = n * (w + 1) / workers; i < e; ++i) fn(i);
            } catch (...) {
                errs[w] = std::current_exception();
            }
        });
    for (std::thread& t : pool) t.join();
    for (const std::exception_ptr& e : errs)
        if (e) std::rethrow_exception(e);
}

}  // namespace

BatchEncoding BertTokenizer::encode_batch(const std::vector<std::string>& texts) const
{
    std::vector<Encoding> encs(texts.size());
    parallel_rows(texts.size(), [&](size_t i) { encs[i] = encode(texts[i], nullptr); });
    return pad_batch(encs);
}

BatchEncoding BertTokenizer::encode_batch_pairs(const std::vector<std::pair<std::string, std::string>>& pairs) const
{
    std::vector<Encoding> encs(pairs.size());
    parallel_rows(pairs.size(), [&](size_t i) { encs[i] = encode(pairs[i].first, &pairs[i].second); });
    return pad_batch(encs);
}

}  // namespace kjarni
