#include "unicode.h"

#include <algorithm>
#include <cstddef>

namespace kjarni {
namespace unicode {

namespace {

struct UnicodeMapEntry {
    uint32_t cp;
    uint32_t offset;
    uint32_t len;
};

#include "unicode_tables.inc"

bool in_ranges(const uint32_t (*r)[2], size_t n, uint32_t cp)
{
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (cp < r[mid][0]) hi = mid;
        else if (cp > r[mid][1]) lo = mid + 1;
        else return true;
    }
    return false;
}

const UnicodeMapEntry* find_entry(const UnicodeMapEntry* tab, size_t n, uint32_t cp)
{
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (cp < tab[mid].cp) hi = mid;
        else if (cp > tab[mid].cp) lo = mid + 1;
        else return &tab[mid];
    }
    return nullptr;
}

uint32_t ccc(uint32_t cp)
{
    size_t lo = 0, hi = kCcc_len;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (cp < kCcc[mid][0]) hi = mid;
        else if (cp > kCcc[mid][1]) lo = mid + 1;
        else return kCcc[mid][2];
    }
    return 0;
}

}  // namespace

bool decode_utf8(const char* s, size_t len, std::vector<uint32_t>& out)
{
    out.clear();
    out.reserve(len);
    const unsigned char* p = reinterpret_cast<const unsigned char*>(s);
    size_t i = 0;
    while (i < len) {
        const unsigned char c = p[i];
        if (c < 0x80) {
            out.push_back(c);
            ++i;
            continue;
        }
        int n;
        uint32_t cp;
        if (c >= 0xC2 && c <= 0xDF) { n = 1; cp = c & 0x1F; }
        else if (c >= 0xE0 && c <= 0xEF) { n = 2; cp = c & 0x0F; }
        else if (c >= 0xF0 && c <= 0xF4) { n = 3; cp = c & 0x07; }
        else return false;
        for (int k = 1; k <= n; ++k) {
            if (i + (size_t)k >= len) return false;
            const unsigned char cc = p[i + k];
            if ((cc & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (cc & 0x3F);
        }
        if (n == 2 && (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF))) return false;
        if (n == 3 && (cp < 0x10000 || cp > 0x10FFFF)) return false;
        out.push_back(cp);
        i += (size_t)n + 1;
    }
    return true;
}

bool is_valid_utf8(const char* s, size_t len)
{
    std::vector<uint32_t> tmp;
    return decode_utf8(s, len, tmp);
}

void append_utf8(std::string& out, uint32_t cp)
{
    if (cp < 0x80) {
        out.push_back((char)cp);
    } else if (cp < 0x800) {
        out.push_back((char)(0xC0 | (cp >> 6)));
        out.push_back((char)(0x80 | (cp & 0x3F)));
    } else if (cp < 0x10000) {
        out.push_back((char)(0xE0 | (cp >> 12)));
        out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
        out.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
        out.push_back((char)(0xF0 | (cp >> 18)));
        out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
        out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
        out.push_back((char)(0x80 | (cp & 0x3F)));
    }
}

std::string encode_utf8(const std::vector<uint32_t>& cps)
{
    std::string s;
    s.reserve(cps.size());
    for (uint32_t cp : cps) append_utf8(s, cp);
    return s;
}

int clean_class(uint32_t cp)
{
    if (in_ranges(kCleanDrop, kCleanDrop_len, cp)) return 1;
    if (in_ranges(kCleanSpace, kCleanSpace_len, cp)) return 2;
    return 0;
}

bool is_cjk(uint32_t cp) { return in_ranges(kCjk, kCjk_len, cp); }
bool is_mark_nonspacing(uint32_t cp) { return in_ranges(kMarkNonspacing, kMarkNonspacing_len, cp); }
bool is_whitespace(uint32_t cp) { return in_ranges(kWhitespace, kWhitespace_len, cp); }
bool is_bert_punctuation(uint32_t cp) { return in_ranges(kPunctuation, kPunctuation_len, cp); }
bool is_alphanumeric(uint32_t cp)
{
    if (cp < 0x80) return (cp >= '0' && cp <= '9') || (cp >= 'a' && cp <= 'z') || (cp >= 'A' && cp <= 'Z');
    return in_ranges(kAlnum, kAlnum_len, cp);
}

int combining_class(uint32_t cp) { return cp < 0x300 ? 0 : (int)ccc(cp); }

void nfd(const std::vector<uint32_t>& in, std::vector<uint32_t>& out)
{
    out.clear();
    out.reserve(in.size() + 8);
    constexpr uint32_t S_BASE = 0xAC00, L_BASE = 0x1100, V_BASE = 0x1161, T_BASE = 0x11A7;
    for (uint32_t cp : in) {
        if (cp >= S_BASE && cp < S_BASE + 11172) {  // Hangul syllables decompose algorithmically
            const uint32_t si = cp - S_BASE;
            out.push_back(L_BASE + si / 588);
            out.push_back(V_BASE + (si % 588) / 28);
            if (si % 28) out.push_back(T_BASE + si % 28);
        } else if (const UnicodeMapEntry* e = (cp >= 0xC0 ? find_entry(kDecomp, kDecomp_len, cp) : nullptr)) {
            for (uint32_t k = 0; k < e->len; ++k) out.push_back(kDecomp_pool[e->offset + k]);
        } else {
            out.push_back(cp);
        }
    }
    // Canonical ordering: stable sort of each run of non-starters by combining class.
    size_t i = 0;
    const size_t n = out.size();
    while (i < n) {
        if (out[i] < 0x300 || ccc(out[i]) == 0) {
            ++i;
            continue;
        }
        size_t j = i;
        while (j < n && out[j] >= 0x300 && ccc(out[j]) != 0) ++j;
        if (j - i > 1)
            std::stable_sort(out.begin() + (ptrdiff_t)i, out.begin() + (ptrdiff_t)j,
                             [](uint32_t a, uint32_t b) { return ccc(a) < ccc(b); });
        i = j;
    }
}

void lowercase(const std::vector<uint32_t>& in, std::vector<uint32_t>& out)
{
    out.clear();
    out.reserve(in.size());
    for (uint32_t cp : in) {
        if (cp < 0x80) {
            out.push_back((cp >= 'A' && cp <= 'Z') ? cp + 32 : cp);
        } else if (const UnicodeMapEntry* e = find_entry(kLower, kLower_len, cp)) {
            for (uint32_t k = 0; k < e->len; ++k) out.push_back(kLower_pool[e->offset + k]);
        } else {
            out.push_back(cp);
        }
    }
}

// Rust str::to_lowercase: per-char mapping except U+03A3, which becomes the final form when it
// follows (Case_Ignorable* Cased) and is not followed by (Case_Ignorable* Cased) -- Unicode's
// Final_Sigma condition, library/alloc/src/str.rs map_uppercase_sigma.
void lowercase_str(const std::vector<uint32_t>& in, std::vector<uint32_t>& out)
{
    out.clear();
    out.reserve(in.size());
    auto cased = [](uint32_t c) { return in_ranges(kCased, kCased_len, c); };
    auto ignorable = [](uint32_t c) { return in_ranges(kCaseIgnorable, kCaseIgnorable_len, c); };
    const size_t n = in.size();
    for (size_t i = 0; i < n; ++i) {
        const uint32_t cp = in[i];
        if (cp == 0x3A3) {
            size_t b = i;
            while (b > 0 && ignorable(in[b - 1])) --b;
            const bool before = b > 0 && cased(in[b - 1]);
            size_t a = i + 1;
            while (a < n && ignorable(in[a])) ++a;
            const bool after = a < n && cased(in[a]);
            out.push_back(before && !after ? 0x3C2 : 0x3C3);
        } else if (cp < 0x80) {
            out.push_back((cp >= 'A' && cp <= 'Z') ? cp + 32 : cp);
        } else if (const UnicodeMapEntry* e = find_entry(kLower, kLower_len, cp)) {
            for (uint32_t k = 0; k < e->len; ++k) out.push_back(kLower_pool[e->offset + k]);
        } else {
            out.push_back(cp);
        }
    }
}

}  // namespace unicode
}  // namespace kjarni
