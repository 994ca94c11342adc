#include "host_util.h"
#include "index.h"

#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <thread>
#include <atomic>
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <tuple>

#include "json.h"
#include "unicode.h"

namespace kjarni {

namespace {

bool is_dir(const std::string& p)
{
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

// ---- bincode 1.x reader / writer (little-endian, fixed-width ints, u64 lengths) ----
struct BinReader {
    const uint8_t* p;
    const uint8_t* end;
    void need(uint64_t n) const
    {
        if ((uint64_t)(end - p) < n) throw std::runtime_error("bincode: unexpected end of data");
    }
    // `count` records of `size` bytes must still be in the buffer (no overflow: a corrupt count must not size an allocation).
    void need_records(uint64_t count, uint64_t size) const
    {
        if (size != 0 && count > (uint64_t)(end - p) / size) throw std::runtime_error("bincode: unexpected end of data");
    }
    uint64_t u64()
    {
        need(8);
        uint64_t v;
        std::memcpy(&v, p, 8);
        p += 8;
        return v;
    }
    float f32()
    {
        need(4);
        float v;
        std::memcpy(&v, p, 4);
        p += 4;
        return v;
    }
    std::string str()
    {
        const uint64_t n = u64();
        need(n);
        std::string s(reinterpret_cast<const char*>(p), (size_t)n);
        p += n;
        return s;
    }
};

void put_u64(std::string& o, uint64_t v) { o.append(reinterpret_cast<const char*>(&v), 8); }
void put_f32(std::string& o, float v) { o.append(reinterpret_cast<const char*>(&v), 4); }
void put_str(std::string& o, const std::string& s)
{
    put_u64(o, s.size());
    o.append(s);
}

}  // namespace

// ---------------------------------------------------------------------------------- BM25

// text.to_lowercase().split(|c| !c.is_alphanumeric()).filter(|s| s.len() >= 2)   (byte length)
std::vector<std::string> Bm25Index::tokenize(const std::string& text)
{
    std::vector<uint32_t> cps, low;
    if (!unicode::decode_utf8(text.data(), text.size(), cps)) throw std::runtime_error("invalid UTF-8");
    unicode::lowercase_str(cps, low);
    std::vector<std::string> out;
    std::string cur;
    auto flush = [&] {
        if (cur.size() >= 2) out.push_back(cur);
        cur.clear();
    };
    for (uint32_t cp : low) {
        if (unicode::is_alphanumeric(cp)) unicode::append_utf8(cur, cp);
        else flush();
    }
    flush();
    return out;
}

void Bm25Index::add_document(size_t doc_id, const std::string& text)
{
    const std::vector<std::string> tokens = tokenize(text);
    if (doc_id >= doc_lengths.size()) doc_lengths.resize(doc_id + 1, 0);
    doc_lengths[doc_id] = tokens.size();
    std::vector<std::pair<std::string, uint64_t>> counts;  // first-seen order
    std::unordered_map<std::string, size_t> pos;
    for (const std::string& t : tokens) {
        auto it = pos.find(t);
        if (it == pos.end()) {
            pos.emplace(t, counts.size());
            counts.emplace_back(t, 1);
        } else {
            counts[it->second].second += 1;
        }
    }
    for (const auto& kv : counts) {
        inverted_index[kv.first].emplace_back((uint64_t)doc_id, kv.second);
        doc_frequencies[kv.first] += 1;
    }
    total_docs = std::max(total_docs, doc_id + 1);
    total_length += tokens.size();
    avg_doc_length = (float)total_length / (float)total_docs;
}

size_t Bm25Index::term_frequency(const std::string& term, size_t doc_id) const
{
    auto it = inverted_index.find(term);
    if (it == inverted_index.end()) return 0;
    for (const auto& p : it->second)
        if (p.first == doc_id) return (size_t)p.second;
    return 0;
}

float Bm25Index::score(const std::vector<std::string>& q, size_t doc_id) const
{
    float score = 0.0f;
    const float doc_length = (float)doc_lengths[doc_id];
    const float length_norm = 1.0f - b + b * (doc_length / avg_doc_length);
    for (const std::string& term : q) {
        const float tf = (float)term_frequency(term, doc_id);
        if (tf == 0.0f) continue;
        auto it = doc_frequencies.find(term);
        const float df = it == doc_frequencies.end() ? 0.0f : (float)it->second;
        if (df == 0.0f) continue;
        const float idf = std::log(((float)total_docs - df + 0.5f) / (df + 0.5f) + 1.0f);
        const float ntf = (tf * (k1 + 1.0f)) / (tf + k1 * length_norm);
        score += idf * ntf;
    }
    return score;
}

std::vector<std::pair<size_t, float>> Bm25Index::search(const std::string& query, size_t limit) const
{
    std::vector<std::pair<size_t, float>> res;
    if (total_docs == 0) return res;
    const std::vector<std::string> q = tokenize(query);
    if (q.empty()) return res;
    // Same result as scoring every document with score() (bm25.rs:96-101): each document's sum is accumulated term by
    // term in query order, so the f32 additions happen in the order score() makes them, but every posting list is
    // walked once instead of once per (document, term).  Documents without a length slot cannot be scored either way.
    const size_t n = std::min(total_docs, doc_lengths.size());  // total_docs comes from the file
    std::vector<float> acc(n, 0.0f);
    std::vector<uint32_t> term_seen(n, 0);  // term_frequency() takes the first posting of a document
    std::vector<size_t> touched;
    for (size_t t = 0; t < q.size(); ++t) {
        auto it = inverted_index.find(q[t]);
        if (it == inverted_index.end()) continue;
        auto dit = doc_frequencies.find(q[t]);
        const float df = dit == doc_frequencies.end() ? 0.0f : (float)dit->second;
        const float idf = std::log(((float)total_docs - df + 0.5f) / (df + 0.5f) + 1.0f);
        for (const auto& p : it->second) {
            const size_t d = (size_t)p.first;
            if (d >= n) continue;
            if (term_seen[d] == 0) touched.push_back(d);
            if (term_seen[d] == t + 1) continue;
            term_seen[d] = (uint32_t)(t + 1);
            const float tf = (float)(size_t)p.second;
            if (tf == 0.0f || df == 0.0f) continue;
            const float length_norm = 1.0f - b + b * ((float)doc_lengths[d] / avg_doc_length);
            acc[d] += idf * ((tf * (k1 + 1.0f)) / (tf + k1 * length_norm));
        }
    }
    for (size_t d : touched)
        if (acc[d] > 0.0f) res.emplace_back(d, acc[d]);
    const auto better = [](const auto& x, const auto& y) {
        if (x.second != y.second) return x.second > y.second;
        return x.first < y.first;
    };
    if (res.size() > limit) {  // a total order on (score, id): the best `limit` are the head of the full sort
        std::partial_sort(res.begin(), res.begin() + (std::ptrdiff_t)limit, res.end(), better);
        res.resize(limit);
    } else {
        std::sort(res.begin(), res.end(), better);
    }
    return res;
}

Bm25Index Bm25Index::from_bincode(const uint8_t* data, size_t len)
{
    BinReader r{data, data + len};
    Bm25Index ix;
    uint64_t n = r.u64();
    for (uint64_t i = 0; i < n; ++i) {
        std::string k = r.str();
        ix.doc_frequencies[std::move(k)] = r.u64();
    }
    n = r.u64();
    r.need_records(n, 8);
    ix.doc_lengths.resize((size_t)n);
    for (uint64_t i = 0; i < n; ++i) ix.doc_lengths[(size_t)i] = r.u64();
    ix.avg_doc_length = r.f32();
    ix.total_docs = (size_t)r.u64();
    n = r.u64();
    for (uint64_t i = 0; i < n; ++i) {
        std::string k = r.str();
        const uint64_t m = r.u64();
        r.need_records(m, 16);
        auto& v = ix.inverted_index[std::move(k)];
        v.reserve((size_t)m);
        for (uint64_t j = 0; j < m; ++j) {
            const uint64_t d = r.u64(), c = r.u64();
            v.emplace_back(d, c);
        }
    }
    ix.k1 = r.f32();
    ix.b = r.f32();
    ix.epsilon = r.f32();
    n = r.u64();  // token_to_docs: HashMap<String, HashSet<usize>> (unused by search)
    for (uint64_t i = 0; i < n; ++i) {
        (void)r.str();
        const uint64_t m = r.u64();
        r.need_records(m, 8);
        r.p += m * 8;
    }
    ix.total_length = r.p < r.end ? r.u64() : 0;  // #[serde(default)]
    return ix;
}

std::string Bm25Index::to_bincode() const
{
    std::string o;
    put_u64(o, doc_frequencies.size());
    for (const auto& kv : doc_frequencies) {
        put_str(o, kv.first);
        put_u64(o, kv.second);
    }
    put_u64(o, doc_lengths.size());
    for (uint64_t v : doc_lengths) put_u64(o, v);
    put_f32(o, avg_doc_length);
    put_u64(o, total_docs);
    put_u64(o, inverted_index.size());
    for (const auto& kv : inverted_index) {
        put_str(o, kv.first);
        put_u64(o, kv.second.size());
        for (const auto& p : kv.second) {
            put_u64(o, p.first);
            put_u64(o, p.second);
        }
    }
    put_f32(o, k1);
    put_f32(o, b);
    put_f32(o, epsilon);
    put_u64(o, 0);
    put_u64(o, total_length);
    return o;
}

// ---------------------------------------------------------------------------------- RRF

std::vector<std::pair<size_t, float>> hybrid_search(const std::vector<std::pair<size_t, float>>& keyword,
                                                    const std::vector<std::pair<size_t, float>>& semantic, size_t limit)
{
    std::vector<std::pair<size_t, float>> comb;  // insertion order; small lists
    auto add = [&](size_t idx, float s) {
        for (auto& e : comb)
            if (e.first == idx) {
                e.second += s;
                return;
            }
        comb.emplace_back(idx, s);
    };
    const float k = 60.0f;
    for (size_t rank = 0; rank < keyword.size(); ++rank) add(keyword[rank].first, 1.0f / (k + (float)(rank + 1)));
    for (size_t rank = 0; rank < semantic.size(); ++rank) add(semantic[rank].first, 1.0f / (k + (float)(rank + 1)));
    std::stable_sort(comb.begin(), comb.end(), [](const auto& x, const auto& y) {
        if (x.second != y.second) return x.second > y.second;
        return x.first < y.first;
    });
    if (comb.size() > limit) comb.resize(limit);
    return comb;
}

// ---------------------------------------------------------------------------------- glob

// Restatement of the glob-match crate (0.2.1, Cargo.lock) as a recursive matcher over bytes:
//   !pat        leading '!'s negate the result
//   \x          x literally; a trailing backslash makes the pattern invalid (no match)
//   ?           one byte (the crate matches over &[u8]), never '/'
//   [a-z] [!..] byte class with ranges and escapes; a first ']' is literal
//   *           any run without '/'
//   **          a whole segment ("**" bounded by pattern start or '/' on the left and '/' or the end
//               on the right) spans separators, and "**/" also matches nothing (a/**/b ~ a/b);
//               a trailing "**" spans separators whatever precedes it; elsewhere it acts as '*'
//   {a,b}       alternatives, nestable
namespace {

bool unescape(const char*& p, const char* pe, unsigned char& c)
{
    c = (unsigned char)*p;
    if (c == '\\') {
        if (p + 1 >= pe) return false;
        ++p;
        c = (unsigned char)*p;
    }
    return true;
}

// What follows a brace alternative: the pattern text after the closing '}'.
struct Cont {
    const char* p;
    const char* pe;
    const Cont* next;
};

// `p0` is the first byte of the whole pattern (after any '!'): a "**" there has nothing to its left.
bool glob_rec(const char* p0, const char* p, const char* pe, const Cont* cont, const char* s0, const char* s,
              const char* se)
{
    while (p < pe) {
        const char c = *p;
        if (c == '*') {
            const bool two = p + 1 < pe && p[1] == '*';
            if (two) {
                const bool left_ok = p == p0 || p[-1] == '/';
                const char* np = p + 2;
                while (left_ok && np + 2 < pe && np[0] == '/' && np[1] == '*' && np[2] == '*' &&
                       (np + 3 == pe || np[3] == '/'))
                    np += 3;  // "**/**" collapses
                if (np == pe && !cont) return true;  // trailing "**"
                if (left_ok && np < pe && *np == '/') {
                    ++np;  // the segment is optional: try every segment start from here on
                    for (const char* t = s; t <= se; ++t)
                        if ((t == s0 || t[-1] == '/' || t == s) && glob_rec(p0, np, pe, cont, s0, t, se))
                            return true;
                    return false;
                }
                // not a whole segment: behaves as '*'
                for (const char* t = s;; ++t) {
                    if (glob_rec(p0, np, pe, cont, s0, t, se)) return true;
                    if (t >= se || *t == '/') return false;
                }
            }
            for (const char* t = s;; ++t) {
                if (glob_rec(p0, p + 1, pe, cont, s0, t, se)) return true;
                if (t >= se || *t == '/') return false;
            }
        } else if (c == '?') {
            if (s >= se || *s == '/') return false;
            ++s;
            ++p;
        } else if (c == '[') {
            if (s >= se) return false;
            ++p;
            bool neg = false;
            if (p < pe && (*p == '^' || *p == '!')) {
                neg = true;
                ++p;
            }
            const unsigned char ch = (unsigned char)*s;
            bool first = true, hit = false;
            while (p < pe && (first || *p != ']')) {
                unsigned char lo, hi;
                if (!unescape(p, pe, lo)) return false;
                ++p;
                if (p + 1 < pe && *p == '-' && p[1] != ']') {
                    ++p;
                    if (!unescape(p, pe, hi)) return false;
                    ++p;
                } else {
                    hi = lo;
                }
                if (lo <= ch && ch <= hi) hit = true;
                first = false;
            }
            if (p >= pe) return false;  // unterminated class
            ++p;
            if (hit == neg) return false;
            ++s;
        } else if (c == '{') {
            // find the matching '}' and the top-level commas
            int depth = 0;
            const char* close = nullptr;
            std::vector<const char*> cuts;
            for (const char* q = p; q < pe; ++q) {
                if (*q == '\\') {
                    ++q;
                    continue;
                }
                if (*q == '[') {  // a class is opaque
                    const char* r = q + 1;
                    if (r < pe && (*r == '!' || *r == '^')) ++r;
                    if (r < pe && *r == ']') ++r;
                    while (r < pe && *r != ']') r += (*r == '\\') ? 2 : 1;
                    q = r;
                    continue;
                }
                if (*q == '{') ++depth;
                else if (*q == '}') {
                    if (--depth == 0) {
                        close = q;
                        break;
                    }
                } else if (*q == ',' && depth == 1) cuts.push_back(q);
            }
            if (!close || depth > 10) return false;  // unbalanced / nested too deep: invalid pattern
            cuts.push_back(close);
            const Cont rest{close + 1, pe, cont};
            const char* a = p + 1;
            for (const char* cut : cuts) {
                if (glob_rec(p0, a, cut, &rest, s0, s, se)) return true;
                a = cut + 1;
            }
            return false;
        } else {
            unsigned char lit;
            if (!unescape(p, pe, lit)) return false;
            if (s >= se || (unsigned char)*s != lit) return false;
            ++s;
            ++p;
        }
    }
    if (cont) return glob_rec(p0, cont->p, cont->pe, cont->next, s0, s, se);
    return s == se;
}

}  // namespace

bool glob_match(const std::string& pattern, const std::string& path)
{
    const char* p = pattern.data();
    const char* pe = p + pattern.size();
    bool negated = false;
    while (p < pe && *p == '!') {
        negated = !negated;
        ++p;
    }
    // "!**/x": the '!' is what precedes the stars, so they are not a whole segment
    const char* p0 = p == pattern.data() ? p : nullptr;
    const bool m = glob_rec(p0, p, pe, nullptr, path.data(), path.data(), path.data() + path.size());
    return m != negated;
}

bool MetadataFilter::matches(const Metadata& md) const
{
    for (const auto& kv : must_match) {
        auto it = md.find(kv.first);
        if (it == md.end() || it->second != kv.second) return false;
    }
    for (const auto& kv : must_not_match) {
        auto it = md.find(kv.first);
        if (it != md.end() && it->second == kv.second) return false;
    }
    if (!source_patterns.empty()) {
        auto it = md.find("source");
        if (it == md.end()) return false;
        const std::string& source = it->second;
        // Path::file_name(): last component (ignoring trailing slashes); falls back to the whole string
        std::string trimmed = source;
        while (trimmed.size() > 1 && trimmed.back() == '/') trimmed.pop_back();
        const size_t slash = trimmed.rfind('/');
        std::string filename = slash == std::string::npos ? trimmed : trimmed.substr(slash + 1);
        if (filename.empty() || filename == "..") filename = source;
        bool any = false;
        for (const std::string& pat : source_patterns) {
            const bool has_sep = pat.find('/') != std::string::npos;
            if (glob_match(pat, has_sep ? source : filename)) {
                any = true;
                break;
            }
        }
        if (!any) return false;
    }
    return true;
}

// ---------------------------------------------------------------------------------- Segment

std::unique_ptr<Segment> Segment::open(const std::string& dir)
{
    std::unique_ptr<Segment> s(new Segment());
    s->dir_ = dir;
    static std::atomic<uint64_t> next_uid{1};
    s->uid_ = next_uid.fetch_add(1, std::memory_order_relaxed);
    const Json meta = Json::parse(slurp(dir + "/segment.json"));
    s->doc_count_ = (size_t)meta.get_int("doc_count", 0);
    s->dimension_ = (size_t)meta.get_int("dimension", 0);

    const std::string vpath = dir + "/vectors.bin";
    int fd = ::open(vpath.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open " + vpath);
    struct stat st;
    if (fstat(fd, &st) != 0) {
        ::close(fd);
        throw std::runtime_error("cannot stat " + vpath);
    }
    s->map_len_ = (size_t)st.st_size;
    if (s->map_len_ > 0) {
        s->map_ = mmap(nullptr, s->map_len_, PROT_READ, MAP_PRIVATE, fd, 0);
        if (s->map_ == MAP_FAILED) {
            s->map_ = nullptr;
            ::close(fd);
            throw std::runtime_error("mmap failed: " + vpath);
        }
        s->vectors_ = static_cast<const float*>(s->map_);
    }
    ::close(fd);

    const std::string idx = slurp(dir + "/docs.idx");
    BinReader r{reinterpret_cast<const uint8_t*>(idx.data()), reinterpret_cast<const uint8_t*>(idx.data()) + idx.size()};
    const uint64_t n = r.u64();
    r.need_records(n, 8);
    s->doc_offsets_.resize((size_t)n);
    for (uint64_t i = 0; i < n; ++i) s->doc_offsets_[(size_t)i] = r.u64();

    const std::string bm = slurp(dir + "/bm25.bin");
    s->bm25_ = Bm25Index::from_bincode(reinterpret_cast<const uint8_t*>(bm.data()), bm.size());
    return s;
}

namespace {
// (size, mtime, ctime, inode) of what a parsed Segment is revalidated by.
struct SegmentStamp {
    uint64_t v[24] = {};
    bool operator==(const SegmentStamp& o) const { return std::memcmp(v, o.v, sizeof v) == 0; }
};
// What a query re-checks of a parsed segment: the segment DIRECTORY (a file created, removed or renamed in it changes its
// mtime; a directory removed and re-created is another inode), segment.json, vectors.bin -- and docs.bin and metadata.jsonl,
// which stay MAPPED for the lifetime of the cached Segment: a truncation or rewrite of those by another process would
// otherwise end in SIGBUS (or stale text) instead of the reference's "corrupt offsets" error on the next query.  Segments are
// immutable once written -- the reference's writer and this one create a segment directory and never touch it again
// (kjarni-rag/src/segment.rs:90-170).  Five stat calls per segment and query (no open / pread / close per hit).
bool stamp_of(const std::string& dir, SegmentStamp& out)
{
    // (the directory is resolved ONCE -- O_PATH -- and its files are stat'ed relative to it: five path walks down a deep
    // index tree become one)
    static const char* const kFiles[5] = {"", "segment.json", "vectors.bin", "docs.bin", "metadata.jsonl"};
    const int dfd = ::open(dir.c_str(), O_PATH | O_DIRECTORY | O_CLOEXEC);
    if (dfd < 0) return false;  // Segment::open would throw
    bool ok = true;
    for (int i = 0; i < 5 && ok; ++i) {
        struct stat st;
        const int rc = i == 0 ? ::fstat(dfd, &st) : ::fstatat(dfd, kFiles[i], &st, 0);
        if (rc != 0) {
            if (i < 3) ok = false;  // Segment::open would throw
            continue;               // (read lazily: a missing file is that lookup's error; its stamp stays zero)
        }
        out.v[4 * i] = (uint64_t)st.st_size;
        out.v[4 * i + 1] = (uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec;
        out.v[4 * i + 2] = (uint64_t)st.st_ctim.tv_sec * 1000000000ull + (uint64_t)st.st_ctim.tv_nsec;
        out.v[4 * i + 3] = (uint64_t)st.st_ino;
    }
    ::close(dfd);
    return ok;
}
struct CachedSegment {
    SegmentStamp stamp;
    std::shared_ptr<const Segment> seg;
};
std::mutex g_segment_mu;
std::unordered_map<std::string, CachedSegment> g_segments;
}  // namespace

std::shared_ptr<const Segment> Segment::open_shared(const std::string& dir)
{
    SegmentStamp stamp;
    if (!stamp_of(dir, stamp)) return std::shared_ptr<const Segment>(open(dir).release());  // throws the open error
    {
        std::lock_guard<std::mutex> lock(g_segment_mu);
        auto it = g_segments.find(dir);
        if (it != g_segments.end() && it->second.stamp == stamp) return it->second.seg;
    }
    std::shared_ptr<const Segment> seg(open(dir).release());
    std::lock_guard<std::mutex> lock(g_segment_mu);
    if (g_segments.size() >= 4096) {  // drop what no reader holds (deleted or superseded indexes)
        for (auto it = g_segments.begin(); it != g_segments.end();)
            it = it->second.seg.use_count() == 1 ? g_segments.erase(it) : std::next(it);
    }
    g_segments[dir] = CachedSegment{stamp, seg};
    return seg;
}

void Segment::forget_under(const std::string& root)
{
    std::lock_guard<std::mutex> lock(g_segment_mu);
    for (auto it = g_segments.begin(); it != g_segments.end();)
        it = it->first.compare(0, root.size(), root) == 0 ? g_segments.erase(it) : std::next(it);
}

Segment::~Segment()
{
    if (map_) munmap(map_, map_len_);
    if (docs_map_) munmap(const_cast<char*>(docs_map_), (size_t)docs_size_);
    if (meta_map_) munmap(const_cast<char*>(meta_map_), (size_t)meta_size_);
}

namespace {
void pread_all(int fd, char* dst, size_t n, uint64_t off, const std::string& what)
{
    size_t done = 0;
    while (done < n) {
        const ssize_t r = ::pread(fd, dst + done, n - done, (off_t)(off + done));
        if (r <= 0) throw std::runtime_error("short read in " + what);
        done += (size_t)r;
    }
}
}  // namespace

namespace {
struct ScopedFd {
    int fd;
    explicit ScopedFd(const std::string& path) : fd(::open(path.c_str(), O_RDONLY | O_CLOEXEC)) {}
    ~ScopedFd()
    {
        if (fd >= 0) ::close(fd);
    }
    ScopedFd(const ScopedFd&) = delete;
    ScopedFd& operator=(const ScopedFd&) = delete;
};
uint64_t fd_size(int fd)
{
    struct stat st;
    return fstat(fd, &st) == 0 ? (uint64_t)st.st_size : 0;
}
}  // namespace

// docs.bin / metadata.jsonl of a parsed segment are mapped on first use and stay mapped while the Segment lives (segments are
// immutable; a changed file makes a new Segment): a hit's text and metadata are then copies out of the page cache with no
// system call -- ten hits were sixty open / pread / close calls per query.  No descriptor is held (an index may have thousands
// of segments).  A file that cannot be mapped (empty, or mmap fails) is read with pread as before.
namespace {
const char* map_whole(const std::string& path, uint64_t& size_out)
{
    const ScopedFd f(path);
    if (f.fd < 0) throw std::runtime_error("cannot open " + path);
    size_out = fd_size(f.fd);
    if (size_out == 0) return nullptr;
    void* m = mmap(nullptr, (size_t)size_out, PROT_READ, MAP_PRIVATE, f.fd, 0);
    return m == MAP_FAILED ? nullptr : static_cast<const char*>(m);
}
void read_range(const char* map, const std::string& path, char* dst, size_t n, uint64_t off)
{
    if (n == 0) return;
    if (map) {
        std::memcpy(dst, map + off, n);
        return;
    }
    const ScopedFd f(path);
    if (f.fd < 0) throw std::runtime_error("cannot open " + path);
    pread_all(f.fd, dst, n, off, path);
}
}  // namespace

std::string Segment::get_document(size_t doc_id) const
{
    if (doc_id >= doc_count_ || doc_id >= doc_offsets_.size()) throw std::runtime_error("Document ID out of range");
    const std::string path = dir_ + "/docs.bin";
    std::call_once(docs_once_, [&] { docs_map_ = map_whole(path, docs_size_); });
    const uint64_t start = doc_offsets_[doc_id];
    // the next document's offset, or the file's end; -1 for the newline
    const uint64_t next = doc_id + 1 < doc_offsets_.size() ? doc_offsets_[doc_id + 1] : docs_size_;
    if (next == 0 || next - 1 < start || next - 1 > docs_size_) throw std::runtime_error("corrupt document offsets");
    const uint64_t end = next - 1;
    std::string buf((size_t)(end - start), '\0');
    read_range(docs_map_, path, buf.empty() ? nullptr : &buf[0], buf.size(), start);
    if (!unicode::is_valid_utf8(buf.data(), buf.size())) throw std::runtime_error("Invalid UTF-8 in document");
    return buf;
}

Metadata Segment::get_metadata(size_t doc_id) const
{
    const std::string path = dir_ + "/metadata.jsonl";
    std::call_once(meta_once_, [&] {  // one pass for the line starts; later lookups read one line
        meta_map_ = map_whole(path, meta_size_);
        std::vector<char> chunk(meta_map_ ? 0 : (1 << 16));
        uint64_t off = 0;
        bool at_line_start = true;
        while (off < meta_size_) {
            const size_t want = (size_t)std::min<uint64_t>(1 << 16, meta_size_ - off);
            const char* p = meta_map_ ? meta_map_ + off : chunk.data();
            if (!meta_map_) read_range(nullptr, path, chunk.data(), want, off);
            for (size_t i = 0; i < want; ++i) {
                if (at_line_start) meta_offsets_.push_back(off + i);
                at_line_start = p[i] == '\n';
            }
            off += want;
        }
    });
    if (doc_id >= meta_offsets_.size()) throw std::runtime_error("Document ID out of range");
    const uint64_t start = meta_offsets_[doc_id];
    uint64_t end = doc_id + 1 < meta_offsets_.size() ? meta_offsets_[doc_id + 1] : meta_size_;
    std::string line((size_t)(end - start), '\0');
    read_range(meta_map_, path, line.empty() ? nullptr : &line[0], line.size(), start);
    while (!line.empty() && line.back() == '\n') line.pop_back();
    const Json j = Json::parse(line);
    if (!j.is_object()) throw std::runtime_error("Invalid JSON: metadata is not an object");
    Metadata md;
    for (const auto& kv : j.obj) {
        if (!kv.second.is_string()) throw std::runtime_error("Invalid JSON: metadata values must be strings");
        md[kv.first] = kv.second.str;
    }
    return md;
}

// ---------------------------------------------------------------------------------- IndexReader

std::unique_ptr<IndexReader> IndexReader::open(const std::string& root)
{
    std::unique_ptr<IndexReader> r(new IndexReader());
    const Json cfg = Json::parse(slurp(root + "/config.json"));
    r->dimension_ = (size_t)cfg.get_int("dimension", 0);
    const std::string segdir = root + "/segments";
    if (is_dir(segdir)) {
        std::vector<std::string> names;
        if (DIR* d = opendir(segdir.c_str())) {
            while (dirent* e = readdir(d)) {
                const std::string n = e->d_name;
                if (n == "." || n == "..") continue;
                // (d_type where the file system reports it: no stat per entry)
                if (e->d_type == DT_DIR || ((e->d_type == DT_UNKNOWN || e->d_type == DT_LNK) && is_dir(segdir + "/" + n))) names.push_back(n);
            }
            closedir(d);
        }
        std::sort(names.begin(), names.end());  // entries.sort_by_key(file_name)
        for (const std::string& n : names) {
            try {
                r->segments_.push_back(Segment::open_shared(segdir + "/" + n));
            } catch (const std::exception&) {
                // index_reader.rs:178-181: a segment that fails to load is skipped with a warning
            }
        }
    }
    for (const auto& s : r->segments_) r->total_docs_ += s->doc_count();
    return r;
}

size_t IndexReader::local_to_global(size_t seg, size_t local) const
{
    size_t off = 0;
    for (size_t i = 0; i < seg; ++i) off += segments_[i]->doc_count();
    return off + local;
}

bool IndexReader::global_to_local(size_t global, size_t& seg, size_t& local) const
{
    size_t off = 0;
    for (size_t i = 0; i < segments_.size(); ++i) {
        if (global < off + segments_[i]->doc_count()) {
            seg = i;
            local = global - off;
            return true;
        }
        off += segments_[i]->doc_count();
    }
    return false;
}

std::vector<SearchHit> IndexReader::convert(const std::vector<std::tuple<size_t, size_t, float>>& rs) const
{
    std::vector<SearchHit> out;
    for (const auto& t : rs) {
        const size_t seg = std::get<0>(t), doc = std::get<1>(t);
        try {  // filter_map(... .ok()?): a document whose text or metadata cannot be read is dropped
            SearchHit h;
            h.score = std::get<2>(t);
            h.document_id = local_to_global(seg, doc);
            h.text = segments_[seg]->get_document(doc);
            h.metadata = segments_[seg]->get_metadata(doc);
            out.push_back(std::move(h));
        } catch (const std::exception&) {
        }
    }
    return out;
}

static void sort_desc_truncate(std::vector<std::tuple<size_t, size_t, float>>& all, size_t limit)
{
    std::stable_sort(all.begin(), all.end(), [](const auto& x, const auto& y) { return std::get<2>(x) > std::get<2>(y); });
    if (all.size() > limit) all.resize(limit);
}

std::vector<SearchHit> IndexReader::search_semantic(const float* query, size_t limit, const SegmentScanFn& scan) const
{
    std::vector<std::tuple<size_t, size_t, float>> all;
    std::vector<const Segment*> segs;
    for (const auto& s : segments_) segs.push_back(s.get());
    const std::vector<SegmentHits> per_segment = scan(segs, query, limit);
    for (size_t si = 0; si < per_segment.size() && si < segments_.size(); ++si)
        for (const auto& r : per_segment[si]) all.emplace_back(si, r.first, r.second);
    sort_desc_truncate(all, limit);
    return convert(all);
}

static std::atomic<size_t> g_keyword_parallel_min_docs{50000};
void set_keyword_parallel_min_docs(size_t docs) { g_keyword_parallel_min_docs.store(docs, std::memory_order_relaxed); }

std::vector<SearchHit> IndexReader::search_keywords(const std::string& query, size_t limit) const
{
    // Segments are independent (bm25.rs scores a segment's documents against that segment's statistics): a large index is
    // walked by a few host threads, each a contiguous run of segments; the hits are merged in segment order as before.
    const size_t nseg = segments_.size();
    std::vector<std::vector<std::pair<size_t, float>>> per(nseg);
    size_t docs = 0;
    for (const auto& sg : segments_) docs += sg->doc_count();
    size_t workers = std::min<size_t>({nseg / 2, (size_t)8, (size_t)std::max(1u, std::thread::hardware_concurrency())});
    if (docs < g_keyword_parallel_min_docs.load(std::memory_order_relaxed)) workers = 1;
    if (workers < 2) {
        for (size_t si = 0; si < nseg; ++si) per[si] = segments_[si]->search_keywords(query, limit);
    } else {
        std::vector<std::thread> pool;
        std::vector<std::exception_ptr> errs(workers);
        for (size_t w = 0; w < workers; ++w)
            pool.emplace_back([&, w] {
                try {
                    for (size_t si = nseg * w / workers, e = nseg * (w + 1) / workers; si < e; ++si)
                        per[si] = segments_[si]->search_keywords(query, limit);
                } catch (...) {
                    errs[w] = std::current_exception();
                }
            });
        for (std::thread& t : pool) t.join();
        for (const std::exception_ptr& e : errs)
            if (e) std::rethrow_exception(e);
    }
    std::vector<std::tuple<size_t, size_t, float>> all;
    for (size_t si = 0; si < nseg; ++si)
        for (const auto& r : per[si]) all.emplace_back(si, r.first, r.second);
    sort_desc_truncate(all, limit);
    return convert(all);
}

std::vector<SearchHit> IndexReader::search_hybrid(const std::string& query, const float* query_emb, size_t limit,
                                                  const SegmentScanFn& scan) const
{
    const std::vector<SearchHit> kw = search_keywords(query, limit * 2);
    const std::vector<SearchHit> sem = search_semantic(query_emb, limit * 2, scan);
    std::vector<std::pair<size_t, float>> kwi, semi;
    for (const SearchHit& h : kw) kwi.emplace_back(h.document_id, h.score);
    for (const SearchHit& h : sem) semi.emplace_back(h.document_id, h.score);
    std::vector<SearchHit> out;
    for (const auto& f : hybrid_search(kwi, semi, limit)) {
        size_t seg = 0, local = 0;
        if (!global_to_local(f.first, seg, local)) continue;
        try {
            SearchHit h;
            h.score = f.second;
            h.document_id = f.first;
            h.text = segments_[seg]->get_document(local);
            h.metadata = segments_[seg]->get_metadata(local);
            out.push_back(std::move(h));
        } catch (const std::exception&) {
        }
    }
    return out;
}

std::vector<SearchHit> IndexReader::apply_filter(std::vector<SearchHit> hits, const MetadataFilter& f, size_t limit) const
{
    std::vector<SearchHit> out;
    for (SearchHit& h : hits) {
        if (out.size() >= limit) break;
        if (f.matches(h.metadata)) out.push_back(std::move(h));
    }
    return out;
}

// serde_json string escaping: ", \\ and control characters; everything else verbatim.
std::string json_escape(const std::string& s)
{
    std::string o = "\"";
    for (unsigned char c : s) {
        switch (c) {
        case '"': o += "\\\""; break;
        case '\\': o += "\\\\"; break;
        case '\n': o += "\\n"; break;
        case '\r': o += "\\r"; break;
        case '\t': o += "\\t"; break;
        case '\b': o += "\\b"; break;
        case '\f': o += "\\f"; break;
        default:
            if (c < 0x20) {
                char buf[8];
                std::snprintf(buf, sizeof buf, "\\u%04x", c);
                o += buf;
            } else {
                o.push_back((char)c);
            }
        }
    }
    return o + "\"";
}

std::string metadata_to_json(const Metadata& md)
{
    std::string o = "{";
    bool first = true;
    for (const auto& kv : md) {
        if (!first) o += ",";
        first = false;
        o += json_escape(kv.first) + ":" + json_escape(kv.second);
    }
    return o + "}";
}

}  // namespace kjarni
