// kjarni_searcher_* (crates/kjarni-ffi/src/searcher.rs:12-491) on top of the GPU cosine scan.
//
// Searcher::search_with_options (crates/kjarni/src/searcher/model.rs:96-187) re-opens the index on
// every call; so does this.  What is cached across calls is the DEVICE copy of each segment's
// vectors.bin (keyed by path, validated by size + mtime), so the scan streams from HBM instead of
// re-crossing PCIe per query.  BM25, rank fusion, filters and result shaping are host logic
// (index.cpp), exactly the parts the reference also runs on the CPU.
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <numeric>

#include "host_util.h"
#include "ffi_common.h"
#include "index.h"
#include "pipeline.h"
#include "unicode.h"

using namespace kjarni;

namespace {

// Device copy of one segment's vectors.bin, revalidated on every scan by the file's identity
// (size, mtime, ctime, inode -- a segment re-created under the same name is a different file).
struct DeviceSegment {
    float* vectors = nullptr;
    size_t bytes = 0;
    int64_t mtime_ns = 0, ctime_ns = 0;
    uint64_t inode = 0;
    uint64_t last_used = 0;  // scan counter: least-recently-used copies go first when the budget is exceeded
};

void fill_results(const std::vector<SearchHit>& hits, KjarniSearchResults* out)
{
    out->results = nullptr;
    out->len = 0;
    if (hits.empty()) return;
    auto* arr = static_cast<KjarniSearchResult*>(std::calloc(hits.size(), sizeof(KjarniSearchResult)));
    if (!arr) throw std::bad_alloc();
    for (size_t i = 0; i < hits.size(); ++i) {
        arr[i].score = hits[i].score;
        arr[i].document_id = hits[i].document_id;
        arr[i].text = dup_cstr(hits[i].text);
        arr[i].metadata_json = dup_cstr(metadata_to_json(hits[i].metadata));
    }
    out->results = arr;
    out->len = hits.size();
}

struct IndexOpenFailed : std::runtime_error {
    using std::runtime_error::runtime_error;
};

std::unique_ptr<IndexReader> open_index(const std::string& path)
{
    try {
        return IndexReader::open(path);
    } catch (const std::exception& e) {
        throw IndexOpenFailed(std::string("Search failed: ") + e.what());
    }
}

class SegmentScanner;
std::mutex g_scanners_mu;
std::vector<SegmentScanner*> g_scanners;  // live scanners, so that kjarni_index_delete can drop their device copies

// Segment::search_vectors (kjarni-rag/src/segment.rs:307-337) on the GPU, with the device copies of
// the segments it has seen.
class SegmentScanner {
public:
    explicit SegmentScanner(int device) : device_(device)
    {
        {
            std::lock_guard<std::mutex> lock(g_scanners_mu);
            g_scanners.push_back(this);
        }
        // The copies may take half of the device's memory; past that, copies no scan has touched longest are dropped
        // (a long-lived searcher that walks many indexes must not grow until hipMalloc fails).
        size_t free_b = 0, total_b = 0;
        if (hipSetDevice(device_) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0)
            budget_bytes_ = total_b / 2;
        else
            (void)hipGetLastError();
    }
    void set_budget(size_t bytes) { budget_bytes_ = bytes; }
    size_t cached_bytes() const { return cached_bytes_; }
    SegmentScanner(const SegmentScanner&) = delete;
    SegmentScanner& operator=(const SegmentScanner&) = delete;
    ~SegmentScanner()
    {
        {
            std::lock_guard<std::mutex> lock(g_scanners_mu);
            g_scanners.erase(std::remove(g_scanners.begin(), g_scanners.end(), this), g_scanners.end());
        }
        (void)hipSetDevice(device_);
        for (auto& kv : cache_)
            if (kv.second.vectors) (void)hipFree(kv.second.vectors);
        if (work_) (void)hipFree(work_);
    }

    // One query against every segment: the kernels of all segments are enqueued back to back and the host waits once.
    std::vector<SegmentHits> scan(const std::vector<const Segment*>& segs, const float* query, size_t query_dim, size_t limit)
    {
        std::vector<SegmentHits> out(segs.size());
        if (limit == 0 || query_dim == 0) return out;
        float qn = 0.0f;
        for (size_t i = 0; i < query_dim; ++i) qn += query[i] * query[i];
        if (std::sqrt(qn) < 1e-9f) return out;                          // segment.rs:315-317

        std::lock_guard<std::mutex> lock(mu_);
        hip_check(hipSetDevice(device_), "hipSetDevice");
        const uint64_t tick = ++tick_;
        auto pad = [](size_t b) { return 256 * ((b + 255) / 256); };
        struct Plan {
            size_t seg, n;
            int k;
            DeviceSegment* ds;
            size_t s_off, w_off, r_off;  // scores, top-k workspace, slot in the result arrays
        };
        std::vector<Plan> plans;
        size_t cursor = pad(query_dim * 4), slots = 0;
        for (size_t si = 0; si < segs.size(); ++si) {
            const Segment& seg = *segs[si];
            const size_t n = seg.doc_count(), dim = seg.dimension();
            if (n == 0 || dim != query_dim) continue;                        // query.len() != dimension -> empty
            if (seg.vectors_bytes() < n * dim * sizeof(float)) continue;     // get_embedding() would return None
            struct stat st;
            const std::string vpath = seg.dir() + "/vectors.bin";
            if (::stat(vpath.c_str(), &st) != 0) continue;
            const int64_t mt = (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
            const int64_t ct = (int64_t)st.st_ctim.tv_sec * 1000000000ll + st.st_ctim.tv_nsec;
            DeviceSegment& ds = cache_[vpath];
            const size_t bytes = n * dim * sizeof(float);
            if (!ds.vectors || ds.bytes != bytes || ds.mtime_ns != mt || ds.ctime_ns != ct || ds.inode != (uint64_t)st.st_ino) {
                drop(ds);
                make_room(bytes, tick);
                hip_check(hipMalloc((void**)&ds.vectors, bytes), "hipMalloc(segment vectors)");
                ds.bytes = bytes;
                cached_bytes_ += bytes;
                hip_check(hipMemcpy(ds.vectors, seg.vectors(), bytes, hipMemcpyHostToDevice), "H2D segment vectors");
                ds.mtime_ns = mt;
                ds.ctime_ns = ct;
                ds.inode = (uint64_t)st.st_ino;
            }
            ds.last_used = tick;
            Plan p;
            p.seg = si;
            p.n = n;
            p.k = (int)std::min(limit, n);
            p.ds = &ds;
            p.s_off = cursor;
            cursor += pad(n * 4);
            p.w_off = cursor;
            cursor += pad(cosine_topk_workspace_bytes(1, (int64_t)n, p.k));
            p.r_off = slots;
            slots += (size_t)p.k;
            plans.push_back(p);
        }
        if (plans.empty()) return out;
        const size_t i_off = cursor, o_off = i_off + pad(slots * 8), total = o_off + pad(slots * 4);
        if (total > work_bytes_) {
            if (work_) {
                hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
                (void)hipFree(work_);
                work_ = nullptr;
                work_bytes_ = 0;
            }
            hip_check(hipMalloc((void**)&work_, total), "hipMalloc(scan workspace)");
            work_bytes_ = total;
        }
        float* q_d = reinterpret_cast<float*>(work_);
        int64_t* i_d = reinterpret_cast<int64_t*>(work_ + i_off);
        float* o_d = reinterpret_cast<float*>(work_ + o_off);
        hip_check(hipMemcpyAsync(q_d, query, query_dim * 4, hipMemcpyHostToDevice, nullptr), "H2D query");
        for (const Plan& p : plans) {
            float* s_d = reinterpret_cast<float*>(work_ + p.s_off);
            hip_check(launch_cosine_scores(q_d, 1, p.ds->vectors, (int64_t)p.n, (int)query_dim, /*segment mode*/ 1, s_d, nullptr),
                      "cosine_scores");
            hip_check(launch_cosine_topk(s_d, 1, (int64_t)p.n, p.k, work_ + p.w_off, i_d + p.r_off, o_d + p.r_off, nullptr),
                      "cosine_topk");
        }
        std::vector<int64_t> idx(slots);
        std::vector<float> sc(slots);
        hip_check(hipMemcpyAsync(idx.data(), i_d, slots * 8, hipMemcpyDeviceToHost, nullptr), "D2H idx");
        hip_check(hipMemcpyAsync(sc.data(), o_d, slots * 4, hipMemcpyDeviceToHost, nullptr), "D2H scores");
        hip_check(hipStreamSynchronize(nullptr), "hipStreamSynchronize");
        for (const Plan& p : plans)
            for (int i = 0; i < p.k; ++i)
                if (idx[p.r_off + (size_t)i] >= 0) out[p.seg].emplace_back((size_t)idx[p.r_off + (size_t)i], sc[p.r_off + (size_t)i]);
        return out;
    }

    // Drops the copies of every segment under `root` (the index there is being deleted or rebuilt).
    void forget_under(const std::string& root)
    {
        std::lock_guard<std::mutex> lock(mu_);
        (void)hipSetDevice(device_);
        const std::string prefix = root.empty() || root.back() == '/' ? root : root + "/";
        for (auto it = cache_.begin(); it != cache_.end();)
            if (it->first.compare(0, prefix.size(), prefix) == 0) {
                drop(it->second);
                it = cache_.erase(it);
            } else {
                ++it;
            }
    }

private:
    void drop(DeviceSegment& ds)
    {
        if (!ds.vectors) return;
        (void)hipDeviceSynchronize();  // earlier scans may still be reading it
        (void)hipFree(ds.vectors);
        ds.vectors = nullptr;
        cached_bytes_ -= ds.bytes;
        ds.bytes = 0;
    }
    // Evicts least-recently-used copies that the current scan (tick) has not touched until `incoming` more bytes fit.
    void make_room(size_t incoming, uint64_t tick)
    {
        while (budget_bytes_ && cached_bytes_ + incoming > budget_bytes_) {
            auto victim = cache_.end();
            for (auto it = cache_.begin(); it != cache_.end(); ++it)
                if (it->second.vectors && it->second.last_used != tick &&
                    (victim == cache_.end() || it->second.last_used < victim->second.last_used))
                    victim = it;
            if (victim == cache_.end()) return;  // everything resident belongs to this scan
            drop(victim->second);
            cache_.erase(victim);
        }
    }

    int device_;
    std::mutex mu_;
    std::map<std::string, DeviceSegment> cache_;
    size_t cached_bytes_ = 0, budget_bytes_ = 0;
    uint64_t tick_ = 0;
    uint8_t* work_ = nullptr;
    size_t work_bytes_ = 0;
};

struct ResolvedOptions {
    KjarniSearchMode mode;
    size_t top_k;
    bool use_reranker;
    bool has_threshold = false;
    float threshold = 0.0f;
    MetadataFilter filter;
};

// option sentinels: searcher.rs:293-310
ResolvedOptions resolve_options(const KjarniSearchOptions* options, KjarniSearchMode default_mode, size_t default_top_k,
                                bool default_rerank)
{
    ResolvedOptions r;
    r.mode = default_mode;
    r.top_k = default_top_k;
    r.use_reranker = default_rerank;
    if (!options) return r;
    if (options->mode >= 0)
        r.mode = options->mode == 0 ? KJARNI_SEARCH_KEYWORD : options->mode == 1 ? KJARNI_SEARCH_SEMANTIC : KJARNI_SEARCH_HYBRID;
    if (options->top_k > 0) r.top_k = options->top_k;
    if (options->use_reranker >= 0) r.use_reranker = options->use_reranker != 0;
    if (options->threshold > 0.0f) {
        r.has_threshold = true;
        r.threshold = options->threshold;
    }
    if (options->source_pattern && valid_utf8(options->source_pattern))
        r.filter.source_patterns.push_back(options->source_pattern);
    if (options->filter_key && options->filter_value && valid_utf8(options->filter_key) &&
        valid_utf8(options->filter_value))
        r.filter.must_match[options->filter_key] = options->filter_value;
    return r;
}

// The retrieval half of Searcher::search_with_options (crates/kjarni/src/searcher/model.rs:120-160):
// mode dispatch, 3x over-fetch + filter.  `query_emb` may be null in keyword mode.
std::vector<SearchHit> retrieve(const IndexReader& reader, KjarniSearchMode mode, const std::string& query,
                                const float* query_emb, size_t fetch_k, const MetadataFilter& filter,
                                const SegmentScanFn& scan)
{
    const bool filtered = !filter.empty();
    const size_t k_eff = filtered ? fetch_k * 3 : fetch_k;
    std::vector<SearchHit> results;
    if (mode == KJARNI_SEARCH_KEYWORD)
        results = reader.search_keywords(query, k_eff);
    else if (mode == KJARNI_SEARCH_SEMANTIC)
        results = reader.search_semantic(query_emb, k_eff, scan);
    else
        results = reader.search_hybrid(query, query_emb, k_eff, scan);
    if (filtered) results = reader.apply_filter(std::move(results), filter, fetch_k);
    return results;
}

void apply_threshold_and_limit(std::vector<SearchHit>& results, const ResolvedOptions& o)
{
    if (o.has_threshold)
        results.erase(std::remove_if(results.begin(), results.end(), [&](const SearchHit& h) { return !(h.score >= o.threshold); }),
                      results.end());
    if (results.size() > o.top_k) results.resize(o.top_k);
}

}  // namespace

namespace kjarni {
void forget_device_segments_under(const std::string& root)
{
    std::lock_guard<std::mutex> lock(g_scanners_mu);
    for (SegmentScanner* sc : g_scanners) sc->forget_under(root);
}
}  // namespace kjarni

struct KjarniSearcher {
    std::unique_ptr<Pipeline> embedder;
    std::unique_ptr<Pipeline> reranker;  // optional
    KjarniSearchMode default_mode = KJARNI_SEARCH_HYBRID;
    size_t default_top_k = 10;
    std::unique_ptr<SegmentScanner> scanner;
};

KJARNI_EXPORT void kjarni_search_results_free(const KjarniSearchResults* r)
{
    if (!r) return;
    if (r->results && r->len > 0) {
        for (size_t i = 0; i < r->len; ++i) {
            std::free(r->results[i].text);
            std::free(r->results[i].metadata_json);
        }
        std::free(r->results);
    }
}

KJARNI_EXPORT KjarniSearchOptions kjarni_search_options_default(void)
{
    KjarniSearchOptions o;
    std::memset(&o, 0, sizeof o);
    o.mode = -1;
    o.top_k = 0;
    o.use_reranker = -1;
    o.threshold = 0.0f;
    return o;
}

KJARNI_EXPORT KjarniSearcherConfig kjarni_searcher_config_default(void)
{
    KjarniSearcherConfig c;
    std::memset(&c, 0, sizeof c);
    c.device = KJARNI_DEVICE_CPU;
    c.default_mode = KJARNI_SEARCH_HYBRID;
    c.default_top_k = 10;
    c.quiet = 0;
    return c;
}

KJARNI_EXPORT KjarniErrorCode kjarni_searcher_new(const KjarniSearcherConfig* config, KjarniSearcher** out)
{
    if (!out) return KJARNI_ERROR_NULL_POINTER;
    const KjarniSearcherConfig dflt = kjarni_searcher_config_default();
    const KjarniSearcherConfig& c = config ? *config : dflt;
    for (const char* s : {c.cache_dir, c.model_name, c.rerank_model})
        if (s && !valid_utf8(s)) return KJARNI_ERROR_INVALID_UTF8;
    // searcher.rs:226-229: every build error is reported as LoadFailed (a missing GPU keeps its own code).
    try {
        auto h = std::make_unique<KjarniSearcher>();
        h->embedder = load_pipeline(c.cache_dir, c.model_name, nullptr, "minilm-l6-v2", Want::Embedding);
        if (c.rerank_model) h->reranker = load_pipeline(c.cache_dir, c.rerank_model, nullptr, "", Want::Reranking);
        h->scanner = std::make_unique<SegmentScanner>(h->embedder->model().device());
        h->default_mode = c.default_mode;
        if (c.default_top_k > 0) h->default_top_k = c.default_top_k;
        *out = h.release();
        return KJARNI_OK;
    } catch (const GpuUnavailable& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_GPU_UNAVAILABLE;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_LOAD_FAILED;
    } catch (...) {
        set_last_error("unknown error");
        return KJARNI_ERROR_LOAD_FAILED;
    }
}

KJARNI_EXPORT void kjarni_searcher_free(KjarniSearcher* s) { delete s; }

KJARNI_EXPORT KjarniErrorCode kjarni_searcher_search_with_options(KjarniSearcher* s, const char* index_path,
                                                                  const char* query, const KjarniSearchOptions* options,
                                                                  KjarniSearchResults* out)
{
    if (!s || !index_path || !query || !out) return KJARNI_ERROR_NULL_POINTER;
    out->results = nullptr;
    out->len = 0;
    if (!valid_utf8(index_path) || !valid_utf8(query)) return KJARNI_ERROR_INVALID_UTF8;
    try {
        const ResolvedOptions o = resolve_options(options, s->default_mode, s->default_top_k, s->reranker != nullptr);

        // Searcher::search_with_options (crates/kjarni/src/searcher/model.rs:96-187)
        std::unique_ptr<IndexReader> reader = open_index(index_path);
        const size_t model_dim = (size_t)s->embedder->config().hidden;
        if (reader->dimension() != model_dim)
            throw InvalidConfig("Index dimension (" + std::to_string(reader->dimension()) +
                                ") doesn't match model dimension (" + std::to_string(model_dim) + ")");
        const bool rerank = o.use_reranker && s->reranker;
        const size_t fetch_k = rerank ? o.top_k * 5 : o.top_k;
        const SegmentScanFn scan = [&](const std::vector<const Segment*>& segs, const float* q, size_t limit) {
            return s->scanner->scan(segs, q, model_dim, limit);
        };
        std::vector<float> q;
        if (o.mode != KJARNI_SEARCH_KEYWORD)  // embedder.embed(query): mean pool, normalised (embedder/model.rs:118-140)
            q = embed_texts(*s->embedder, {std::string(query)}, POOL_MEAN, true);
        std::vector<SearchHit> results = retrieve(*reader, o.mode, query, q.data(), fetch_k, o.filter, scan);

        if (rerank) {
            std::vector<std::string> texts;
            for (const SearchHit& h : results) texts.push_back(h.text);
            if (!texts.empty()) {
                const std::vector<float> scores = rerank_scores(*s->reranker, query, texts);
                std::vector<size_t> order(texts.size());
                std::iota(order.begin(), order.end(), (size_t)0);
                std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return scores[a] > scores[b]; });
                std::vector<SearchHit> nr;
                for (size_t i = 0; i < order.size() && i < o.top_k; ++i) {
                    SearchHit h = results[order[i]];
                    h.score = scores[order[i]];
                    nr.push_back(std::move(h));
                }
                results.swap(nr);
            }
        }
        apply_threshold_and_limit(results, o);
        fill_results(results, out);
        return KJARNI_OK;
    } catch (const InvalidConfig& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_INVALID_CONFIG;  // DimensionMismatch
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_INFERENCE_FAILED;
    } catch (...) {
        set_last_error("unknown error");
        return KJARNI_ERROR_INFERENCE_FAILED;
    }
}

KJARNI_EXPORT KjarniErrorCode kjarni_searcher_search(KjarniSearcher* s, const char* index_path, const char* query,
                                                     KjarniSearchResults* out)
{
    const KjarniSearchOptions o = kjarni_search_options_default();
    return kjarni_searcher_search_with_options(s, index_path, query, &o, out);
}

// Searcher::search_keywords (model.rs:79-89): BM25 only, no model and no GPU involved.
KJARNI_EXPORT KjarniErrorCode kjarni_search_keywords(const char* index_path, const char* query, size_t top_k,
                                                     KjarniSearchResults* out)
{
    if (!index_path || !query || !out) return KJARNI_ERROR_NULL_POINTER;
    out->results = nullptr;
    out->len = 0;
    if (!valid_utf8(index_path) || !valid_utf8(query)) return KJARNI_ERROR_INVALID_UTF8;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        std::unique_ptr<IndexReader> reader = open_index(index_path);
        fill_results(reader->search_keywords(query, top_k), out);
    });
}

KJARNI_EXPORT bool kjarni_searcher_has_reranker(const KjarniSearcher* s) { return s && s->reranker; }
KJARNI_EXPORT KjarniSearchMode kjarni_searcher_default_mode(const KjarniSearcher* s) { return s ? s->default_mode : KJARNI_SEARCH_HYBRID; }
KJARNI_EXPORT size_t kjarni_searcher_default_top_k(const KjarniSearcher* s) { return s ? s->default_top_k : 10; }

namespace {
size_t copy_name(const std::string& name, char* buf, size_t buf_len)
{
    const size_t required = name.size() + 1;
    if (!buf || buf_len == 0) return required;
    const size_t n = std::min(name.size(), buf_len - 1);
    std::memcpy(buf, name.data(), n);
    buf[n] = '\0';
    return required;
}
}  // namespace

KJARNI_EXPORT size_t kjarni_searcher_model_name(const KjarniSearcher* s, char* buf, size_t buf_len)
{
    if (!s) return 0;
    return copy_name(s->embedder->model_name, buf, buf_len);
}

KJARNI_EXPORT size_t kjarni_searcher_reranker_model(const KjarniSearcher* s, char* buf, size_t buf_len)
{
    if (!s || !s->reranker) return 0;
    return copy_name(s->reranker->model_name, buf, buf_len);
}

// ---- test / tooling hooks for the host-side search logic (kjarni_hip.h) ----------------------------

KJARNI_EXPORT KjarniErrorCode kjarni_bm25_tokenize(const char* text, KjarniStringArray* out)
{
    if (!text || !out) return KJARNI_ERROR_NULL_POINTER;
    out->strings = nullptr;
    out->len = 0;
    if (!valid_utf8(text)) return KJARNI_ERROR_INVALID_UTF8;
    return guarded(KJARNI_ERROR_UNKNOWN, [&] {
        const std::vector<std::string> toks = Bm25Index::tokenize(text);
        if (toks.empty()) return;
        char** arr = static_cast<char**>(std::calloc(toks.size(), sizeof(char*)));
        if (!arr) throw std::bad_alloc();
        for (size_t i = 0; i < toks.size(); ++i) arr[i] = dup_cstr(toks[i]);
        out->strings = arr;
        out->len = toks.size();
    });
}

KJARNI_EXPORT int32_t kjarni_glob_match(const char* pattern, const char* path)
{
    if (!pattern || !path) return 0;
    return glob_match(pattern, path) ? 1 : 0;
}

KJARNI_EXPORT KjarniErrorCode kjarni_rrf_fuse(const size_t* keyword_ids, size_t n_keyword, const size_t* semantic_ids,
                                              size_t n_semantic, size_t limit, size_t* ids_out, float* scores_out,
                                              size_t* n_out)
{
    if (!ids_out || !scores_out || !n_out || (n_keyword && !keyword_ids) || (n_semantic && !semantic_ids))
        return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_UNKNOWN, [&] {
        std::vector<std::pair<size_t, float>> kw, sem;
        for (size_t i = 0; i < n_keyword; ++i) kw.emplace_back(keyword_ids[i], 0.0f);
        for (size_t i = 0; i < n_semantic; ++i) sem.emplace_back(semantic_ids[i], 0.0f);
        const auto r = hybrid_search(kw, sem, limit);
        for (size_t i = 0; i < r.size(); ++i) {
            ids_out[i] = r[i].first;
            scores_out[i] = r[i].second;
        }
        *n_out = r.size();
    });
}

// Retrieval over an on-disk index with a caller-supplied query embedding: the Searcher's path minus
// the encoder.  Keyword mode touches neither a model nor the GPU.
KJARNI_EXPORT KjarniErrorCode kjarni_hip_index_search(const char* index_path, const char* text_query,
                                                      const float* query_emb, size_t dim,
                                                      const KjarniSearchOptions* options, KjarniSearchResults* out)
{
    if (!index_path || !out) return KJARNI_ERROR_NULL_POINTER;
    out->results = nullptr;
    out->len = 0;
    if (!valid_utf8(index_path) || (text_query && !valid_utf8(text_query))) return KJARNI_ERROR_INVALID_UTF8;
    try {
        const ResolvedOptions o = resolve_options(options, KJARNI_SEARCH_HYBRID, 10, false);
        if (o.mode != KJARNI_SEARCH_SEMANTIC && !text_query) return KJARNI_ERROR_NULL_POINTER;
        if (o.mode != KJARNI_SEARCH_KEYWORD && !query_emb) return KJARNI_ERROR_NULL_POINTER;
        std::unique_ptr<IndexReader> reader = open_index(index_path);
        if (o.mode != KJARNI_SEARCH_KEYWORD && reader->dimension() != dim)
            throw InvalidConfig("Index dimension (" + std::to_string(reader->dimension()) +
                                ") doesn't match query dimension (" + std::to_string(dim) + ")");
        static std::mutex mu;
        static std::unique_ptr<SegmentScanner> scanner;
        const SegmentScanFn scan = [&](const std::vector<const Segment*>& segs, const float* q, size_t limit) {
            {
                std::lock_guard<std::mutex> lock(mu);
                if (!scanner) {
                    int n = 0;
                    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw GpuUnavailable("no usable HIP device");
                    scanner = std::make_unique<SegmentScanner>(0);
                }
            }
            return scanner->scan(segs, q, dim, limit);
        };
        std::vector<SearchHit> results =
            retrieve(*reader, o.mode, text_query ? text_query : "", query_emb, o.top_k, o.filter, scan);
        apply_threshold_and_limit(results, o);
        fill_results(results, out);
        return KJARNI_OK;
    } catch (const GpuUnavailable& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_GPU_UNAVAILABLE;
    } catch (const InvalidConfig& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_INVALID_CONFIG;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_INFERENCE_FAILED;
    } catch (...) {
        set_last_error("unknown error");
        return KJARNI_ERROR_INFERENCE_FAILED;
    }
}
