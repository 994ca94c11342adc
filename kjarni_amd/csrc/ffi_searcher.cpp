// kjarni_searcher_* (crates/kjarni-ffi/src/searcher.rs:12-491) on top of the GPU cosine scan.
//
// Searcher::search_with_options (crates/kjarni/src/searcher/model.rs:96-187) re-opens the index on
// every call; so does this.  What is cached across calls is the DEVICE copy of each segment's
// vectors.bin (keyed by path, validated by size + mtime), so the scan streams from HBM instead of
// re-crossing PCIe per query.  BM25, rank fusion, filters and result shaping are host logic
// (index.cpp), exactly the parts the reference also runs on the CPU.
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
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

// Where a query's time went (thread-local, the calling thread's last retrieval): kjarni_hip_search_breakdown().
struct SearchBreakdown {
    double open_us = 0, scan_us = 0, device_us = 0, hits_us = 0, total_us = 0;
};
thread_local SearchBreakdown t_breakdown;
inline double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

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

// Segment::search_vectors (kjarni-rag/src/segment.rs:307-337) for all the segments of an index at once.
//
// The reference scans segment after segment and merges (index_reader.rs:207-228: every segment's top `limit`, then one sort
// by score).  Segments are immutable once written, so the scanner keeps ONE device image per index: the segments' vectors.bin
// concatenated in segment order, [sum of doc_count, dim].  A query is then one fused scan + selection over that image
// (launch_cosine_search: a single pass, the scores never exist in memory) on a stream of the scanner's own -- two launches and
// two small copies per query, whatever the number of segments.  The global top `limit` with ties by ascending row equals the
// reference's merge of per-segment lists (a stable sort of segment-ordered, index-ordered lists), and a row maps back to
// (segment, local id) through the image's prefix offsets.
//
// An image is keyed by the uids of the parsed Segment objects it was built from (index.h: a changed file makes a new Segment,
// hence a new uid), so a query validates it by comparing a few integers -- no stat() here; the reader's own revalidation decides
// which Segment objects a query sees.  Queries on one scanner run concurrently: each leases a context (stream, workspace,
// pinned staging) for its launches; images are reference-counted, so an eviction never frees what a running scan reads.
class SegmentScanner {
public:
    explicit SegmentScanner(int device) : device_(device)
    {
        {
            std::lock_guard<std::mutex> lock(g_scanners_mu);
            g_scanners.push_back(this);
        }
        // The images may take half of the device's memory; past that, images no scan has touched longest are dropped
        // (a long-lived searcher that walks many indexes must not grow until hipMalloc fails).
        DeviceGuard guard;
        size_t free_b = 0, total_b = 0;
        if (hipSetDevice(device_) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0)
            budget_bytes_ = total_b / 2;
        else
            (void)hipGetLastError();
    }
    void set_budget(size_t bytes) { budget_bytes_ = bytes; }
    size_t cached_bytes() const
    {
        std::lock_guard<std::mutex> lock(mu_);
        size_t b = 0;
        for (const auto& im : images_) b += im->bytes;
        return b;
    }
    SegmentScanner(const SegmentScanner&) = delete;
    SegmentScanner& operator=(const SegmentScanner&) = delete;
    ~SegmentScanner()
    {
        {
            std::lock_guard<std::mutex> lock(g_scanners_mu);
            g_scanners.erase(std::remove(g_scanners.begin(), g_scanners.end(), this), g_scanners.end());
        }
        DeviceGuard guard;
        (void)hipSetDevice(device_);
        images_.clear();
        for (auto& c : contexts_) {
            if (c->stream) {
                (void)hipStreamSynchronize(c->stream);
                (void)hipStreamDestroy(c->stream);
            }
            if (c->work) (void)hipFree(c->work);
            if (c->io) (void)hipFree(c->io);
            if (c->pin) (void)hipHostFree(c->pin);
        }
    }

    // One query against every segment: ONE fused scan over the index's device image.
    std::vector<SegmentHits> scan(const std::vector<const Segment*>& segs, const float* query, size_t query_dim, size_t limit)
    {
        std::vector<SegmentHits> out(segs.size());
        if (limit == 0 || query_dim == 0) return out;
        float qn = 0.0f;
        for (size_t i = 0; i < query_dim; ++i) qn += query[i] * query[i];
        if (std::sqrt(qn) < 1e-9f) return out;                          // segment.rs:315-317

        // the segments a query of this width can match, in index order
        std::vector<size_t> use;
        std::vector<uint64_t> uids;
        for (size_t si = 0; si < segs.size(); ++si) {
            const Segment& seg = *segs[si];
            const size_t n = seg.doc_count(), dim = seg.dimension();
            if (n == 0 || dim != query_dim) continue;                        // query.len() != dimension -> empty
            if (seg.vectors_bytes() < n * dim * sizeof(float)) continue;     // get_embedding() would return None
            use.push_back(si);
            uids.push_back(seg.uid());
        }
        if (use.empty()) return out;

        DeviceGuard guard;  // (the caller's current device comes back)
        hip_check(hipSetDevice(device_), "hipSetDevice");
        const std::shared_ptr<Image> image = image_for(segs, use, uids, query_dim);
        const size_t total = image->offsets.back();
        if (total >= (size_t)0xFFFFFFFFu) throw std::runtime_error("index too large for one scan (2^32 rows)");
        const int k = (int)std::min(limit, total);

        ContextLease lease(*this);
        Context& c = lease.ctx();
        // (image->vectors and c.io come from hipMalloc: 256-byte aligned, so the one-query bound applies -- 4 MB per context on
        // the fused pass instead of 36 MB + 4 bytes per document)
        const size_t need = cosine_search_one_query_workspace_bytes((int64_t)total, (int)query_dim, k);
        if (need > c.work_bytes) {
            if (c.work) {
                hip_check(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
                (void)hipFree(c.work);
                c.work = nullptr;
                c.work_bytes = 0;
            }
            hip_check(hipMalloc((void**)&c.work, need), "hipMalloc(scan workspace)");
            c.work_bytes = need;
        }
        // staging: [query | idx[k] | score[k]] in one pinned buffer and one device buffer
        auto pad = [](size_t b) { return 256 * ((b + 255) / 256); };
        const size_t q_bytes = pad(query_dim * 4), i_off = q_bytes, s_off = i_off + pad((size_t)k * 8), io_bytes = s_off + pad((size_t)k * 4);
        if (io_bytes > c.io_bytes) {
            if (c.io) {
                hip_check(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
                (void)hipFree(c.io);
                (void)hipHostFree(c.pin);
                c.io = nullptr;
                c.pin = nullptr;
                c.io_bytes = 0;
            }
            hip_check(hipMalloc((void**)&c.io, io_bytes), "hipMalloc(scan staging)");
            hip_check(hipHostMalloc((void**)&c.pin, io_bytes, hipHostMallocDefault), "hipHostMalloc(scan staging)");
            c.io_bytes = io_bytes;
        }
        const double t0 = now_us();
        std::memcpy(c.pin, query, query_dim * 4);
        hip_check(hipMemcpyAsync(c.io, c.pin, query_dim * 4, hipMemcpyHostToDevice, c.stream), "H2D query");
        hip_check(launch_cosine_search(reinterpret_cast<const float*>(c.io), 1, image->vectors, (int64_t)total, (int)query_dim,
                                       /*segment mode*/ 1, k, c.work, reinterpret_cast<int64_t*>(c.io + i_off),
                                       reinterpret_cast<float*>(c.io + s_off), c.stream),
                  "cosine_search");
        hip_check(hipMemcpyAsync(c.pin + i_off, c.io + i_off, (s_off - i_off) + (size_t)k * 4, hipMemcpyDeviceToHost, c.stream), "D2H hits");
        hip_check(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
        t_breakdown.device_us += now_us() - t0;
        const int64_t* idx = reinterpret_cast<const int64_t*>(c.pin + i_off);
        const float* sc = reinterpret_cast<const float*>(c.pin + s_off);
        for (int i = 0; i < k; ++i) {
            if (idx[i] < 0) continue;
            const size_t row = (size_t)idx[i];
            const size_t u = (size_t)(std::upper_bound(image->offsets.begin(), image->offsets.end(), row) - image->offsets.begin()) - 1;
            out[use[u]].emplace_back(row - image->offsets[u], sc[i]);
        }
        return out;
    }

    // Drops the images that hold a segment under `root` (the index there is being deleted or rebuilt).
    void forget_under(const std::string& root)
    {
        const std::string prefix = root.empty() || root.back() == '/' ? root : root + "/";
        std::lock_guard<std::mutex> lock(mu_);
        DeviceGuard guard;
        (void)hipSetDevice(device_);
        for (auto it = images_.begin(); it != images_.end();) {
            bool hit = false;
            for (const std::string& d : (*it)->dirs) hit = hit || d.compare(0, prefix.size(), prefix) == 0;
            it = hit ? images_.erase(it) : std::next(it);  // (freed when the last running scan lets go of it)
        }
    }

private:
    struct Image {
        std::vector<uint64_t> uids;    // Segment::uid() of the segments, in order
        std::vector<std::string> dirs;
        std::vector<size_t> offsets;   // first row of segment i; back() = rows in all
        float* vectors = nullptr;
        size_t bytes = 0, dim = 0;
        uint64_t last_used = 0;
        bool ready = false, failed = false;  // (under the scanner's mu_) uploaded / its upload threw
        ~Image()
        {
            if (vectors) (void)hipFree(vectors);  // (runs on a thread whose current device is the scanner's)
        }
    };
    struct Context {
        hipStream_t stream = nullptr;
        uint8_t *work = nullptr, *io = nullptr, *pin = nullptr;
        size_t work_bytes = 0, io_bytes = 0;
        bool busy = false;
    };
    static constexpr int kMaxContexts = 8;
    class ContextLease {
    public:
        explicit ContextLease(SegmentScanner& s) : s_(s), c_(nullptr)
        {
            std::unique_lock<std::mutex> lock(s.mu_);
            for (;;) {
                for (auto& c : s.contexts_)
                    if (!c->busy) c_ = c.get();
                if (!c_ && (int)s.contexts_.size() < kMaxContexts) {
                    s.contexts_.push_back(std::make_unique<Context>());
                    c_ = s.contexts_.back().get();
                }
                if (c_) break;
                s.ctx_cv_.wait(lock);
            }
            c_->busy = true;
            lock.unlock();
            if (!c_->stream && hipStreamCreateWithFlags(&c_->stream, hipStreamNonBlocking) != hipSuccess) {
                (void)hipGetLastError();
                release();
                throw std::runtime_error("hipStreamCreate failed");
            }
        }
        ~ContextLease() { release(); }
        Context& ctx() { return *c_; }

    private:
        void release()
        {
            if (!c_) return;
            std::lock_guard<std::mutex> lock(s_.mu_);
            c_->busy = false;
            c_ = nullptr;
            s_.ctx_cv_.notify_one();
        }
        SegmentScanner& s_;
        Context* c_;
    };

    // The image of this segment list: the resident one, or a new upload.  The upload (hipMalloc + a synchronous copy of the
    // whole index) runs OUTSIDE mu_: the new image is registered first as a placeholder (`ready` false) that later queries
    // against the SAME index wait on, while queries against images that are already resident keep being served.
    std::shared_ptr<Image> image_for(const std::vector<const Segment*>& segs, const std::vector<size_t>& use,
                                     const std::vector<uint64_t>& uids, size_t dim)
    {
        std::shared_ptr<Image> im;
        {
            std::unique_lock<std::mutex> lock(mu_);
            const uint64_t tick = ++tick_;
            for (auto& x : images_)
                if (x->dim == dim && x->uids == uids) {
                    x->last_used = tick;
                    std::shared_ptr<Image> found = x;
                    image_cv_.wait(lock, [&] { return found->ready || found->failed; });
                    if (found->failed) break;  // its upload failed in another thread: try again below
                    return found;
                }
            im = std::make_shared<Image>();
            im->uids = uids;
            im->dim = dim;
            im->offsets.push_back(0);
            for (size_t u : use) {
                im->dirs.push_back(segs[u]->dir());
                im->offsets.push_back(im->offsets.back() + segs[u]->doc_count());
            }
            im->bytes = im->offsets.back() * dim * sizeof(float);
            // a superseded image of the same index (a segment was added: the old list is a prefix or shares directories) goes
            // first, then least-recently-used ones until the new image fits the budget (images still uploading are kept)
            for (auto it = images_.begin(); it != images_.end();) {
                bool same_index = (*it)->failed;
                for (const std::string& d : (*it)->dirs) same_index = same_index || d == im->dirs.front();
                it = (same_index && ((*it)->ready || (*it)->failed)) ? images_.erase(it) : std::next(it);
            }
            while (budget_bytes_ && !images_.empty()) {
                size_t held = 0;
                for (const auto& x : images_) held += x->bytes;
                if (held + im->bytes <= budget_bytes_) break;
                auto victim = images_.end();
                for (auto it = images_.begin(); it != images_.end(); ++it)
                    if ((*it)->ready && (victim == images_.end() || (*it)->last_used < (*victim)->last_used)) victim = it;
                if (victim == images_.end()) break;
                images_.erase(victim);
            }
            im->last_used = tick;
            images_.push_back(im);
        }
        try {
            hip_check(hipMalloc((void**)&im->vectors, im->bytes ? im->bytes : 4), "hipMalloc(index image)");
            for (size_t i = 0; i < use.size(); ++i) {
                const Segment& seg = *segs[use[i]];
                hip_check(hipMemcpy(im->vectors + im->offsets[i] * dim, seg.vectors(), seg.doc_count() * dim * sizeof(float),
                                    hipMemcpyHostToDevice),
                          "H2D segment vectors");
            }
        } catch (...) {
            std::lock_guard<std::mutex> lock(mu_);
            im->failed = true;
            for (auto it = images_.begin(); it != images_.end(); ++it)
                if (*it == im) {
                    images_.erase(it);
                    break;
                }
            image_cv_.notify_all();
            throw;
        }
        {
            std::lock_guard<std::mutex> lock(mu_);
            im->ready = true;
        }
        image_cv_.notify_all();
        return im;
    }

    int device_;
    mutable std::mutex mu_;
    std::condition_variable ctx_cv_;
    std::condition_variable image_cv_;  // an image's upload finished (or failed)
    std::vector<std::shared_ptr<Image>> images_;
    std::vector<std::unique_ptr<Context>> contexts_;
    size_t budget_bytes_ = 0;
    uint64_t tick_ = 0;
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
        t_breakdown = SearchBreakdown();
        const double t_begin = now_us();
        std::unique_ptr<IndexReader> reader = open_index(index_path);
        t_breakdown.open_us = now_us() - t_begin;
        const size_t model_dim = (size_t)s->embedder->config().hidden;
        if (reader->dimension() != model_dim)
            throw InvalidConfig("Index dimension (" + std::to_string(reader->dimension()) +
                                ") doesn't match model dimension (" + std::to_string(model_dim) + ")");
        const bool rerank = o.use_reranker && s->reranker;
        const size_t fetch_k = rerank ? o.top_k * 5 : o.top_k;
        const SegmentScanFn scan = [&](const std::vector<const Segment*>& segs, const float* q, size_t limit) {
            const double t0 = now_us();
            std::vector<SegmentHits> hits = s->scanner->scan(segs, q, model_dim, limit);
            t_breakdown.scan_us += now_us() - t0;
            return hits;
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
        t_breakdown.total_us = now_us() - t_begin;
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
        t_breakdown = SearchBreakdown();
        const double t_begin = now_us();
        std::unique_ptr<IndexReader> reader = open_index(index_path);
        t_breakdown.open_us = now_us() - t_begin;
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
            const double t0 = now_us();
            std::vector<SegmentHits> hits = scanner->scan(segs, q, dim, limit);
            t_breakdown.scan_us += now_us() - t0;
            return hits;
        };
        std::vector<SearchHit> results =
            retrieve(*reader, o.mode, text_query ? text_query : "", query_emb, o.top_k, o.filter, scan);
        apply_threshold_and_limit(results, o);
        fill_results(results, out);
        t_breakdown.total_us = now_us() - t_begin;
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

// Where the calling thread's last kjarni_searcher_search* / kjarni_hip_index_search call spent its time, microseconds:
// out[0] re-opening the index (revalidation of the parsed segments), out[1] the scan callback (image lookup, copies, launches,
// wait), out[2] of it between the query's H2D copy and the hits' arrival, out[3] the whole call -- the rest is host work: BM25,
// rank fusion, reading the hits' documents and metadata back, (for a Searcher) tokenising and embedding the query.
KJARNI_EXPORT void kjarni_hip_set_keyword_parallel_min_docs(size_t docs) { set_keyword_parallel_min_docs(docs); }

KJARNI_EXPORT void kjarni_hip_search_breakdown(double* out, size_t n)
{
    const double v[4] = {t_breakdown.open_us, t_breakdown.scan_us, t_breakdown.device_us, t_breakdown.total_us};
    for (size_t i = 0; i < n && i < 4; ++i) out[i] = v[i];
}
