// kjarni_indexer_* / kjarni_index_* (crates/kjarni-ffi/src/indexer.rs:14-675) and the cancel tokens
// of crates/kjarni-ffi/src/callback.rs:46-101.
//
// Indexer::create / add (crates/kjarni/src/indexer/model.rs:160-726) walk the inputs, chunk every
// file, embed the chunks `batch_size` at a time and stream them into segments.  Here the walk, the
// progress events and the cancellation points are the reference's, one for one; the embedding work
// behind them is coalesced: the chunks of several reference-sized batches are encoded in one pass of
// the GPU encoder (sorted by length so a short chunk is not padded to the longest) and reach the
// segment writer in their original order.  The files on disk are the ones the reference writes.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <memory>
#include <numeric>

#include "host_util.h"
#include "ffi_common.h"
#include "index.h"
#include "index_writer.h"
#include "json.h"
#include "pipeline.h"
#include "unicode.h"

using namespace kjarni;

// callback.rs:46-101
struct KjarniCancelToken {
    std::atomic<bool> flag{false};
};

KJARNI_EXPORT KjarniCancelToken* kjarni_cancel_token_new(void) { return new (std::nothrow) KjarniCancelToken(); }
KJARNI_EXPORT void kjarni_cancel_token_cancel(KjarniCancelToken* t)
{
    if (t) t->flag.store(true, std::memory_order_seq_cst);
}
KJARNI_EXPORT bool kjarni_cancel_token_is_cancelled(const KjarniCancelToken* t)
{
    return t ? t->flag.load(std::memory_order_seq_cst) : false;
}
KJARNI_EXPORT void kjarni_cancel_token_reset(KjarniCancelToken* t)
{
    if (t) t->flag.store(false, std::memory_order_seq_cst);
}
KJARNI_EXPORT void kjarni_cancel_token_free(KjarniCancelToken* t) { delete t; }

namespace {

std::string trim(const std::string& s)  // str::trim (ASCII + Unicode White_Space; the lists here are ASCII)
{
    size_t a = 0, b = s.size();
    while (a < b && std::strchr(" \t\n\r\v\f", s[a])) ++a;
    while (b > a && std::strchr(" \t\n\r\v\f", s[b - 1])) --b;
    return s.substr(a, b - a);
}

std::vector<std::string> split_commas(const char* s)
{
    std::vector<std::string> out;
    const std::string str(s);
    size_t pos = 0;
    for (;;) {
        const size_t hit = str.find(',', pos);
        out.push_back(trim(str.substr(pos, hit == std::string::npos ? std::string::npos : hit - pos)));
        if (hit == std::string::npos) break;
        pos = hit + 1;
    }
    return out;
}

std::string lower(const std::string& s)
{
    std::vector<uint32_t> cps, low;
    if (!unicode::decode_utf8(s.data(), s.size(), cps)) return s;
    unicode::lowercase_str(cps, low);
    std::string o;
    for (uint32_t cp : low) unicode::append_utf8(o, cp);
    return o;
}

// IndexerError (crates/kjarni/src/indexer/types.rs:42-67) -> code (indexer.rs:268-279)
struct IndexerFailure : std::runtime_error {
    KjarniErrorCode code;
    IndexerFailure(KjarniErrorCode c, const std::string& m) : std::runtime_error(m), code(c) {}
};
IndexerFailure cancelled() { return {KJARNI_ERROR_CANCELLED, "Operation cancelled"}; }
IndexerFailure indexing_failed(const std::string& m) { return {KJARNI_ERROR_INFERENCE_FAILED, "Indexing failed: " + m}; }

size_t device_batch_chunks()
{
    if (const char* e = std::getenv("KJARNI_HIP_INDEX_DEVICE_BATCH")) {
        const long v = std::atol(e);
        if (v > 0) return (size_t)v;
    }
    return 2048;
}

// KJARNI_HIP_INDEX_TIMING=1: where a run's wall time goes, per stage, on stderr (measurements)
struct StageTimes {
    std::atomic<int64_t> load_us{0}, blocked_us{0}, device_us{0}, write_us{0}, commit_us{0}, tok_wait_us{0}, embed_us{0};
};
inline int64_t now_us()
{
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline bool index_timing()
{
    static const bool on = std::getenv("KJARNI_HIP_INDEX_TIMING") != nullptr;
    return on;
}

using ReportFn = std::function<void(KjarniProgressStage, size_t, size_t, const char*)>;
using CancelFn = std::function<bool()>;

}  // namespace

struct KjarniIndexer {
    std::unique_ptr<Pipeline> embedder;
    LoaderConfig loader;
    size_t chunk_size = 512, chunk_overlap = 50, batch_size = 32, max_docs_per_segment = 10000;
    bool quiet = false;

    // Embeds what the reference's flush_batch calls have handed over so far and streams it to the writer.
    struct Sink {
        KjarniIndexer& ix;
        IndexWriter& writer;
        size_t device_batch;
        std::vector<std::string> texts;
        std::vector<Metadata> metas;

        size_t accept(std::vector<std::string>& batch_texts, std::vector<Metadata>& batch_metas)
        {
            const size_t count = batch_texts.size();
            for (size_t i = 0; i < count; ++i) {
                texts.push_back(std::move(batch_texts[i]));
                metas.push_back(std::move(batch_metas[i]));
            }
            batch_texts.clear();
            batch_metas.clear();
            if (texts.size() >= device_batch) drain();
            return count;
        }

        // Three stages: the caller keeps loading and splitting files; one device batch is in flight on a worker thread (tokenise ->
        // GPU; inside the batch the next length group is tokenised while the GPU works on the current one); the batch before it is
        // being written (IndexWriter::add: the BM25 postings, the segment files) on a third thread -- the writer is the slowest host
        // stage per chunk, and serial after the GPU stage it left the device idle half of the time.  Batches are written in order;
        // at most one waits behind the one being written.
        StageTimes* times = nullptr;
        std::future<void> pending;  // the device stage of the newest batch
        std::future<void> writing;  // the write stage of the batch before it (owned by the device stage's thread while that runs)

        void wait()
        {
            const int64_t t0 = now_us();
            if (pending.valid()) pending.get();  // rethrows what the batch threw
            if (times) times->blocked_us += now_us() - t0;
        }

        void drain()
        {
            wait();
            if (texts.empty()) return;
            auto job_texts = std::make_shared<std::vector<std::string>>(std::move(texts));
            auto job_metas = std::make_shared<std::vector<Metadata>>(std::move(metas));
            texts.clear();
            metas.clear();
            pending = std::async(std::launch::async, [this, job_texts, job_metas] { process(job_texts, job_metas); });
        }

        void finish()  // everything handed over so far is on disk when this returns
        {
            drain();
            wait();
            if (writing.valid()) writing.get();
        }

        ~Sink()
        {
            try {
                wait();
            } catch (...) {
            }
            try {
                if (writing.valid()) writing.get();
            } catch (...) {
            }
        }

        void process(std::shared_ptr<std::vector<std::string>> texts_p, std::shared_ptr<std::vector<Metadata>> metas_p)
        {
            std::vector<std::string>& texts = *texts_p;
            const size_t n = texts.size();
            if (n == 0) return;
            const int64_t t_dev = now_us();
            const size_t H = (size_t)ix.embedder->config().hidden;
            auto emb_p = std::make_shared<std::vector<float>>(n * H);
            std::vector<float>& emb = *emb_p;
            // length-sorted groups: every group is padded to ITS longest member only
            std::vector<size_t> order(n);
            std::iota(order.begin(), order.end(), (size_t)0);
            std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return texts[a].size() < texts[b].size(); });
            const size_t group = 256;
            auto tokenize = [&](size_t s) {
                std::vector<std::string> g;
                for (size_t i = s, e = std::min(n, s + group); i < e; ++i) g.push_back(texts[order[i]]);
                return ix.embedder->tokenizer.encode_batch(g);
            };
            std::future<BatchEncoding> next = std::async(std::launch::async, tokenize, (size_t)0);
            for (size_t s = 0; s < n; s += group) {
                const size_t e = std::min(n, s + group);
                std::vector<float> out;
                try {
                    const int64_t t0 = now_us();
                    const BatchEncoding be = next.get();
                    const int64_t t1 = now_us();
                    if (e < n) next = std::async(std::launch::async, tokenize, e);
                    // Embedder::embed_batch: mean pool, L2-normalised (crates/kjarni/src/embedder/model.rs:142-160)
                    out = embed_encoding(*ix.embedder, be, POOL_MEAN, true);
                    if (times) {
                        times->tok_wait_us += t1 - t0;
                        times->embed_us += now_us() - t1;
                    }
                } catch (const std::exception& ex) {
                    if (next.valid()) next.wait();
                    throw IndexerFailure(KJARNI_ERROR_INFERENCE_FAILED, std::string("Failed to load embedder: ") + ex.what());
                }
                for (size_t i = s; i < e; ++i) std::memcpy(&emb[order[i] * H], &out[(i - s) * H], H * sizeof(float));
            }
            if (times) times->device_us += now_us() - t_dev;
            if (writing.valid()) writing.get();  // (in order, and no more than one batch waiting to be written)
            IndexWriter* w = &writer;
            StageTimes* tm = times;
            writing = std::async(std::launch::async, [w, tm, texts_p, metas_p, emb_p, n, H] {
                const int64_t t0 = now_us();
                try {
                    for (size_t i = 0; i < n; ++i) w->add((*texts_p)[i], &(*emb_p)[i * H], H, &(*metas_p)[i]);
                } catch (const std::exception& ex) {
                    throw indexing_failed(ex.what());
                }
                if (tm) tm->write_us += now_us() - t0;
            });
        }
    };

    // The shared body of create_impl / add_impl (model.rs:318-484, 629-726).
    void run(IndexWriter& writer, const std::vector<std::string>& inputs, const ReportFn& report, const CancelFn& is_cancelled,
             const char* commit_msg, size_t& total_docs, size_t& total_chunks, size_t& files_processed, size_t& files_skipped)
    {
        report(KJARNI_PROGRESS_SCANNING, 0, 0, "Discovering files...");
        std::vector<std::string> files;
        try {
            files = collect_files(loader, inputs);
        } catch (const PathNotFound& e) {
            throw IndexerFailure(KJARNI_ERROR_MODEL_NOT_FOUND, std::string("Path not found: ") + e.what());
        }
        const size_t total_files = files.size();
        if (is_cancelled()) throw cancelled();

        const DocumentLoader doc_loader(loader);
        StageTimes times;
        Sink sink{*this, writer, std::max(device_batch_chunks(), std::max<size_t>(batch_size, 1)), {}, {}, index_timing() ? &times : nullptr, {}, {}};
        std::vector<std::string> batch_texts;
        std::vector<Metadata> batch_metas;

        for (size_t file_idx = 0; file_idx < total_files; ++file_idx) {
            if (is_cancelled()) throw cancelled();
            report(KJARNI_PROGRESS_LOADING, file_idx, total_files, files[file_idx].c_str());
            std::vector<Chunk> chunks;
            const int64_t t_load = now_us();
            try {
                chunks = doc_loader.load_file(files[file_idx]);
            } catch (const std::exception& e) {
                ++files_skipped;
                if (!quiet) std::fprintf(stderr, "Warning: Failed to load %s: %s\n", files[file_idx].c_str(), e.what());
                continue;
            }
            times.load_us += now_us() - t_load;
            total_chunks += chunks.size();
            ++files_processed;
            for (Chunk& c : chunks) {
                batch_texts.push_back(std::move(c.text));
                batch_metas.push_back(std::move(c.metadata));
                if (batch_texts.size() >= batch_size) {
                    if (is_cancelled()) throw cancelled();
                    report(KJARNI_PROGRESS_EMBEDDING, total_docs, 0, nullptr);
                    total_docs += sink.accept(batch_texts, batch_metas);
                }
            }
        }
        if (!batch_texts.empty()) {
            report(KJARNI_PROGRESS_EMBEDDING, total_docs, 0, nullptr);
            total_docs += sink.accept(batch_texts, batch_metas);
        }
        sink.finish();
        report(KJARNI_PROGRESS_COMMITTING, total_docs, total_docs, commit_msg);
        const int64_t t_commit = now_us();
        try {
            writer.commit();
        } catch (const std::exception& e) {
            throw indexing_failed(e.what());
        }
        times.commit_us += now_us() - t_commit;
        if (index_timing())
            std::fprintf(stderr, "indexer stages (ms): load + split %.1f | caller blocked on the device stage %.1f | device stage %.1f (waiting for the tokeniser %.1f, embedding %.1f) | write stage %.1f | commit %.1f\n",
                         times.load_us / 1e3, times.blocked_us / 1e3, times.device_us / 1e3, times.tok_wait_us / 1e3, times.embed_us / 1e3, times.write_us / 1e3,
                         times.commit_us / 1e3);
    }

    KjarniIndexStats create(const std::string& index_path, const std::vector<std::string>& inputs, bool force,
                            const ReportFn& report, const CancelFn& is_cancelled)
    {
        if (inputs.empty()) throw IndexerFailure(KJARNI_ERROR_INVALID_CONFIG, "No input paths specified");
        if (path_exists(index_path)) {
            if (!force)
                throw IndexerFailure(KJARNI_ERROR_INVALID_CONFIG,
                                     "Index already exists at " + index_path + ". Use force=true to overwrite.");
            try {
                remove_dir_all(index_path);
            } catch (const std::exception& e) {
                throw indexing_failed(e.what());
            }
        }
        const auto start = std::chrono::steady_clock::now();
        const size_t dimension = (size_t)embedder->config().hidden;
        IndexConfig config;
        config.dimension = dimension;
        config.max_docs_per_segment = max_docs_per_segment;
        config.has_embedding_model = true;
        config.embedding_model = embedder->model_name;
        std::unique_ptr<IndexWriter> writer;
        try {
            writer = IndexWriter::open(index_path, config);
        } catch (const std::exception& e) {
            throw indexing_failed(e.what());
        }
        KjarniIndexStats st;
        std::memset(&st, 0, sizeof st);
        run(*writer, inputs, report, is_cancelled, "Finalizing index...", st.documents_indexed, st.chunks_created,
            st.files_processed, st.files_skipped);
        st.dimension = dimension;
        st.size_bytes = directory_size(index_path);
        st.elapsed_ms = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start).count();
        return st;
    }

    size_t add(const std::string& index_path, const std::vector<std::string>& inputs, const ReportFn& report,
               const CancelFn& is_cancelled)
    {
        if (inputs.empty()) return 0;
        if (!path_exists(index_path)) throw IndexerFailure(KJARNI_ERROR_MODEL_NOT_FOUND, "Index not found at " + index_path);
        std::unique_ptr<IndexWriter> writer;
        try {
            writer = IndexWriter::open_existing(index_path);
        } catch (const std::exception& e) {
            throw indexing_failed(e.what());
        }
        const size_t model_dim = (size_t)embedder->config().hidden;
        if (writer->dimension() != model_dim)
            throw IndexerFailure(KJARNI_ERROR_INVALID_CONFIG, "Dimension mismatch: index has " + std::to_string(writer->dimension()) +
                                                                  ", model produces " + std::to_string(model_dim));
        size_t total_docs = 0, chunks = 0, processed = 0, skipped = 0;
        run(*writer, inputs, report, is_cancelled, "Finalizing...", total_docs, chunks, processed, skipped);
        return total_docs;
    }
};

namespace {

template <class F>
KjarniErrorCode indexer_guarded(F&& fn)
{
    try {
        fn();
        return KJARNI_OK;
    } catch (const IndexerFailure& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::exception& e) {
        set_last_error(std::string("Indexing failed: ") + e.what());
        return KJARNI_ERROR_INFERENCE_FAILED;
    } catch (...) {
        set_last_error("unknown error");
        return KJARNI_ERROR_INFERENCE_FAILED;
    }
}

KjarniErrorCode parse_inputs(const char* const* inputs, size_t n, std::vector<std::string>& out)
{
    out.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        if (!inputs[i]) return KJARNI_ERROR_NULL_POINTER;
        if (!valid_utf8(inputs[i])) return KJARNI_ERROR_INVALID_UTF8;
        out.emplace_back(inputs[i]);
    }
    return KJARNI_OK;
}

ReportFn make_report(KjarniProgressCallbackFn cb, void* user_data)
{
    return [cb, user_data](KjarniProgressStage stage, size_t current, size_t total, const char* msg) {
        if (!cb) return;
        KjarniProgress p;
        p.stage = stage;
        p.current = current;
        p.total = total;
        p.message = msg;
        cb(p, user_data);
    };
}

}  // namespace

KJARNI_EXPORT KjarniIndexerConfig kjarni_indexer_config_default(void)
{
    KjarniIndexerConfig c;
    std::memset(&c, 0, sizeof c);
    c.device = KJARNI_DEVICE_CPU;
    c.chunk_size = 512;
    c.chunk_overlap = 50;
    c.batch_size = 32;
    c.recursive = 1;
    c.include_hidden = 0;
    c.max_file_size = 10u * 1024 * 1024;
    c.quiet = 0;
    return c;
}

KJARNI_EXPORT KjarniErrorCode kjarni_indexer_new(const KjarniIndexerConfig* config, KjarniIndexer** out)
{
    if (!out) return KJARNI_ERROR_NULL_POINTER;
    const KjarniIndexerConfig dflt = kjarni_indexer_config_default();
    const KjarniIndexerConfig& c = config ? *config : dflt;
    for (const char* s : {c.model_name, c.cache_dir, c.extensions, c.exclude_patterns})
        if (s && !valid_utf8(s)) return KJARNI_ERROR_INVALID_UTF8;
    try {
        auto h = std::make_unique<KjarniIndexer>();
        h->chunk_size = c.chunk_size;
        h->chunk_overlap = c.chunk_overlap;
        h->batch_size = c.batch_size;
        h->quiet = c.quiet != 0;
        h->loader.splitter.chunk_size = c.chunk_size;
        h->loader.splitter.chunk_overlap = c.chunk_overlap;
        h->loader.recursive = c.recursive != 0;
        h->loader.include_hidden = c.include_hidden != 0;
        h->loader.has_max_file_size = true;  // builder default Some(10 MiB); 0 keeps the default (indexer.rs:221-223)
        h->loader.max_file_size = c.max_file_size > 0 ? c.max_file_size : 10u * 1024 * 1024;
        if (c.extensions)
            for (const std::string& e : split_commas(c.extensions)) {
                std::string x = lower(e);  // builder.rs:96-101: lowercase, strip leading dots
                x.erase(0, x.find_first_not_of('.'));
                h->loader.extensions.push_back(x);
            }
        if (c.exclude_patterns)
            for (const std::string& p : split_commas(c.exclude_patterns)) h->loader.exclude_patterns.push_back(p);
        // The reference builds the TextSplitter inside create()/add(), where an invalid chunking
        // configuration PANICS (splitter.rs:51-56; abort in release builds).  Reject it here instead.
        if (const char* e = h->loader.splitter.validate()) throw InvalidConfig(std::string("Invalid SplitterConfig: ") + e);
        h->embedder = load_pipeline(c.cache_dir, c.model_name, nullptr, "minilm-l6-v2", Want::Embedding);
        *out = h.release();
        return KJARNI_OK;
    } catch (const GpuUnavailable& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_GPU_UNAVAILABLE;
    } catch (const InvalidConfig& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_INVALID_CONFIG;
    } catch (const std::exception& e) {
        set_last_error(std::string("Failed to load embedder: ") + e.what());
        return KJARNI_ERROR_LOAD_FAILED;  // indexer.rs:227-230
    } catch (...) {
        set_last_error("unknown error");
        return KJARNI_ERROR_LOAD_FAILED;
    }
}

KJARNI_EXPORT void kjarni_indexer_free(KjarniIndexer* indexer) { delete indexer; }

KJARNI_EXPORT KjarniErrorCode kjarni_indexer_create_with_callback(KjarniIndexer* indexer, const char* index_path,
                                                                  const char* const* inputs, size_t num_inputs, int32_t force,
                                                                  KjarniProgressCallbackFn progress_callback, void* user_data,
                                                                  const KjarniCancelToken* cancel_token, KjarniIndexStats* out)
{
    if (!indexer || !index_path || !inputs || !out) return KJARNI_ERROR_NULL_POINTER;
    if (!valid_utf8(index_path)) return KJARNI_ERROR_INVALID_UTF8;
    std::vector<std::string> in;
    if (const KjarniErrorCode rc = parse_inputs(inputs, num_inputs, in)) return rc;
    return indexer_guarded([&] {
        *out = indexer->create(index_path, in, force != 0, make_report(progress_callback, user_data),
                               [cancel_token] { return kjarni_cancel_token_is_cancelled(cancel_token); });
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_indexer_create(KjarniIndexer* indexer, const char* index_path, const char* const* inputs,
                                                    size_t num_inputs, int32_t force, KjarniIndexStats* out)
{
    return kjarni_indexer_create_with_callback(indexer, index_path, inputs, num_inputs, force, nullptr, nullptr, nullptr, out);
}

KJARNI_EXPORT KjarniErrorCode kjarni_indexer_add_with_callback(KjarniIndexer* indexer, const char* index_path,
                                                               const char* const* inputs, size_t num_inputs,
                                                               KjarniProgressCallbackFn progress_callback, void* user_data,
                                                               const KjarniCancelToken* cancel_token, size_t* documents_added)
{
    if (!indexer || !index_path || !inputs || !documents_added) return KJARNI_ERROR_NULL_POINTER;
    if (num_inputs == 0) {
        *documents_added = 0;
        return KJARNI_OK;
    }
    if (!valid_utf8(index_path)) return KJARNI_ERROR_INVALID_UTF8;
    std::vector<std::string> in;
    if (const KjarniErrorCode rc = parse_inputs(inputs, num_inputs, in)) return rc;
    *documents_added = 0;
    return indexer_guarded([&] {
        *documents_added = indexer->add(index_path, in, make_report(progress_callback, user_data),
                                        [cancel_token] { return kjarni_cancel_token_is_cancelled(cancel_token); });
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_indexer_add(KjarniIndexer* indexer, const char* index_path, const char* const* inputs,
                                                 size_t num_inputs, size_t* documents_added)
{
    return kjarni_indexer_add_with_callback(indexer, index_path, inputs, num_inputs, nullptr, nullptr, nullptr, documents_added);
}

// Indexer::info (model.rs:95-127): no model involved.
KJARNI_EXPORT KjarniErrorCode kjarni_index_info(const char* index_path, KjarniIndexInfo* out)
{
    if (!index_path || !out) return KJARNI_ERROR_NULL_POINTER;
    if (!valid_utf8(index_path)) return KJARNI_ERROR_INVALID_UTF8;
    try {
        const std::string path(index_path);
        if (!path_exists(path)) throw std::runtime_error("Index not found at " + path);
        std::unique_ptr<IndexReader> reader;
        try {
            reader = IndexReader::open(path);
        } catch (const std::exception& e) {
            throw std::runtime_error(std::string("Indexing failed: ") + e.what());
        }
        std::string model;
        bool has_model = false;
        try {
            const IndexConfig cfg = IndexConfig::from_json([&] {
                FILE* f = std::fopen((path + "/config.json").c_str(), "rb");
                if (!f) throw std::runtime_error("no config.json");
                std::string s;
                char buf[4096];
                size_t n;
                while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
                std::fclose(f);
                return s;
            }());
            has_model = cfg.has_embedding_model;
            model = cfg.embedding_model;
        } catch (const std::exception&) {
            has_model = false;  // a config that IndexConfig cannot parse just yields None
        }
        KjarniIndexInfo info;
        std::memset(&info, 0, sizeof info);
        info.path = dup_cstr(path);
        info.document_count = reader->len();
        info.segment_count = reader->segment_count();
        info.dimension = reader->dimension();
        info.size_bytes = directory_size(path);
        info.embedding_model = (has_model && model.find('\0') == std::string::npos) ? dup_cstr(model) : nullptr;
        *out = info;
        return KJARNI_OK;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_MODEL_NOT_FOUND;  // indexer.rs:597-600: every info error
    } catch (...) {
        set_last_error("unknown error");
        return KJARNI_ERROR_MODEL_NOT_FOUND;
    }
}

// By VALUE, as the Rust source declares it (indexer.rs:85-93).
KJARNI_EXPORT void kjarni_index_info_free(KjarniIndexInfo info)
{
    std::free(info.path);
    std::free(info.embedding_model);
}

// Indexer::delete (model.rs:130-135)
KJARNI_EXPORT KjarniErrorCode kjarni_index_delete(const char* index_path)
{
    if (!index_path) return KJARNI_ERROR_NULL_POINTER;
    if (!valid_utf8(index_path)) return KJARNI_ERROR_INVALID_UTF8;
    try {
        if (!path_exists(index_path)) throw std::runtime_error(std::string("Index not found at ") + index_path);
        forget_device_segments_under(index_path);  // HBM copies of its vectors.bin in live Searchers
        try {
            remove_dir_all(index_path);
        } catch (const std::exception& e) {
            throw std::runtime_error(std::string("Indexing failed: ") + e.what());
        }
        return KJARNI_OK;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_INFERENCE_FAILED;  // indexer.rs:617-620
    }
}

KJARNI_EXPORT size_t kjarni_indexer_model_name(const KjarniIndexer* indexer, char* buf, size_t buf_len)
{
    if (!indexer) return 0;
    const std::string& name = indexer->embedder->model_name;
    const size_t required = name.size() + 1;
    if (!buf || buf_len == 0) return required;
    const size_t n = std::min(name.size(), buf_len - 1);
    std::memcpy(buf, name.data(), n);
    buf[n] = '\0';
    return required;
}

KJARNI_EXPORT size_t kjarni_indexer_dimension(const KjarniIndexer* indexer)
{
    return indexer ? (size_t)indexer->embedder->config().hidden : 0;
}

KJARNI_EXPORT size_t kjarni_indexer_chunk_size(const KjarniIndexer* indexer) { return indexer ? indexer->chunk_size : 0; }

// ---- host-side pieces of the indexing pipeline, one at a time (kjarni_hip.h; parity tests) ---------

KJARNI_EXPORT KjarniErrorCode kjarni_text_split(const char* text, size_t chunk_size, size_t chunk_overlap,
                                                const char* separator, KjarniStringArray* out)
{
    if (!text || !out) return KJARNI_ERROR_NULL_POINTER;
    out->strings = nullptr;
    out->len = 0;
    if (!valid_utf8(text) || (separator && !valid_utf8(separator))) return KJARNI_ERROR_INVALID_UTF8;
    return guarded(KJARNI_ERROR_UNKNOWN, [&] {
        SplitterConfig sc;
        sc.chunk_size = chunk_size;
        sc.chunk_overlap = chunk_overlap;
        if (separator) sc.separator = separator;
        if (const char* e = sc.validate()) throw InvalidConfig(std::string("Invalid SplitterConfig: ") + e);
        const std::vector<std::string> chunks = TextSplitter(sc).split(text);
        if (chunks.empty()) return;
        char** arr = static_cast<char**>(std::calloc(chunks.size(), sizeof(char*)));
        if (!arr) throw std::bad_alloc();
        for (size_t i = 0; i < chunks.size(); ++i) arr[i] = dup_cstr(chunks[i]);
        out->strings = arr;
        out->len = chunks.size();
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_collect_files(const KjarniIndexerConfig* config, const char* const* inputs, size_t num_inputs,
                                                   KjarniStringArray* out)
{
    if (!inputs || !out) return KJARNI_ERROR_NULL_POINTER;
    out->strings = nullptr;
    out->len = 0;
    const KjarniIndexerConfig dflt = kjarni_indexer_config_default();
    const KjarniIndexerConfig& c = config ? *config : dflt;
    std::vector<std::string> in;
    if (const KjarniErrorCode rc = parse_inputs(inputs, num_inputs, in)) return rc;
    try {
        LoaderConfig lc;
        lc.recursive = c.recursive != 0;
        lc.include_hidden = c.include_hidden != 0;
        lc.has_max_file_size = true;
        lc.max_file_size = c.max_file_size > 0 ? c.max_file_size : 10u * 1024 * 1024;
        if (c.extensions)
            for (const std::string& e : split_commas(c.extensions)) {
                std::string x = lower(e);
                x.erase(0, x.find_first_not_of('.'));
                lc.extensions.push_back(x);
            }
        if (c.exclude_patterns)
            for (const std::string& p : split_commas(c.exclude_patterns)) lc.exclude_patterns.push_back(p);
        const std::vector<std::string> files = collect_files(lc, in);
        if (files.empty()) return KJARNI_OK;
        char** arr = static_cast<char**>(std::calloc(files.size(), sizeof(char*)));
        if (!arr) throw std::bad_alloc();
        for (size_t i = 0; i < files.size(); ++i) arr[i] = dup_cstr(files[i]);
        out->strings = arr;
        out->len = files.size();
        return KJARNI_OK;
    } catch (const PathNotFound& e) {
        set_last_error(std::string("Path not found: ") + e.what());
        return KJARNI_ERROR_MODEL_NOT_FOUND;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_INFERENCE_FAILED;
    }
}

// IndexWriter driven with caller-supplied embeddings: `texts[i]` with `embeddings[i*dim .. ]` and
// `metadata_json[i]` (a flat JSON object of strings, or NULL).  append = 0 creates (IndexWriter::open),
// append = 1 opens an existing index (IndexWriter::open_existing).
KJARNI_EXPORT KjarniErrorCode kjarni_index_write(const char* index_path, size_t dimension, size_t max_docs_per_segment,
                                                 const char* embedding_model, const char* const* texts,
                                                 const char* const* metadata_json, const float* embeddings, size_t n,
                                                 int32_t append)
{
    if (!index_path || (n && (!texts || !embeddings))) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        std::unique_ptr<IndexWriter> w;
        if (append) {
            w = IndexWriter::open_existing(index_path);
            if (w->dimension() != dimension) throw InvalidConfig("Dimension mismatch");
        } else {
            IndexConfig cfg;
            cfg.dimension = dimension;
            if (max_docs_per_segment) cfg.max_docs_per_segment = max_docs_per_segment;
            if (embedding_model) {
                cfg.has_embedding_model = true;
                cfg.embedding_model = embedding_model;
            }
            w = IndexWriter::open(index_path, cfg);
        }
        for (size_t i = 0; i < n; ++i) {
            Metadata md;
            if (metadata_json && metadata_json[i]) {
                const Json j = Json::parse(metadata_json[i]);
                for (const auto& kv : j.obj) md[kv.first] = kv.second.as_string();
            }
            w->add(texts[i], embeddings + i * dimension, dimension, &md);
        }
        w->commit();
    });
}

// ---- by-value twins of the frees ---------------------------------------------------------------
// The reference's stale cbindgen header and its C# / Python bindings declare the frees BY VALUE
// (crates/kjarni-ffi/include/kjarni.h:440-455, bindings/csharp/Kjarni/Native.cs:376-394) while the
// Rust source takes pointers.  A binding generated from that header can bind these names instead
// (e.g. DllImport EntryPoint = "kjarni_float_array_free_by_value") and keep its declarations.
KJARNI_EXPORT void kjarni_float_array_free_by_value(KjarniFloatArray arr) { kjarni_float_array_free(&arr); }
KJARNI_EXPORT void kjarni_float_2d_array_free_by_value(KjarniFloat2DArray arr) { kjarni_float_2d_array_free(&arr); }
KJARNI_EXPORT void kjarni_string_array_free_by_value(KjarniStringArray arr) { kjarni_string_array_free(&arr); }
KJARNI_EXPORT void kjarni_class_results_free_by_value(KjarniClassResults results) { kjarni_class_results_free(&results); }
KJARNI_EXPORT void kjarni_rerank_results_free_by_value(KjarniRerankResults results) { kjarni_rerank_results_free(&results); }
KJARNI_EXPORT void kjarni_search_results_free_by_value(KjarniSearchResults results) { kjarni_search_results_free(&results); }
