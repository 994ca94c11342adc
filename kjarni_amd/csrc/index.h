// Read side of the reference's segmented on-disk index + its host-side ranking
// logic: BM25, reciprocal-rank fusion, metadata filter.
//
//   <root>/config.json                                  crates/kjarni-rag/src/config.rs:5-13
//   <root>/segments/<dir>/segment.json                  kjarni-rag/src/segment.rs:11-18
//                         vectors.bin   raw LE f32 [doc_count, dimension]   segment.rs:103-106, 240-262
//                         docs.bin      texts joined by '\n'                segment.rs:108-112
//                         docs.idx      bincode Vec<u64> of text offsets    segment.rs:162-164
//                         bm25.bin      bincode Bm25Index                   segment.rs:166-168
//                         metadata.jsonl one JSON object per document       segment.rs:114-116
//
// Everything here is host logic (integer / small-float work the reference also runs
// on the CPU).  The cosine scan over vectors.bin -- Segment::search_vectors, the hot
// loop -- is NOT here: IndexReader::search_semantic takes a callback that runs it on
// the GPU (kernels of cosine.hip).
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace kjarni {

// crates/kjarni-search/src/bm25.rs:42-189
class Bm25Index {
public:
    static std::vector<std::string> tokenize(const std::string& text);  // bm25.rs:191-197
    void add_document(size_t doc_id, const std::string& text);          // bm25.rs:108-139
    std::vector<std::pair<size_t, float>> search(const std::string& query, size_t limit) const;  // bm25.rs:84-107
    float score(const std::vector<std::string>& query_tokens, size_t doc_id) const;               // bm25.rs:141-175
    size_t term_frequency(const std::string& term, size_t doc_id) const;                          // bm25.rs:177-188

    // bincode 1.x layout of the struct (fixed-width LE ints, u64 lengths, fields in declaration order).
    static Bm25Index from_bincode(const uint8_t* data, size_t len);
    std::string to_bincode() const;

    size_t total_docs = 0;
    float avg_doc_length = 0.0f;
    std::vector<uint64_t> doc_lengths;
    std::unordered_map<std::string, uint64_t> doc_frequencies;
    std::unordered_map<std::string, std::vector<std::pair<uint64_t, uint64_t>>> inverted_index;
    float k1 = 1.2f, b = 0.75f, epsilon = 0.25f;
    uint64_t total_length = 0;
};

// crates/kjarni-search/src/hybrid.rs:3-31 (k = 60).  Ties: ascending id (the reference
// sorts HashMap entries, whose tie order is unspecified).
std::vector<std::pair<size_t, float>> hybrid_search(const std::vector<std::pair<size_t, float>>& keyword,
                                                    const std::vector<std::pair<size_t, float>>& semantic,
                                                    size_t limit);

// glob-match 0.2 semantics used by MetadataFilter (`*` within a component, `**`, `?`, `[..]`, `{a,b}`).
bool glob_match(const std::string& pattern, const std::string& path);

using Metadata = std::map<std::string, std::string>;

// crates/kjarni-rag/src/index_reader.rs:13-101
struct MetadataFilter {
    std::map<std::string, std::string> must_match, must_not_match;
    std::vector<std::string> source_patterns;
    bool empty() const { return must_match.empty() && must_not_match.empty() && source_patterns.empty(); }
    bool matches(const Metadata& md) const;
};

struct SearchHit {
    float score = 0.0f;
    size_t document_id = 0;  // global id across segments
    std::string text;
    Metadata metadata;
};

// crates/kjarni-rag/src/segment.rs:200-345
class Segment {
public:
    static std::unique_ptr<Segment> open(const std::string& dir);
    // Same segment, parsed once per process: queries re-open the index (Searcher::search does), so the parsed
    // BM25 postings / offsets / mmap are kept and revalidated by every file's (size, mtime) before reuse.
    static std::shared_ptr<const Segment> open_shared(const std::string& dir);
    static void forget_under(const std::string& root);  // the index at `root` is being removed
    ~Segment();
    size_t doc_count() const { return doc_count_; }
    size_t dimension() const { return dimension_; }
    const float* vectors() const { return vectors_; }  // mmap of vectors.bin
    size_t vectors_bytes() const { return map_len_; }
    const std::string& dir() const { return dir_; }
    // Unique per parsed Segment object in this process (never reused): the key of device-side copies of its vectors.
    uint64_t uid() const { return uid_; }
    std::string get_document(size_t doc_id) const;     // segment.rs:264-289
    Metadata get_metadata(size_t doc_id) const;        // segment.rs:292-304
    std::vector<std::pair<size_t, float>> search_keywords(const std::string& q, size_t limit) const
    {
        return bm25_.search(q, limit);
    }
    const Bm25Index& bm25() const { return bm25_; }

private:
    std::string dir_;
    uint64_t uid_ = 0;
    size_t doc_count_ = 0, dimension_ = 0;
    const float* vectors_ = nullptr;
    void* map_ = nullptr;
    size_t map_len_ = 0;
    std::vector<uint64_t> doc_offsets_;
    Bm25Index bm25_;
    // Hits are read out of mappings of docs.bin / metadata.jsonl made on first use (no descriptor is held: an index may have
    // thousands of segments).  The file sizes and the byte offset of every metadata.jsonl line are taken once per parsed segment
    // (a changed segment makes a new Segment, see open_shared).
    mutable std::once_flag docs_once_, meta_once_;
    mutable uint64_t docs_size_ = 0, meta_size_ = 0;
    mutable const char *docs_map_ = nullptr, *meta_map_ = nullptr;  // mapped on first use (null: read with pread)
    mutable std::vector<uint64_t> meta_offsets_;
};

// Scans every segment for the query: per segment, up to `limit` (local doc id, score), score descending
// (Segment::search_vectors, segment.rs:307-337).  Supplied by the GPU side, which takes the whole list so that
// all segments of a query are enqueued before the one synchronisation.
using SegmentHits = std::vector<std::pair<size_t, float>>;

// Documents from which IndexReader::search_keywords walks the segments on several host threads (kjarni_hip.h).
void set_keyword_parallel_min_docs(size_t docs);
using SegmentScanFn = std::function<std::vector<SegmentHits>(const std::vector<const Segment*>& segments,
                                                             const float* query, size_t limit)>;

// crates/kjarni-rag/src/index_reader.rs:104-347
class IndexReader {
public:
    static std::unique_ptr<IndexReader> open(const std::string& root);
    size_t dimension() const { return dimension_; }
    size_t len() const { return total_docs_; }
    size_t segment_count() const { return segments_.size(); }
    const Segment& segment(size_t i) const { return *segments_[i]; }

    std::vector<SearchHit> search_semantic(const float* query, size_t limit, const SegmentScanFn& scan) const;
    std::vector<SearchHit> search_keywords(const std::string& query, size_t limit) const;
    std::vector<SearchHit> search_hybrid(const std::string& query, const float* query_emb, size_t limit,
                                         const SegmentScanFn& scan) const;
    // *_filtered: fetch 3x, filter, take `limit` (index_reader.rs:107-158)
    std::vector<SearchHit> apply_filter(std::vector<SearchHit> hits, const MetadataFilter& f, size_t limit) const;

private:
    size_t local_to_global(size_t seg, size_t local) const;
    bool global_to_local(size_t global, size_t& seg, size_t& local) const;
    std::vector<SearchHit> convert(const std::vector<std::tuple<size_t, size_t, float>>& r) const;

    size_t dimension_ = 0, total_docs_ = 0;
    std::vector<std::shared_ptr<const Segment>> segments_;
};

std::string metadata_to_json(const Metadata& md);  // serde_json::to_string(&HashMap), keys sorted
std::string json_escape(const std::string& s);

}  // namespace kjarni
