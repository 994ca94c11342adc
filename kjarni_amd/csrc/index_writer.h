// Write side of the reference's segmented on-disk index, and the document pipeline in front of
// it: file discovery, loading, chunking.
//
//   TextSplitter      crates/kjarni-rag/src/splitter.rs:44-205
//   DocumentLoader    crates/kjarni-rag/src/loader.rs:9-110, 181-198
//   collect_files     crates/kjarni/src/indexer/model.rs:727-810
//   SegmentBuilder    crates/kjarni-rag/src/segment.rs:21-197
//   IndexWriter       crates/kjarni-rag/src/index_writer.rs:12-191
//
// Host logic throughout (byte/integer work the reference also runs on the CPU); the embeddings that
// go into vectors.bin come from the GPU encoder (ffi_indexer.cpp).
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "index.h"

namespace kjarni {

// splitter.rs:5-42
struct SplitterConfig {
    size_t chunk_size = 1000;     // characters
    size_t chunk_overlap = 200;   // characters
    std::string separator = "\n\n";
    // nullptr when valid, else the reference's message (splitter.rs:33-41)
    const char* validate() const;
};

// splitter.rs:44-205.  Lengths compared against chunk_size are BYTE lengths where the reference
// uses str::len (section and running-chunk sizes) and CHARACTER counts where it collects chars
// (oversized sections, overlap suffix).
class TextSplitter {
public:
    explicit TextSplitter(SplitterConfig config);  // throws std::invalid_argument (the reference panics)
    std::vector<std::string> split(const std::string& text) const;
    size_t estimate_chunks(const std::string& text) const;
    const SplitterConfig& config() const { return config_; }

private:
    std::string overlap_suffix(const std::string& text) const;
    void split_large_text(const std::string& text, std::vector<std::string>& out) const;
    SplitterConfig config_;
};

// loader.rs:24-32
struct LoaderConfig {
    SplitterConfig splitter;
    bool recursive = true;
    std::vector<std::string> extensions;        // empty = TEXT_EXTENSIONS
    std::vector<std::string> exclude_patterns;  // glob, matched against the whole path
    bool include_hidden = false;
    bool has_max_file_size = false;
    size_t max_file_size = 0;
};

struct Chunk {
    std::string text;
    Metadata metadata;  // ChunkMetadata::to_hashmap: source, chunk_index, total_chunks
};

bool is_default_text_extension(const std::string& ext);  // loader.rs:9-21
// Path::extension + to_lowercase, then the configured / default list (loader.rs:181-198).
bool is_supported_file(const LoaderConfig& config, const std::string& path);

struct PathNotFound : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// indexer/model.rs:727-810.  Directory entries are visited in byte order of their names (the
// reference takes the OS's readdir order, which is unspecified).
std::vector<std::string> collect_files(const LoaderConfig& config, const std::vector<std::string>& inputs);

class DocumentLoader {
public:
    explicit DocumentLoader(const LoaderConfig& config) : splitter_(config.splitter) {}
    // loader.rs:85-110.  Throws when the file cannot be read or is not valid UTF-8.
    std::vector<Chunk> load_file(const std::string& path) const;

private:
    TextSplitter splitter_;
};

// config.rs:5-29
struct IndexConfig {
    size_t dimension = 384;
    size_t max_docs_per_segment = 10000;
    size_t max_segment_memory = 100u * 1024 * 1024;
    bool has_embedding_model = false;
    std::string embedding_model;
    bool has_model_name = false;
    std::string model_name;
    bool has_created_at = false;
    uint64_t created_at = 0;
    uint32_t version = 1;

    std::string to_json_pretty() const;
    static IndexConfig from_json(const std::string& text);
};

struct SegmentMeta {
    uint64_t id = 0;
    size_t doc_count = 0, dimension = 0;
    uint64_t created_at = 0, total_bytes = 0;
};

// segment.rs:21-197: streams vectors / texts / metadata to temp files, keeps BM25 in memory.
class SegmentBuilder {
public:
    SegmentBuilder(const std::string& temp_dir, size_t dimension, size_t max_docs);
    ~SegmentBuilder();
    SegmentBuilder(const SegmentBuilder&) = delete;
    SegmentBuilder& operator=(const SegmentBuilder&) = delete;

    size_t add(const std::string& text, const float* embedding, size_t embedding_len, const Metadata* metadata);
    bool is_full() const { return doc_count_ >= max_docs_; }
    bool empty() const { return doc_count_ == 0; }
    size_t len() const { return doc_count_; }
    SegmentMeta flush(const std::string& segment_dir, uint64_t segment_id);

private:
    void close_files();
    size_t dimension_, max_docs_;
    std::string temp_dir_, vectors_path_, docs_path_, metadata_path_;
    FILE* vectors_ = nullptr;
    FILE* docs_ = nullptr;
    FILE* meta_ = nullptr;
    std::vector<uint64_t> doc_offsets_;
    uint64_t current_offset_ = 0;
    Bm25Index bm25_;
    size_t doc_count_ = 0;
};

// index_writer.rs:12-191
class IndexWriter {
public:
    static std::unique_ptr<IndexWriter> open(const std::string& root, const IndexConfig& config);
    static std::unique_ptr<IndexWriter> open_existing(const std::string& root);

    void add(const std::string& text, const float* embedding, size_t embedding_len, const Metadata* metadata);
    void commit();
    size_t len() const { return total_docs_; }
    size_t dimension() const { return config_.dimension; }

private:
    IndexWriter() = default;
    static uint64_t find_next_segment_id(const std::string& root);
    void flush_current_segment();
    std::string root_;
    IndexConfig config_;
    std::unique_ptr<SegmentBuilder> current_;
    uint64_t next_segment_id_ = 0;
    size_t total_docs_ = 0;
};

// Sum of the sizes of all regular files under `path` (indexer/model.rs:25-36).
uint64_t directory_size(const std::string& path);
// std::fs::remove_dir_all
void remove_dir_all(const std::string& path);
bool path_exists(const std::string& path);

}  // namespace kjarni
