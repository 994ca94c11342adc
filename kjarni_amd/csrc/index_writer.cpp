#include "index_writer.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <system_error>

#include "index.h"
#include "json.h"
#include "unicode.h"

namespace fs = std::filesystem;

namespace kjarni {

namespace {

// Byte offsets of the character boundaries of a valid UTF-8 string (plus the end).
std::vector<size_t> char_starts(const std::string& s)
{
    std::vector<size_t> st;
    st.reserve(s.size() + 1);
    for (size_t i = 0; i < s.size(); ++i)
        if (((unsigned char)s[i] & 0xC0) != 0x80) st.push_back(i);
    st.push_back(s.size());
    return st;
}

void write_file(const std::string& path, const std::string& data)
{
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    if (!f) throw std::runtime_error("cannot create " + path);
    f.write(data.data(), (std::streamsize)data.size());
    f.close();
    if (!f) throw std::runtime_error("cannot write " + path);
}

std::string read_file(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::string data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (f.bad()) throw std::runtime_error("cannot read " + path);
    return data;
}

std::string join_path(const std::string& base, const std::string& name)
{
    if (base.empty()) return name;
    return base.back() == '/' ? base + name : base + "/" + name;  // PathBuf::push
}

// Path::file_name of a path that names a file.
std::string file_name_of(const std::string& path)
{
    size_t end = path.size();
    while (end > 1 && path[end - 1] == '/') --end;
    const size_t slash = path.rfind('/', end ? end - 1 : 0);
    return slash == std::string::npos ? path.substr(0, end) : path.substr(slash + 1, end - slash - 1);
}

}  // namespace

// ------------------------------------------------------------------------------------ splitter

const char* SplitterConfig::validate() const
{
    if (chunk_size == 0) return "chunk_size must be greater than 0";
    if (chunk_overlap >= chunk_size) return "chunk_overlap must be less than chunk_size";
    return nullptr;
}

TextSplitter::TextSplitter(SplitterConfig config) : config_(std::move(config))
{
    if (const char* e = config_.validate()) throw std::invalid_argument(std::string("Invalid SplitterConfig: ") + e);
    if (config_.separator.empty()) throw std::invalid_argument("Invalid SplitterConfig: separator must not be empty");
}

std::string TextSplitter::overlap_suffix(const std::string& text) const
{
    const std::vector<size_t> st = char_starts(text);
    const size_t n = st.size() - 1;
    if (n <= config_.chunk_overlap) return text;
    return text.substr(st[n - config_.chunk_overlap]);
}

void TextSplitter::split_large_text(const std::string& text, std::vector<std::string>& out) const
{
    const std::vector<size_t> st = char_starts(text);
    const size_t n = st.size() - 1;
    if (n == 0) return;
    const size_t cs = config_.chunk_size, ov = config_.chunk_overlap;
    size_t start = 0;
    while (start < n) {
        const size_t end = std::min(start + cs, n);
        out.push_back(text.substr(st[start], st[end] - st[start]));
        if (end >= n) break;
        const size_t step = (ov > 0 && ov < cs) ? cs - ov : cs;
        start = (start + step > start) ? start + step : start + 1;
    }
}

std::vector<std::string> TextSplitter::split(const std::string& text) const
{
    std::vector<std::string> chunks;
    if (text.empty()) return chunks;
    const std::string& sep = config_.separator;
    std::string current;
    size_t pos = 0;
    for (;;) {
        const size_t hit = text.find(sep, pos);
        const size_t end = hit == std::string::npos ? text.size() : hit;
        if (end > pos) {
            const size_t len = end - pos;
            if (len > config_.chunk_size) {
                if (!current.empty()) {
                    chunks.push_back(current);
                    current.clear();
                }
                split_large_text(text.substr(pos, len), chunks);
            } else {
                const size_t would_be = current.empty() ? len : current.size() + sep.size() + len;
                if (would_be > config_.chunk_size && !current.empty()) {
                    chunks.push_back(current);
                    if (config_.chunk_overlap > 0) current = overlap_suffix(current);
                    else current.clear();
                }
                if (!current.empty()) current += sep;
                current.append(text, pos, len);
            }
        }
        if (hit == std::string::npos) break;
        pos = hit + sep.size();
    }
    if (!current.empty()) chunks.push_back(std::move(current));
    return chunks;
}

size_t TextSplitter::estimate_chunks(const std::string& text) const
{
    if (text.empty()) return 0;
    const size_t eff = config_.chunk_size - config_.chunk_overlap;
    if (eff == 0) return 1;
    const size_t chars = char_starts(text).size() - 1;
    return (chars + eff - 1) / eff;
}

// ------------------------------------------------------------------------------------ loader

bool is_default_text_extension(const std::string& ext)
{
    static const char* const kExt[] = {
        "txt", "md", "markdown", "rst", "org",
        "json", "yaml", "yml", "toml", "xml", "csv",
        "html", "htm", "css",
        "rs", "py", "js", "ts", "go", "java", "c", "cpp", "h", "hpp",
        "cs", "rb", "sh", "bash", "zsh", "fish", "ps1",
        "sql", "r", "scala", "kt", "swift", "m", "mm",
        "lua", "pl", "php", "ex", "exs", "clj", "hs",
    };
    for (const char* e : kExt)
        if (ext == e) return true;
    return false;
}

bool is_supported_file(const LoaderConfig& config, const std::string& path)
{
    const std::string name = file_name_of(path);
    if (name.empty() || name == "..") return false;
    const size_t dot = name.rfind('.');
    if (dot == std::string::npos || dot == 0) return false;  // Path::extension: "foo", ".foo" have none
    std::string ext = name.substr(dot + 1);
    std::vector<uint32_t> cps, low;
    if (!unicode::decode_utf8(ext.data(), ext.size(), cps)) return false;  // to_str() == None
    unicode::lowercase_str(cps, low);
    ext.clear();
    for (uint32_t cp : low) unicode::append_utf8(ext, cp);
    if (config.extensions.empty()) return is_default_text_extension(ext);
    return std::find(config.extensions.begin(), config.extensions.end(), ext) != config.extensions.end();
}

namespace {

void walk(const LoaderConfig& config, const std::string& dir, std::vector<std::string>& files)
{
    std::vector<std::string> names;
    std::error_code ec;
    for (fs::directory_iterator it(dir, ec), end; !ec && it != end; it.increment(ec))
        names.push_back(it->path().filename().string());
    std::sort(names.begin(), names.end());
    for (const std::string& name : names) {
        const std::string path = join_path(dir, name);
        std::error_code e2;
        const fs::file_status lst = fs::symlink_status(path, e2);
        if (e2) continue;
        if (fs::is_directory(lst)) {  // WalkDir does not follow directory symlinks
            if (config.recursive) walk(config, path, files);
            continue;
        }
        if (!fs::is_regular_file(path, e2) || e2) continue;  // Path::is_file follows file symlinks
        if (!config.include_hidden && !name.empty() && name[0] == '.') continue;
        bool excluded = false;
        for (const std::string& pat : config.exclude_patterns)
            if (glob_match(pat, path)) {
                excluded = true;
                break;
            }
        if (excluded) continue;
        if (config.has_max_file_size) {
            const uintmax_t sz = fs::file_size(path, e2);
            if (!e2 && sz > config.max_file_size) continue;
        }
        if (is_supported_file(config, path)) files.push_back(path);
    }
}

}  // namespace

std::vector<std::string> collect_files(const LoaderConfig& config, const std::vector<std::string>& inputs)
{
    std::vector<std::string> files;
    for (const std::string& input : inputs) {
        std::error_code ec;
        if (!fs::exists(input, ec)) throw PathNotFound(input);
        if (fs::is_regular_file(input, ec)) {
            if (is_supported_file(config, input)) files.push_back(input);  // no hidden/exclude/size checks here
        } else if (fs::is_directory(input, ec)) {
            walk(config, input, files);
        }
    }
    return files;
}

std::vector<Chunk> DocumentLoader::load_file(const std::string& path) const
{
    const std::string content = read_file(path);
    if (!unicode::is_valid_utf8(content.data(), content.size()))
        throw std::runtime_error("stream did not contain valid UTF-8");  // fs::read_to_string
    std::vector<std::string> texts = splitter_.split(content);
    const size_t total = texts.size();
    std::vector<Chunk> chunks;
    chunks.reserve(total);
    for (size_t i = 0; i < total; ++i) {
        Chunk c;
        c.text = std::move(texts[i]);
        c.metadata["source"] = path;
        c.metadata["chunk_index"] = std::to_string(i);
        c.metadata["total_chunks"] = std::to_string(total);
        chunks.push_back(std::move(c));
    }
    return chunks;
}

// ------------------------------------------------------------------------------------ config

std::string IndexConfig::to_json_pretty() const
{
    auto opt_str = [](bool has, const std::string& s) { return has ? json_escape(s) : std::string("null"); };
    std::string o = "{\n";
    o += "  \"dimension\": " + std::to_string(dimension) + ",\n";
    o += "  \"max_docs_per_segment\": " + std::to_string(max_docs_per_segment) + ",\n";
    o += "  \"max_segment_memory\": " + std::to_string(max_segment_memory) + ",\n";
    o += "  \"embedding_model\": " + opt_str(has_embedding_model, embedding_model) + ",\n";
    o += "  \"model_name\": " + opt_str(has_model_name, model_name) + ",\n";
    o += "  \"created_at\": " + (has_created_at ? std::to_string(created_at) : std::string("null")) + ",\n";
    o += "  \"version\": " + std::to_string(version) + "\n}";
    return o;
}

IndexConfig IndexConfig::from_json(const std::string& text)
{
    const Json j = Json::parse(text);
    if (!j.is_object()) throw std::runtime_error("config.json: expected an object");
    IndexConfig c;
    // serde: every field without a default is required
    for (const char* k : {"dimension", "max_docs_per_segment", "max_segment_memory", "version"})
        if (!j.find(k) || !j.find(k)->is_number()) throw std::runtime_error(std::string("config.json: missing field `") + k + "`");
    c.dimension = (size_t)j.get_int("dimension", 0);
    c.max_docs_per_segment = (size_t)j.get_int("max_docs_per_segment", 0);
    c.max_segment_memory = (size_t)j.get_int("max_segment_memory", 0);
    c.version = (uint32_t)j.get_int("version", 1);
    if (const Json* v = j.find("embedding_model"); v && v->is_string()) {
        c.has_embedding_model = true;
        c.embedding_model = v->as_string();
    }
    if (const Json* v = j.find("model_name"); v && v->is_string()) {
        c.has_model_name = true;
        c.model_name = v->as_string();
    }
    if (const Json* v = j.find("created_at"); v && v->is_number()) {
        c.has_created_at = true;
        c.created_at = (uint64_t)v->as_int();
    }
    return c;
}

// ------------------------------------------------------------------------------------ segment builder

SegmentBuilder::SegmentBuilder(const std::string& temp_dir, size_t dimension, size_t max_docs)
    : dimension_(dimension), max_docs_(max_docs), temp_dir_(temp_dir)
{
    fs::create_directories(temp_dir);
    vectors_path_ = join_path(temp_dir, "vectors.bin.tmp");
    docs_path_ = join_path(temp_dir, "docs.bin.tmp");
    metadata_path_ = join_path(temp_dir, "metadata.jsonl.tmp");
    vectors_ = std::fopen(vectors_path_.c_str(), "wb");
    docs_ = std::fopen(docs_path_.c_str(), "wb");
    meta_ = std::fopen(metadata_path_.c_str(), "wb");
    if (!vectors_ || !docs_ || !meta_) {
        close_files();
        throw std::runtime_error("cannot create segment temp files in " + temp_dir);
    }
    std::setvbuf(vectors_, nullptr, _IOFBF, 64 * 1024);
    std::setvbuf(docs_, nullptr, _IOFBF, 64 * 1024);
    std::setvbuf(meta_, nullptr, _IOFBF, 32 * 1024);
    doc_offsets_.reserve(std::min<size_t>(max_docs, 1u << 20));
}

SegmentBuilder::~SegmentBuilder() { close_files(); }

void SegmentBuilder::close_files()
{
    for (FILE** f : {&vectors_, &docs_, &meta_})
        if (*f) {
            std::fclose(*f);
            *f = nullptr;
        }
}

size_t SegmentBuilder::add(const std::string& text, const float* embedding, size_t embedding_len, const Metadata* metadata)
{
    if (embedding_len != dimension_)
        throw std::runtime_error("Embedding dimension mismatch: got " + std::to_string(embedding_len) + ", expected " +
                                 std::to_string(dimension_));
    const size_t doc_id = doc_count_;
    // x86-64 / gfx950 hosts are little-endian: the in-memory f32 bytes are the to_le_bytes() image
    if (embedding_len && std::fwrite(embedding, sizeof(float), embedding_len, vectors_) != embedding_len)
        throw std::runtime_error("write failed: " + vectors_path_);
    doc_offsets_.push_back(current_offset_);
    if ((!text.empty() && std::fwrite(text.data(), 1, text.size(), docs_) != text.size()) || std::fputc('\n', docs_) == EOF)
        throw std::runtime_error("write failed: " + docs_path_);
    current_offset_ += (uint64_t)text.size() + 1;
    static const Metadata kEmpty;
    const std::string mj = metadata_to_json(metadata ? *metadata : kEmpty);
    if (std::fwrite(mj.data(), 1, mj.size(), meta_) != mj.size() || std::fputc('\n', meta_) == EOF)
        throw std::runtime_error("write failed: " + metadata_path_);
    bm25_.add_document(doc_id, text);
    ++doc_count_;
    return doc_id;
}

SegmentMeta SegmentBuilder::flush(const std::string& segment_dir, uint64_t segment_id)
{
    for (FILE* f : {vectors_, docs_, meta_})
        if (f && std::fflush(f) != 0) throw std::runtime_error("flush failed in " + temp_dir_);
    close_files();
    fs::create_directories(segment_dir);
    fs::rename(vectors_path_, join_path(segment_dir, "vectors.bin"));
    fs::rename(docs_path_, join_path(segment_dir, "docs.bin"));
    fs::rename(metadata_path_, join_path(segment_dir, "metadata.jsonl"));

    std::string idx;  // bincode Vec<u64>
    const uint64_t n = doc_offsets_.size();
    idx.append(reinterpret_cast<const char*>(&n), 8);
    if (n) idx.append(reinterpret_cast<const char*>(doc_offsets_.data()), n * 8);
    write_file(join_path(segment_dir, "docs.idx"), idx);
    write_file(join_path(segment_dir, "bm25.bin"), bm25_.to_bincode());

    SegmentMeta meta;
    meta.id = segment_id;
    meta.doc_count = doc_count_;
    meta.dimension = dimension_;
    meta.created_at = (uint64_t)std::chrono::duration_cast<std::chrono::seconds>(
                          std::chrono::system_clock::now().time_since_epoch()).count();
    meta.total_bytes = (uint64_t)fs::file_size(join_path(segment_dir, "vectors.bin")) +
                       (uint64_t)fs::file_size(join_path(segment_dir, "docs.bin"));
    std::string j = "{\n";
    j += "  \"id\": " + std::to_string(meta.id) + ",\n";
    j += "  \"doc_count\": " + std::to_string(meta.doc_count) + ",\n";
    j += "  \"dimension\": " + std::to_string(meta.dimension) + ",\n";
    j += "  \"created_at\": " + std::to_string(meta.created_at) + ",\n";
    j += "  \"total_bytes\": " + std::to_string(meta.total_bytes) + "\n}";
    write_file(join_path(segment_dir, "segment.json"), j);

    std::error_code ec;
    fs::remove(temp_dir_, ec);  // only if empty
    return meta;
}

// ------------------------------------------------------------------------------------ index writer

uint64_t IndexWriter::find_next_segment_id(const std::string& root)
{
    const std::string dir = join_path(root, "segments");
    std::error_code ec;
    if (!fs::exists(dir, ec)) return 0;
    uint64_t max_id = 0;
    for (const auto& entry : fs::directory_iterator(dir)) {
        const std::string name = entry.path().filename().string();
        if (name.rfind("seg_", 0) != 0) continue;
        const std::string digits = name.substr(4);
        // str::parse::<u64>: optional leading '+', then decimal digits only, no overflow
        size_t i = (!digits.empty() && digits[0] == '+') ? 1 : 0;
        if (i >= digits.size()) continue;
        uint64_t v = 0;
        bool ok = true;
        for (; i < digits.size() && ok; ++i) {
            const char c = digits[i];
            if (c < '0' || c > '9' || v > (UINT64_MAX - (uint64_t)(c - '0')) / 10) ok = false;
            else v = v * 10 + (uint64_t)(c - '0');
        }
        if (ok && v != UINT64_MAX) max_id = std::max(max_id, v + 1);
    }
    return max_id;
}

std::unique_ptr<IndexWriter> IndexWriter::open(const std::string& root, const IndexConfig& config)
{
    std::unique_ptr<IndexWriter> w(new IndexWriter());
    w->root_ = root;
    w->config_ = config;
    fs::create_directories(root);
    fs::create_directories(join_path(root, "segments"));
    write_file(join_path(root, "config.json"), config.to_json_pretty());
    w->next_segment_id_ = find_next_segment_id(root);
    return w;
}

std::unique_ptr<IndexWriter> IndexWriter::open_existing(const std::string& root)
{
    std::unique_ptr<IndexWriter> w(new IndexWriter());
    w->root_ = root;
    w->config_ = IndexConfig::from_json(read_file(join_path(root, "config.json")));
    w->next_segment_id_ = find_next_segment_id(root);
    for (const auto& entry : fs::directory_iterator(join_path(root, "segments"))) {
        if (!fs::is_directory(entry.path())) continue;
        const std::string mp = (entry.path() / "segment.json").string();
        if (!fs::exists(mp)) continue;
        const Json m = Json::parse(read_file(mp));
        for (const char* k : {"id", "doc_count", "dimension", "created_at", "total_bytes"})
            if (!m.find(k) || !m.find(k)->is_number()) throw std::runtime_error(mp + ": missing field `" + k + "`");
        w->total_docs_ += (size_t)m.get_int("doc_count", 0);
    }
    return w;
}

void IndexWriter::add(const std::string& text, const float* embedding, size_t embedding_len, const Metadata* metadata)
{
    if (!current_) {
        const std::string temp = join_path(join_path(root_, "temp"), "seg_" + std::to_string(next_segment_id_));
        current_ = std::make_unique<SegmentBuilder>(temp, config_.dimension, config_.max_docs_per_segment);
    }
    current_->add(text, embedding, embedding_len, metadata);
    ++total_docs_;
    if (current_->is_full()) flush_current_segment();
}

void IndexWriter::flush_current_segment()
{
    if (!current_) return;
    std::unique_ptr<SegmentBuilder> b = std::move(current_);
    if (b->empty()) return;
    char name[32];
    std::snprintf(name, sizeof name, "seg_%06llu", (unsigned long long)next_segment_id_);
    b->flush(join_path(join_path(root_, "segments"), name), next_segment_id_);
    ++next_segment_id_;
}

void IndexWriter::commit()
{
    flush_current_segment();
    std::string j = "{\n";
    j += "  \"total_docs\": " + std::to_string(total_docs_) + ",\n";
    j += "  \"segment_count\": " + std::to_string(next_segment_id_) + ",\n";
    j += "  \"dimension\": " + std::to_string(config_.dimension) + "\n}";
    write_file(join_path(root_, "index.json"), j);
    std::error_code ec;
    fs::remove_all(join_path(root_, "temp"), ec);
}

// ------------------------------------------------------------------------------------ misc

uint64_t directory_size(const std::string& path)
{
    uint64_t total = 0;
    std::error_code ec;
    if (fs::is_regular_file(fs::symlink_status(path, ec))) return (uint64_t)fs::file_size(path, ec);
    for (fs::recursive_directory_iterator it(path, fs::directory_options::skip_permission_denied, ec), end; !ec && it != end;
         it.increment(ec)) {
        std::error_code e2;
        if (fs::is_regular_file(it->symlink_status(e2)) && !e2) {
            const uintmax_t sz = fs::file_size(it->path(), e2);
            if (!e2) total += (uint64_t)sz;
        }
    }
    return total;
}

void remove_dir_all(const std::string& path)
{
    Segment::forget_under(path);  // parsed segments of this index must not outlive it
    std::error_code ec;
    fs::remove_all(path, ec);
    if (ec) throw std::runtime_error("cannot remove " + path + ": " + ec.message());
}

bool path_exists(const std::string& path)
{
    std::error_code ec;
    return fs::exists(path, ec);
}

}  // namespace kjarni
