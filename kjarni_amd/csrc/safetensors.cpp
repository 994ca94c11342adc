#include "safetensors.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "json.h"

namespace kjarni {

SafeTensors::~SafeTensors()
{
    for (const auto& m : maps_) munmap(m.first, m.second);
}

void SafeTensors::open(const std::string& path) { map_file(path, tensors_); }

void SafeTensors::open_dir(const std::string& dir)
{
    const std::string index_path = dir + "/model.safetensors.index.json";
    struct stat ist;
    if (::stat(index_path.c_str(), &ist) != 0) {
        open(dir + "/model.safetensors");
        return;
    }
    std::string text;
    {
        FILE* f = std::fopen(index_path.c_str(), "rb");
        if (!f) throw std::runtime_error("failed to read index file: " + index_path);
        char buf[65536];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
        std::fclose(f);
    }
    const Json index = Json::parse(text);
    const Json* wm = index.find("weight_map");
    if (!wm || !wm->is_object()) throw std::runtime_error("invalid index.json: missing 'weight_map' object");
    std::map<std::string, std::map<std::string, TensorView>> shards;  // file name -> its tensors (sorted, deduplicated)
    for (const auto& kv : wm->obj)
        if (kv.second.is_string()) {
            const std::string& file = kv.second.str;
            if (file.empty() || file.find('/') != std::string::npos || file.find("..") != std::string::npos)
                throw std::runtime_error("invalid index.json: shard name '" + file + "'");
            shards[file];
        }
    for (auto& sh : shards) map_file(dir + "/" + sh.first, sh.second);
    for (const auto& kv : wm->obj) {
        if (!kv.second.is_string()) continue;
        const auto& shard = shards[kv.second.str];
        auto it = shard.find(kv.first);
        if (it == shard.end()) throw std::runtime_error("tensor '" + kv.first + "' is not in the shard the index maps it to (" + kv.second.str + ")");
        tensors_[kv.first] = it->second;
    }
}

void SafeTensors::map_file(const std::string& path, std::map<std::string, TensorView>& tensors_)
{
    void* map_ = nullptr;
    size_t map_len_ = 0;
    int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open " + path);
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 8) {
        ::close(fd);
        throw std::runtime_error("not a safetensors file (too short): " + path);
    }
    map_len_ = (size_t)st.st_size;
    map_ = mmap(nullptr, map_len_, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map_ == MAP_FAILED) throw std::runtime_error("mmap failed: " + path);
    maps_.emplace_back(map_, map_len_);
    const uint8_t* base = static_cast<const uint8_t*>(map_);
    uint64_t hlen = 0;
    std::memcpy(&hlen, base, 8);  // little-endian host
    if (hlen > map_len_ - 8) throw std::runtime_error("corrupt safetensors header length: " + path);
    Json header = Json::parse(reinterpret_cast<const char*>(base + 8), (size_t)hlen);
    if (!header.is_object()) throw std::runtime_error("safetensors header is not an object: " + path);
    const uint8_t* data0 = base + 8 + hlen;
    const size_t data_len = map_len_ - 8 - (size_t)hlen;
    for (const auto& kv : header.obj) {
        if (kv.first == "__metadata__") continue;
        const Json& e = kv.second;
        TensorView tv;
        tv.dtype = e.get_string("dtype", "");
        const Json* shp = e.find("shape");
        if (shp && shp->is_array())
            for (const Json& d : shp->arr) tv.shape.push_back(d.as_int());
        const Json* off = e.find("data_offsets");
        if (!off || !off->is_array() || off->arr.size() != 2)
            throw std::runtime_error("tensor without data_offsets: " + kv.first);
        const uint64_t b = (uint64_t)off->arr[0].as_int(), en = (uint64_t)off->arr[1].as_int();
        if (en < b || en > data_len) throw std::runtime_error("tensor out of file bounds: " + kv.first);
        tv.data = data0 + b;
        tv.nbytes = (size_t)(en - b);
        tensors_.emplace(kv.first, std::move(tv));
    }
}

const TensorView& SafeTensors::get(const std::string& name) const
{
    auto it = tensors_.find(name);
    if (it == tensors_.end()) throw std::runtime_error("tensor not found: " + name);
    return it->second;
}

static inline float half_to_float(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1F;
    uint32_t man = h & 0x3FF;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {
            int e = -1;
            do {
                man <<= 1;
                ++e;
            } while ((man & 0x400) == 0);
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FF) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

std::vector<int64_t> SafeTensors::read_f32(const std::string& name, std::vector<float>& out) const
{
    const TensorView& tv = get(name);
    const int64_t n = tv.numel();
    out.resize((size_t)n);
    if (tv.dtype == "F32") {
        if (tv.nbytes != (size_t)n * 4) throw std::runtime_error("size mismatch: " + name);
        std::memcpy(out.data(), tv.data, tv.nbytes);
    } else if (tv.dtype == "F16") {
        if (tv.nbytes != (size_t)n * 2) throw std::runtime_error("size mismatch: " + name);
        const uint16_t* s = reinterpret_cast<const uint16_t*>(tv.data);
        for (int64_t i = 0; i < n; ++i) out[(size_t)i] = half_to_float(s[i]);
    } else if (tv.dtype == "BF16") {
        if (tv.nbytes != (size_t)n * 2) throw std::runtime_error("size mismatch: " + name);
        const uint16_t* s = reinterpret_cast<const uint16_t*>(tv.data);
        for (int64_t i = 0; i < n; ++i) {
            uint32_t bits = (uint32_t)s[i] << 16;
            std::memcpy(&out[(size_t)i], &bits, 4);
        }
    } else if (tv.dtype == "F64") {
        if (tv.nbytes != (size_t)n * 8) throw std::runtime_error("size mismatch: " + name);
        const double* s = reinterpret_cast<const double*>(tv.data);
        for (int64_t i = 0; i < n; ++i) out[(size_t)i] = (float)s[i];
    } else {
        throw std::runtime_error("unsupported dtype " + tv.dtype + " for tensor " + name);
    }
    return tv.shape;
}

}  // namespace kjarni
