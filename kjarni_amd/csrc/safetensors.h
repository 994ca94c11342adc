// Read-only safetensors view (mmap).  Counterpart of ModelWeights
// (crates/kjarni-transformers/src/weights/model_weights.rs:45-282,
// weights/mmap_cache.rs:12): tensors are looked up by their HF names and
// converted to f32 on request (F32 is the hot-path dtype; F16/BF16 files are
// widened so they still load).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace kjarni {

struct TensorView {
    std::string dtype;  // "F32", "F16", "BF16", ...
    std::vector<int64_t> shape;
    const uint8_t* data = nullptr;
    size_t nbytes = 0;
    int64_t numel() const
    {
        int64_t n = 1;
        for (int64_t d : shape) n *= d;
        return n;
    }
};

class SafeTensors {
public:
    SafeTensors() = default;
    ~SafeTensors();
    SafeTensors(const SafeTensors&) = delete;
    SafeTensors& operator=(const SafeTensors&) = delete;

    // Throws std::runtime_error with a readable message on any format problem.
    void open(const std::string& path);
    // A model directory (weights/safetensors_loader.rs:46-129): `model.safetensors.index.json` + the shards its
    // weight_map names when the index exists, else the single `model.safetensors`.  A sharded model exposes the tensors the
    // weight_map lists, each read from the shard it is mapped to.
    void open_dir(const std::string& dir);
    size_t shard_count() const { return maps_.size(); }
    bool contains(const std::string& name) const { return tensors_.count(name) != 0; }
    const TensorView& get(const std::string& name) const;
    // Copies the tensor as f32 into out (resized); returns its shape.
    std::vector<int64_t> read_f32(const std::string& name, std::vector<float>& out) const;
    const std::map<std::string, TensorView>& tensors() const { return tensors_; }

private:
    void map_file(const std::string& path, std::map<std::string, TensorView>& into);

    std::vector<std::pair<void*, size_t>> maps_;
    std::map<std::string, TensorView> tensors_;
};

}  // namespace kjarni
