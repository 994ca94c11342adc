// A model's device memory as a few large hipMalloc blocks instead of one hipMalloc per tensor.
// A loader that uploads tensor by tensor leaves a decode stage's operands (weights, bias, LayerNorm parameters, the rows it
// reads and writes) in seven or eight allocations scattered over the address space, 200+ per model.  Handing the tensors out of
// 64 MiB blocks in upload order keeps a layer's operands next to each other and the allocator's bookkeeping to a handful of
// entries.  Measured: no effect on the decode steps (docs/history/r06.md 2b) -- kept for the tidier address space and the
// faster load / unload.  Requests above 16 MiB keep an allocation of their own.
#pragma once
#include <hip/hip_runtime.h>

#include <stdexcept>
#include <string>
#include <vector>

namespace kjarni {

class DeviceArena {
public:
    DeviceArena() = default;
    DeviceArena(const DeviceArena&) = delete;
    DeviceArena& operator=(const DeviceArena&) = delete;
    ~DeviceArena() { release(); }

    // 256-byte aligned, never null; throws std::runtime_error when the device is out of memory
    void* alloc(size_t bytes)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes == 0) bytes = 256;
        if (bytes > kOwn) return fresh(bytes);
        if (bytes > left_) {
            cur_ = static_cast<char*>(fresh(kBlock));
            left_ = kBlock;
        }
        void* p = cur_;
        cur_ += bytes;
        left_ -= bytes;
        return p;
    }
    void release()
    {
        for (void* p : blocks_) (void)hipFree(p);
        blocks_.clear();
        cur_ = nullptr;
        left_ = 0;
    }

private:
    static constexpr size_t kBlock = (size_t)64 << 20, kOwn = (size_t)16 << 20;
    void* fresh(size_t bytes)
    {
        void* p = nullptr;
        const hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess || !p) throw std::runtime_error(std::string("hipMalloc: ") + hipGetErrorString(e));
        blocks_.push_back(p);
        return p;
    }
    std::vector<void*> blocks_;
    char* cur_ = nullptr;
    size_t left_ = 0;
};

}  // namespace kjarni
