// EncoderGroup: replicas of one encoder on several HIP devices of the node, driven from ONE process.
//
// The reference is single-device (no communication layer); this is the fan-out
// BASELINE.json's north_star asks for behind the unchanged C ABI, so that a C# / Go caller of
// kjarni_embedder_encode_batch / kjarni_reranker_rerank (crates/kjarni-ffi/src/embedder.rs:20-275,
// reranker.rs:10-322) reaches every GPU: rows are independent, so a batch is cut into balanced contiguous row
// blocks, one per device, weights replicated, one host thread + one stream per device, results written
// straight into the caller's buffer.  Device-resident consumers get the same sharding with an RCCL all-gather
// of the output slabs over xGMI (allgather_*), the only collective on the path.
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "encoder.h"

namespace kjarni {

// KJARNI_HIP_DEVICES="0,1,2" (default: every visible device; the older KJARNI_HIP_DEVICE=k means {k}).
// A device may be listed more than once (two workers on one GPU: how the 1-GPU test box exercises the fan-out).
std::vector<int> devices_from_env();

class EncoderGroup {
public:
    static std::unique_ptr<EncoderGroup> load(const std::string& dir, const std::vector<int>& devices);
    ~EncoderGroup();
    EncoderGroup(const EncoderGroup&) = delete;
    EncoderGroup& operator=(const EncoderGroup&) = delete;

    size_t size() const { return replicas_.size(); }
    EncoderModel& replica(size_t i) { return *replicas_[i]; }
    const EncoderConfig& config() const { return replicas_[0]->config(); }
    int device(size_t i) const { return replicas_[i]->device(); }

    // Balanced contiguous partition of `rows` over `parts`: floor(rows/parts) each, the first rows % parts one more.
    static void shard(int64_t rows, size_t parts, size_t i, int64_t* start, int64_t* count);
    // The collective of the device-resident form as a list of operations (pure host arithmetic, testable without a device):
    // rank `rank` of `parts` receives `floats` values of root `root`'s block at float offset `offset` of its full buffer.  Blocks
    // of equal size: ONE in-place all-gather per rank (root = -1, its own block's offset and size); otherwise one broadcast per
    // non-empty block and rank, in root order.  Every rank's operations cover [0, rows * width) exactly once.
    struct GatherOp {
        int32_t rank;     // the communicator rank that issues the call
        int32_t root;     // -1: ncclAllGather (in place); >= 0: ncclBroadcast from this root
        int64_t offset;   // in floats, into out_dev[rank] (send buffer for the all-gather / root, receive buffer otherwise)
        int64_t floats;   // element count of the call
    };
    static std::vector<GatherOp> gather_plan(int64_t rows_total, size_t parts, int64_t width);
    // Replicas a batch of `batch` sentences is spread over (a device's share is never below kMinRowsPerDevice,
    // so a single sentence does not pay a thread hop per GPU).
    size_t fanout(int64_t batch) const;
    static constexpr int64_t kMinRowsPerDevice = 8;

    // Host pointers in, host pointers out; returns when `out` is complete.  Callable from any number of threads.
    void embed_host(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch, int seq,
                    PoolMode pool, bool normalize, float mask_value, float* out);
    void logits_host(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch, int seq,
                     float mask_value, float* out);
    void hidden_states_host(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch, int seq,
                            float mask_value, float* out);

    // Device-resident form.  Replica i holds ITS row block of the inputs (ids_dev[i] etc.: shard(batch_total, size(), i)
    // rows, device i) and a FULL output buffer out_dev[i] = [batch_total, width] on device i.  Every replica computes
    // its block in place in its buffer, then the blocks are all-gathered so that every device holds the whole
    // output: ncclAllGather over xGMI when the devices are distinct (grouped ncclBroadcast when the blocks differ by
    // a row), peer copies when a device is listed twice.  Returns after the collective has completed.
    void allgather_embed(const uint32_t* const* ids_dev, const uint32_t* const* mask_dev, const uint32_t* const* type_dev,
                         int64_t batch_total, int seq, PoolMode pool, bool normalize, float mask_value,
                         float* const* out_dev);
    void allgather_logits(const uint32_t* const* ids_dev, const uint32_t* const* mask_dev, const uint32_t* const* type_dev,
                          int64_t batch_total, int seq, float mask_value, float* const* out_dev);
    const char* transport();  // "rccl" | "memcpy" (what allgather_* uses for this device list)

private:
    EncoderGroup() = default;
    struct Worker {
        std::thread thread;
        std::mutex mu;
        std::condition_variable cv;
        std::deque<std::function<void()>> queue;
        bool stop = false;
    };
    // Runs fn(i) for i < n on worker i (i = 0 inline on the caller) and rethrows the first failure.
    void parallel(size_t n, const std::function<void(size_t)>& fn);
    template <class F>
    void fan_out(int64_t batch, F&& per_block);
    void gather(float* const* out_dev, int64_t rows_total, int64_t width);
    bool ensure_rccl();

    std::vector<std::unique_ptr<EncoderModel>> replicas_;
    std::vector<std::unique_ptr<Worker>> workers_;
    std::vector<hipStream_t> streams_;  // one per replica, for the device-resident form and its collective
    std::atomic<size_t> next_{0};       // rotating first replica for calls that use only some devices
    std::mutex coll_mu_;                // one collective at a time per group
    bool distinct_ = true;
    bool rccl_tried_ = false;
    std::string transport_note_;        // why RCCL is not the transport, when its initialisation failed
    void* rccl_ = nullptr;              // RcclApi*, group.cpp
    std::vector<void*> comms_;
};

}  // namespace kjarni
