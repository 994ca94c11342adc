#include "group.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <exception>
#include <set>

namespace kjarni {

std::vector<int> devices_from_env()
{
    const int n = visible_device_count();
    std::vector<int> out;
    if (const char* list = std::getenv("KJARNI_HIP_DEVICES")) {
        const char* p = list;
        while (*p) {
            char* end = nullptr;
            const long v = std::strtol(p, &end, 10);
            if (end == p) throw InvalidDeviceList(std::string("KJARNI_HIP_DEVICES is not a comma-separated list of device indices: '") + list + "'");
            out.push_back((int)v);
            p = end;
            while (*p == ',' || *p == ' ') ++p;
        }
        if (out.empty()) throw InvalidDeviceList("KJARNI_HIP_DEVICES is empty");
    } else if (const char* one = std::getenv("KJARNI_HIP_DEVICE")) {
        out.push_back(std::atoi(one));
    } else {
        // Under a one-process-per-GPU launcher (torchrun and friends export LOCAL_RANK) every rank sees every GPU:
        // replicas on all of them from every rank would be N^2 models.  A rank keeps to its own device.
        const char* lr = std::getenv("LOCAL_RANK");
        if (lr && *lr) {
            char* end = nullptr;
            const long v = std::strtol(lr, &end, 10);
            if (end != lr && v >= 0 && (n == 0 || v < n)) {
                out.push_back((int)v);
                return out;
            }
        }
        for (int i = 0; i < n; ++i) out.push_back(i);
        if (out.empty()) out.push_back(0);  // EncoderModel::load reports the missing GPU
    }
    return out;
}

void EncoderGroup::shard(int64_t rows, size_t parts, size_t i, int64_t* start, int64_t* count)
{
    const int64_t p = (int64_t)std::max<size_t>(parts, 1), base = rows / p, rem = rows % p, k = (int64_t)i;
    *start = k * base + std::min(k, rem);
    *count = base + (k < rem ? 1 : 0);
}

size_t EncoderGroup::fanout(int64_t batch) const
{
    const int64_t by_rows = batch / kMinRowsPerDevice;
    return (size_t)std::max<int64_t>(1, std::min<int64_t>((int64_t)replicas_.size(), by_rows));
}

std::unique_ptr<EncoderGroup> EncoderGroup::load(const std::string& dir, const std::vector<int>& devices)
{
    if (devices.empty()) throw InvalidDeviceList("empty device list");
    std::unique_ptr<EncoderGroup> g(new EncoderGroup());
    for (int d : devices) g->replicas_.push_back(EncoderModel::load(dir, d));
    g->distinct_ = std::set<int>(devices.begin(), devices.end()).size() == devices.size();
    // worker 0 is the calling thread
    for (size_t i = 1; i < devices.size(); ++i) {
        g->workers_.push_back(std::make_unique<Worker>());
        Worker* w = g->workers_.back().get();
        w->thread = std::thread([w] {
            for (;;) {
                std::function<void()> job;
                {
                    std::unique_lock<std::mutex> lock(w->mu);
                    w->cv.wait(lock, [w] { return w->stop || !w->queue.empty(); });
                    if (w->queue.empty()) return;  // stop requested and nothing left
                    job = std::move(w->queue.front());
                    w->queue.pop_front();
                }
                job();
            }
        });
    }
    return g;
}

namespace {

// RCCL is resolved at run time: the library that is already in the process (torch ships its own librccl next
// to its HIP runtime) or the ROCm one; a host without RCCL still loads libkjarni_ffi.so and only the
// device-resident all-gather reports the failure.
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi* open_rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL)) != nullptr) break;
        if (!h)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
        if (!h) return;
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(h, "ncclAllGather"));
        api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(dlsym(h, "ncclBroadcast"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(h, "ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        if (api.CommInitAll && api.CommDestroy && api.AllGather && api.Broadcast && api.GroupStart && api.GroupEnd)
            api.handle = h;
    });
    return api.handle ? &api : nullptr;
}

void nccl_check(const RcclApi& r, ncclResult_t e, const char* what)
{
    if (e != ncclSuccess)
        throw HipError(std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(e) : "RCCL error"));
}

}  // namespace

EncoderGroup::~EncoderGroup()
{
    for (auto& w : workers_) {
        {
            std::lock_guard<std::mutex> lock(w->mu);
            w->stop = true;
        }
        w->cv.notify_all();
        if (w->thread.joinable()) w->thread.join();
    }
    if (rccl_)
        for (void* c : comms_)
            if (c) (void)static_cast<RcclApi*>(rccl_)->CommDestroy(static_cast<ncclComm_t>(c));
    for (size_t i = 0; i < streams_.size(); ++i)
        if (streams_[i]) {
            (void)hipSetDevice(replicas_[i]->device());
            (void)hipStreamDestroy(streams_[i]);
        }
}

void EncoderGroup::parallel(size_t n, const std::function<void(size_t)>& fn)
{
    if (n <= 1) {
        if (n == 1) fn(0);
        return;
    }
    struct Join {
        std::mutex mu;
        std::condition_variable cv;
        size_t pending;
        std::exception_ptr error;
    } join;
    join.pending = n - 1;
    for (size_t i = 1; i < n; ++i) {
        Worker& w = *workers_[i - 1];
        {
            std::lock_guard<std::mutex> lock(w.mu);
            w.queue.emplace_back([&join, &fn, i] {
                std::exception_ptr err;
                try {
                    fn(i);
                } catch (...) {
                    err = std::current_exception();
                }
                std::lock_guard<std::mutex> l(join.mu);
                if (err && !join.error) join.error = err;
                if (--join.pending == 0) join.cv.notify_one();
            });
        }
        w.cv.notify_one();
    }
    std::exception_ptr mine;
    try {
        fn(0);
    } catch (...) {
        mine = std::current_exception();
    }
    {
        std::unique_lock<std::mutex> lock(join.mu);
        join.cv.wait(lock, [&join] { return join.pending == 0; });
    }
    if (mine) std::rethrow_exception(mine);
    if (join.error) std::rethrow_exception(join.error);
}

template <class F>
void EncoderGroup::fan_out(int64_t batch, F&& per_block)
{
    if (batch <= 0) return;
    const size_t n = fanout(batch), all = replicas_.size();
    // Calls that do not use every device start at a rotating replica, so that many small requests (one
    // sentence to classify, from many host threads) spread over the GPUs instead of queueing on the first.
    const size_t first = n < all ? next_.fetch_add(n, std::memory_order_relaxed) % all : 0;
    parallel(n, [&](size_t i) {
        int64_t start, count;
        shard(batch, n, i, &start, &count);
        if (count > 0) per_block(*replicas_[(first + i) % all], start, count);
    });
}

void EncoderGroup::embed_host(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch, int seq,
                              PoolMode pool, bool normalize, float mask_value, float* out)
{
    const int64_t H = config().hidden;
    fan_out(batch, [&](EncoderModel& m, int64_t start, int64_t count) {
        m.embed_host(ids + start * seq, mask + start * seq, type_ids ? type_ids + start * seq : nullptr, count, seq, pool,
                     normalize, mask_value, out + start * H);
    });
}

void EncoderGroup::logits_host(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch, int seq,
                               float mask_value, float* out)
{
    const int64_t L = config().num_labels;
    fan_out(batch, [&](EncoderModel& m, int64_t start, int64_t count) {
        m.logits_host(ids + start * seq, mask + start * seq, type_ids ? type_ids + start * seq : nullptr, count, seq,
                      mask_value, out + start * L);
    });
}

void EncoderGroup::hidden_states_host(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch,
                                      int seq, float mask_value, float* out)
{
    const int64_t H = config().hidden;
    fan_out(batch, [&](EncoderModel& m, int64_t start, int64_t count) {
        m.hidden_states_host(ids + start * seq, mask + start * seq, type_ids ? type_ids + start * seq : nullptr, count, seq,
                             mask_value, out + start * seq * H);
    });
}

bool EncoderGroup::ensure_rccl()
{
    if (!distinct_) return false;
    if (rccl_tried_) return rccl_ != nullptr;
    rccl_tried_ = true;
    RcclApi* r = open_rccl();
    if (!r) return false;
    std::vector<int> devs;
    for (auto& m : replicas_) devs.push_back(m->device());
    std::vector<ncclComm_t> comms(devs.size());
    const ncclResult_t e = r->CommInitAll(comms.data(), (int)devs.size(), devs.data());
    if (e != ncclSuccess) {  // the copy transport takes over -- and transport() says why
        transport_note_ = std::string("memcpy (ncclCommInitAll failed: ") + (r->GetErrorString ? r->GetErrorString(e) : "RCCL error") + ")";
        return false;
    }
    comms_.assign(comms.begin(), comms.end());
    rccl_ = r;
    return true;
}

const char* EncoderGroup::transport()
{
    std::lock_guard<std::mutex> lock(coll_mu_);
    DeviceGuard guard;
    if (ensure_rccl()) return "rccl";
    return transport_note_.empty() ? "memcpy" : transport_note_.c_str();
}

std::vector<EncoderGroup::GatherOp> EncoderGroup::gather_plan(int64_t rows_total, size_t parts, int64_t width)
{
    std::vector<GatherOp> ops;
    if (parts == 0 || rows_total < 0 || width <= 0) return ops;
    std::vector<int64_t> start(parts), count(parts);
    for (size_t i = 0; i < parts; ++i) shard(rows_total, parts, i, &start[i], &count[i]);
    const bool even = rows_total % (int64_t)parts == 0;
    for (size_t i = 0; i < parts; ++i) {
        if (even) {
            ops.push_back({(int32_t)i, -1, start[i] * width, count[i] * width});
        } else {
            for (size_t root = 0; root < parts; ++root)
                if (count[root] > 0) ops.push_back({(int32_t)i, (int32_t)root, start[root] * width, count[root] * width});
        }
    }
    return ops;
}

// Every out_dev[i] holds block i (rows shard(rows_total, n, i)) at its place; afterwards every buffer holds all
// blocks.  Called with the compute of block i already enqueued on streams_[i].
void EncoderGroup::gather(float* const* out_dev, int64_t rows_total, int64_t width)
{
    const size_t n = replicas_.size();
    std::vector<int64_t> start(n), count(n);
    for (size_t i = 0; i < n; ++i) shard(rows_total, n, i, &start[i], &count[i]);
    if (ensure_rccl()) {
        const RcclApi& r = *static_cast<RcclApi*>(rccl_);
        const std::vector<GatherOp> plan = gather_plan(rows_total, n, width);
        nccl_check(r, r.GroupStart(), "ncclGroupStart");
        // Nothing throws between GroupStart and GroupEnd: the first failure is remembered, the bracket is always closed
        // and the streams drained, so a failed collective leaves neither an open RCCL group on this thread nor
        // unsynchronised streams behind.
        ncclResult_t first = ncclSuccess;
        const char* first_what = nullptr;
        auto note = [&](ncclResult_t e, const char* what) {
            if (e != ncclSuccess && first == ncclSuccess) {
                first = e;
                first_what = what;
            }
        };
        for (const GatherOp& op : plan) {
            if (first != ncclSuccess) break;
            const size_t i = (size_t)op.rank;
            ncclComm_t comm = static_cast<ncclComm_t>(comms_[i]);
            if (op.root < 0) {
                // in place: the send buffer is this rank's slot of the receive buffer
                note(r.AllGather(out_dev[i] + op.offset, out_dev[i], (size_t)op.floats, ncclFloat, comm, streams_[i]), "ncclAllGather");
            } else {
                note(r.Broadcast(out_dev[i] + op.offset, out_dev[i] + op.offset, (size_t)op.floats, ncclFloat, op.root, comm, streams_[i]),
                     "ncclBroadcast");
            }
        }
        note(r.GroupEnd(), "ncclGroupEnd");
        hipError_t sync_err = hipSuccess;
        for (size_t i = 0; i < n; ++i) {
            hipError_t e = hipSetDevice(replicas_[i]->device());
            if (e == hipSuccess) e = hipStreamSynchronize(streams_[i]);
            if (e != hipSuccess && sync_err == hipSuccess) sync_err = e;
        }
        nccl_check(r, first, first_what ? first_what : "rccl");
        hip_check(sync_err, "hipStreamSynchronize(all-gather)");
        return;
    }
    // Copy transport (a device listed twice, or no RCCL in the process): wait for every block, then each
    // destination pulls the blocks it does not own.
    for (size_t i = 0; i < n; ++i) {
        hip_check(hipSetDevice(replicas_[i]->device()), "hipSetDevice");
        hip_check(hipStreamSynchronize(streams_[i]), "hipStreamSynchronize(block)");
    }
    for (size_t dst = 0; dst < n; ++dst) {
        hip_check(hipSetDevice(replicas_[dst]->device()), "hipSetDevice");
        for (size_t src = 0; src < n; ++src) {
            if (src == dst || count[src] == 0 || out_dev[src] == out_dev[dst]) continue;
            const size_t bytes = (size_t)(count[src] * width) * sizeof(float);
            hip_check(hipMemcpyPeerAsync(out_dev[dst] + start[src] * width, replicas_[dst]->device(),
                                         out_dev[src] + start[src] * width, replicas_[src]->device(), bytes, streams_[dst]),
                      "hipMemcpyPeerAsync");
        }
    }
    for (size_t i = 0; i < n; ++i) {
        hip_check(hipSetDevice(replicas_[i]->device()), "hipSetDevice");
        hip_check(hipStreamSynchronize(streams_[i]), "hipStreamSynchronize(copy)");
    }
}

void EncoderGroup::allgather_embed(const uint32_t* const* ids_dev, const uint32_t* const* mask_dev,
                                   const uint32_t* const* type_dev, int64_t batch_total, int seq, PoolMode pool,
                                   bool normalize, float mask_value, float* const* out_dev)
{
    if (batch_total <= 0) return;
    std::lock_guard<std::mutex> lock(coll_mu_);
    DeviceGuard guard;
    const size_t n = replicas_.size();
    const int64_t H = config().hidden;
    if (streams_.empty()) {
        streams_.assign(n, nullptr);
        for (size_t i = 0; i < n; ++i) {
            hip_check(hipSetDevice(replicas_[i]->device()), "hipSetDevice");
            hip_check(hipStreamCreateWithFlags(&streams_[i], hipStreamNonBlocking), "hipStreamCreate");
        }
    }
    parallel(n, [&](size_t i) {
        int64_t start, count;
        shard(batch_total, n, i, &start, &count);
        if (count > 0)
            replicas_[i]->embed(ids_dev[i], mask_dev[i], type_dev ? type_dev[i] : nullptr, count, seq, pool, normalize,
                                mask_value, out_dev[i] + start * H, streams_[i]);
    });
    gather(out_dev, batch_total, H);
}

void EncoderGroup::allgather_logits(const uint32_t* const* ids_dev, const uint32_t* const* mask_dev,
                                    const uint32_t* const* type_dev, int64_t batch_total, int seq, float mask_value,
                                    float* const* out_dev)
{
    if (batch_total <= 0) return;
    std::lock_guard<std::mutex> lock(coll_mu_);
    DeviceGuard guard;
    const size_t n = replicas_.size();
    const int64_t L = config().num_labels;
    if (streams_.empty()) {
        streams_.assign(n, nullptr);
        for (size_t i = 0; i < n; ++i) {
            hip_check(hipSetDevice(replicas_[i]->device()), "hipSetDevice");
            hip_check(hipStreamCreateWithFlags(&streams_[i], hipStreamNonBlocking), "hipStreamCreate");
        }
    }
    parallel(n, [&](size_t i) {
        int64_t start, count;
        shard(batch_total, n, i, &start, &count);
        if (count > 0)
            replicas_[i]->logits(ids_dev[i], mask_dev[i], type_dev ? type_dev[i] : nullptr, count, seq, mask_value,
                                 out_dev[i] + start * L, streams_[i]);
    });
    gather(out_dev, batch_total, L);
}

}  // namespace kjarni
