// EncoderModel: a BERT-family encoder resident in HBM + the launch sequence of
// its forward pass.  The HIP analogue of the reference's CPU engine
// (CpuTransformerEncoder, crates/kjarni-transformers/src/cpu/encoder/
// transformer_encoder.rs:30-368) and of its wgpu plug-in point
// (GpuTransformerEncoder, cpu/encoder/gpu.rs:34-383).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <exception>
#include <functional>
#include <thread>
#include <condition_variable>
#include <cstdint>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "kernels.h"

namespace kjarni {

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct GpuUnavailable : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct InvalidDeviceList : std::runtime_error {  // KJARNI_HIP_DEVICES / an explicit device list that cannot be used
    using std::runtime_error::runtime_error;
};

void hip_check(hipError_t e, const char* what);

// Entry points select their model's device with hipSetDevice, which is state of the CALLING host thread: a caller that
// shares the thread with other HIP code (torch, its own kernels) must find its current device unchanged afterwards.
class DeviceGuard {
public:
    DeviceGuard() { ok_ = hipGetDevice(&prev_) == hipSuccess; }
    ~DeviceGuard() { if (ok_) (void)hipSetDevice(prev_); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;

private:
    int prev_ = 0;
    bool ok_ = false;
};
int visible_device_count();  // 0 when no HIP device / runtime is usable

struct EncoderConfig {
    std::string model_type;  // "bert" | "distilbert" | "roberta" | "distilroberta" | "mpnet" | "nomic_bert"
    int hidden = 0, layers = 0, heads = 0, inter = 0, vocab = 0, max_pos = 0, type_vocab = 0;
    int pos_offset = 0;
    float eps = 1e-12f;
    GemmEpilogue ffn_act = EPI_BIAS_GELU;
    bool gated_ffn = false;   // SwiGLU: down(silu(gate(x)) * up(x)) (transformer_encoder.rs:163-172)
    float rope_theta = 0.0f;  // > 0: no position table, Q / K rotated instead (transformer_encoder.rs:222-227)
    // classification head (cpu/encoder/classifier.rs:103-202), 0 = none
    int num_labels = 0;
    int head_kind = 0;  // 0 none, 1 dense+tanh (bert.pooler / classifier.dense), 2 dense+relu (pre_classifier), 3 classifier only
    std::vector<std::string> labels;  // id2label order
    std::string config_json;
};

struct DeviceLayer {
    float *wqkv = nullptr, *bqkv = nullptr, *wo = nullptr, *bo = nullptr, *ln1_g = nullptr,
          *ln1_b = nullptr, *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr,
          *ln2_g = nullptr, *ln2_b = nullptr;
    float* wg = nullptr;  // SwiGLU gate [inter, hidden]; w1 is then the up projection
};

// Per-kernel HIP-event timing on the launch stream (bench.py's roofline leg).
// Off by default; when on, every launch of the forward pass is bracketed by two
// events that are resolved at profile_end().
enum KernelKind : int {
    KK_EMBED_LN = 0, KK_GEMM_QKV, KK_ATTENTION, KK_GEMM_OUT, KK_LAYERNORM, KK_GEMM_FC1, KK_GEMM_FC2,
    KK_POOL, KK_HEAD, KK_ROPE, KK_COUNT
};

struct KernelStat {
    const char* kind;    // e.g. "gemm_fc1"
    const char* symbol;  // kernel function the launches ran
    uint64_t launches = 0;
    double total_ms = 0.0;
    double flops = 0.0;  // algorithmic FLOPs summed over the launches
    double bytes = 0.0;  // algorithmic bytes (operands read once + outputs written once)
};

// Activations + staging area of ONE call in flight.  A model owns a small pool of these: the weights are
// immutable after load, so calls on one handle from several host threads run concurrently, each on its own
// workspace (the reference's model types are Send + Sync and nothing serialises calls on a handle,
// crates/kjarni-ffi/src/lib.rs:25-32, kjarni-transformers/src/traits.rs:33).
struct Workspace {
    int64_t tokens = 0, sentences = 0;  // capacity of the activation buffers (0 whenever they are not all allocated)
    float *hidden = nullptr, *qkv = nullptr, *ctx = nullptr, *mid = nullptr, *feat = nullptr;
    float* split = nullptr;  // partial tiles of the mid-size GEMM route (gemm.hip), gemm_scratch_floats() floats
    size_t split_floats = 0;
    // ragged batches (packed rows, rowops.hip): row -> padded token index for `tokens` rows; per-chunk prefix sums of the
    // sentence lengths; the device-side length scan of the device-pointer entry points
    int32_t* tok_src = nullptr;
    int32_t* cu = nullptr;
    size_t cu_ints = 0;
    uint32_t* lens = nullptr;       // device: kept tokens per sentence (packing mode 2)
    uint32_t* lens_host = nullptr;  // pinned: its read-back, then the chunks' prefix sums on their way up
    size_t lens_cap = 0;
    void* stage = nullptr;  // ids / mask / types in, outputs back, for the host-pointer entry points
    size_t stage_bytes = 0;
    // small host-pointer calls (one sentence, a handful): inputs gathered in pinned host memory and sent as ONE copy, outputs
    // written by the last kernel straight into the pinned buffer (kPinnedStageBytes, device-mapped)
    uint8_t* pin = nullptr;
    uint8_t* pin_dev = nullptr;
    hipStream_t stream = nullptr;  // this workspace's own stream (host-pointer entry points run on it)
    hipEvent_t done = nullptr;     // recorded behind the last launch that used the buffers
    hipStream_t done_stream = nullptr;
    bool done_pending = false;
};

// How a ragged call is cut into chunks of packed rows (host side; built once per call).
struct PackPlan {
    struct Chunk {
        int64_t b0, nb;      // sentences [b0, b0 + nb) of the call
        int64_t tokens;      // kept tokens = packed rows of the chunk
        int max_len;         // the longest sentence
        double sum_len_sq;   // sum of len^2 (attention work)
        size_t cu_off;       // this chunk's nb + 1 prefix sums inside `cu`
    };
    bool packed = false;
    std::vector<Chunk> chunks;
    std::vector<int32_t> cu;
    int64_t max_tokens = 0, max_sentences = 0;
};
// The packed rows of one chunk as the kernels see them (device pointers).
struct PackView {
    const int32_t* cu;
    const int32_t* tok_src;
    int64_t tokens;
    int max_len;
    double sum_len_sq;
};

class EncoderModel {
public:
    // Loads <dir>/config.json + <dir>/model.safetensors onto `device`.
    static std::unique_ptr<EncoderModel> load(const std::string& dir, int device);
    ~EncoderModel();
    EncoderModel(const EncoderModel&) = delete;
    EncoderModel& operator=(const EncoderModel&) = delete;

    const EncoderConfig& config() const { return cfg_; }
    int device() const { return device_; }
    size_t weight_bytes() const { return weight_bytes_; }
    void set_chunk_tokens(int64_t t) { if (t > 0) chunk_tokens_ = t; }
    int64_t chunk_tokens() const { return chunk_tokens_; }
    // Ragged batches run over the kept tokens only (embed / logits).  0: never (every call takes the padded layout);
    // 1 (default): the host-pointer entry points, whose mask is already on the host; 2: the device-pointer entry points too --
    // those then read 4 bytes per sentence back and SYNCHRONISE `stream` once per call before the layers are enqueued (not
    // graph-capturable, blocks the calling thread behind whatever the stream holds), which is why it is not the default.
    void set_packing(int mode) { packing_ = mode < 0 ? 0 : (mode > 2 ? 2 : mode); }
    int packing() const { return packing_; }

    // All pointers are DEVICE pointers on this model's device; work is enqueued on `stream` and not synchronised
    // (exception: packing mode 2, above).  ids/mask/type_ids: u32 [batch, seq].
    // R1: get_hidden_states_batch_from_ids (cpu/encoder/traits.rs:66-139).
    void hidden_states(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids,
                       int64_t batch, int seq, float mask_value, float* out, hipStream_t stream);
    // R1 + R11: hidden states -> pool -> optional L2 (traits.rs:203-225,
    // sentence_encoder/model.rs:201-218).  out: [batch, hidden].
    void embed(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch,
               int seq, PoolMode pool, bool normalize, float mask_value, float* out,
               hipStream_t stream);
    // R12: forward_tokens + classification head -> logits [batch, num_labels]
    // (cross_encoder/model.rs:170-240, sequence_classifier/mod.rs:265-346).
    void logits(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch,
                int seq, float mask_value, float* out, hipStream_t stream);

    // kinds_mask: bit k enables KernelKind k (all ones = every launch of the forward pass).
    void profile_begin(uint32_t kinds_mask = 0xFFFFFFFFu);
    // Synchronises the device, resolves the events; returns KK_COUNT entries.
    std::vector<KernelStat> profile_end();

    // The same three on HOST pointers: ids / mask / type_ids are staged on the leased workspace's own stream,
    // the result is copied back and that stream (only) is synchronised before returning.
    void hidden_states_host(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch,
                            int seq, float mask_value, float* out);
    void embed_host(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch, int seq,
                    PoolMode pool, bool normalize, float mask_value, float* out);
    void logits_host(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch, int seq,
                     float mask_value, float* out);

    // Thread safety: every entry point above may be called concurrently from any number of host threads and on
    // any streams.  A call leases one workspace for its launches; when the pool (kMaxWorkspaces) is exhausted
    // the call waits for a lease.  A workspace handed from one stream to another is ordered by an event, so
    // two streams never see each other's activations.  The profiler is a single-caller tool.
    static constexpr int kMaxWorkspaces = 4;

    // Small host-pointer calls that arrive while another small call of the same kind is on the device are COMBINED: a call
    // becomes the leader when none is running (an uncontended call runs at once, exactly as before), and a leader takes every
    // compatible call that queued up meanwhile along as ONE packed batch -- N threads that each embed or classify one sentence
    // then cost one forward pass per round, not N x ~45 launches (the reference serialises nothing on a handle,
    // kjarni-ffi/src/lib.rs:25-32; on a 256-CU part a single-sentence forward occupies a fraction of the chip).  Results equal
    // the solo call's to rounding (the rows take the packed layout and another tile route).  On by default.
    void set_combining(bool on) { combining_ = on; }
    bool combining() const { return combining_; }
    // Mid-size host-pointer calls (2 304 .. 8 192 kept tokens) run as two halves on two workspaces / streams, the second on a
    // helper thread: one half's launch gaps, prologues and output bursts fall under the other's matrix work (encoder.cpp).
    void set_two_lanes(bool on) { two_lanes_ = on; }

private:
    EncoderModel() = default;
    float* upload(const std::vector<float>& host);

    class Lease {
    public:
        Lease(EncoderModel& m, hipStream_t stream, bool own_stream);
        ~Lease();
        Lease(const Lease&) = delete;
        Lease& operator=(const Lease&) = delete;
        Workspace& ws() { return *ws_; }
        hipStream_t stream() const { return stream_; }

    private:
        DeviceGuard guard_;  // first member: the caller's device comes back after everything else is released
        EncoderModel& m_;
        Workspace* ws_;
        hipStream_t stream_;
    };
    void reserve(Workspace& ws, int64_t tokens, int64_t sentences);
    void* reserve_stage(Workspace& ws, size_t bytes);
    template <class F>
    void run_host(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch, int seq,
                  size_t out_floats, float* out, bool out_written_once, F&& body);
    // Runs embeddings + all layers for `batch` sentences into hidden (device, [batch*seq, H]).
    // pack != null: the chunk's rows are its kept tokens only (mask is not read; hidden is [pack->tokens, H]).
    void forward_chunk(Workspace& ws, const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids,
                       int64_t batch, int seq, float mask_value, float* hidden, hipStream_t stream,
                       const PackView* pack = nullptr);
    void embed_on(Workspace& ws, const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch,
                  int seq, PoolMode pool, bool normalize, float mask_value, float* out, hipStream_t stream,
                  const PackPlan& plan);
    void logits_on(Workspace& ws, const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch,
                   int seq, float mask_value, float* out, hipStream_t stream, const PackPlan& plan);
    // Decides whether the call runs on packed rows and, if so, cuts it into chunks and uploads the prefix sums.
    // mask_host: the same mask on the host when the caller has it (no device round trip), else null.
    void plan_packing(Workspace& ws, const uint32_t* mask_dev, const uint32_t* mask_host, int64_t batch, int seq,
                      hipStream_t stream, PackPlan& plan);
    int64_t sentences_per_chunk(int seq) const;
    // LayerNorm folded into the residual GEMMs' epilogue when the kernel covers this model's row width.
    bool fuse_layernorm() const;

    struct PendingEvent {
        int kind;
        hipEvent_t start, stop;
    };
    hipEvent_t prof_start(int kind, hipStream_t stream, double flops, double bytes);  // null: not timed
    void prof_stop(hipEvent_t stop, hipStream_t stream);
    std::mutex prof_mu_;
    bool prof_on_ = false;
    uint32_t prof_mask_ = 0xFFFFFFFFu;
    std::vector<PendingEvent> prof_pending_;
    std::vector<hipEvent_t> prof_pool_;
    KernelStat prof_stats_[KK_COUNT];

    EncoderConfig cfg_;
    int device_ = 0;
    size_t weight_bytes_ = 0;
    std::atomic<int64_t> chunk_tokens_{262144};
    std::atomic<int> packing_{1};
    std::atomic<bool> combining_{false};  // opt-in (kjarni_hip_encoder_set_combining / KJARNI_HIP_COMBINE=1)

    // ---- combining of concurrent small host-pointer calls (encoder.cpp, "call combining") ----
    struct CombineReq {
        int kind;  // 0 embed, 1 logits
        const uint32_t *ids, *mask, *type_ids;
        int64_t batch;
        int seq;
        int pool;
        bool normalize;
        float mask_value;
        float* out;
        size_t out_per_row;
        bool done = false;
        bool taken = false;  // in some leader's batch (its owner then only waits)
        std::exception_ptr error;
        bool compatible(const CombineReq& o) const
        {
            return kind == o.kind && pool == o.pool && normalize == o.normalize && mask_value == o.mask_value &&
                   (type_ids != nullptr) == (o.type_ids != nullptr);
        }
    };
    void submit_small(CombineReq& req);
    void run_combined(const std::vector<CombineReq*>& reqs);
    void embed_host_now(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch, int seq, PoolMode pool,
                        bool normalize, float mask_value, float* out);
    void logits_host_now(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch, int seq, float mask_value,
                         float* out);
    // ---- two lanes for mid-size host-pointer calls (encoder.cpp, "two lanes") ----
    // A persistent helper thread that runs the second half of a call on a workspace (and stream) of its own.
    class Lane {
    public:
        ~Lane();
        bool try_begin(std::function<void()> task);  // false: the lane is busy with another call (the caller runs alone)
        void wait();                                  // joins the task, rethrows what it threw
    private:
        void loop();
        std::thread thread_;
        std::mutex mu_;
        std::condition_variable cv_;
        std::function<void()> task_;
        std::exception_ptr error_;
        bool busy_ = false, has_task_ = false, stop_ = false;
    };
    Lane lane_[2];  // (a call runs as up to three parts: the caller's + two helpers')
    std::atomic<bool> two_lanes_{true};
    template <class F>
    bool run_two_lanes(const uint32_t* mask, int64_t batch, int seq, F&& half);
    std::mutex combine_mu_;
    std::condition_variable combine_cv_;
    std::vector<CombineReq*> combine_queue_;
    int combine_leaders_ = 0;
    std::vector<void*> allocs_;

    float *word_ = nullptr, *pos_ = nullptr, *type_ = nullptr, *emb_ln_g_ = nullptr,
          *emb_ln_b_ = nullptr;
    float *rope_cos_ = nullptr, *rope_sin_ = nullptr;  // [max_pos, head_dim]
    std::vector<DeviceLayer> layers_;
    float *head_dense_w_ = nullptr, *head_dense_b_ = nullptr, *head_cls_w_ = nullptr,
          *head_cls_b_ = nullptr;

    // workspace pool
    std::mutex ws_mu_;
    std::condition_variable ws_cv_;
    std::vector<std::unique_ptr<Workspace>> ws_all_;
    std::vector<Workspace*> ws_free_;  // LIFO: a single-threaded caller keeps getting the same (warm) workspace
};

}  // namespace kjarni
