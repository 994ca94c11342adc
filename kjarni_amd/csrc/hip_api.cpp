// Token-level C ABI (include/kjarni_hip.h).
#include <mutex>
#include <sys/stat.h>

#include <cmath>
#include <limits>
#include <vector>

#include "../../include/kjarni_hip.h"
#include "ffi_common.h"
#include "group.h"
#include "tuning.h"

using namespace kjarni;

struct KjarniHipEncoder {
    std::unique_ptr<EncoderModel> model;
};

namespace {

bool file_exists(const std::string& p)
{
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

// cpu/strategy.rs:43-44 (scratch buffers iff tokens <= 1 or tokens >= 1000) decides
// between the -inf (no-alloc) and -1e9 (alloc) mask fills on the embed path.
float resolve_fill(KjarniHipMaskFill fill, int64_t tokens, bool logits_path)
{
    const float neg_inf = -std::numeric_limits<float>::infinity();
    switch (fill) {
    case KJARNI_HIP_MASK_NEG_1E9: return -1e9f;
    case KJARNI_HIP_MASK_NEG_INF: return neg_inf;
    default: break;
    }
    if (logits_path) return -1e9f;  // forward_tokens: always encoder.forward (alloc path)
    return (tokens <= 1 || tokens >= 1000) ? neg_inf : -1e9f;
}

PoolMode pool_mode(KjarniHipPooling p)
{
    switch (p) {
    case KJARNI_HIP_POOL_MEAN: return POOL_MEAN;
    case KJARNI_HIP_POOL_CLS: return POOL_CLS;
    case KJARNI_HIP_POOL_MAX: return POOL_MAX;
    case KJARNI_HIP_POOL_LAST_TOKEN: return POOL_LAST;
    }
    throw InvalidConfig("unknown pooling strategy");
}

void check_shape(const EncoderModel& m, int64_t batch, int32_t seq)
{
    if (batch < 0 || seq < 0) throw InvalidConfig("negative batch or sequence length");
    if (seq > m.config().max_pos)
        throw InvalidConfig("sequence length " + std::to_string(seq) + " exceeds max_position_embeddings " +
                            std::to_string(m.config().max_pos));
}

struct DeviceBuf {
    void* p = nullptr;
    explicit DeviceBuf(size_t bytes) { hip_check(hipMalloc(&p, bytes ? bytes : 4), "hipMalloc"); }
    ~DeviceBuf()
    {
        if (p) (void)hipFree(p);
    }
    DeviceBuf(const DeviceBuf&) = delete;
    DeviceBuf& operator=(const DeviceBuf&) = delete;
};

void use_device(int32_t device)
{
    const int n = visible_device_count();
    if (n <= 0) throw GpuUnavailable("no HIP device is visible");
    if (device < 0 || device >= n) throw GpuUnavailable("HIP device index out of range");
    hip_check(hipSetDevice(device), "hipSetDevice");
}

}  // namespace

KJARNI_EXPORT int32_t kjarni_hip_device_count(void) { return visible_device_count(); }

KJARNI_EXPORT KjarniErrorCode kjarni_hip_encoder_load(const char* model_dir, int32_t device,
                                                      KjarniHipEncoder** out)
{
    if (!model_dir || !out) return KJARNI_ERROR_NULL_POINTER;
    *out = nullptr;
    return guarded(KJARNI_ERROR_LOAD_FAILED, [&] {
        const std::string dir(model_dir);
        if (!file_exists(dir + "/config.json") ||
            !(file_exists(dir + "/model.safetensors") || file_exists(dir + "/model.safetensors.index.json")))
            throw ModelNotFound("model files not found in '" + dir +
                                "' (need config.json and model.safetensors)");
        auto h = std::make_unique<KjarniHipEncoder>();
        h->model = EncoderModel::load(dir, device);
        *out = h.release();
    });
}

KJARNI_EXPORT void kjarni_hip_encoder_free(KjarniHipEncoder* enc) { delete enc; }

KJARNI_EXPORT int32_t kjarni_hip_encoder_hidden_size(const KjarniHipEncoder* e) { return e ? e->model->config().hidden : 0; }
KJARNI_EXPORT int32_t kjarni_hip_encoder_num_layers(const KjarniHipEncoder* e) { return e ? e->model->config().layers : 0; }
KJARNI_EXPORT int32_t kjarni_hip_encoder_max_seq_len(const KjarniHipEncoder* e) { return e ? e->model->config().max_pos : 0; }
KJARNI_EXPORT int32_t kjarni_hip_encoder_vocab_size(const KjarniHipEncoder* e) { return e ? e->model->config().vocab : 0; }
KJARNI_EXPORT int32_t kjarni_hip_encoder_num_labels(const KjarniHipEncoder* e) { return e ? e->model->config().num_labels : 0; }
KJARNI_EXPORT int32_t kjarni_hip_encoder_device(const KjarniHipEncoder* e) { return e ? e->model->device() : -1; }

KJARNI_EXPORT KjarniErrorCode kjarni_hip_encoder_set_chunk_tokens(KjarniHipEncoder* enc, int64_t tokens)
{
    if (!enc) return KJARNI_ERROR_NULL_POINTER;
    enc->model->set_chunk_tokens(tokens);
    return KJARNI_OK;
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_encoder_set_packing(KjarniHipEncoder* enc, int32_t on)
{
    if (!enc) return KJARNI_ERROR_NULL_POINTER;
    enc->model->set_packing((int)on);
    return KJARNI_OK;
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_encoder_set_combining(KjarniHipEncoder* enc, int32_t on)
{
    if (!enc) return KJARNI_ERROR_NULL_POINTER;
    enc->model->set_combining(on != 0);
    return KJARNI_OK;
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_encoder_set_two_lanes(KjarniHipEncoder* enc, int32_t on)
{
    if (!enc) return KJARNI_ERROR_NULL_POINTER;
    enc->model->set_two_lanes(on != 0);
    return KJARNI_OK;
}

KJARNI_EXPORT int32_t kjarni_hip_set_f32_on_bf16(int32_t on)
{
    const int32_t before = kjarni::get_f32_on_bf16() ? 1 : 0;
    kjarni::set_f32_on_bf16(on != 0);
    return before;
}

KJARNI_EXPORT int32_t kjarni_hip_get_f32_on_bf16(void) { return kjarni::get_f32_on_bf16() ? 1 : 0; }

KJARNI_EXPORT KjarniErrorCode kjarni_hip_clock_probe(uint64_t* out_dev, uint32_t spin_us, void* stream)
{
    if (!out_dev) return KJARNI_ERROR_NULL_POINTER;
    const uint32_t us = spin_us == 0 ? 20u : (spin_us > 10000u ? 10000u : spin_us);
    return kjarni::launch_clock_probe(out_dev, us * 100u, (hipStream_t)stream) == hipSuccess ? KJARNI_OK
                                                                                             : KJARNI_ERROR_INFERENCE_FAILED;
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_selftest_reductions(int32_t device, uint32_t waves, uint32_t seed, uint32_t* mismatches_out)
{
    if (!mismatches_out) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        if (waves == 0 || waves > (1u << 22)) throw InvalidConfig("1 .. 4 194 304 waves");
        use_device(device);
        DeviceBuf d(4);
        hip_check(hipMemset(d.p, 0, 4), "memset");
        hip_check(kjarni::launch_reduction_selftest((unsigned*)d.p, waves, seed, nullptr), "reduction self-test");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        hip_check(hipMemcpy(mismatches_out, d.p, 4, hipMemcpyDeviceToHost), "D2H");
    });
}

namespace {
std::mutex g_measurement_mu;
constexpr int kMaxMeasurementDevices = 64;
hipStream_t g_measurement_stream[kMaxMeasurementDevices] = {};   // one per device, made on first use on that device
}  // namespace

KJARNI_EXPORT void* kjarni_hip_measurement_stream(void)
{
    // one non-blocking stream per DEVICE, made on first use with that device current (a measurement aid: see
    // kjarni_hip_clock_trace); a caller whose current device differs gets that device's own stream, never another's
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxMeasurementDevices) return nullptr;
    std::lock_guard<std::mutex> lock(g_measurement_mu);
    if (!g_measurement_stream[dev] && hipStreamCreateWithFlags(&g_measurement_stream[dev], hipStreamNonBlocking) != hipSuccess)
        g_measurement_stream[dev] = nullptr;
    return g_measurement_stream[dev];
}

KJARNI_EXPORT void kjarni_hip_measurement_stream_release(void)
{
    // (a process has a handful of hardware queues and HIP deals its streams over them: while this stream exists, one of the
    // encoder's own streams may share a queue with it -- a 64-sentence call, three parts on three streams, 1.72 -> 2.07 ms)
    std::lock_guard<std::mutex> lock(g_measurement_mu);
    int before = -1;
    (void)hipGetDevice(&before);
    for (int dev = 0; dev < kMaxMeasurementDevices; ++dev) {
        if (!g_measurement_stream[dev]) continue;
        (void)hipSetDevice(dev);
        (void)hipStreamSynchronize(g_measurement_stream[dev]);
        (void)hipStreamDestroy(g_measurement_stream[dev]);
        g_measurement_stream[dev] = nullptr;
    }
    if (before >= 0) (void)hipSetDevice(before);
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_clock_trace(uint64_t* out_dev, uint32_t samples, uint32_t window_us, void* stream)
{
    if (!out_dev) return KJARNI_ERROR_NULL_POINTER;
    // (one launch stays below ten seconds: 1 .. 4096 windows of 10 us .. 1 s)
    if (samples == 0 || samples > 4096u || window_us < 10u || window_us > 1000000u || (uint64_t)samples * window_us > 10000000ull)
        return KJARNI_ERROR_INVALID_CONFIG;
    return kjarni::launch_clock_trace(out_dev, samples, window_us * 100u, (hipStream_t)stream) == hipSuccess ? KJARNI_OK
                                                                                                           : KJARNI_ERROR_INFERENCE_FAILED;
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_encoder_hidden_states(KjarniHipEncoder* enc, const uint32_t* ids_dev,
                                                               const uint32_t* mask_dev,
                                                               const uint32_t* type_ids_dev, int64_t batch,
                                                               int32_t seq, KjarniHipMaskFill fill,
                                                               float* hidden_out_dev, void* stream)
{
    if (!enc || !ids_dev || !mask_dev || !hidden_out_dev) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        check_shape(*enc->model, batch, seq);
        enc->model->hidden_states(ids_dev, mask_dev, type_ids_dev, batch, seq,
                                  resolve_fill(fill, batch * seq, false), hidden_out_dev,
                                  static_cast<hipStream_t>(stream));
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_encoder_embed(KjarniHipEncoder* enc, const uint32_t* ids_dev,
                                                       const uint32_t* mask_dev, const uint32_t* type_ids_dev,
                                                       int64_t batch, int32_t seq, KjarniHipPooling pooling,
                                                       int32_t normalize, KjarniHipMaskFill fill,
                                                       float* out_dev, void* stream)
{
    if (!enc || !ids_dev || !mask_dev || !out_dev) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        check_shape(*enc->model, batch, seq);
        enc->model->embed(ids_dev, mask_dev, type_ids_dev, batch, seq, pool_mode(pooling), normalize != 0,
                          resolve_fill(fill, batch * seq, false), out_dev, static_cast<hipStream_t>(stream));
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_encoder_logits(KjarniHipEncoder* enc, const uint32_t* ids_dev,
                                                        const uint32_t* mask_dev, const uint32_t* type_ids_dev,
                                                        int64_t batch, int32_t seq, KjarniHipMaskFill fill,
                                                        float* logits_out_dev, void* stream)
{
    if (!enc || !ids_dev || !mask_dev || !logits_out_dev) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        check_shape(*enc->model, batch, seq);
        enc->model->logits(ids_dev, mask_dev, type_ids_dev, batch, seq, resolve_fill(fill, batch * seq, true),
                           logits_out_dev, static_cast<hipStream_t>(stream));
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_encoder_hidden_states_host(KjarniHipEncoder* enc, const uint32_t* ids,
                                                                    const uint32_t* mask,
                                                                    const uint32_t* type_ids, int64_t batch,
                                                                    int32_t seq, KjarniHipMaskFill fill,
                                                                    float* hidden_out)
{
    if (!enc || !ids || !mask || !hidden_out) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        check_shape(*enc->model, batch, seq);
        enc->model->hidden_states_host(ids, mask, type_ids, batch, seq, resolve_fill(fill, batch * seq, false), hidden_out);
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_encoder_embed_host(KjarniHipEncoder* enc, const uint32_t* ids,
                                                            const uint32_t* mask, const uint32_t* type_ids,
                                                            int64_t batch, int32_t seq, KjarniHipPooling pooling,
                                                            int32_t normalize, KjarniHipMaskFill fill, float* out)
{
    if (!enc || !ids || !mask || !out) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        check_shape(*enc->model, batch, seq);
        enc->model->embed_host(ids, mask, type_ids, batch, seq, pool_mode(pooling), normalize != 0,
                               resolve_fill(fill, batch * seq, false), out);
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_encoder_logits_host(KjarniHipEncoder* enc, const uint32_t* ids,
                                                             const uint32_t* mask, const uint32_t* type_ids,
                                                             int64_t batch, int32_t seq, KjarniHipMaskFill fill,
                                                             float* logits_out)
{
    if (!enc || !ids || !mask || !logits_out) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        check_shape(*enc->model, batch, seq);
        enc->model->logits_host(ids, mask, type_ids, batch, seq, resolve_fill(fill, batch * seq, true), logits_out);
    });
}

// ---- several devices in one process (group.h) ----------------------------------------------------------

struct KjarniHipEncoderGroup {
    std::unique_ptr<EncoderGroup> group;
};

KJARNI_EXPORT KjarniErrorCode kjarni_hip_group_load(const char* model_dir, const int32_t* devices, size_t n_devices,
                                                    KjarniHipEncoderGroup** out)
{
    if (!model_dir || !out || (n_devices > 0 && !devices)) return KJARNI_ERROR_NULL_POINTER;
    *out = nullptr;
    return guarded(KJARNI_ERROR_LOAD_FAILED, [&] {
        const std::string dir(model_dir);
        if (!file_exists(dir + "/config.json") ||
            !(file_exists(dir + "/model.safetensors") || file_exists(dir + "/model.safetensors.index.json")))
            throw ModelNotFound("model files not found in '" + dir + "' (need config.json and model.safetensors)");
        std::vector<int> devs(devices, devices + n_devices);
        if (devs.empty()) devs = devices_from_env();
        auto h = std::make_unique<KjarniHipEncoderGroup>();
        h->group = EncoderGroup::load(dir, devs);
        *out = h.release();
    });
}

KJARNI_EXPORT void kjarni_hip_group_free(KjarniHipEncoderGroup* g) { delete g; }
KJARNI_EXPORT size_t kjarni_hip_group_size(const KjarniHipEncoderGroup* g) { return g ? g->group->size() : 0; }
KJARNI_EXPORT int32_t kjarni_hip_group_device(const KjarniHipEncoderGroup* g, size_t i)
{
    return (g && i < g->group->size()) ? g->group->device(i) : -1;
}
KJARNI_EXPORT int32_t kjarni_hip_group_hidden_size(const KjarniHipEncoderGroup* g) { return g ? g->group->config().hidden : 0; }
KJARNI_EXPORT int32_t kjarni_hip_group_num_labels(const KjarniHipEncoderGroup* g) { return g ? g->group->config().num_labels : 0; }
KJARNI_EXPORT const char* kjarni_hip_group_transport(KjarniHipEncoderGroup* g)
{
    if (!g) return "";
    const char* t = "";
    (void)guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] { t = g->group->transport(); });
    return t;
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_group_shard(const KjarniHipEncoderGroup* g, int64_t rows, size_t i, int64_t* start_out,
                                                     int64_t* count_out)
{
    if (!g || !start_out || !count_out) return KJARNI_ERROR_NULL_POINTER;
    if (i >= g->group->size() || rows < 0) return KJARNI_ERROR_INVALID_CONFIG;
    EncoderGroup::shard(rows, g->group->size(), i, start_out, count_out);
    return KJARNI_OK;
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_group_gather_plan(int64_t rows, size_t n, int64_t width, KjarniHipGatherOp* ops_out, size_t cap,
                                                           size_t* count_out)
{
    if (!count_out || (cap && !ops_out)) return KJARNI_ERROR_NULL_POINTER;
    if (rows < 0 || n == 0 || n > 4096 || width <= 0) return KJARNI_ERROR_INVALID_CONFIG;
    const std::vector<EncoderGroup::GatherOp> plan = EncoderGroup::gather_plan(rows, n, width);
    *count_out = plan.size();
    for (size_t i = 0; i < plan.size() && i < cap; ++i) ops_out[i] = KjarniHipGatherOp{plan[i].rank, plan[i].root, plan[i].offset, plan[i].floats};
    return KJARNI_OK;
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_group_embed_host(KjarniHipEncoderGroup* g, const uint32_t* ids, const uint32_t* mask,
                                                          const uint32_t* type_ids, int64_t batch, int32_t seq,
                                                          KjarniHipPooling pooling, int32_t normalize, KjarniHipMaskFill fill,
                                                          float* out)
{
    if (!g || !ids || !mask || !out) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        check_shape(g->group->replica(0), batch, seq);
        g->group->embed_host(ids, mask, type_ids, batch, seq, pool_mode(pooling), normalize != 0,
                             resolve_fill(fill, batch * seq, false), out);
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_group_logits_host(KjarniHipEncoderGroup* g, const uint32_t* ids, const uint32_t* mask,
                                                           const uint32_t* type_ids, int64_t batch, int32_t seq,
                                                           KjarniHipMaskFill fill, float* logits_out)
{
    if (!g || !ids || !mask || !logits_out) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        check_shape(g->group->replica(0), batch, seq);
        if (g->group->config().head_kind == 0) throw InvalidConfig("model has no classification head");
        g->group->logits_host(ids, mask, type_ids, batch, seq, resolve_fill(fill, batch * seq, true), logits_out);
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_group_embed_allgather(KjarniHipEncoderGroup* g, const uint32_t* const* ids_dev,
                                                               const uint32_t* const* mask_dev,
                                                               const uint32_t* const* type_ids_dev, int64_t batch_total,
                                                               int32_t seq, KjarniHipPooling pooling, int32_t normalize,
                                                               KjarniHipMaskFill fill, float* const* out_dev)
{
    if (!g || !ids_dev || !mask_dev || !out_dev) return KJARNI_ERROR_NULL_POINTER;
    for (size_t i = 0; i < g->group->size(); ++i)
        if (!ids_dev[i] || !mask_dev[i] || !out_dev[i]) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        check_shape(g->group->replica(0), batch_total, seq);
        g->group->allgather_embed(ids_dev, mask_dev, type_ids_dev, batch_total, seq, pool_mode(pooling), normalize != 0,
                                  resolve_fill(fill, batch_total * seq, false), out_dev);
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_group_logits_allgather(KjarniHipEncoderGroup* g, const uint32_t* const* ids_dev,
                                                                const uint32_t* const* mask_dev,
                                                                const uint32_t* const* type_ids_dev, int64_t batch_total,
                                                                int32_t seq, KjarniHipMaskFill fill, float* const* logits_out_dev)
{
    if (!g || !ids_dev || !mask_dev || !logits_out_dev) return KJARNI_ERROR_NULL_POINTER;
    for (size_t i = 0; i < g->group->size(); ++i)
        if (!ids_dev[i] || !mask_dev[i] || !logits_out_dev[i]) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        check_shape(g->group->replica(0), batch_total, seq);
        if (g->group->config().head_kind == 0) throw InvalidConfig("model has no classification head");
        g->group->allgather_logits(ids_dev, mask_dev, type_ids_dev, batch_total, seq, resolve_fill(fill, batch_total * seq, true),
                                   logits_out_dev);
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_encoder_profile_begin(KjarniHipEncoder* enc)
{
    if (!enc) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] { enc->model->profile_begin(); });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_encoder_profile_begin_kinds(KjarniHipEncoder* enc, uint32_t kinds_mask)
{
    if (!enc) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] { enc->model->profile_begin(kinds_mask); });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_encoder_profile_end(KjarniHipEncoder* enc, KjarniHipKernelStat* stats_out,
                                                             size_t capacity, size_t* count_out)
{
    if (!enc || !stats_out || !count_out) return KJARNI_ERROR_NULL_POINTER;
    *count_out = 0;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        const std::vector<KernelStat> st = enc->model->profile_end();
        size_t n = 0;
        for (const KernelStat& k : st) {
            if (n >= capacity) break;
            stats_out[n].kind = k.kind;
            stats_out[n].symbol = k.symbol;
            stats_out[n].launches = k.launches;
            stats_out[n].total_ms = k.total_ms;
            stats_out[n].flops = k.flops;
            stats_out[n].bytes = k.bytes;
            ++n;
        }
        *count_out = n;
    });
}

// ---- single operators -------------------------------------------------------------

namespace {

template <class F>
void time_launches(int32_t iters, float* ms_out, F&& launch)
{
    launch();  // the run whose result is returned
    if (iters > 0) {
        hipEvent_t a, b;
        hip_check(hipEventCreate(&a), "hipEventCreate");
        hip_check(hipEventCreate(&b), "hipEventCreate");
        hip_check(hipEventRecord(a, nullptr), "hipEventRecord");
        for (int32_t i = 0; i < iters; ++i) launch();
        hip_check(hipEventRecord(b, nullptr), "hipEventRecord");
        hip_check(hipEventSynchronize(b), "hipEventSynchronize");
        float ms = 0.0f;
        hip_check(hipEventElapsedTime(&ms, a, b), "hipEventElapsedTime");
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
        if (ms_out) *ms_out = ms / (float)iters;
    }
    hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
}

}  // namespace

KJARNI_EXPORT KjarniErrorCode kjarni_hip_op_linear(int32_t device, const float* x, const float* w, const float* bias,
                                                   const float* residual, int64_t m, int32_t k, int32_t n,
                                                   KjarniHipEpilogue epilogue, float* y, int32_t iters, float* ms_out)
{
    if (!x || !w || !y) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        if (m < 0 || k <= 0 || n <= 0) throw InvalidConfig("invalid GEMM dimensions");
        if ((epilogue == KJARNI_HIP_EPI_BIAS_RESIDUAL || epilogue == KJARNI_HIP_EPI_BIAS_MUL_SILU) && !residual)
            throw InvalidConfig("residual epilogue without residual");
        use_device(device);
        if (m == 0) return;
        const size_t xb = (size_t)m * k * 4, wb = (size_t)n * k * 4, yb = (size_t)m * n * 4;
        // A guard band of one tile's rows behind the output: the tile kernels compute whole 128-row tiles and leave the rows past m
        // to a bounds check (a predicate, or the extent of a buffer descriptor) -- a wrong one would write here, silently.
        const size_t guard_floats = (size_t)128 * n;
        DeviceBuf xd(xb), wd(wb), bd((size_t)n * 4), rd(residual ? yb : 4), yd(yb + guard_floats * 4);
        hip_check(hipMemcpy(xd.p, x, xb, hipMemcpyHostToDevice), "H2D x");
        hip_check(hipMemcpy(wd.p, w, wb, hipMemcpyHostToDevice), "H2D w");
        if (bias) hip_check(hipMemcpy(bd.p, bias, (size_t)n * 4, hipMemcpyHostToDevice), "H2D bias");
        if (residual) hip_check(hipMemcpy(rd.p, residual, yb, hipMemcpyHostToDevice), "H2D residual");
        constexpr uint32_t kGuard = 0x7fc0beefu;  // (a NaN no epilogue produces)
        hip_check(hipMemsetD32((hipDeviceptr_t)((char*)yd.p + yb), (int)kGuard, guard_floats), "guard band");
        // the scratch slab the encoder lends its GEMMs, so that the op takes the route the model takes at this row count
        const size_t sf = gemm_scratch_floats(m, n);
        DeviceBuf sd(sf * 4);
        const GemmScratch sc{(float*)sd.p, sf};
        time_launches(iters, ms_out, [&] {
            hip_check(launch_gemm((const float*)xd.p, k, (const float*)wd.p, bias ? (const float*)bd.p : nullptr,
                                  residual ? (const float*)rd.p : nullptr, n, (float*)yd.p, n, m, n, k,
                                  (GemmEpilogue)epilogue, nullptr, sc),
                      "gemm");
        });
        hip_check(hipMemcpy(y, yd.p, yb, hipMemcpyDeviceToHost), "D2H y");
        std::vector<uint32_t> guard(guard_floats);
        hip_check(hipMemcpy(guard.data(), (char*)yd.p + yb, guard_floats * 4, hipMemcpyDeviceToHost), "D2H guard band");
        for (size_t i = 0; i < guard_floats; ++i)
            if (guard[i] != kGuard) throw std::runtime_error("GEMM wrote past the last output row (guard band touched at float " + std::to_string(i) + ")");
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_op_linear_bf16_weights(int32_t device, const float* x, const uint16_t* w_bf16,
                                                                const float* bias, const float* residual, int64_t m, int32_t k,
                                                                int32_t n, KjarniHipEpilogue epilogue, float* y, int32_t iters,
                                                                float* ms_out)
{
    if (!x || !w_bf16 || !y) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        if (m < 0 || k <= 0 || n <= 0 || n % 128 != 0 || k % 64 != 0) throw InvalidConfig("invalid GEMM dimensions (n % 128, k % 64)");
        if ((epilogue == KJARNI_HIP_EPI_BIAS_RESIDUAL || epilogue == KJARNI_HIP_EPI_BIAS_MUL_SILU) && !residual)
            throw InvalidConfig("residual epilogue without residual");
        use_device(device);
        if (m == 0) return;
        const size_t xb = (size_t)m * k * 4, wb = (size_t)n * k * 2, yb = (size_t)m * n * 4;
        DeviceBuf xd(xb), wd(wb), bd((size_t)n * 4), rd(residual ? yb : 4), yd(yb);
        hip_check(hipMemcpy(xd.p, x, xb, hipMemcpyHostToDevice), "H2D x");
        hip_check(hipMemcpy(wd.p, w_bf16, wb, hipMemcpyHostToDevice), "H2D w");
        if (bias) hip_check(hipMemcpy(bd.p, bias, (size_t)n * 4, hipMemcpyHostToDevice), "H2D bias");
        if (residual) hip_check(hipMemcpy(rd.p, residual, yb, hipMemcpyHostToDevice), "H2D residual");
        time_launches(iters, ms_out, [&] {
            hip_check(launch_gemm_bf16_weights((const float*)xd.p, k, wd.p, bias ? (const float*)bd.p : nullptr,
                                               residual ? (const float*)rd.p : nullptr, n, (float*)yd.p, n, m, n, k,
                                               (GemmEpilogue)epilogue, nullptr),
                      "gemm (bf16 weights)");
        });
        hip_check(hipMemcpy(y, yd.p, yb, hipMemcpyDeviceToHost), "D2H y");
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_op_attention(int32_t device, const float* qkv, const uint32_t* mask,
                                                      int64_t batch, int32_t seq, int32_t heads, int32_t head_dim,
                                                      float mask_value, float* ctx, int32_t iters, float* ms_out)
{
    if (!qkv || !ctx) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        if (batch < 0 || seq <= 0 || heads <= 0 || head_dim <= 0) throw InvalidConfig("invalid attention dimensions");
        use_device(device);
        if (batch == 0) return;
        const size_t T = (size_t)batch * seq, H = (size_t)heads * head_dim;
        DeviceBuf qd(T * 3 * H * 4), md(T * 4), cd(T * H * 4);
        hip_check(hipMemcpy(qd.p, qkv, T * 3 * H * 4, hipMemcpyHostToDevice), "H2D qkv");
        if (mask) hip_check(hipMemcpy(md.p, mask, T * 4, hipMemcpyHostToDevice), "H2D mask");
        time_launches(iters, ms_out, [&] {
            hip_check(launch_attention((const float*)qd.p, mask ? (const uint32_t*)md.p : nullptr, batch, seq, heads,
                                       head_dim, mask_value, (float*)cd.p, nullptr),
                      "attention");
        });
        hip_check(hipMemcpy(ctx, cd.p, T * H * 4, hipMemcpyDeviceToHost), "D2H ctx");
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_op_attention_biased(int32_t device, const float* qkv, const uint32_t* mask,
                                                             const float* position_bias, int32_t bias_seq, int64_t batch,
                                                             int32_t seq, int32_t heads, int32_t head_dim, int32_t scale_qk,
                                                             float mask_value, float* ctx)
{
    if (!qkv || !ctx) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        if (batch < 0 || seq <= 0 || heads <= 0 || head_dim <= 0) throw InvalidConfig("invalid attention dimensions");
        if (position_bias && bias_seq < seq) throw InvalidConfig("position bias is smaller than the sequence");
        use_device(device);
        if (batch == 0) return;
        const size_t T = (size_t)batch * seq, H = (size_t)heads * head_dim;
        const size_t bias_bytes = position_bias ? (size_t)heads * bias_seq * bias_seq * 4 : 4;
        DeviceBuf qd(T * 3 * H * 4), md(T * 4), cd(T * H * 4), bd(bias_bytes);
        hip_check(hipMemcpy(qd.p, qkv, T * 3 * H * 4, hipMemcpyHostToDevice), "H2D qkv");
        if (mask) hip_check(hipMemcpy(md.p, mask, T * 4, hipMemcpyHostToDevice), "H2D mask");
        if (position_bias) hip_check(hipMemcpy(bd.p, position_bias, bias_bytes, hipMemcpyHostToDevice), "H2D position bias");
        hip_check(launch_attention_biased((const float*)qd.p, mask ? (const uint32_t*)md.p : nullptr,
                                          position_bias ? (const float*)bd.p : nullptr, bias_seq, batch, seq, heads, head_dim,
                                          scale_qk != 0, mask_value, (float*)cd.p, nullptr),
                  "attention (position bias)");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        hip_check(hipMemcpy(ctx, cd.p, T * H * 4, hipMemcpyDeviceToHost), "D2H ctx");
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_op_pool(int32_t device, const float* hidden_states, const uint32_t* mask, int64_t batch,
                                                 int32_t seq, int32_t hidden, KjarniHipPooling pooling, int32_t normalize, float* out)
{
    if (!hidden_states || !out) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        if (batch < 0 || seq <= 0 || hidden <= 0 || hidden > 1024) throw InvalidConfig("invalid pooling dimensions");
        const PoolMode mode = pool_mode(pooling);   // (throws InvalidConfig for a value outside the enum)
        use_device(device);
        if (batch == 0) return;
        const size_t T = (size_t)batch * seq;
        DeviceBuf hd(T * hidden * 4), md(T * 4), od((size_t)batch * hidden * 4);
        hip_check(hipMemcpy(hd.p, hidden_states, T * hidden * 4, hipMemcpyHostToDevice), "H2D hidden states");
        std::vector<uint32_t> ones;
        if (!mask) ones.assign(T, 1u);
        hip_check(hipMemcpy(md.p, mask ? mask : ones.data(), T * 4, hipMemcpyHostToDevice), "H2D mask");
        hip_check(launch_pool((const float*)hd.p, (const uint32_t*)md.p, batch, seq, hidden, mode, normalize,
                              (float*)od.p, nullptr),
                  "pool");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        hip_check(hipMemcpy(out, od.p, (size_t)batch * hidden * 4, hipMemcpyDeviceToHost), "D2H pooled");
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_op_layer_norm(int32_t device, const float* x, const float* gamma,
                                                       const float* beta, float eps, int64_t rows, int32_t hidden,
                                                       float* y, int32_t iters, float* ms_out)
{
    if (!x || !gamma || !beta || !y) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        if (rows < 0 || hidden <= 0) throw InvalidConfig("invalid LayerNorm dimensions");
        use_device(device);
        if (rows == 0) return;
        const size_t b = (size_t)rows * hidden * 4;
        DeviceBuf xd(b), gd((size_t)hidden * 4), bd((size_t)hidden * 4), yd(b);
        hip_check(hipMemcpy(xd.p, x, b, hipMemcpyHostToDevice), "H2D x");
        hip_check(hipMemcpy(gd.p, gamma, (size_t)hidden * 4, hipMemcpyHostToDevice), "H2D gamma");
        hip_check(hipMemcpy(bd.p, beta, (size_t)hidden * 4, hipMemcpyHostToDevice), "H2D beta");
        time_launches(iters, ms_out, [&] {
            hip_check(launch_layernorm((const float*)xd.p, (const float*)gd.p, (const float*)bd.p, eps, rows, hidden,
                                       (float*)yd.p, nullptr),
                      "layernorm");
        });
        hip_check(hipMemcpy(y, yd.p, b, hipMemcpyDeviceToHost), "D2H y");
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_op_linear_layer_norm(int32_t device, const float* x, const float* w,
                                                              const float* bias, const float* residual, const float* gamma,
                                                              const float* beta, float eps, int64_t m, int32_t k, int32_t n,
                                                              float* y, int32_t iters, float* ms_out)
{
    if (!x || !w || !residual || !gamma || !beta || !y) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        if (m < 0 || k <= 0 || n <= 0) throw InvalidConfig("invalid GEMM dimensions");
        use_device(device);
        if (m == 0) return;
        const size_t xb = (size_t)m * k * 4, wb = (size_t)n * k * 4, yb = (size_t)m * n * 4, nb = (size_t)n * 4;
        DeviceBuf xd(xb), wd(wb), bd(nb), rd(yb), gd(nb), ed(nb), yd(yb);
        hip_check(hipMemcpy(xd.p, x, xb, hipMemcpyHostToDevice), "H2D x");
        hip_check(hipMemcpy(wd.p, w, wb, hipMemcpyHostToDevice), "H2D w");
        if (bias) hip_check(hipMemcpy(bd.p, bias, nb, hipMemcpyHostToDevice), "H2D bias");
        const float* bias_d = bias ? (const float*)bd.p : nullptr;
        hip_check(hipMemcpy(rd.p, residual, yb, hipMemcpyHostToDevice), "H2D residual");
        hip_check(hipMemcpy(gd.p, gamma, nb, hipMemcpyHostToDevice), "H2D gamma");
        hip_check(hipMemcpy(ed.p, beta, nb, hipMemcpyHostToDevice), "H2D beta");
        const size_t sf = gemm_scratch_floats(m, n);
        DeviceBuf sd(sf * 4);
        const GemmScratch sc{(float*)sd.p, sf};
        const bool fused = gemm_residual_layernorm_supported(n, k) || gemm_mid_layernorm_supported(m, n, k);
        time_launches(iters, ms_out, [&] {
            if (fused) {
                hip_check(launch_gemm_residual_layernorm((const float*)xd.p, k, (const float*)wd.p, bias_d,
                                                         (const float*)rd.p, n, (const float*)gd.p, (const float*)ed.p, eps,
                                                         (float*)yd.p, n, m, n, k, nullptr, sc),
                          "gemm + layernorm");
            } else {
                hip_check(launch_gemm((const float*)xd.p, k, (const float*)wd.p, bias_d, (const float*)rd.p, n,
                                      (float*)yd.p, n, m, n, k, EPI_BIAS_RESIDUAL, nullptr, sc),
                          "gemm");
                hip_check(launch_layernorm((const float*)yd.p, (const float*)gd.p, (const float*)ed.p, eps, m, n, (float*)yd.p,
                                           nullptr),
                          "layernorm");
            }
        });
        hip_check(hipMemcpy(y, yd.p, yb, hipMemcpyDeviceToHost), "D2H y");
    });
}

#ifdef KJARNI_TUNING
#include "whisper_kernels.h"
// Kernel A/B switches (tuning.h): exported by the tuning build only (kjarni_amd/lib/libkjarni_ffi_tuning.so, tools/).
namespace kjarni { namespace tune { std::atomic<int> g_gemm{0}, g_attention{0}, g_cosine{0}; } }
KJARNI_EXPORT void kjarni_hip_set_gemm_variant(int32_t variant) { kjarni::tune::g_gemm = variant; }
KJARNI_EXPORT void kjarni_hip_set_attention_variant(int32_t variant) { kjarni::tune::g_attention = variant; }
KJARNI_EXPORT void kjarni_hip_set_cosine_variant(int32_t variant) { kjarni::tune::g_cosine = variant; }
KJARNI_EXPORT int32_t kjarni_hip_attention_stamps(uint64_t* out16, int32_t reset)
{
    return out16 && kjarni::attention_stamps(reinterpret_cast<unsigned long long*>(out16), reset) == hipSuccess ? 0 : 5;
}
#endif

// ---- cosine scan ----------------------------------------------------------------

KJARNI_EXPORT KjarniErrorCode kjarni_hip_cosine_scores(int32_t device, const float* queries_dev,
                                                       int32_t n_queries, const float* corpus_dev,
                                                       int64_t n_docs, int32_t dim, KjarniHipCosineMode mode,
                                                       float* scores_out_dev, void* stream)
{
    if (!queries_dev || !corpus_dev || !scores_out_dev) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        if (dim <= 0 || n_queries < 0 || n_docs < 0) throw InvalidConfig("invalid scan dimensions");
        use_device(device);
        hip_check(launch_cosine_scores(queries_dev, n_queries, corpus_dev, n_docs, dim, (int)mode,
                                       scores_out_dev, static_cast<hipStream_t>(stream)),
                  "cosine_scores");
    });
}

KJARNI_EXPORT size_t kjarni_hip_cosine_topk_workspace_bytes(int32_t n_queries, int64_t n_docs, int32_t k)
{
    if (n_queries <= 0 || n_docs <= 0 || k <= 0) return 256;
    return cosine_topk_workspace_bytes(n_queries, n_docs, k);
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_cosine_topk(int32_t device, const float* scores_dev, int32_t n_queries,
                                                     int64_t n_docs, int32_t k, void* workspace_dev,
                                                     int64_t* idx_out_dev, float* score_out_dev, void* stream)
{
    if (!scores_dev || !workspace_dev || !idx_out_dev || !score_out_dev) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        use_device(device);
        hip_check(launch_cosine_topk(scores_dev, n_queries, n_docs, k, workspace_dev, idx_out_dev,
                                     score_out_dev, static_cast<hipStream_t>(stream)),
                  "cosine_topk");
    });
}

KJARNI_EXPORT size_t kjarni_hip_cosine_search_workspace_bytes(int32_t n_queries, int64_t n_docs, int32_t dim, int32_t k)
{
    if (n_queries <= 0 || n_docs <= 0 || k <= 0 || dim <= 0) return 256;
    return cosine_search_workspace_bytes(n_queries, n_docs, dim, k);
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_cosine_search(int32_t device, const float* queries_dev, int32_t n_queries,
                                                       const float* corpus_dev, int64_t n_docs, int32_t dim,
                                                       KjarniHipCosineMode mode, int32_t k, void* workspace_dev,
                                                       int64_t* idx_out_dev, float* score_out_dev, void* stream)
{
    if (!queries_dev || !corpus_dev || !workspace_dev || !idx_out_dev || !score_out_dev) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        if (dim <= 0 || n_queries < 0 || n_docs < 0 || k < 0) throw InvalidConfig("invalid search dimensions");
        use_device(device);
        hip_check(launch_cosine_search(queries_dev, n_queries, corpus_dev, n_docs, dim, (int)mode, k, workspace_dev, idx_out_dev,
                                       score_out_dev, (hipStream_t)stream),
                  "cosine_search");
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_cosine_search_host(int32_t device, const float* queries,
                                                            int32_t n_queries, const float* corpus,
                                                            int64_t n_docs, int32_t dim, KjarniHipCosineMode mode,
                                                            int32_t k, int64_t* idx_out, float* score_out,
                                                            int64_t* n_hits_out)
{
    if (!queries || !corpus || !idx_out || !score_out) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        if (dim <= 0 || n_queries < 0 || n_docs < 0 || k < 0) throw InvalidConfig("invalid search dimensions");
        if (n_hits_out) *n_hits_out = 0;
        if (n_queries == 0 || n_docs == 0 || k == 0) return;
        use_device(device);
        const size_t qb = (size_t)n_queries * dim * 4, cb = (size_t)n_docs * dim * 4;
        DeviceBuf q_d(qb), c_d(cb);
        DeviceBuf ws(cosine_search_workspace_bytes(n_queries, n_docs, dim, k));
        DeviceBuf i_d((size_t)n_queries * k * 8), o_d((size_t)n_queries * k * 4);
        hip_check(hipMemcpy(q_d.p, queries, qb, hipMemcpyHostToDevice), "H2D queries");
        hip_check(hipMemcpy(c_d.p, corpus, cb, hipMemcpyHostToDevice), "H2D corpus");
        hip_check(launch_cosine_search((const float*)q_d.p, n_queries, (const float*)c_d.p, n_docs, dim, (int)mode, k, ws.p,
                                       (int64_t*)i_d.p, (float*)o_d.p, nullptr),
                  "cosine_search");
        hip_check(hipMemcpy(idx_out, i_d.p, (size_t)n_queries * k * 8, hipMemcpyDeviceToHost), "D2H idx");
        hip_check(hipMemcpy(score_out, o_d.p, (size_t)n_queries * k * 4, hipMemcpyDeviceToHost), "D2H scores");
        if (mode == KJARNI_HIP_COSINE_SEGMENT) {
            // segment.rs:313-317: a query whose norm is < 1e-9 has no hits.
            for (int32_t j = 0; j < n_queries; ++j) {
                float s2 = 0.0f;
                for (int32_t i = 0; i < dim; ++i) s2 += queries[(size_t)j * dim + i] * queries[(size_t)j * dim + i];
                if (std::sqrt(s2) < 1e-9f)
                    for (int32_t i = 0; i < k; ++i) {
                        idx_out[(size_t)j * k + i] = -1;
                        score_out[(size_t)j * k + i] = -std::numeric_limits<float>::infinity();
                    }
            }
        }
        if (n_hits_out) *n_hits_out = (k < n_docs) ? k : n_docs;
    });
}

// ---- device memory helpers ------------------------------------------------------

KJARNI_EXPORT KjarniErrorCode kjarni_hip_malloc(int32_t device, size_t bytes, void** out_dev)
{
    if (!out_dev) return KJARNI_ERROR_NULL_POINTER;
    *out_dev = nullptr;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        use_device(device);
        hip_check(hipMalloc(out_dev, bytes ? bytes : 4), "hipMalloc");
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_free(int32_t device, void* ptr_dev)
{
    if (!ptr_dev) return KJARNI_OK;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        use_device(device);
        hip_check(hipFree(ptr_dev), "hipFree");
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_memcpy_h2d(int32_t device, void* dst_dev, const void* src, size_t bytes)
{
    if (!dst_dev || !src) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        use_device(device);
        hip_check(hipMemcpy(dst_dev, src, bytes, hipMemcpyHostToDevice), "hipMemcpy H2D");
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_memcpy_d2h(int32_t device, void* dst, const void* src_dev, size_t bytes)
{
    if (!dst || !src_dev) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        use_device(device);
        hip_check(hipMemcpy(dst, src_dev, bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H");
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_synchronize(int32_t device)
{
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        use_device(device);
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
    });
}
