// stage_lab: what does one STAGE of a one-token decode step cost, and what moves it?
// A decode step is a chain of dependent launches, each a small GEMV whose input row (2 KB) was written by the launch before
// it (docs/history/r06.md section 2).  This probe replays such a chain as a hipGraph -- 48 stages of
// y = W_s x + b over K = 512, N = 512 f32 (1 MB of weights per stage, 24 different matrices in rotation), ping-pong rows --
// in several geometries and memory policies, next to two floors: a chain of empty kernels, and a chain of kernels that only
// carry the row (load x, store y).
// Standalone (tools/ only): hipcc --offload-arch=gfx950 -O3 -std=c++17 -o stage_lab stage_lab.hip && ./stage_lab
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

constexpr int K = 512, N = 512, KCH = K / 256;   // the homogeneous chains; the mixed chain below has its own shapes

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// The same sum, bit for bit, without the LDS crossbar (ds_bpermute_b32 is ~100 cycles of latency per step): the butterfly's
// partner of lane i at offset 32 / 16 comes from v_permlane32_swap / v_permlane16_swap (gfx950), at 8 / 4 / 2 / 1 from a DPP
// rotation of the 16-lane row (row_ror:n hands lane i the value of lane (i + n) % 16; after the steps before it that lane
// holds what lane i ^ n holds) -- and a + b == b + a.
__device__ __forceinline__ float wave_sum_fast(float v)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    {
        const unsigned u = __float_as_uint(v);
        const u32x2 t = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v = __uint_as_float(t[0]) + __uint_as_float(t[1]);
    }
    {
        const unsigned u = __float_as_uint(v);
        const u32x2 t = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = __uint_as_float(t[0]) + __uint_as_float(t[1]);
    }
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x128, 0xF, 0xF, false));  // row_ror:8
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x124, 0xF, 0xF, false));  // row_ror:4
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x122, 0xF, 0xF, false));  // row_ror:2
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x121, 0xF, 0xF, false));  // row_ror:1
    return v;
}

__global__ void reduce_check_kernel(const float* __restrict__ x, unsigned* __restrict__ mismatches)
{
    const float v = x[blockIdx.x * 64 + threadIdx.x];
    const float a = wave_sum(v), b = wave_sum_fast(v);
    if (__float_as_uint(a) != __float_as_uint(b)) atomicAdd(mismatches, 1u);
}

__device__ __forceinline__ f32x4 load_sc1(const float* p)
{
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void store_sc1(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

__global__ void empty_kernel(const float* x, float* y) {}

__global__ __launch_bounds__(256) void carry_kernel(const float* __restrict__ x, float* __restrict__ y)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) y[i] = x[i] + 1.0f;
}

// WAVES waves per workgroup, COLS output columns per wave; XP: 0 plain row loads / stores, 1 sc1 (write-through stores, L1-bypassing loads);
// WP: 0 plain weight loads, 1 non-temporal; LN: normalise the row first (two wave reductions more, as the real stages do)
template <int WAVES, int COLS, int XP, int WP, bool LN, bool FAST = false>
__global__ __launch_bounds__(64 * WAVES) void gemv_stage(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ y)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * WAVES + wave) * COLS;
    if (n0 >= N) return;
    f32x4 xv[KCH], w[COLS][KCH];
#pragma unroll
    for (int j = 0; j < KCH; ++j) xv[j] = XP ? load_sc1(x + (lane + 64 * j) * 4) : *reinterpret_cast<const f32x4*>(x + (lane + 64 * j) * 4);
    float b[COLS];
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
        b[c] = bias[n0 + c];
        const f32x4* w4 = reinterpret_cast<const f32x4*>(W + (size_t)(n0 + c) * K);
#pragma unroll
        for (int j = 0; j < KCH; ++j) w[c][j] = WP ? __builtin_nontemporal_load(w4 + lane + 64 * j) : w4[lane + 64 * j];
    }
    if (LN) {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < KCH; ++j) s += (xv[j][0] + xv[j][1]) + (xv[j][2] + xv[j][3]);
        const float mu = (FAST ? wave_sum_fast(s) : wave_sum(s)) / (float)K;
        float v = 0.0f;
#pragma unroll
        for (int j = 0; j < KCH; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) v = fmaf(xv[j][c] - mu, xv[j][c] - mu, v);
        const float rstd = 1.0f / sqrtf((FAST ? wave_sum_fast(v) : wave_sum(v)) / (float)K + 1e-5f);
#pragma unroll
        for (int j = 0; j < KCH; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) xv[j][c] = (xv[j][c] - mu) * rstd;
    }
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < KCH; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = fmaf(xv[j][e], w[c][j][e], acc);
        const float v = (FAST ? wave_sum_fast(acc) : wave_sum(acc)) * 0.05f + b[c];
        if (lane == 0) {
            if (XP) store_sc1(y + n0 + c, v);
            else y[n0 + c] = v;
        }
    }
}

// A stage of the MIXED chain: y[0 .. n_out) = W x (+ LayerNorm first) over k = 256 KC floats, one column per wave, 4 waves per workgroup
template <int KC, bool LN>
__global__ __launch_bounds__(256) void mixed_stage(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                                   float* __restrict__ y, int n_out)
{
    constexpr int KK = 256 * KC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    if (n >= n_out) return;
    f32x4 xv[KC], w[KC];
#pragma unroll
    for (int j = 0; j < KC; ++j) xv[j] = *reinterpret_cast<const f32x4*>(x + (lane + 64 * j) * 4);
    const float b = bias[n & 511];
    const f32x4* w4 = reinterpret_cast<const f32x4*>(W + (size_t)n * KK);
#pragma unroll
    for (int j = 0; j < KC; ++j) w[j] = __builtin_nontemporal_load(w4 + lane + 64 * j);
    if (LN) {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < KC; ++j) s += (xv[j][0] + xv[j][1]) + (xv[j][2] + xv[j][3]);
        const float mu = wave_sum_fast(s) / (float)KK;
        float v = 0.0f;
#pragma unroll
        for (int j = 0; j < KC; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) v = fmaf(xv[j][c] - mu, xv[j][c] - mu, v);
        const float rstd = 1.0f / sqrtf(wave_sum_fast(v) / (float)KK + 1e-5f);
#pragma unroll
        for (int j = 0; j < KC; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) xv[j][c] = (xv[j][c] - mu) * rstd;
    }
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < KC; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = fmaf(xv[j][e], w[j][e], acc);
    const float v = wave_sum_fast(acc) * 0.02f + b;
    if (lane == 0) y[n] = v;
}

struct Bufs {
    float *x[2], *W, *bias, *Wbig;   // Wbig: 768 MB -- a stage's weights 16 MB apart, 48 stages: nothing of it survives in the L2s / MALL
};

template <typename F>
static double time_chain(const char* name, F enqueue_stage, int stages)
{
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < stages; ++i) enqueue_stage(i, s);
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 20; ++i) CHECK(hipGraphLaunch(ge, s));
    CHECK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 200;
    CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) CHECK(hipGraphLaunch(ge, s));
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps / stages;
    printf("%-64s %7.3f us per stage\n", name, us);
    fflush(stdout);
    CHECK(hipGraphExecDestroy(ge));
    CHECK(hipGraphDestroy(g));
    CHECK(hipStreamDestroy(s));
    return us;
}

// A LayerNorm + GEMV stage (512 -> n_out) with PAD one-cycle, four-byte no-ops executed before its loads (WHERE = 0) or between
// its loads and its arithmetic (WHERE = 1): what does a KB of code cost a stage that starts with a cold instruction cache?
// ID makes distinct copies of the same code at different addresses.
template <int ID, int PAD, int WHERE>
__global__ __launch_bounds__(256) void padded_stage(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                                    float* __restrict__ y, int n_out)
{
    constexpr int KC = 2, KK = 512;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    if (n >= n_out) return;
    if (PAD > 0 && WHERE == 0) asm volatile(".rept %0\n\ts_nop 0\n\t.endr" ::"n"(PAD));
    f32x4 xv[KC], w[KC];
#pragma unroll
    for (int j = 0; j < KC; ++j) xv[j] = *reinterpret_cast<const f32x4*>(x + (lane + 64 * j) * 4);
    const float b = bias[n & 511];
    const f32x4* w4 = reinterpret_cast<const f32x4*>(W + (size_t)n * KK);
#pragma unroll
    for (int j = 0; j < KC; ++j) w[j] = __builtin_nontemporal_load(w4 + lane + 64 * j);
    if (PAD > 0 && WHERE == 1) asm volatile(".rept %0\n\ts_nop 0\n\t.endr" ::"n"(PAD));
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < KC; ++j) s += (xv[j][0] + xv[j][1]) + (xv[j][2] + xv[j][3]);
    const float mu = wave_sum_fast(s) / (float)KK;
    float v = 0.0f;
#pragma unroll
    for (int j = 0; j < KC; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) v = fmaf(xv[j][c] - mu, xv[j][c] - mu, v);
    const float rstd = 1.0f / sqrtf(wave_sum_fast(v) / (float)KK + 1e-5f);
#pragma unroll
    for (int j = 0; j < KC; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) xv[j][c] = (xv[j][c] - mu) * rstd;
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < KC; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = fmaf(xv[j][e], w[j][e], acc);
    const float r = wave_sum_fast(acc) * 0.02f + b + (float)ID * 0.0f;
    if (lane == 0) y[n] = r;
}

template <int PAD, int WHERE>
static void run_padded(const Bufs& b, int stages, bool rotate)
{
    char name[128];
    snprintf(name, sizeof name, "LN 512 -> 1536, %4d no-ops %s, %s", PAD, WHERE ? "after the loads " : "before the loads",
             rotate ? "8 copies of the code in turn" : "the same kernel every stage");
    time_chain(name, [&](int i, hipStream_t s) {
        const float* W = b.W + (size_t)(i % 20) * N * K;
#define PS(ID) hipLaunchKernelGGL((padded_stage<ID, PAD, WHERE>), dim3(384), dim3(256), 0, s, b.x[i & 1], W, b.bias, b.x[(i + 1) & 1], 1536)
        switch (rotate ? i & 7 : 0) {
        case 0: PS(0); break;
        case 1: PS(1); break;
        case 2: PS(2); break;
        case 3: PS(3); break;
        case 4: PS(4); break;
        case 5: PS(5); break;
        case 6: PS(6); break;
        default: PS(7); break;
        }
#undef PS
    }, stages);
}

template <int WAVES, int COLS, int XP, int WP, bool LN, bool FAST = false>
static void run_gemv(const char* name, const Bufs& b, int stages)
{
    const int grid = N / (WAVES * COLS);
    time_chain(name, [&](int i, hipStream_t s) {
        hipLaunchKernelGGL((gemv_stage<WAVES, COLS, XP, WP, LN, FAST>), dim3(grid), dim3(64 * WAVES), 0, s, b.x[i & 1], b.W + (size_t)(i % 24) * N * K,
                           b.bias, b.x[(i + 1) & 1]);
    }, stages);
}

int main(int argc, char** argv)
{
    Bufs b;
    CHECK(hipMalloc(&b.x[0], 2048 * 4));
    CHECK(hipMalloc(&b.x[1], 2048 * 4));
    CHECK(hipMemset(b.x[0], 0, 2048 * 4));
    CHECK(hipMemset(b.x[1], 0, 2048 * 4));
    CHECK(hipMalloc(&b.W, (size_t)24 * N * K * 4));
    CHECK(hipMalloc(&b.bias, N * 4));
    CHECK(hipMalloc(&b.Wbig, (size_t)768 << 20));
    CHECK(hipMemset(b.Wbig, 0, (size_t)768 << 20));
    std::vector<float> h((size_t)24 * N * K);
    unsigned s = 12345;
    for (auto& v : h) {
        s = s * 1664525u + 1013904223u;
        v = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f;
    }
    CHECK(hipMemcpy(b.W, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(b.x[0], h.data(), N * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(b.bias, h.data() + 999, N * 4, hipMemcpyHostToDevice));
    const int stages = 48;
    {   // the DPP / permlane-swap reduction against the ds_bpermute butterfly, bit for bit, on 2^20 random waves
        unsigned* mm;
        CHECK(hipMalloc(&mm, 4));
        CHECK(hipMemset(mm, 0, 4));
        hipLaunchKernelGGL(reduce_check_kernel, dim3((unsigned)(h.size() / 64)), dim3(64), 0, 0, b.W, mm);
        unsigned host = 1;
        CHECK(hipMemcpy(&host, mm, 4, hipMemcpyDeviceToHost));
        printf("wave_sum_fast vs wave_sum on %zu waves of random data: %u mismatching lanes\n", h.size() / 64, host);
    }
    if (argc > 1 && !strcmp(argv[1], "pad")) {   // the cost of code to a stage (instruction cache cold at every launch?)
        for (int rep = 0; rep < 2; ++rep) {
            run_padded<0, 0>(b, stages, false);
            run_padded<0, 0>(b, stages, true);
            run_padded<64, 0>(b, stages, false);
            run_padded<256, 0>(b, stages, false);
            run_padded<256, 0>(b, stages, true);
            run_padded<256, 1>(b, stages, false);
            run_padded<1024, 0>(b, stages, false);
            run_padded<1024, 0>(b, stages, true);
            run_padded<1024, 1>(b, stages, false);
        }
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        time_chain("empty kernels, 128 x 256 threads", [&](int i, hipStream_t st) { hipLaunchKernelGGL(empty_kernel, dim3(128), dim3(256), 0, st, b.x[i & 1], b.x[(i + 1) & 1]); }, stages);
        time_chain("empty kernels, 1 x 64 threads", [&](int i, hipStream_t st) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, b.x[i & 1], b.x[(i + 1) & 1]); }, stages);
        time_chain("carry the row only (2 x 256 threads: load, add, store)", [&](int i, hipStream_t st) { hipLaunchKernelGGL(carry_kernel, dim3(2), dim3(256), 0, st, b.x[i & 1], b.x[(i + 1) & 1]); }, stages);
        // the Whisper decoder layer's GEMV shapes in their order (no attention): LN + QKV 512 -> 1536, out 512 -> 512, LN + Q 512 -> 512,
        // out 512 -> 512, LN + FC1 512 -> 2048, FC2 2048 -> 512; weights of a stage at a different place every time (6 layers' worth)
        time_chain("MIXED chain: Whisper layer GEMV shapes (6 stages x 8 = 48)", [&](int i, hipStream_t st) {
            const float* Wp = b.W + (size_t)((i * 5) % 20) * N * K;   // (up to 4 MB per stage inside the 24 MB block)
            const float* xin = b.x[i & 1];
            float* yout = b.x[(i + 1) & 1];
            switch (i % 6) {
            case 0: hipLaunchKernelGGL((mixed_stage<2, true>), dim3(384), dim3(256), 0, st, xin, Wp, b.bias, yout, 1536); break;
            case 1: hipLaunchKernelGGL((mixed_stage<2, false>), dim3(128), dim3(256), 0, st, xin, Wp, b.bias, yout, 512); break;
            case 2: hipLaunchKernelGGL((mixed_stage<2, true>), dim3(128), dim3(256), 0, st, xin, Wp, b.bias, yout, 512); break;
            case 3: hipLaunchKernelGGL((mixed_stage<2, false>), dim3(128), dim3(256), 0, st, xin, Wp, b.bias, yout, 512); break;
            case 4: hipLaunchKernelGGL((mixed_stage<2, true>), dim3(512), dim3(256), 0, st, xin, Wp, b.bias, yout, 2048); break;
            default: hipLaunchKernelGGL((mixed_stage<8, false>), dim3(128), dim3(256), 0, st, xin, Wp, b.bias, yout, 512); break;
            }
        }, stages);
        time_chain("MIXED chain, weights 16 MB apart in a 768 MB pool (cold every replay)", [&](int i, hipStream_t st) {
            const float* Wp = b.Wbig + (size_t)i * (4u << 20);   // (floats: 16 MB apart)
            const float* xin = b.x[i & 1];
            float* yout = b.x[(i + 1) & 1];
            switch (i % 6) {
            case 0: hipLaunchKernelGGL((mixed_stage<2, true>), dim3(384), dim3(256), 0, st, xin, Wp, b.bias, yout, 1536); break;
            case 1: hipLaunchKernelGGL((mixed_stage<2, false>), dim3(128), dim3(256), 0, st, xin, Wp, b.bias, yout, 512); break;
            case 2: hipLaunchKernelGGL((mixed_stage<2, true>), dim3(128), dim3(256), 0, st, xin, Wp, b.bias, yout, 512); break;
            case 3: hipLaunchKernelGGL((mixed_stage<2, false>), dim3(128), dim3(256), 0, st, xin, Wp, b.bias, yout, 512); break;
            case 4: hipLaunchKernelGGL((mixed_stage<2, true>), dim3(512), dim3(256), 0, st, xin, Wp, b.bias, yout, 2048); break;
            default: hipLaunchKernelGGL((mixed_stage<8, false>), dim3(128), dim3(256), 0, st, xin, Wp, b.bias, yout, 512); break;
            }
        }, stages);
        {   // the same mixed chain with every stage's weights AND its bias in hipMalloc allocations of their own (as a loader that
            // uploads tensor by tensor leaves them), against one pool: page-table reach
            static std::vector<float*> Ws, Bs;
            if (Ws.empty()) {
                const size_t sizes[6] = {(size_t)1536 * 512, (size_t)512 * 512, (size_t)512 * 512, (size_t)512 * 512, (size_t)2048 * 512, (size_t)512 * 2048};
                for (int i = 0; i < stages; ++i) {
                    float *w, *bb;
                    CHECK(hipMalloc(&w, sizes[i % 6] * 4));
                    CHECK(hipMemset(w, 0, sizes[i % 6] * 4));
                    CHECK(hipMalloc(&bb, 2048 * 4));
                    CHECK(hipMemset(bb, 0, 2048 * 4));
                    Ws.push_back(w);
                    Bs.push_back(bb);
                }
            }
            time_chain("MIXED chain, every stage's weights and bias in their OWN hipMalloc", [&](int i, hipStream_t st) {
                const float* Wp = Ws[i];
                const float* xin = b.x[i & 1];
                float* yout = b.x[(i + 1) & 1];
                switch (i % 6) {
                case 0: hipLaunchKernelGGL((mixed_stage<2, true>), dim3(384), dim3(256), 0, st, xin, Wp, Bs[i], yout, 1536); break;
                case 1: hipLaunchKernelGGL((mixed_stage<2, false>), dim3(128), dim3(256), 0, st, xin, Wp, Bs[i], yout, 512); break;
                case 2: hipLaunchKernelGGL((mixed_stage<2, true>), dim3(128), dim3(256), 0, st, xin, Wp, Bs[i], yout, 512); break;
                case 3: hipLaunchKernelGGL((mixed_stage<2, false>), dim3(128), dim3(256), 0, st, xin, Wp, Bs[i], yout, 512); break;
                case 4: hipLaunchKernelGGL((mixed_stage<2, true>), dim3(512), dim3(256), 0, st, xin, Wp, Bs[i], yout, 2048); break;
                default: hipLaunchKernelGGL((mixed_stage<8, false>), dim3(128), dim3(256), 0, st, xin, Wp, Bs[i], yout, 512); break;
                }
            }, stages);
        }
        time_chain("same shape every stage: LN 512 -> 1536 (384 WG)", [&](int i, hipStream_t st) {
            hipLaunchKernelGGL((mixed_stage<2, true>), dim3(384), dim3(256), 0, st, b.x[i & 1], b.W + (size_t)((i * 5) % 20) * N * K, b.bias, b.x[(i + 1) & 1], 1536); }, stages);
        time_chain("same shape every stage: LN 512 -> 2048 (512 WG)", [&](int i, hipStream_t st) {
            hipLaunchKernelGGL((mixed_stage<2, true>), dim3(512), dim3(256), 0, st, b.x[i & 1], b.W + (size_t)((i * 5) % 20) * N * K, b.bias, b.x[(i + 1) & 1], 2048); }, stages);
        time_chain("same shape every stage: 2048 -> 512 (128 WG, 4 MB)", [&](int i, hipStream_t st) {
            hipLaunchKernelGGL((mixed_stage<8, false>), dim3(128), dim3(256), 0, st, b.x[i & 1], b.W + (size_t)((i * 5) % 20) * N * K, b.bias, b.x[(i + 1) & 1], 512); }, stages);
        run_gemv<4, 1, 0, 1, true>("gemv LN, 128 WG x 4 waves x 1 col, plain row, nt weights (production)", b, stages);
        run_gemv<4, 1, 0, 1, true, true>("gemv LN, 128 WG x 4 waves x 1 col, DPP / permlane reductions", b, stages);
        run_gemv<4, 1, 0, 1, false>("gemv    , 128 WG x 4 waves x 1 col, plain row, nt weights", b, stages);
        run_gemv<4, 1, 0, 1, false, true>("gemv    , 128 WG x 4 waves x 1 col, DPP / permlane reductions", b, stages);
        run_gemv<4, 1, 0, 0, true>("gemv LN, 128 WG x 4 waves x 1 col, plain row, plain weights", b, stages);
        run_gemv<4, 1, 1, 1, true>("gemv LN, 128 WG x 4 waves x 1 col, sc1 row, nt weights", b, stages);
        run_gemv<4, 2, 0, 1, true>("gemv LN,  64 WG x 4 waves x 2 cols, plain row, nt weights", b, stages);
        run_gemv<4, 4, 0, 1, true>("gemv LN,  32 WG x 4 waves x 4 cols, plain row, nt weights", b, stages);
        run_gemv<8, 1, 0, 1, true>("gemv LN,  64 WG x 8 waves x 1 col, plain row, nt weights", b, stages);
        run_gemv<16, 1, 0, 1, true>("gemv LN,  32 WG x 16 waves x 1 col, plain row, nt weights", b, stages);
        run_gemv<2, 1, 0, 1, true>("gemv LN, 256 WG x 2 waves x 1 col, plain row, nt weights", b, stages);
        run_gemv<1, 1, 0, 1, true>("gemv LN, 512 WG x 1 wave x 1 col, plain row, nt weights", b, stages);
        run_gemv<1, 2, 0, 1, true>("gemv LN, 256 WG x 1 wave x 2 cols, plain row, nt weights", b, stages);
        run_gemv<4, 2, 1, 1, true>("gemv LN,  64 WG x 4 waves x 2 cols, sc1 row, nt weights", b, stages);
    }
    return 0;
}
