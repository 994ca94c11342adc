// stage_lab: what does one STAGE of a one-token decode step cost, and what moves it?
// A decode step is a chain of dependent launches, each a small GEMV whose input row (2 KB) was written by the launch before
// it (docs/history/r06.md section 2).  This probe replays such a chain as a hipGraph -- 48 stages of
// y = W_s x + b over K = 512, N = 512 f32 (1 MB of weights per stage, 24 different matrices in rotation), ping-pong rows --
// in several geometries and memory policies, next to two floors: a chain of empty kernels, and a chain of kernels that only
// carry the row (load x, store y).
// Standalone (tools/ only): hipcc --offload-arch=gfx950 -O3 -std=c++17 -o stage_lab stage_lab.hip && ./stage_lab
#include <hip/hip_runtime.h>

#include <cstdio>
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

constexpr int K = 512, N = 512, KCH = K / 256;

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

struct Bufs {
    float *x[2], *W, *bias;
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

template <int WAVES, int COLS, int XP, int WP, bool LN, bool FAST = false>
static void run_gemv(const char* name, const Bufs& b, int stages)
{
    const int grid = N / (WAVES * COLS);
    time_chain(name, [&](int i, hipStream_t s) {
        hipLaunchKernelGGL((gemv_stage<WAVES, COLS, XP, WP, LN, FAST>), dim3(grid), dim3(64 * WAVES), 0, s, b.x[i & 1], b.W + (size_t)(i % 24) * N * K,
                           b.bias, b.x[(i + 1) & 1]);
    }, stages);
}

int main()
{
    Bufs b;
    CHECK(hipMalloc(&b.x[0], N * 4));
    CHECK(hipMalloc(&b.x[1], N * 4));
    CHECK(hipMalloc(&b.W, (size_t)24 * N * K * 4));
    CHECK(hipMalloc(&b.bias, N * 4));
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
    for (int rep = 0; rep < 2; ++rep) {
        time_chain("empty kernels, 128 x 256 threads", [&](int i, hipStream_t st) { hipLaunchKernelGGL(empty_kernel, dim3(128), dim3(256), 0, st, b.x[i & 1], b.x[(i + 1) & 1]); }, stages);
        time_chain("empty kernels, 1 x 64 threads", [&](int i, hipStream_t st) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, b.x[i & 1], b.x[(i + 1) & 1]); }, stages);
        time_chain("carry the row only (2 x 256 threads: load, add, store)", [&](int i, hipStream_t st) { hipLaunchKernelGGL(carry_kernel, dim3(2), dim3(256), 0, st, b.x[i & 1], b.x[(i + 1) & 1]); }, stages);
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
