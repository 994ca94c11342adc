// Issue rate of the two f32 MFMA shapes on gfx950, one wave per SIMD (256-thread workgroups, one per CU), operands in registers:
//   v_mfma_f32_16x16x4_f32 with NACC independent 4-register accumulators, v_mfma_f32_32x32x2_f32 with NACC / 4 sixteen-register ones.
// Prints wall time per launch and shader cycles per MFMA (s_memtime around the loop, median workgroup).
// hipcc --offload-arch=gfx950 -O3 -o tools/lab/mfma_rate tools/lab/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256, 2) void k16(float* out, unsigned long long* cyc, int iters, float a0, float b0)
{
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = a0 + threadIdx.x * 0.001f + i; b[i] = b0 + i * 0.5f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[(q + i) & 3], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123456.789f) out[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(256, 2) void k32(float* out, unsigned long long* cyc, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = a0 + threadIdx.x * 0.001f + i; b[i] = b0 + i * 0.5f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[(q + i) & 3], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123456.789f) out[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class F>
void run(const char* name, F launch, int mfma_per_iter, double flop_per_mfma, int iters, int grid, unsigned long long* cyc_d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 200; ++w) launch();
    hipEventRecord(e0);
    const int reps = 50;
    for (int w = 0; w < reps; ++w) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(grid);
    hipMemcpy(c.data(), cyc_d, grid * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double n = (double)iters * mfma_per_iter;
    const double us = ms * 1e3 / reps;
    std::printf("%-28s %8.2f us/launch  %6.2f cycles (memtime ticks) per MFMA (median WG)  %6.1f TFLOP/s  wall ns/MFMA %.2f\n", name, us,
                c[grid / 2] / n, n * flop_per_mfma * 4 * grid / (us * 1e-6) / 1e12, us * 1e3 / n);
}

int main()
{
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 1024);
    hipMalloc(&cyc, 4096 * 8);
    const int iters = 24;
    for (int grid : {256, 512, 1024}) {  // 1, 2, 4 waves per SIMD (256-thread workgroups, as many per CU as fit)
        std::printf("grid %d\n", grid);
        run("16x16x4, 24 accumulators", [&] { hipLaunchKernelGGL(k16<24>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.f, 2.f); }, 96, 2.0 * 16 * 16 * 4, iters, grid, cyc);
        run("16x16x4, 18 accumulators", [&] { hipLaunchKernelGGL(k16<18>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.f, 2.f); }, 72, 2.0 * 16 * 16 * 4, iters, grid, cyc);
        run("16x16x4, 4 accumulators", [&] { hipLaunchKernelGGL(k16<4>, dim3(grid), dim3(256), 0, 0, out, cyc, iters * 6, 1.f, 2.f); }, 16, 2.0 * 16 * 16 * 4, iters * 6, grid, cyc);
        run("32x32x2, 6 accumulators", [&] { hipLaunchKernelGGL(k32<6>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.f, 2.f); }, 24, 2.0 * 32 * 32 * 2, iters, grid, cyc);
        run("32x32x2, 4 accumulators", [&] { hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.f, 2.f); }, 16, 2.0 * 32 * 32 * 2, iters, grid, cyc);
        run("32x32x2, 1 accumulator", [&] { hipLaunchKernelGGL(k32<1>, dim3(grid), dim3(256), 0, 0, out, cyc, iters * 4, 1.f, 2.f); }, 4, 2.0 * 32 * 32 * 2, iters * 4, grid, cyc);
    }
    return 0;
}
