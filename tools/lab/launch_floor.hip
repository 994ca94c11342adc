// What does a small dependent launch cost before it does anything?  Back-to-back launches on one stream of (a) an empty kernel,
// (b) one that makes one trip to memory (every thread loads 16 bytes another launch wrote, adds, stores), for the grid / block /
// LDS shapes the few-rows route uses.  Build: hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                       \
    do {                                                               \
        hipError_t e_ = (x);                                           \
        if (e_ != hipSuccess) {                                        \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); \
            std::exit(1);                                              \
        }                                                              \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void empty_kernel(float* p)
{
    extern __shared__ float lds[];
    if (p == nullptr) lds[threadIdx.x] = 0.f;
}

template <int BLOCK, int TRIPS>
__global__ __launch_bounds__(BLOCK) void trip_kernel(const float* __restrict__ in, float* __restrict__ out, int n4)
{
    extern __shared__ float lds[];
    int i = (blockIdx.x * BLOCK + threadIdx.x) % n4;
    f32x4 v = *reinterpret_cast<const f32x4*>(in + (size_t)i * 4);
#pragma unroll
    for (int t = 1; t < TRIPS; ++t) {  // dependent trips: the next address depends on the loaded value (always 0 offset in practice)
        const int j = (i + (int)(v[0] * 0.0f) + t * 4099) % n4;
        const f32x4 w = *reinterpret_cast<const f32x4*>(in + (size_t)j * 4);
        v += w;
    }
    *reinterpret_cast<f32x4*>(out + (size_t)i * 4) = v;
}

template <typename F>
static float time_us(int iters, F&& launch)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 50; ++i) launch(i);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) launch(i);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / iters;
}

int main()
{
    const int n4 = 1 << 18;  // 4 MB per buffer
    float *x, *y;
    CHECK(hipMalloc(&x, (size_t)n4 * 16));
    CHECK(hipMalloc(&y, (size_t)n4 * 16));
    CHECK(hipMemset(x, 0, (size_t)n4 * 16));
    CHECK(hipMemset(y, 0, (size_t)n4 * 16));
    const int iters = 2000;
    struct Shape { int grid, lds; };
    const Shape shapes[] = {{12, 65536}, {48, 65536}, {144, 65536}, {192, 65536}, {144, 0}, {576, 0}};
    for (const Shape& s : shapes) {
        const float e1024 = time_us(iters, [&](int) { hipLaunchKernelGGL(empty_kernel<1024>, dim3(s.grid), dim3(1024), s.lds, 0, x); });
        const float t1 = time_us(iters, [&](int i) {
            hipLaunchKernelGGL((trip_kernel<1024, 1>), dim3(s.grid), dim3(1024), s.lds, 0, (i & 1) ? y : x, (i & 1) ? x : y, n4);
        });
        const float t2 = time_us(iters, [&](int i) {
            hipLaunchKernelGGL((trip_kernel<1024, 2>), dim3(s.grid), dim3(1024), s.lds, 0, (i & 1) ? y : x, (i & 1) ? x : y, n4);
        });
        const float t3 = time_us(iters, [&](int i) {
            hipLaunchKernelGGL((trip_kernel<1024, 3>), dim3(s.grid), dim3(1024), s.lds, 0, (i & 1) ? y : x, (i & 1) ? x : y, n4);
        });
        std::printf("grid %4d x 1024 threads, %5d B LDS: empty %.2f us | 1 trip %.2f | 2 trips %.2f | 3 trips %.2f\n", s.grid, s.lds, e1024, t1,
                    t2, t3);
    }
    for (const Shape& s : {Shape{48, 0}, Shape{576, 0}, Shape{2304, 0}}) {
        const float e = time_us(iters, [&](int) { hipLaunchKernelGGL(empty_kernel<256>, dim3(s.grid), dim3(256), s.lds, 0, x); });
        const float t1 = time_us(iters, [&](int i) {
            hipLaunchKernelGGL((trip_kernel<256, 1>), dim3(s.grid), dim3(256), s.lds, 0, (i & 1) ? y : x, (i & 1) ? x : y, n4);
        });
        const float t2 = time_us(iters, [&](int i) {
            hipLaunchKernelGGL((trip_kernel<256, 2>), dim3(s.grid), dim3(256), s.lds, 0, (i & 1) ? y : x, (i & 1) ? x : y, n4);
        });
        std::printf("grid %4d x  256 threads, %5d B LDS: empty %.2f us | 1 trip %.2f | 2 trips %.2f\n", s.grid, s.lds, e, t1, t2);
    }
    return 0;
}
