// What does a grid-wide barrier cost on this part, next to the ~4.2 us a dependent kernel launch costs?
// A persistent kernel of G co-resident workgroups runs `iters` rounds of: every thread stores a value another workgroup will
// read, release fence, one arrive per workgroup on a global counter, spin until all have arrived, acquire, read a value a
// different workgroup wrote (checked).  Build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));            \
            std::exit(1);                                                         \
        }                                                                         \
    } while (0)

__global__ __launch_bounds__(256) void barrier_kernel(unsigned* counter, float* buf, int iters, int payload, unsigned* errors)
{
    const int tid = threadIdx.x, wg = blockIdx.x, G = gridDim.x;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        // payload floats per thread written, then read from the next workgroup's slice after the barrier
        for (int p = 0; p < payload; ++p) buf[((size_t)wg * 256 + tid) * payload + p] = (float)(it * 7 + wg + p);
        __threadfence();  // release: this workgroup's stores are visible device-wide before it arrives
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(it + 1) * (unsigned)G;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        __threadfence();  // acquire side for the other threads of the workgroup
        const int src = (wg + 1) % G;
        for (int p = 0; p < payload; ++p) {
            const float v = __builtin_nontemporal_load(&buf[((size_t)src * 256 + tid) * payload + p]);
            if (v != (float)(it * 7 + src + p)) ++bad;
        }
    }
    if (bad) atomicAdd(errors, bad);
}

__global__ void tiny_kernel(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? std::atoi(argv[1]) : 2000;
    unsigned *counter, *errors;
    float* buf;
    CHECK(hipMalloc(&counter, 4));
    CHECK(hipMalloc(&errors, 4));
    CHECK(hipMalloc(&buf, (size_t)1024 * 256 * 16 * 4));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int G : {64, 256, 512, 1024}) {
        for (int payload : {0, 1, 8}) {
            CHECK(hipMemset(counter, 0, 4));
            CHECK(hipMemset(errors, 0, 4));
            hipLaunchKernelGGL(barrier_kernel, dim3(G), dim3(256), 0, 0, counter, buf, 50, payload, errors);  // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemset(counter, 0, 4));
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(barrier_kernel, dim3(G), dim3(256), 0, 0, counter, buf, iters, payload, errors);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            unsigned bad = 0;
            CHECK(hipMemcpy(&bad, errors, 4, hipMemcpyDeviceToHost));
            std::printf("grid barrier: %4d workgroups, %d floats/thread exchanged: %.3f us per round, %u stale reads\n", G, payload,
                        ms * 1e3 / iters, bad);
        }
    }
    // dependent tiny launches for comparison
    CHECK(hipMemset(buf, 0, 4));
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, 0, buf);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, 0, buf);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    std::printf("dependent one-thread launches (same stream): %.3f us each\n", ms * 1e3 / iters);
    return 0;
}
