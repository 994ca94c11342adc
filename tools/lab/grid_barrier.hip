// What does a grid-wide barrier cost on this part, next to the ~4.2 us a dependent kernel launch costs?
// A persistent kernel of G co-resident workgroups runs `iters` rounds of: every thread stores a value another workgroup will
// read, release fence, one arrive per workgroup on a global counter, spin until all have arrived, acquire, read a value a
// different workgroup wrote (checked).  Two forms: the flat counter with __threadfence() on both sides (round 3), and the
// XCD-hierarchical barrier the guide prices at 4-10 us (round 4).  Build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip
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
        // (two payload buffers by round parity: a fast writer of round it + 1 must not overwrite what a slow reader of round it is
        // still reading -- with one buffer the check below counted that race as stale reads)
        float* pb = buf + (size_t)(it & 1) * 1024 * 256 * 8;
        for (int p = 0; p < payload; ++p) pb[((size_t)wg * 256 + tid) * payload + p] = (float)(it * 7 + wg + p);
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
            const float v = __builtin_nontemporal_load(&pb[((size_t)src * 256 + tid) * payload + p]);
            if (v != (float)(it * 7 + src + p)) ++bad;
        }
    }
    if (bad) atomicAdd(errors, bad);
}

// The guide's form (MI355X_MICROARCH.md, price list, row barrier-xcd): arrivals are counted per XCD (8 counters on lines of
// their own, indexed by the hardware XCC id), so 32 workgroups, not 256, contend on one word; the last arriver of an XCD is
// its leader: ONE release fence (its XCD's L2 is written back once, not once per workgroup), one arrive on the top counter,
// a relaxed poll of it, one acquire, then it raises its XCD's generation word; everybody else polls that word relaxed and
// acquires once.  Payload stores are plain; only lane 0 of a workgroup fences (after every wave's vmcnt drain + the barrier).
struct XcdBarrier {
    unsigned xcd_count[8][32];  // [xcc][0]: arrivals of this round (128-byte lines of their own)
    unsigned xcd_gen[8][32];    // [xcc][0]: generation the XCD has been released to
    unsigned top[32];           // [0]: XCDs arrived, monotone
    unsigned xcd_size[8][32];   // [xcc][0]: workgroups resident on the XCD (census in round 0)
};

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

__global__ __launch_bounds__(256) void barrier_xcd_kernel(XcdBarrier* b, float* buf, int iters, int payload, unsigned* errors)
{
    const int tid = threadIdx.x, wg = blockIdx.x, G = gridDim.x;
    const unsigned x = xcc_id();
    // census: how many workgroups sit on each XCD (placement is not ours to assume); a flat barrier on `top` ends it
    if (tid == 0) {
        __hip_atomic_fetch_add(&b->xcd_size[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&b->top[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(&b->top[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    const unsigned mine = __hip_atomic_load(&b->xcd_size[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned n_xcd = 0;
    for (int i = 0; i < 8; ++i) n_xcd += __hip_atomic_load(&b->xcd_size[i][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        float* pb = buf + (size_t)(it & 1) * 1024 * 256 * 8;
        for (int p = 0; p < payload; ++p) pb[((size_t)wg * 256 + tid) * payload + p] = (float)(it * 7 + wg + p);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave
        __syncthreads();
        if (tid == 0) {
            const unsigned gen = (unsigned)it + 1u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // (this workgroup's dirty lines: the L2 write-back is per XCD, the fence per CU)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned arrived = __hip_atomic_fetch_add(&b->xcd_count[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            if (arrived == gen * mine) {  // the XCD's last arriver leads
                __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(&b->top[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * n_xcd) __builtin_amdgcn_s_sleep(1);
                __hip_atomic_store(&b->xcd_gen[x][0], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(&b->xcd_gen[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const int src = (wg + 1) % G;
        for (int p = 0; p < payload; ++p) {
            const float v = pb[((size_t)src * 256 + tid) * payload + p];
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
    XcdBarrier* xb;
    CHECK(hipMalloc(&xb, sizeof(XcdBarrier)));
    for (int G : {64, 256, 512, 1024}) {
        for (int payload : {0, 1, 8}) {
            CHECK(hipMemset(xb, 0, sizeof(XcdBarrier)));
            CHECK(hipMemset(errors, 0, 4));
            hipLaunchKernelGGL(barrier_xcd_kernel, dim3(G), dim3(256), 0, 0, xb, buf, 50, payload, errors);  // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemset(xb, 0, sizeof(XcdBarrier)));
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(barrier_xcd_kernel, dim3(G), dim3(256), 0, 0, xb, buf, iters, payload, errors);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            unsigned bad = 0;
            CHECK(hipMemcpy(&bad, errors, 4, hipMemcpyDeviceToHost));
            std::printf("XCD-hierarchical barrier: %4d workgroups, %d floats/thread exchanged: %.3f us per round, %u stale reads\n", G, payload,
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
