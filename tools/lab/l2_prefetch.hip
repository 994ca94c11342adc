// Does what one launch READS survive in the XCDs' L2s into the next launch, and what is a cold trip worth?
// A chain of dependent launches; launch i's workgroups each read their 16 KB slice of window i of a 1 GB buffer (cold: no window
// is read twice) and store a checksum.  Variants: (a) cold; (b) every launch re-reads ONE window (L2-warm, same workgroup ->
// same slice -> same XCD); (c) cold windows, but launch i - 1 carried extra workgroups that read window i with the SAME
// workgroup -> slice mapping modulo 8 (so the lines land in the L2 of the XCD that will want them); (d) as (c) with the
// prefetching workgroups' mapping rotated by one XCD (the lines land in the WRONG L2: what a prefetch into the memory-side cache
// alone is worth).  Build: hipcc --offload-arch=gfx950 -O3 -o l2_prefetch l2_prefetch.hip
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
constexpr int kSlice = 16 * 1024;  // bytes a workgroup reads (1024 threads x 16 B)

// workgroups [0, main): read slice (wg) of window `cur`, store a checksum; workgroups [main, 2 main): read slice ((wg - main) + rot)
// of window `next` and store nothing that matters.
__global__ __launch_bounds__(1024) void chain_kernel(const float* __restrict__ buf, size_t cur_off, size_t next_off, int main_wgs, int rot,
                                                     float* __restrict__ out)
{
    const int wg = blockIdx.x;
    if (wg < main_wgs) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(buf + cur_off + (size_t)wg * (kSlice / 4) + threadIdx.x * 4);
        out[(size_t)wg * 1024 + threadIdx.x] = v[0] + v[1] + v[2] + v[3] + out[(size_t)wg * 1024 + threadIdx.x] * 0.0f;
    } else {
        const int s = (wg - main_wgs + rot) % main_wgs;
        const f32x4 v = *reinterpret_cast<const f32x4*>(buf + next_off + (size_t)s * (kSlice / 4) + threadIdx.x * 4);
        if (v[0] == 123456.789f) out[0] = v[1];
    }
}

int main()
{
    const size_t total = (size_t)1 << 30;  // 1 GB
    float *buf, *out;
    CHECK(hipMalloc(&buf, total));
    CHECK(hipMalloc(&out, (size_t)512 * 1024 * 4));
    CHECK(hipMemset(buf, 0, total));
    CHECK(hipMemset(out, 0, (size_t)512 * 1024 * 4));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int main_wgs : {48, 144, 192}) {
        const size_t window = (size_t)main_wgs * kSlice;            // bytes per window
        const int n_windows = (int)(total / window);
        const int iters = n_windows < 1000 ? n_windows - 2 : 1000;
        for (int variant = 0; variant < 4; ++variant) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                // (as a graph: stream launches from the host are 3 us apart whatever the kernels do)
                hipStream_t st;
                CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
                hipGraph_t g;
                hipGraphExec_t ge;
                CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                for (int i = 0; i < iters; ++i) {
                    const size_t cur = variant == 1 ? 0 : (size_t)i * (window / 4);
                    const size_t next = (size_t)(i + 1) * (window / 4);
                    const int grid = variant >= 2 ? 2 * main_wgs : main_wgs;
                    hipLaunchKernelGGL(chain_kernel, dim3(grid), dim3(1024), 0, st, buf, cur, next, main_wgs, variant == 3 ? 1 : 0, out);
                }
                CHECK(hipStreamEndCapture(st, &g));
                CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CHECK(hipGraphLaunch(ge, st));
                CHECK(hipStreamSynchronize(st));
                CHECK(hipEventRecord(a, st));
                CHECK(hipGraphLaunch(ge, st));
                CHECK(hipEventRecord(b, st));
                CHECK(hipEventSynchronize(b));
                CHECK(hipGraphExecDestroy(ge));
                CHECK(hipGraphDestroy(g));
                CHECK(hipStreamDestroy(st));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, a, b));
                best = ms < best ? ms : best;
            }
            const char* names[4] = {"cold windows", "one window again and again (L2-warm)", "cold, prefetched by the launch before (same XCD)",
                                    "cold, prefetched by the launch before (next XCD)"};
            std::printf("%3d workgroups x 16 KB: %-52s %.2f us per launch\n", main_wgs, names[variant], best * 1e3f / iters);
        }
    }
    return 0;
}
