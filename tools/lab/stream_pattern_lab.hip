// What the corpus stream of the bf16 filter pass can reach: 16 rows x 64 bytes per load instruction (the MFMA B-operand dealing:
// lane (n = lane % 16, g = lane / 16) reads 16 bytes at row n, byte 128 s + 16 g (+ 64)) against 1 KB contiguous per instruction,
// same tile (16 rows x 1 536 bytes per wave), same number of requests in flight, nothing but a sum behind the loads.
// Standalone: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o stream_pattern_lab stream_pattern_lab.hip && ./stream_pattern_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int DIM = 384, NK = 12;

template <int PATTERN, int AUX>   // 0: rows x 64 B (filter), 1: contiguous
__global__ __launch_bounds__(256, 3) void stream_kernel(const float* __restrict__ corpus, int64_t n_docs, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int n16 = lane & 15, g = lane >> 4;
    const int64_t tiles = n_docs >> 4, waves = (int64_t)gridDim.x * 4, wave = (int64_t)blockIdx.x * 4 + wid;
    const uint32_t voff = PATTERN == 0 ? (uint32_t)((n16 * DIM + 4 * g) * 4) : (uint32_t)(lane * 16);
    float total = 0.0f;
    for (int64_t t = wave; t < tiles; t += waves) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(corpus + (t << 4) * DIM), 0, 16 * DIM * 4, 0x00020000);
        f32x4 x[2 * NK];
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            if (PATTERN == 0) {
                x[2 * s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + 128 * s, 0, AUX));
                x[2 * s + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + 128 * s + 64, 0, AUX));
            } else {
                x[2 * s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + 2048 * s, 0, AUX));
                x[2 * s + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + 2048 * s + 1024, 0, AUX));
            }
        }
#pragma unroll
        for (int s = 0; s < 2 * NK; ++s) total += (x[s][0] + x[s][1]) + (x[s][2] + x[s][3]);
    }
    if (total == 123.456f) out[0] = total;
}

template <int PATTERN, int AUX>
static void run(const char* name, const float* corpus, int64_t n, float* out, int grid)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<PATTERN, AUX>), dim3(grid), dim3(256), 0, 0, corpus, n, out);
    CHECK(hipEventRecord(e0, 0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<PATTERN, AUX>), dim3(grid), dim3(256), 0, 0, corpus, n, out);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-58s grid %4d  %8.1f us  %6.2f TB/s\n", name, grid, ms * 1e3 / reps, (double)n * DIM * 4 / (ms * 1e-3 / reps) / 1e12);
}

int main()
{
    const int64_t n = 1000000;
    float *corpus, *out;
    CHECK(hipMalloc(&corpus, n * DIM * 4));
    CHECK(hipMalloc(&out, 4));
    CHECK(hipMemset(corpus, 0x11, n * DIM * 4));
    for (int rep = 0; rep < 2; ++rep)
        for (int grid : {512, 768, 1024, 2048}) {
            run<0, 0>("rows x 64 B per instruction, plain", corpus, n, out, grid);
            run<0, 2>("rows x 64 B per instruction, nt", corpus, n, out, grid);
            run<1, 0>("1 KB contiguous per instruction, plain", corpus, n, out, grid);
            run<1, 2>("1 KB contiguous per instruction, nt", corpus, n, out, grid);
        }
    return 0;
}
