// direct_lab: the K-loop of the LayerNorm projections (gemm_nt_f32_mfma_ln: a workgroup owns 64 rows x 384 columns, four
// waves of 64 x 96) in two forms on the same operands, without the LayerNorm epilogue:
//   staged   the production loop: [rows][16] operand tiles staged global -> registers -> LDS (double-buffered, one barrier
//            per K-step), fragments read back from LDS;
//   direct   no LDS, no barrier: every wave loads its MFMA fragments straight from global memory (each lane 32 contiguous
//            bytes of its row per K-step: k 0..7 on lanes 0-31, k 8..15 on lanes 32-63), the A rows re-read by all four waves
//            (L1 / L2), the W rows private to their wave; the next K-step's fragments in flight under this one's MFMAs.
// Standalone (tools/ only): hipcc --offload-arch=gfx950 -O3 -std=c++17 -o direct_lab direct_lab.hip && ./direct_lab
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

constexpr int NT = 3, BM = 64, BN = 128 * NT, BK = 16, STRIDE = BK + 4, STAGE = (BM + BN) * STRIDE;

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__device__ __forceinline__ f32x4 ld16(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, int soff_bytes)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, soff_bytes, 0));
}

// accumulators out in the MFMA layout (4-byte stores: a lab epilogue, the same for both forms)
__device__ __forceinline__ void store_acc(const f32x16 (&acc)[2][NT], float* Y, int64_t m0, int64_t M, int wid, int l31, int half, int stores)
{
    if (stores) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t m = m0 + i * 32 + acc_row(r, half);
                    if (m < M) Y[m * BN + wid * 32 * NT + j * 32 + l31] = acc[i][j][r];
                }
    } else {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 123456.789f) Y[0] = s;
    }
}

__global__ __launch_bounds__(256, 2) void staged_kernel(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ Y,
                                                        int64_t M, int K, int64_t total_tiles, int stores)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, half = lane >> 5;
    for (int64_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const int64_t m0 = tile * BM;
        const int ld_grp = tid >> 3;
        const int ld_row = (ld_grp >> 2) * 8 + (ld_grp & 3) + 4 * ((tid >> 2) & 1), ld_c4 = tid & 3;
        const int64_t rows_a = (M - m0 < BM) ? (M - m0) : BM;
        const __amdgpu_buffer_rsrc_t rsrcA =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + m0 * K), 0, (int)(((rows_a - 1) * K + K) * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, (int)((int64_t)BN * K * 4), 0x00020000);
        const uint32_t offA = (uint32_t)(((int64_t)ld_row * K + ld_c4 * 4) * 4);
        uint32_t offW[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) offW[i] = (uint32_t)(((int64_t)(ld_row + 64 * i) * K + ld_c4 * 4) * 4);
        f32x4 ga, gb[6];
        const int st_off = ld_row * STRIDE + ld_c4 * 4;
        f32x16 acc[2][NT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        struct Fr {
            f32x4 a[2], b[NT];
        };
        const int a_off = l31 * STRIDE + half * 4;
        const int b_off = BM * STRIDE + (wid * 32 * NT + l31) * STRIDE + half * 4;
        auto read_frag = [&](Fr& f, int stage, int kk) {
            const float* base = smem + stage * STAGE + kk * 8;
            f.a[0] = *reinterpret_cast<const f32x4*>(base + a_off);
            f.a[1] = *reinterpret_cast<const f32x4*>(base + a_off + 32 * STRIDE);
#pragma unroll
            for (int j = 0; j < NT; ++j) f.b[j] = *reinterpret_cast<const f32x4*>(base + b_off + j * 32 * STRIDE);
        };
        auto mfma_phase = [&](const Fr& f) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][c], f.b[j][c], acc[i][j], 0, 0, 0);
        };
        const int nk = K / BK;
        ga = ld16(rsrcA, offA, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i) gb[i] = ld16(rsrcW, offW[i], 0);
        *reinterpret_cast<f32x4*>(smem + st_off) = ga;
#pragma unroll
        for (int i = 0; i < 6; ++i) *reinterpret_cast<f32x4*>(smem + BM * STRIDE + st_off + 64 * i * STRIDE) = gb[i];
        ga = ld16(rsrcA, offA, BK * 4);
#pragma unroll
        for (int i = 0; i < 6; ++i) gb[i] = ld16(rsrcW, offW[i], BK * 4);
        __syncthreads();
        Fr fr[2];
        read_frag(fr[0], 0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            const bool st = kt + 1 < nk, ldn = kt + 2 < nk;
            read_frag(fr[1], cur, 1);
            float* wbase = smem + (cur ^ 1) * STAGE;
            if (st) *reinterpret_cast<f32x4*>(wbase + st_off) = ga;
            if (ldn) ga = ld16(rsrcA, offA, (kt + 2) * BK * 4);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                if (st) *reinterpret_cast<f32x4*>(wbase + BM * STRIDE + st_off + 64 * i * STRIDE) = gb[i];
                if (ldn) gb[i] = ld16(rsrcW, offW[i], (kt + 2) * BK * 4);
            }
            mfma_phase(fr[0]);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 + NT, 0);
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            if (st) read_frag(fr[0], cur ^ 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_phase(fr[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        store_acc(acc, Y, m0, M, wid, l31, half, stores);
        __syncthreads();
    }
}

// DEPTH: K-steps of fragments in flight ahead of the one being multiplied (1 or 2)
template <int DEPTH>
__global__ __launch_bounds__(256, 2) void direct_kernel(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ Y,
                                                        int64_t M, int K, int64_t total_tiles, int stores)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, half = lane >> 5;
    struct Fr {
        f32x4 a[2][2], b[NT][2];  // [tile][q]: k = 8 half + 4 q + c of the K-step
    };
    for (int64_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const int64_t m0 = tile * BM;
        const int64_t rows_a = (M - m0 < BM) ? (M - m0) : BM;
        const __amdgpu_buffer_rsrc_t rsrcA =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + m0 * K), 0, (int)(((rows_a - 1) * K + K) * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, (int)((int64_t)BN * K * 4), 0x00020000);
        uint32_t offA[2], offW[NT];
#pragma unroll
        for (int i = 0; i < 2; ++i) offA[i] = (uint32_t)(((int64_t)(i * 32 + l31) * K + half * 8) * 4);
#pragma unroll
        for (int j = 0; j < NT; ++j) offW[j] = (uint32_t)(((int64_t)(wid * 32 * NT + j * 32 + l31) * K + half * 8) * 4);
        f32x16 acc[2][NT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        auto load = [&](Fr& f, int k0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int i = 0; i < 2; ++i) f.a[i][q] = ld16(rsrcA, offA[i] + q * 16, k0 * 4);
#pragma unroll
                for (int j = 0; j < NT; ++j) f.b[j][q] = ld16(rsrcW, offW[j] + q * 16, k0 * 4);
            }
        };
        auto mfma = [&](const Fr& f) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][q][c], f.b[j][q][c], acc[i][j], 0, 0, 0);
        };
        auto spread = [&]() {  // 10 loads dealt out between the 48 MFMAs of a K-step
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        const int nk = K / BK;  // (a multiple of DEPTH + 1: the lab's shapes)
        Fr fr[DEPTH + 1];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) load(fr[d], d * BK);
        for (int kt = 0; kt < nk; kt += DEPTH + 1) {
#pragma unroll
            for (int u = 0; u < DEPTH + 1; ++u) {
                // (loads past K read zeros through the descriptor's bounds check: no branch around a load)
                load(fr[(u + DEPTH) % (DEPTH + 1)], (kt + u + DEPTH) * BK);
                mfma(fr[u]);
                spread();
            }
        }
        store_acc(acc, Y, m0, M, wid, l31, half, stores);
    }
}

static double run(const char* name, int form, const float* A, const float* W, float* Y, int64_t M, int K, int grid_mode, int stores)
{
    const int64_t tiles = (M + BM - 1) / BM;
    const unsigned grid = (unsigned)(grid_mode ? std::min<int64_t>(tiles, 512) : tiles);
    const size_t lds = 2 * STAGE * sizeof(float);
    auto launch = [&]() {
        if (form == 0) hipLaunchKernelGGL(staged_kernel, dim3(grid), dim3(256), lds, 0, A, W, Y, M, K, tiles, stores);
        else if (form == 1) hipLaunchKernelGGL(direct_kernel<1>, dim3(grid), dim3(256), 0, 0, A, W, Y, M, K, tiles, stores);
        else hipLaunchKernelGGL(direct_kernel<2>, dim3(grid), dim3(256), 0, 0, A, W, Y, M, K, tiles, stores);
    };
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 20;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    const double tf = 2.0 * M * BN * K / (ms * 1e-3) / 1e12;
    printf("%-34s K=%4d grid=%-9s stores=%d  %8.4f ms  %7.2f TFLOP/s (%5.1f %%)\n", name, K, grid_mode ? "resident" : "per-tile", stores, ms, tf,
           tf / 157.3 * 100);
    fflush(stdout);
    return tf;
}

int main()
{
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&staged_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE * 4));
    const int64_t M = 262144;
    const int KMAX = 1536;
    std::vector<float> h((size_t)M * KMAX + 4096);
    unsigned s = 12345;
    for (auto& v : h) {
        s = s * 1664525u + 1013904223u;
        v = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f;
    }
    float *A, *W, *Y, *Y2;
    CHECK(hipMalloc(&A, (size_t)M * KMAX * 4));
    CHECK(hipMalloc(&W, (size_t)BN * KMAX * 4));
    CHECK(hipMalloc(&Y, (size_t)M * BN * 4));
    CHECK(hipMalloc(&Y2, (size_t)M * BN * 4));
    CHECK(hipMemcpy(A, h.data(), (size_t)M * KMAX * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(W, h.data() + 777, (size_t)BN * KMAX * 4, hipMemcpyHostToDevice));
    for (int K : {384, 1536}) {
        // correctness: both forms against a float64 product on sampled rows, and against each other
        CHECK(hipMemset(Y, 0, (size_t)M * BN * 4));
        CHECK(hipMemset(Y2, 0, (size_t)M * BN * 4));
        const int64_t tiles = M / BM;
        hipLaunchKernelGGL(staged_kernel, dim3((unsigned)tiles), dim3(256), 2 * STAGE * sizeof(float), 0, A, W, Y, M, K, tiles, 1);
        hipLaunchKernelGGL(direct_kernel<2>, dim3((unsigned)tiles), dim3(256), 0, 0, A, W, Y2, M, K, tiles, 1);
        CHECK(hipDeviceSynchronize());
        std::vector<float> y((size_t)8 * BN), y2((size_t)8 * BN);
        double worst = 0, worst2 = 0;
        for (int64_t row : {0L, 63L, 64L, 131071L, 262143L}) {
            CHECK(hipMemcpy(y.data(), Y + row * BN, BN * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(y2.data(), Y2 + row * BN, BN * 4, hipMemcpyDeviceToHost));
            for (int n = 0; n < BN; ++n) {
                double ref = 0;
                for (int k = 0; k < K; ++k) ref += (double)h[(size_t)row * K + k] * (double)h[777 + (size_t)n * K + k];
                worst = std::max(worst, std::fabs(ref - y[n]));
                worst2 = std::max(worst2, std::fabs(ref - y2[n]));
            }
        }
        printf("K=%d: max |staged - f64| %.3e, max |direct - f64| %.3e (sampled rows)\n", K, worst, worst2);
        for (int rep = 0; rep < 2; ++rep) {
            run("staged (LDS, barrier per K-step)", 0, A, W, Y, M, K, 0, 0);
            run("direct, 1 K-step ahead", 1, A, W, Y2, M, K, 0, 0);
            run("direct, 2 K-steps ahead", 2, A, W, Y2, M, K, 0, 0);
            run("staged (LDS, barrier per K-step)", 0, A, W, Y, M, K, 1, 0);
            run("direct, 1 K-step ahead", 1, A, W, Y2, M, K, 1, 0);
            run("direct, 2 K-steps ahead", 2, A, W, Y2, M, K, 1, 0);
        }
    }
    return 0;
}
