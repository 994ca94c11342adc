// mfma_lab: where does the fp32 MFMA GEMM main loop lose its last 15 %?
// Standalone probe (tools/ only, not part of the library): the production 128x128x32 K-loop of
// kjarni_amd/csrc/gemm.hip with pieces switched off one at a time, on synthetic operands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_lab mfma_lab.hip && ./mfma_lab
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

constexpr int BM = 128, BN = 128, BK = 32, NKK = 4, STRIDE = BK + 4, TILE = BM * STRIDE;
enum : int { NOGLOBAL = 1, NOBARRIER = 2, NOLDSREAD = 4, SETPRIO = 8, NOSTORE = 16, EARLYREAD = 32 };

struct Frag {
    f32x4 a0, a1, b0, b1;
};

__device__ __forceinline__ void read_frag(Frag& f, const float* pa, const float* pb, int kk)
{
    f.a0 = *reinterpret_cast<const f32x4*>(pa + kk * 8);
    f.a1 = *reinterpret_cast<const f32x4*>(pa + 32 * STRIDE + kk * 8);
    f.b0 = *reinterpret_cast<const f32x4*>(pb + kk * 8);
    f.b1 = *reinterpret_cast<const f32x4*>(pb + 32 * STRIDE + kk * 8);
}

__device__ __forceinline__ void mfma16(f32x16 (&acc)[2][2], const Frag& f)
{
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[c], f.b0[c], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[c], f.b1[c], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[c], f.b0[c], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[c], f.b1[c], acc[1][1], 0, 0, 0);
    }
}

template <int FLAGS>
__global__ __launch_bounds__(256, 2) void lab_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                     float* __restrict__ Y, int64_t M, int N, int K, int n_tiles)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sB = smem + 2 * TILE;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, l31 = lane & 31, half = lane >> 5;
    const int64_t nwg = gridDim.x;
    const int64_t xcd = blockIdx.x % 8, slot = blockIdx.x / 8, q8 = nwg / 8, r8 = nwg % 8;
    const int64_t bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int64_t m0 = (bid / n_tiles) * BM;
    const int n0 = (int)(bid % n_tiles) * BN;
    const int ld_row = tid / 8, ld_c4 = tid % 8;
    const float* ga_ptr[4];
    const float* gb_ptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ga_ptr[i] = A + (m0 + ld_row + 32 * i) * K + ld_c4 * 4;
        gb_ptr[i] = W + (int64_t)(n0 + ld_row + 32 * i) * K + ld_c4 * 4;
    }
    f32x4 ga[4], gb[4];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ga[i] = *reinterpret_cast<const f32x4*>(ga_ptr[i] + k0);
            gb[i] = *reinterpret_cast<const f32x4*>(gb_ptr[i] + k0);
        }
    };
    const int st_off = ld_row * STRIDE + ld_c4 * 4;
    auto store_tiles = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(sA + stage * TILE + st_off + 32 * i * STRIDE) = ga[i];
            *reinterpret_cast<f32x4*>(sB + stage * TILE + st_off + 32 * i * STRIDE) = gb[i];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int nk = K / BK;
    const int a_off = (wr * 64 + l31) * STRIDE + half * 4;
    const int b_off = (wc * 64 + l31) * STRIDE + half * 4;
    load_tiles(0);
    store_tiles(0);
    store_tiles(1);
    if (nk > 1) load_tiles(BK);
    __syncthreads();
    Frag fr[2];
    read_frag(fr[0], sA + a_off, sB + b_off, 0);
    read_frag(fr[1], sA + a_off, sB + b_off, 1);

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const float* pa = sA + cur * TILE + a_off;
        const float* pb = sB + cur * TILE + b_off;
#pragma unroll
        for (int p = 0; p < NKK; ++p) {
            Frag& use = fr[p & 1];
            Frag& nxt = fr[(p + 1) & 1];
            if (!(FLAGS & NOLDSREAD)) {
                if (p + 1 < NKK) {
                    read_frag(nxt, pa, pb, p + 1);
                } else {
                    if (!(FLAGS & NOBARRIER)) __syncthreads();
                    read_frag(nxt, sA + (cur ^ 1) * TILE + a_off, sB + (cur ^ 1) * TILE + b_off, 0);
                }
            } else if (p + 1 == NKK && !(FLAGS & NOBARRIER)) {
                __syncthreads();
            }
            if (p == 1) {
                if (!(FLAGS & NOGLOBAL) && !(FLAGS & NOSTORE)) store_tiles(cur ^ 1);
                if (!(FLAGS & NOGLOBAL) && kt + 2 < nk) load_tiles((kt + 2) * BK);
                if (FLAGS & SETPRIO) __builtin_amdgcn_s_setprio(1);
                mfma16(acc, use);
                if (FLAGS & SETPRIO) __builtin_amdgcn_s_setprio(0);
                if (!(FLAGS & NOLDSREAD)) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                if (!(FLAGS & NOGLOBAL) && !(FLAGS & NOSTORE)) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    }
                }
                if (!(FLAGS & NOGLOBAL)) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
                __builtin_amdgcn_sched_barrier(0);
                if (FLAGS & SETPRIO) __builtin_amdgcn_s_setprio(1);
                mfma16(acc, use);
                if (FLAGS & SETPRIO) __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // minimal epilogue: keep the accumulators live, one 64-byte store per lane
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    Y[(m0 + wr * 64 + l31) * N + n0 + wc * 64 + half] = s;
}

template <int FLAGS>
void run(const char* name, const float* A, const float* W, float* Y, int64_t M, int N, int K, int extra_lds)
{
    const int lds = 2 * 2 * TILE * 4 + extra_lds;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_kernel<FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int n_tiles = N / BN;
    dim3 grid((unsigned)(M / BM * n_tiles));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const double flops = 2.0 * M * N * K;
    const int iters = (int)(0.4 / (flops / 125e12)) + 1;
    for (int rep = 0; rep < 2; ++rep) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(lab_kernel<FLAGS>, grid, dim3(256), lds, 0, A, W, Y, M, N, K, n_tiles);
        CHECK(hipEventRecord(a));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(lab_kernel<FLAGS>, grid, dim3(256), lds, 0, A, W, Y, M, N, K, n_tiles);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        ms /= iters;
        const double tf = flops / (ms * 1e-3) / 1e12;
        printf("%-44s lds=%6d K=%5d N=%5d  %8.4f ms %7.2f TFLOP/s %5.1f%%\n", name, lds, K, N, ms, tf, tf / 157.3 * 100);
        fflush(stdout);
    }
}

int main(int argc, char** argv)
{
    const int64_t M = 131072;
    const int N = 1536, K = 1536;
    std::vector<float> h((size_t)M * K);
    unsigned s = 12345;
    for (auto& v : h) {
        s = s * 1664525u + 1013904223u;
        v = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f;
    }
    float *A, *W, *Y;
    CHECK(hipMalloc(&A, (size_t)M * K * 4));
    CHECK(hipMalloc(&W, (size_t)N * K * 4));
    CHECK(hipMalloc(&Y, (size_t)M * N * 4));
    CHECK(hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(W, h.data() + 777, (size_t)N * K * 4, hipMemcpyHostToDevice));
    run<0>("production loop, 2 WG/CU", A, W, Y, M, N, K, 0);
    run<0>("production loop, 1 WG/CU", A, W, Y, M, N, K, 40 * 1024);
    run<SETPRIO>("setprio around MFMA clusters, 2 WG/CU", A, W, Y, M, N, K, 0);
    run<NOSTORE>("no LDS stores (global loads kept), 2 WG/CU", A, W, Y, M, N, K, 0);
    run<NOGLOBAL>("no global loads / LDS stores, 2 WG/CU", A, W, Y, M, N, K, 0);
    run<NOGLOBAL>("no global loads / LDS stores, 1 WG/CU", A, W, Y, M, N, K, 40 * 1024);
    run<NOGLOBAL | NOBARRIER>("... and no barrier, 2 WG/CU", A, W, Y, M, N, K, 0);
    run<NOGLOBAL | NOBARRIER>("... and no barrier, 1 WG/CU", A, W, Y, M, N, K, 40 * 1024);
    run<NOBARRIER>("no barrier only (racy), 2 WG/CU", A, W, Y, M, N, K, 0);
    run<NOGLOBAL | NOBARRIER | NOLDSREAD>("MFMA only, 2 WG/CU", A, W, Y, M, N, K, 0);
    run<NOGLOBAL | NOBARRIER | NOLDSREAD>("MFMA only, 1 WG/CU", A, W, Y, M, N, K, 40 * 1024);
    return 0;
}
