// mfma_lab: where does the fp32 MFMA GEMM main loop lose its last 15 %?
// Standalone probe (tools/ only, not part of the library): the production 128x128x32 K-loop of
// kjarni_amd/csrc/gemm.hip with pieces switched off one at a time, on synthetic operands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_lab mfma_lab.hip && ./mfma_lab
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <string>
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
enum : int { NOGLOBAL = 1, NOBARRIER = 2, NOLDSREAD = 4, SETPRIO = 8, NOSTORE = 16, EARLYREAD = 32, SPREADW = 64, EPI_LDS = 128,
              EPI_DIRECT = 256, SPREADW2 = 512, CONSTSTORE = 1024, NOSTORE_KEEPLOAD = 2048 };

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
                                                     float* __restrict__ Y, int64_t M, int N, int K, int n_tiles, int stagger)
{
    // One-time phase shift of the second resident workgroup of every CU (first generation only: dispatch follows
    // blockIdx): its tile boundaries then fall inside its neighbour's K-loop for the rest of the launch.
    if (stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512)
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(16);  // 16 * 64 cycles each
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
    f32x4 konst = f32x4{1.0f, 2.0f, 3.0f, (float)tid};
    float sink = 0.0f;
    auto store_piece = [&](int stage, int i) {  // i in 0..7: A pieces 0-3, B pieces 4-7
        if (FLAGS & CONSTSTORE) {  // LDS store traffic without the dependence on the global loads
            *reinterpret_cast<f32x4*>((i < 4 ? sA : sB) + stage * TILE + st_off + 32 * (i & 3) * STRIDE) = konst;
            sink += (i < 4 ? ga[i & 3] : gb[i & 3])[0];
            return;
        }
        if (FLAGS & NOSTORE_KEEPLOAD) {  // the loads and their waits without the LDS store
            sink += (i < 4 ? ga[i & 3] : gb[i & 3])[0];
            return;
        }
        if (i < 4)
            *reinterpret_cast<f32x4*>(sA + stage * TILE + st_off + 32 * i * STRIDE) = ga[i];
        else
            *reinterpret_cast<f32x4*>(sB + stage * TILE + st_off + 32 * (i - 4) * STRIDE) = gb[i - 4];
    };
    auto load_piece = [&](int i, int k0) {
        if (i < 4)
            ga[i] = *reinterpret_cast<const f32x4*>(ga_ptr[i] + k0);
        else
            gb[i - 4] = *reinterpret_cast<const f32x4*>(gb_ptr[i - 4] + k0);
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
            if (FLAGS & (SPREADW | SPREADW2)) {
                // tile kt+1: registers -> LDS[cur^1], a few pieces per phase, each load of tile kt+2 right behind
                // the store that frees its registers; SPREADW: 3 + 3 + 2 over phases 0-2, SPREADW2: 2 per phase
                // over phases 0-2 + 2 after the barrier is impossible (the buffer is read then) -> 4 + 4 over 1-2
                const int first = (FLAGS & SPREADW) ? (p == 0 ? 0 : p == 1 ? 3 : 6) : (p == 1 ? 0 : 4);
                const int count = (FLAGS & SPREADW) ? (p == 0 ? 3 : p == 1 ? 3 : p == 2 ? 2 : 0) : (p == 1 || p == 2 ? 4 : 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < count) {
                        store_piece(cur ^ 1, first + i);
                        load_piece(first + i, (kt + 2 < nk ? kt + 2 : nk - 1) * BK);  // redundant reload at the tail: no branch
                    }
                mfma16(acc, use);
                if (p + 1 < NKK) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < count) {
                        if constexpr ((FLAGS & SPREADW) != 0)
                            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        else
                            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            } else if (p == 1) {
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
    if (FLAGS & EPI_LDS) {
        // the production epilogue: transpose through wave-private LDS, 16-byte row-contiguous stores
        constexpr int ES = 68;
        float* sw = smem + wid * (64 * ES);
        const int e_row = lane >> 4, e_c4 = lane & 15;
        const int n = n0 + wc * 64 + e_c4 * 4;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sw[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * ES + j * 32 + l31] = acc[i][j][r];
        const int64_t m_base = m0 + wr * 64 + e_row;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            f32x4 v = *reinterpret_cast<const f32x4*>(sw + (it * 4 + e_row) * ES + e_c4 * 4);
            *reinterpret_cast<f32x4*>(Y + (m_base + it * 4) * N + n) = v;
        }
        return;
    }
    if (FLAGS & EPI_DIRECT) {
        // straight from the accumulator layout: a register is 32 consecutive columns of one row per lane half
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Y[(m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * N + n0 + wc * 64 + j * 32 + l31] = acc[i][j][r];
        return;
    }
    // minimal epilogue: keep the accumulators live, one 64-byte store per lane
    float s = sink;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    Y[(m0 + wr * 64 + l31) * N + n0 + wc * 64 + half] = s;
}

template <int FLAGS>
void run(const char* name, const float* A, const float* W, float* Y, int64_t M, int N, int K, int extra_lds, int stagger = 0)
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
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(lab_kernel<FLAGS>, grid, dim3(256), lds, 0, A, W, Y, M, N, K, n_tiles, stagger);
        CHECK(hipEventRecord(a));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(lab_kernel<FLAGS>, grid, dim3(256), lds, 0, A, W, Y, M, N, K, n_tiles, stagger);
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


// ---- The tile with WM x 2 waves: WM = 2 is the production shape (128 x 128, two workgroups per CU), WM = 4 the 256 x 128
// eight-wave tile (one workgroup per CU, a quarter less staging per MFMA).  Same loop for both: tile kt + 1 goes registers -> LDS
// in phase 1 of step kt with the loads of tile kt + 2 behind it; EPI: the LDS-transposed epilogue with 16-byte stores.
template <int WM, bool EPI>
__global__ __launch_bounds__(128 * WM, WM == 2 ? 2 : 1) void lab_wm_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                                            float* __restrict__ Y, int64_t M, int N, int K, int n_tiles)
{
    constexpr int TBM = 64 * WM, TA = TBM * STRIDE, TB = BN * STRIDE, NB = 8 / WM;  // B pieces per thread (A: always 4)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;            // [2][TBM][STRIDE]
    float* sB = smem + 2 * TA;   // [2][128][STRIDE]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, l31 = lane & 31, half = lane >> 5;
    const int64_t nwg = gridDim.x;
    const int64_t xcd = blockIdx.x % 8, slot = blockIdx.x / 8, q8 = nwg / 8, r8 = nwg % 8;
    const int64_t bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int64_t m0 = (bid / n_tiles) * TBM;
    const int n0 = (int)(bid % n_tiles) * BN;
    const int ld_row = tid / 8, ld_c4 = tid % 8;  // rows 0 .. 16 WM - 1
    const float* ga_ptr[4];
    const float* gb_ptr[NB];
#pragma unroll
    for (int i = 0; i < 4; ++i) ga_ptr[i] = A + (m0 + ld_row + 16 * WM * i) * K + ld_c4 * 4;
#pragma unroll
    for (int i = 0; i < NB; ++i) gb_ptr[i] = W + (int64_t)(n0 + ld_row + 16 * WM * i) * K + ld_c4 * 4;
    f32x4 ga[4], gb[NB];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ga[i] = *reinterpret_cast<const f32x4*>(ga_ptr[i] + k0);
#pragma unroll
        for (int i = 0; i < NB; ++i) gb[i] = *reinterpret_cast<const f32x4*>(gb_ptr[i] + k0);
    };
    const int st_off = ld_row * STRIDE + ld_c4 * 4;
    auto store_tiles = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(sA + stage * TA + st_off + 16 * WM * i * STRIDE) = ga[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(sB + stage * TB + st_off + 16 * WM * i * STRIDE) = gb[i];
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
    if (nk > 1) load_tiles(BK);
    __syncthreads();
    Frag fr[2];
    read_frag(fr[0], sA + a_off, sB + b_off, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const float* pa = sA + cur * TA + a_off;
        const float* pb = sB + cur * TB + b_off;
#pragma unroll
        for (int p = 0; p < NKK; ++p) {
            Frag& use = fr[p & 1];
            Frag& nxt = fr[(p + 1) & 1];
            if (p + 1 < NKK) {
                read_frag(nxt, pa, pb, p + 1);
            } else {
                __syncthreads();
                read_frag(nxt, sA + (cur ^ 1) * TA + a_off, sB + (cur ^ 1) * TB + b_off, 0);
            }
            if (p == 1) {
                store_tiles(cur ^ 1);
                if (kt + 2 < nk) load_tiles((kt + 2) * BK);
            }
            mfma16(acc, use);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (EPI) {
        constexpr int ES = 68;
        float* sw = smem + wid * (32 * ES);  // (a 32-row round at a time: 8 waves x 32 x 68 floats fit the operand stages)
        const int e_row = lane >> 4, e_c4 = lane & 15;
        const int n = n0 + wc * 64 + e_c4 * 4;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sw[((r & 3) + 8 * (r >> 2) + 4 * half) * ES + j * 32 + l31] = acc[i][j][r];
            const int64_t m_base = m0 + wr * 64 + i * 32 + e_row;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                f32x4 v = *reinterpret_cast<const f32x4*>(sw + (it * 4 + e_row) * ES + e_c4 * 4);
                *reinterpret_cast<f32x4*>(Y + (m_base + it * 4) * N + n) = v;
            }
        }
        return;
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    Y[(m0 + wr * 64 + l31) * N + n0 + wc * 64 + half] = s;
}

template <int WM, bool EPI>
void run_wm(const char* name, const float* A, const float* W, float* Y, int64_t M, int N, int K)
{
    const int lds = 2 * (64 * WM + BN) * STRIDE * 4;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_wm_kernel<WM, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int n_tiles = N / BN;
    dim3 grid((unsigned)(M / (64 * WM) * n_tiles));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const double flops = 2.0 * M * N * K;
    const int iters = (int)(0.4 / (flops / 125e12)) + 1;
    for (int rep = 0; rep < 2; ++rep) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((lab_wm_kernel<WM, EPI>), grid, dim3(128 * WM), lds, 0, A, W, Y, M, N, K, n_tiles);
        CHECK(hipEventRecord(a));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((lab_wm_kernel<WM, EPI>), grid, dim3(128 * WM), lds, 0, A, W, Y, M, N, K, n_tiles);
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

// ---- LDS-DMA variant: global -> LDS directly (no VGPR staging, no ds_write), unpadded [128][32] tiles with the
// 16-byte slot XOR-swizzled by (row >> 1) & 7 (conflict-free ds_read_b128), the swizzle applied on the SOURCE address.
constexpr int DTILE = BM * BK;  // floats per operand tile per stage (16 KiB)
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void read_frag_sw(Frag& f, const float* tA, const float* tB, int rowA, int rowB, int half, int kk)
{
    // logical 16-byte slot s = kk*2 + half of row r sits at physical slot s ^ ((r >> 1) & 7)
    const int s = kk * 2 + half;
    f.a0 = *reinterpret_cast<const f32x4*>(tA + rowA * BK + ((s ^ ((rowA >> 1) & 7)) << 2));
    f.a1 = *reinterpret_cast<const f32x4*>(tA + (rowA + 32) * BK + ((s ^ (((rowA + 32) >> 1) & 7)) << 2));
    f.b0 = *reinterpret_cast<const f32x4*>(tB + rowB * BK + ((s ^ ((rowB >> 1) & 7)) << 2));
    f.b1 = *reinterpret_cast<const f32x4*>(tB + (rowB + 32) * BK + ((s ^ (((rowB + 32) >> 1) & 7)) << 2));
}

template <int SPREAD>
__global__ __launch_bounds__(256, 2) void lab_dma_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                         float* __restrict__ Y, int64_t M, int N, int K, int n_tiles)
{
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    float* sA = smem;               // [2][128][32]
    float* sB = smem + 2 * DTILE;   // [2][128][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the DMA's LDS base goes through M0
    const int wr = wid >> 1, wc = wid & 1, l31 = lane & 31, half = lane >> 5;
    const int64_t nwg = gridDim.x;
    const int64_t xcd = blockIdx.x % 8, slot = blockIdx.x / 8, q8 = nwg / 8, r8 = nwg % 8;
    const int64_t bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int64_t m0 = (bid / n_tiles) * BM;
    const int n0 = (int)(bid % n_tiles) * BN;
    // DMA piece i of a wave: 8 rows x 128 B = 1 KiB.  Wave w stages rows [32w, 32w+32) of both tiles: 4 pieces each.
    // Lane l of a piece lands at LDS row 8*i + l/8, physical slot l%8 and must therefore FETCH logical slot p ^ f(row).
    const int prow = lane >> 3, pslot = lane & 7;
    const float* ga_ptr[4];
    const float* gb_ptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wid * 32 + i * 8 + prow;
        const int ls = pslot ^ ((row >> 1) & 7);
        ga_ptr[i] = A + (m0 + row) * K + ls * 4;
        gb_ptr[i] = W + (int64_t)(n0 + row) * K + ls * 4;
    }
    auto dma_piece = [&](int stage, int i, int k0) {
        // LDS destination: wave-uniform base (piece start); the hardware adds lane * 16
        float* la = sA + stage * DTILE + (wid * 32 + i * 8) * BK;
        float* lb = sB + stage * DTILE + (wid * 32 + i * 8) * BK;
        __builtin_amdgcn_global_load_lds((gptr_t)(ga_ptr[i] + k0), (lptr_t)la, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(gb_ptr[i] + k0), (lptr_t)lb, 16, 0, 0);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int nk = K / BK;
    const int rowA = wr * 64 + l31, rowB = wc * 64 + l31;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(0, i, 0);
    __syncthreads();  // emits vmcnt(0): tile 0 has landed for every wave
    Frag fr[2];
    read_frag_sw(fr[0], sA, sB, rowA, rowB, half, 0);

    auto step = [&](auto more_tag, int kt) {
        constexpr bool MORE = decltype(more_tag)::value;  // tile kt+1 exists
        const int cur = kt & 1;
        const float* tA = sA + cur * DTILE;
        const float* tB = sB + cur * DTILE;
#pragma unroll
        for (int p = 0; p < NKK; ++p) {
            Frag& use = fr[p & 1];
            Frag& nxt = fr[(p + 1) & 1];
            if (p + 1 < NKK) {
                read_frag_sw(nxt, tA, tB, rowA, rowB, half, p + 1);
            } else {
                __syncthreads();  // vmcnt(0) + barrier: everyone's DMA for tile kt+1 landed, everyone done reading cur
                if (MORE) read_frag_sw(nxt, sA + (cur ^ 1) * DTILE, sB + (cur ^ 1) * DTILE, rowA, rowB, half, 0);
            }
            // tile kt+1 -> the other buffer (free since the barrier that ended step kt-1)
            int n_dma = 0;
            if (MORE) {
                if (SPREAD) {
                    if (p < 2) {
                        dma_piece(cur ^ 1, 2 * p, (kt + 1) * BK);
                        dma_piece(cur ^ 1, 2 * p + 1, (kt + 1) * BK);
                        n_dma = 4;
                    }
                } else if (p == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) dma_piece(cur ^ 1, i, (kt + 1) * BK);
                    n_dma = 8;
                }
            }
            mfma16(acc, use);
            if (p + 1 < NKK || MORE) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // DS reads first
            if (n_dma == 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM (the DMA)
                }
            } else if (n_dma == 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int kt = 0;
    for (; kt + 1 < nk; ++kt) step(std::true_type{}, kt);
    step(std::false_type{}, kt);
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    Y[(m0 + wr * 64 + l31) * N + n0 + wc * 64 + half] = s;
}

template <int SPREAD>
void run_dma(const char* name, const float* A, const float* W, float* Y, int64_t M, int N, int K, int extra_lds)
{
    const int lds = 2 * 2 * DTILE * 4 + extra_lds;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_dma_kernel<SPREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int n_tiles = N / BN;
    dim3 grid((unsigned)(M / BM * n_tiles));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const double flops = 2.0 * M * N * K;
    const int iters = (int)(0.4 / (flops / 125e12)) + 1;
    for (int rep = 0; rep < 2; ++rep) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(lab_dma_kernel<SPREAD>, grid, dim3(256), lds, 0, A, W, Y, M, N, K, n_tiles);
        CHECK(hipEventRecord(a));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(lab_dma_kernel<SPREAD>, grid, dim3(256), lds, 0, A, W, Y, M, N, K, n_tiles);
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

// max |difference| of the checksum outputs of two kernels (same K order per accumulator => bit-equal expected)
double compare(const float* Y1, const float* Y2, int64_t M, int N)
{
    std::vector<float> a((size_t)M * N), b((size_t)M * N);
    CHECK(hipMemcpy(a.data(), Y1, a.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), Y2, b.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    size_t nz = 0;
    for (int64_t m = 0; m < M; ++m)
        for (int n = 0; n < N; n += 64)
            for (int h = 0; h < 2; ++h) {
                const size_t i = (size_t)m * N + n + h;
                const double d = std::fabs((double)a[i] - (double)b[i]);
                if (d > worst) worst = d;
                if (a[i] != 0.0f) ++nz;
            }
    printf("checksum compare: max |diff| = %g over %zu non-zero sums\n", worst, nz);
    return worst;
}

// ---- persistent, cross-tile pipelined variant --------------------------------------------------------------
// One workgroup walks several output tiles.  The K-step stream never drains at a tile boundary: while the last
// K-steps of tile t run, the staging registers already fetch the first K-tiles of tile t+1 (no prologue bubble), and
// tile t's results leave straight from (a copy of) the accumulator registers as 4-byte row-segment stores threaded
// between the MFMAs of tile t+1's first K-step (no LDS round trip: the LDS is busy with tile t+1).
template <int DUMMY>
__global__ __launch_bounds__(256, 2) void lab_persist_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                             float* __restrict__ Y, int64_t M, int N, int K, int n_tiles,
                                                             int total_tiles)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sB = smem + 2 * TILE;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, l31 = lane & 31, half = lane >> 5;
    const int ld_row = tid / 8, ld_c4 = tid % 8;
    const int nk = K / BK;
    // tiles of this workgroup: v = blockIdx + i * gridDim, mapped XCD-aware (v % 8 = blockIdx % 8 since gridDim % 8 == 0)
    auto tile_of = [&](int64_t v64, int64_t& m0, int& n0) {
        // 32-bit arithmetic only: a 64-bit division is ~150 instructions on this part, and this runs once per tile
        const uint32_t v = (uint32_t)v64, nwg = (uint32_t)total_tiles;
        const uint32_t xcd = v & 7u, slot = v >> 3, q8 = nwg >> 3, r8 = nwg & 7u;
        const uint32_t bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
        const uint32_t mt = __builtin_amdgcn_readfirstlane(bid / (uint32_t)n_tiles);
        m0 = (int64_t)mt * BM;
        n0 = (int)(__builtin_amdgcn_readfirstlane(bid) - mt * (uint32_t)n_tiles) * BN;
    };
    const int my_tiles = __builtin_amdgcn_readfirstlane((int)(((uint32_t)total_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x));
    if (my_tiles <= 0) return;

    uint32_t offA[4], offW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        offA[i] = (uint32_t)(((int64_t)(ld_row + 32 * i) * K + ld_c4 * 4) * 4);
        offW[i] = offA[i];
    }
    // load cursor: (tile li, K-tile lk) of the NEXT load_piece round, with that tile's descriptors
    int li = 0, lk = 0;
    int64_t lm0;
    int ln0;
    tile_of(blockIdx.x, lm0, ln0);
    auto make_rsrc = [&](const float* p, int64_t rows) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(rows * K * 4), 0x00020000);
    };
    __amdgpu_buffer_rsrc_t rA = make_rsrc(A + lm0 * K, BM), rW = make_rsrc(W + (int64_t)ln0 * K, BN);
    auto advance_load = [&]() {  // after a whole K-tile has been requested
        if (++lk == nk) {
            lk = 0;
            if (li + 1 < my_tiles) {  // past the last tile: keep re-reading its last K-tile (never used)
                ++li;
                tile_of((int64_t)blockIdx.x + (int64_t)li * gridDim.x, lm0, ln0);
                rA = make_rsrc(A + lm0 * K, BM);
                rW = make_rsrc(W + (int64_t)ln0 * K, BN);
            } else {
                lk = nk - 1;
            }
        }
    };
    f32x4 ga[4], gb[4];
    auto ld16 = [](__amdgpu_buffer_rsrc_t r, uint32_t off, int k0) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, k0 * 4, 0));
    };
    auto load_piece = [&](int i) {
        if (i < 4)
            ga[i] = ld16(rA, offA[i], lk * BK);
        else
            gb[i - 4] = ld16(rW, offW[i - 4], lk * BK);
    };
    const int st_off = ld_row * STRIDE + ld_c4 * 4;
    auto store_piece = [&](int stage, int i) {
        if (i < 4)
            *reinterpret_cast<f32x4*>(sA + stage * TILE + st_off + 32 * i * STRIDE) = ga[i];
        else
            *reinterpret_cast<f32x4*>(sB + stage * TILE + st_off + 32 * (i - 4) * STRIDE) = gb[i - 4];
    };
    const int a_off = (wr * 64 + l31) * STRIDE + half * 4;
    const int b_off = (wc * 64 + l31) * STRIDE + half * 4;

    f32x16 acc[2][2], outr[2][2];
    // prologue: K-tile 0 -> LDS[0], K-tile 1 in registers
#pragma unroll
    for (int i = 0; i < 8; ++i) load_piece(i);
    advance_load();
#pragma unroll
    for (int i = 0; i < 8; ++i) store_piece(0, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) load_piece(i);
    advance_load();
    __syncthreads();
    Frag fr[2];
    read_frag(fr[0], sA + a_off, sB + b_off, 0);

    // output cursor (the tile whose results sit in outr): a descriptor over the wave's 64 x 64 block, a constant
    // per-lane offset, and the row segment as a scalar offset -- no vector address arithmetic per store
    __amdgpu_buffer_rsrc_t rY = make_rsrc(Y, 1);
    const uint32_t voffY = (uint32_t)((4 * half * N + l31) * 4);
    auto put = [&](float v, int e) {
        const int i = e >> 5, j = (e >> 4) & 1, r = e & 15;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rY, voffY,
                                              ((i * 32 + (r & 3) + 8 * (r >> 2)) * N + j * 32) * 4, 0);
    };
    int step_parity = 0;
    // One K-step.  FIRST: first K-step of a tile (accumulators start from zero); FLUSH: outr holds a finished tile
    // whose 64 row segments are stored under this step's MFMAs.
    auto step = [&](auto first_tag, auto flush_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr bool FLUSH = decltype(flush_tag)::value;
        const int cur = step_parity;
        const float* pa = sA + cur * TILE + a_off;
        const float* pb = sB + cur * TILE + b_off;
#pragma unroll
        for (int p = 0; p < NKK; ++p) {
            Frag& use = fr[p & 1];
            Frag& nxt = fr[(p + 1) & 1];
            if (p + 1 < NKK) {
                read_frag(nxt, pa, pb, p + 1);
            } else {
                __syncthreads();
                read_frag(nxt, sA + (cur ^ 1) * TILE + a_off, sB + (cur ^ 1) * TILE + b_off, 0);
            }
            constexpr int PP = 3;
#pragma unroll
            for (int i = 0; i < PP; ++i) {
                const int piece = p * PP + i;
                if (p + 1 < NKK && piece < 8) {
                    store_piece(cur ^ 1, piece);
                    load_piece(piece);
                }
            }
            if (FLUSH) {
                // 16 of the 64 row segments per phase
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int e = p * 16 + q;
                    put(outr[e >> 5][(e >> 4) & 1][e & 15], e);
                }
            }
            if (FIRST && p == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (c == 0) {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(use.a0[c], use.b0[c], z, 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(use.a0[c], use.b1[c], z, 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(use.a1[c], use.b0[c], z, 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(use.a1[c], use.b1[c], z, 0, 0, 0);
                    } else {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(use.a0[c], use.b0[c], acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(use.a0[c], use.b1[c], acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(use.a1[c], use.b0[c], acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(use.a1[c], use.b1[c], acc[1][1], 0, 0, 0);
                    }
                }
            } else {
                mfma16(acc, use);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // DS reads first
            if (p + 1 < NKK) {
#pragma unroll
                for (int i = 0; i < PP; ++i)
                    if (p * PP + i < 8) {
                        __builtin_amdgcn_sched_group_barrier(0x008, FLUSH ? 2 : 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
                        if (FLUSH) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x040, 3, 0);  // VMEM writes
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x040, 3, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
                    }
            } else if (FLUSH) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (p + 1 < NKK && p == NKK - 2) advance_load();
        }
        step_parity ^= 1;
    };
    using TT = std::true_type;
    using FF = std::false_type;
    bool pending = false;
    for (int t = 0; t < my_tiles; ++t) {
        if (pending)
            step(TT{}, TT{});
        else
            step(TT{}, FF{});
        for (int kt = 1; kt < nk; ++kt) step(FF{}, FF{});
        // hand the finished tile to the output registers; its stores ride on the next tile's first K-step
        int64_t m0;
        int n0;
        tile_of((int64_t)blockIdx.x + (int64_t)t * gridDim.x, m0, n0);
        rY = __builtin_amdgcn_make_buffer_rsrc(Y + (m0 + wr * 64) * N + n0 + wc * 64, 0, (int)(64 * N * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) outr[i][j] = acc[i][j];
        pending = true;
    }
    // last tile: nothing left to hide the stores under
#pragma unroll
    for (int e = 0; e < 64; ++e) put(outr[e >> 5][(e >> 4) & 1][e & 15], e);
}

void run_persist(const char* name, const float* A, const float* W, float* Y, int64_t M, int N, int K, int blocks)
{
    const int lds = 2 * 2 * TILE * 4;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_persist_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int n_tiles = N / BN;
    const int total = (int)(M / BM * n_tiles);
    dim3 grid((unsigned)(blocks < total ? blocks : total));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const double flops = 2.0 * M * N * K;
    const int iters = (int)(0.4 / (flops / 125e12)) + 1;
    for (int rep = 0; rep < 2; ++rep) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(lab_persist_kernel<0>, grid, dim3(256), lds, 0, A, W, Y, M, N, K, n_tiles, total);
        CHECK(hipEventRecord(a));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(lab_persist_kernel<0>, grid, dim3(256), lds, 0, A, W, Y, M, N, K, n_tiles, total);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        ms /= iters;
        const double tf = flops / (ms * 1e-3) / 1e12;
        printf("%-44s blocks=%4d K=%5d N=%5d  %8.4f ms %7.2f TFLOP/s %5.1f%%\n", name, (int)grid.x, K, N, ms, tf, tf / 157.3 * 100);
        fflush(stdout);
    }
}

double compare_full(const float* Y1, const float* Y2, int64_t M, int N)
{
    std::vector<float> a((size_t)M * N), b((size_t)M * N);
    CHECK(hipMemcpy(a.data(), Y1, a.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), Y2, b.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    size_t nz = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        const double d = std::fabs((double)a[i] - (double)b[i]);
        if (d > worst) worst = d;
        if (a[i] != 0.0f) ++nz;
    }
    printf("full compare: max |diff| = %g over %zu non-zero outputs of %zu\n", worst, nz, a.size());
    return worst;
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
    float* Y2;
    CHECK(hipMalloc(&Y2, (size_t)M * N * 4));
    CHECK(hipMemset(Y, 0, (size_t)M * N * 4));
    CHECK(hipMemset(Y2, 0, (size_t)M * N * 4));
    if (argc > 1 && std::string(argv[1]) == "wm") {  // the eight-wave 256 x 128 tile against the production shape, same loop
        for (int k : {384, 1536}) {
            run_wm<2, false>("128x128, 4 waves, 2 WG/CU, trivial epilogue", A, W, Y, M, 1536, k);
            run_wm<4, false>("256x128, 8 waves, 1 WG/CU, trivial epilogue", A, W, Y2, M, 1536, k);
            run_wm<2, true>("128x128, 4 waves, 2 WG/CU, LDS epilogue", A, W, Y, M, 1536, k);
            run_wm<4, true>("256x128, 8 waves, 1 WG/CU, LDS epilogue", A, W, Y2, M, 1536, k);
            compare_full(Y, Y2, M, 1536);
        }
        return 0;
    }
    for (int n : {1536, 384}) {
        CHECK(hipMemset(Y, 0, (size_t)M * N * 4));
        CHECK(hipMemset(Y2, 0, (size_t)M * N * 4));
        run<SPREADW | EPI_LDS>("spread + LDS epilogue, K=384", A, W, Y, M, n, 384, 0);
        run<SPREADW>("spread + trivial epilogue, K=384", A, W, Y2, M, n, 384, 0);
        run_persist("persistent cross-tile pipeline, K=384", A, W, Y2, M, n, 384, 512);
        compare_full(Y, Y2, M, n);
    }
    run<SPREADW | EPI_LDS>("spread + LDS epilogue, K=1536 N=384", A, W, Y, M, 384, 1536, 0);
    run_persist("persistent cross-tile pipeline, K=1536 N=384", A, W, Y2, M, 384, 1536, 512);
    compare_full(Y, Y2, M, 384);
    return 0;
}
