// Projection GEMM on the fp32 matrix cores:  Y[M,N] = X[M,K] * W[N,K]^T + b (+ epilogue).
//
// Replaces LinearLayer::matmul / matmul_noalloc
// (crates/kjarni-transformers/src/linear_layer/linear_layer.rs:160-282,
//  cpu/ops/matmul.rs:571-686, cpu/kernels/x86/f32.rs:8-127), the fused QKV
// projection (cpu/encoder/qkv_projection.rs:30-138) and the FFN's FC1+act /
// FC2 (cpu/feedforward/standard_new.rs:29-82) with the bias, activation and
// residual add (cpu/encoder/encoder_layer.rs:129-136, 155-163) fused into the
// epilogue.  W keeps the HF [out,in] row-major layout the reference uses, so
// both operands are K-contiguous.
//
// v_mfma_f32_32x32x2_f32 is an exact k-ordered f32 fma chain, i.e. the same
// arithmetic class as the reference's AVX2 FMA accumulation (different
// summation order only).
//
// Tiling: 128x128 block tile, BK = 32, 4 waves in a 2x2 grid, each wave a
// 64x64 sub-tile = 2x2 MFMA tiles of 32x32.  Operand tiles live in LDS as
// [128 rows][32 k] with a 36-float row stride, which makes the 16-byte
// fragment reads (one ds_read_b128 feeds four MFMAs: the k order inside an
// MFMA is free as long as A and B agree) bank-conflict free.  Global->LDS
// staging goes through registers, issued one K-step ahead (double-buffered
// LDS, one barrier per K-step).
#include <type_traits>

#include "device_utils.h"
#include "kernels.h"

namespace kjarni {

namespace {

int g_gemm_variant = 0;  // 0: software-pipelined kernel (default), 1: plain double-buffered loop

constexpr int EPI_GELU_LIBM = 6;  // tuning variant 3: erf-GELU with the libm-grade erff (A/B against the fast form)
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_STRIDE = BK + 4;               // floats
constexpr int TILE_FLOATS = BM * LDS_STRIDE;     // one operand tile
constexpr int GEMM_LDS_BYTES = 2 * 2 * TILE_FLOATS * 4;  // 2 operands x 2 stages
constexpr int EPI_STRIDE = 68;                           // floats; 4 waves x 64 x 68 x 4 B <= GEMM_LDS_BYTES

template <int EPI>
__device__ __forceinline__ float epilogue(float v)
{
    if (EPI == EPI_BIAS_GELU) return gelu_erf_fast(v);
    if (EPI == EPI_GELU_LIBM) return gelu_erf(v);
    if (EPI == EPI_BIAS_GELU_NEW) return gelu_tanh(v);
    if (EPI == EPI_BIAS_RELU) return fmaxf(v, 0.0f);
    if (EPI == EPI_BIAS_TANH) return tanhf(v);
    return v;
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_f32_mfma(const float* __restrict__ A, int64_t lda,
                                                           const float* __restrict__ W,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ R, int64_t ldr,
                                                           float* __restrict__ Y, int64_t ldy,
                                                           int64_t M, int N, int K, int n_tiles)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                    // [2][128][36]
    float* sB = smem + 2 * TILE_FLOATS;  // [2][128][36]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int l31 = lane & 31, half = lane >> 5;

    // Consecutive blocks walk N first so the X row-panel is re-read from L2.
    const int64_t bid = blockIdx.x;
    const int64_t m0 = (bid / n_tiles) * BM;
    const int n0 = (int)(bid % n_tiles) * BN;

    // Global staging: 1024 float4 per operand tile, 4 per thread.
    f32x4 ga[4], gb[4];
    const int ld_row = tid >> 3;  // + 32*i
    const int ld_c4 = tid & 7;

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = ld_row + 32 * i;
            const int64_t m = m0 + row;
            if (m < M)
                ga[i] = *reinterpret_cast<const f32x4*>(A + m * lda + k0 + ld_c4 * 4);
            else
                ga[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            gb[i] = *reinterpret_cast<const f32x4*>(W + (int64_t)(n0 + row) * K + k0 + ld_c4 * 4);
        }
    };
    auto store_tiles = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = ld_row + 32 * i;
            *reinterpret_cast<f32x4*>(sA + stage * TILE_FLOATS + row * LDS_STRIDE + ld_c4 * 4) = ga[i];
            *reinterpret_cast<f32x4*>(sB + stage * TILE_FLOATS + row * LDS_STRIDE + ld_c4 * 4) = gb[i];
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
    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    const int a_off = (wr * 64 + l31) * LDS_STRIDE + half * 4;
    const int b_off = (wc * 64 + l31) * LDS_STRIDE + half * 4;

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);

        const float* pa = sA + cur * TILE_FLOATS + a_off;
        const float* pb = sB + cur * TILE_FLOATS + b_off;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 a0 = *reinterpret_cast<const f32x4*>(pa + kk * 8);
            f32x4 a1 = *reinterpret_cast<const f32x4*>(pa + 32 * LDS_STRIDE + kk * 8);
            f32x4 b0 = *reinterpret_cast<const f32x4*>(pb + kk * 8);
            f32x4 b1 = *reinterpret_cast<const f32x4*>(pb + 32 * LDS_STRIDE + kk * 8);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b0[c], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b1[c], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b0[c], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b1[c], acc[1][1], 0, 0, 0);
            }
        }

        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }

    // Epilogue straight from the accumulators: lanes 0..31 of a register hold
    // 32 consecutive columns of one row (128-byte segments).
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wc * 64 + j * 32 + l31;
        const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wr * 64 + i * 32 + acc_row(r, half);
                if (m < M) {
                    float v = acc[i][j][r] + bv;
                    if (EPI == EPI_BIAS_RESIDUAL) v += R[m * ldr + n];
                    Y[m * ldy + n] = epilogue<EPI>(v);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Software-pipelined variant (same 128x128x32 tile, same arithmetic).
//
// The plain loop above exposes, once per K-step and in every wave at the same
// time (co-resident blocks run the same program in lockstep), the LDS-write ->
// barrier -> global-load issue -> first LDS read chain, and once per 16 MFMAs
// the latency of the fragment reads.  Here a K-step is four phases of 16 MFMAs
// and every memory operation is issued under the MFMAs of an earlier phase:
//   phase 0: read fragments kk=1                       | MFMA kk=0
//   phase 1: read fragments kk=2, write tile t+1 to the
//            other LDS buffer, issue global loads t+2  | MFMA kk=1
//   phase 2: read fragments kk=3                       | MFMA kk=2
//   phase 3: barrier, read fragments kk=0 of tile t+1  | MFMA kk=3 (from registers)
// so the only exposed wait is the barrier skew itself.
// ---------------------------------------------------------------------------
struct Frag {
    f32x4 a0, a1, b0, b1;
};

__device__ __forceinline__ void read_frag(Frag& f, const float* pa, const float* pb, int kk)
{
    f.a0 = *reinterpret_cast<const f32x4*>(pa + kk * 8);
    f.a1 = *reinterpret_cast<const f32x4*>(pa + 32 * LDS_STRIDE + kk * 8);
    f.b0 = *reinterpret_cast<const f32x4*>(pb + kk * 8);
    f.b1 = *reinterpret_cast<const f32x4*>(pb + 32 * LDS_STRIDE + kk * 8);
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

template <int EPI, int DIAG = 0>
__global__ __launch_bounds__(256, 2) void gemm_nt_f32_mfma_pipe(const float* __restrict__ A, int64_t lda,
                                                                const float* __restrict__ W,
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ R, int64_t ldr,
                                                                float* __restrict__ Y, int64_t ldy,
                                                                int64_t M, int N, int K, int n_tiles)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sB = smem + 2 * TILE_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int l31 = lane & 31, half = lane >> 5;

    // Workgroups are dealt round-robin over the 8 XCDs (private L2 each).  Remap so that every
    // XCD walks one contiguous run of tiles (N fastest): the tiles that share an X row panel
    // then hit the same L2 instead of fetching the panel once per XCD.  Bijective for any grid.
    const int64_t nwg = gridDim.x;
    const int64_t xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    const int64_t q8 = nwg / 8, r8 = nwg % 8;
    const int64_t bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int64_t m0 = (bid / n_tiles) * BM;
    const int n0 = (int)(bid % n_tiles) * BN;

    // Global staging pointers; rows past M are clamped (their results are never stored).
    const int ld_row = tid >> 3, ld_c4 = tid & 7;
    const float* ga_ptr[4];
    const float* gb_ptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int64_t m = m0 + ld_row + 32 * i;
        m = m < M ? m : M - 1;
        ga_ptr[i] = A + m * lda + ld_c4 * 4;
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
    const int st_off = ld_row * LDS_STRIDE + ld_c4 * 4;
    auto store_tiles = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(sA + stage * TILE_FLOATS + st_off + 32 * i * LDS_STRIDE) = ga[i];
            *reinterpret_cast<f32x4*>(sB + stage * TILE_FLOATS + st_off + 32 * i * LDS_STRIDE) = gb[i];
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
    const int a_off = (wr * 64 + l31) * LDS_STRIDE + half * 4;
    const int b_off = (wc * 64 + l31) * LDS_STRIDE + half * 4;

    // Prologue: tile 0 -> LDS[0], tile 1 in flight in registers, fragments kk=0 of tile 0.
    load_tiles(0);
    store_tiles(0);
    if (nk > 1) load_tiles(BK);
    __syncthreads();
    Frag f0, f1;
    read_frag(f0, sA + a_off, sB + b_off, 0);

    // One K-step.  STORE: tile kt+1 exists (registers -> other LDS buffer, and its first
    // fragments are fetched after the barrier); LOAD: tile kt+2 exists (global -> registers).
    auto step = [&](auto store_tag, auto load_tag, int kt) {
        constexpr bool STORE = decltype(store_tag)::value;
        constexpr bool LOAD = decltype(load_tag)::value;
        const int cur = kt & 1;
        const float* pa = sA + cur * TILE_FLOATS + a_off;
        const float* pb = sB + cur * TILE_FLOATS + b_off;

        // phase 0
        read_frag(f1, pa, pb, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(acc, f0);
        __builtin_amdgcn_sched_barrier(0);

        // phase 1: memory traffic of the next tiles is interleaved one-for-one with the MFMAs.
        read_frag(f0, pa, pb, 2);
        if (STORE) store_tiles(cur ^ 1);
        if (LOAD) load_tiles((kt + 2) * BK);
        mfma16(acc, f1);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // 4 x DS read
        if (STORE) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
            }
        }
        if (LOAD) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
            }
        }
        __builtin_amdgcn_sched_barrier(0);

        // phase 2
        read_frag(f1, pa, pb, 3);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(acc, f0);
        __builtin_amdgcn_sched_barrier(0);

        // phase 3: everyone is done reading LDS[cur] (fragments are in registers) and has
        // written LDS[cur^1]; cross the barrier, fetch the next tile's first fragments, then
        // run the last 16 MFMAs of this tile from registers.
        __syncthreads();
        if (STORE) read_frag(f0, sA + (cur ^ 1) * TILE_FLOATS + a_off, sB + (cur ^ 1) * TILE_FLOATS + b_off, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(acc, f1);
        __builtin_amdgcn_sched_barrier(0);
    };
    using T = std::true_type;
    using F = std::false_type;
    int kt = 0;
    for (; kt + 2 < nk; ++kt) step(T{}, T{}, kt);
    if (kt + 1 < nk) step(T{}, F{}, kt++);
    step(F{}, F{}, kt);

    if (DIAG == 1) {
        // Diagnostic build only (tools/kernel_bench.py): no epilogue; keeps the accumulators live.
        float sacc = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
        if (sacc == 123456.789f) Y[0] = sacc;
        return;
    }
    // Epilogue through LDS.  Straight from the accumulators a store instruction covers two
    // 128-byte row segments (64 stores + 64 residual loads per wave); transposing the wave's
    // 64x64 tile through its own LDS region turns that into 16-byte accesses per lane:
    // 16 loads + 16 stores of 1 KiB each.  After the last barrier nobody reads the operand
    // tiles any more, so the staging buffers are free.
    float* sw = smem + wid * (64 * EPI_STRIDE);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sw[(i * 32 + acc_row(r, half)) * EPI_STRIDE + j * 32 + l31] = acc[i][j][r];
    const int e_row = lane >> 4, e_c4 = lane & 15;
    const int n = n0 + wc * 64 + e_c4 * 4;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + n);
    const int64_t m_base = m0 + wr * 64 + e_row;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int64_t m = m_base + it * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(sw + (it * 4 + e_row) * EPI_STRIDE + e_c4 * 4);
        if (m < M) {
            v += bv;
            if (EPI == EPI_BIAS_RESIDUAL) v += *reinterpret_cast<const f32x4*>(R + m * ldr + n);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = epilogue<EPI>(v[c]);
            *reinterpret_cast<f32x4*>(Y + m * ldy + n) = v;
        }
    }
}

// Any-shape fallback (odd hidden sizes in tests, tiny heads): 32x32 LDS tiles,
// plain FMA.  Not on the MiniLM/BERT hot path.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_f32_generic(const float* __restrict__ A, int64_t lda,
                                                           const float* __restrict__ W,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ R, int64_t ldr,
                                                           float* __restrict__ Y, int64_t ldy,
                                                           int64_t M, int N, int K)
{
    __shared__ float sA[32][33];
    __shared__ float sB[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // ty 0..7
    const int64_t m0 = (int64_t)blockIdx.y * 32;
    const int n0 = blockIdx.x * 32;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = ty + 8 * i;
            const int k = k0 + tx;
            sA[row][tx] = (m0 + row < M && k < K) ? A[(m0 + row) * lda + k] : 0.0f;
            sB[row][tx] = (n0 + row < N && k < K) ? W[(int64_t)(n0 + row) * K + k] : 0.0f;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const float b = sB[tx][k];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(sA[ty + 8 * i][k], b, acc[i]);
        }
        __syncthreads();
    }
    const int n = n0 + tx;
    if (n < N) {
        const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = m0 + ty + 8 * i;
            if (m < M) {
                float v = acc[i] + bv;
                if (EPI == EPI_BIAS_RESIDUAL) v += R[m * ldr + n];
                Y[m * ldy + n] = epilogue<EPI>(v);
            }
        }
    }
}

template <int EPI>
hipError_t launch_epi(const float* A, int64_t lda, const float* W, const float* bias, const float* R,
                      int64_t ldr, float* Y, int64_t ldy, int64_t M, int N, int K,
                      hipStream_t stream)
{
    const bool aligned = (N % BN == 0) && (K % BK == 0) && (lda % 4 == 0) && (ldy % 4 == 0) &&
                         ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(W) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(Y) & 15) == 0) &&
                         (bias == nullptr || (reinterpret_cast<uintptr_t>(bias) & 15) == 0) &&
                         (R == nullptr || ((ldr % 4 == 0) && (reinterpret_cast<uintptr_t>(R) & 15) == 0));
    if (aligned) {
        // > 64 KiB of dynamic LDS needs the opt-in once per device.
        static bool attr_set[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (!attr_set[dev & 63]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_f32_mfma<EPI>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
            if (e != hipSuccess) return e;
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_f32_mfma_pipe<EPI, 0>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
            if (e != hipSuccess) return e;
            attr_set[dev & 63] = true;
        }
        const int n_tiles = N / BN;
        const int64_t m_tiles = (M + BM - 1) / BM;
        dim3 grid((unsigned)(m_tiles * n_tiles));
        if (g_gemm_variant == 2) {
            static bool diag_attr = false;
            if (!diag_attr) {
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_f32_mfma_pipe<EPI, 1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
                if (e != hipSuccess) return e;
                diag_attr = true;
            }
            hipLaunchKernelGGL((gemm_nt_f32_mfma_pipe<EPI, 1>), grid, dim3(256), GEMM_LDS_BYTES, stream, A, lda, W,
                               bias, R, ldr, Y, ldy, M, N, K, n_tiles);
        } else if (g_gemm_variant == 1) {
            hipLaunchKernelGGL(gemm_nt_f32_mfma<EPI>, grid, dim3(256), GEMM_LDS_BYTES, stream, A, lda, W, bias, R,
                               ldr, Y, ldy, M, N, K, n_tiles);
        } else {
            hipLaunchKernelGGL((gemm_nt_f32_mfma_pipe<EPI, 0>), grid, dim3(256), GEMM_LDS_BYTES, stream, A, lda, W,
                               bias, R, ldr, Y, ldy, M, N, K, n_tiles);
        }
    } else {
        dim3 grid((unsigned)((N + 31) / 32), (unsigned)((M + 31) / 32));
        hipLaunchKernelGGL(gemm_nt_f32_generic<EPI>, grid, dim3(256), 0, stream, A, lda, W, bias, R,
                           ldr, Y, ldy, M, N, K);
    }
    return hipGetLastError();
}

}  // namespace

void set_gemm_variant(int variant) { g_gemm_variant = variant; }
int gemm_variant() { return g_gemm_variant; }

hipError_t launch_gemm(const float* A, int64_t lda, const float* W, const float* bias, const float* R,
                       int64_t ldr, float* Y, int64_t ldy, int64_t M, int N, int K, GemmEpilogue epi,
                       hipStream_t stream)
{
    if (M <= 0 || N <= 0 || K <= 0) return hipSuccess;
    switch (epi) {
    case EPI_BIAS: return launch_epi<EPI_BIAS>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    case EPI_BIAS_GELU:
        if (g_gemm_variant == 3) return launch_epi<EPI_GELU_LIBM>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
        return launch_epi<EPI_BIAS_GELU>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    case EPI_BIAS_GELU_NEW: return launch_epi<EPI_BIAS_GELU_NEW>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    case EPI_BIAS_RELU: return launch_epi<EPI_BIAS_RELU>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    case EPI_BIAS_TANH: return launch_epi<EPI_BIAS_TANH>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    case EPI_BIAS_RESIDUAL: return launch_epi<EPI_BIAS_RESIDUAL>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace kjarni
