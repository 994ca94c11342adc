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
#include "device_utils.h"
#include "kernels.h"

namespace kjarni {

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_STRIDE = BK + 4;               // floats
constexpr int TILE_FLOATS = BM * LDS_STRIDE;     // one operand tile
constexpr int GEMM_LDS_BYTES = 2 * 2 * TILE_FLOATS * 4;  // 2 operands x 2 stages

template <int EPI>
__device__ __forceinline__ float epilogue(float v)
{
    if (EPI == EPI_BIAS_GELU) return gelu_erf(v);
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
    const bool aligned = (N % BN == 0) && (K % BK == 0) && (lda % 4 == 0) &&
                         ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
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
            attr_set[dev & 63] = true;
        }
        const int n_tiles = N / BN;
        const int64_t m_tiles = (M + BM - 1) / BM;
        dim3 grid((unsigned)(m_tiles * n_tiles));
        hipLaunchKernelGGL(gemm_nt_f32_mfma<EPI>, grid, dim3(256), GEMM_LDS_BYTES, stream, A, lda, W,
                           bias, R, ldr, Y, ldy, M, N, K, n_tiles);
    } else {
        dim3 grid((unsigned)((N + 31) / 32), (unsigned)((M + 31) / 32));
        hipLaunchKernelGGL(gemm_nt_f32_generic<EPI>, grid, dim3(256), 0, stream, A, lda, W, bias, R,
                           ldr, Y, ldy, M, N, K);
    }
    return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm(const float* A, int64_t lda, const float* W, const float* bias, const float* R,
                       int64_t ldr, float* Y, int64_t ldy, int64_t M, int N, int K, GemmEpilogue epi,
                       hipStream_t stream)
{
    if (M <= 0 || N <= 0 || K <= 0) return hipSuccess;
    switch (epi) {
    case EPI_BIAS: return launch_epi<EPI_BIAS>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    case EPI_BIAS_GELU: return launch_epi<EPI_BIAS_GELU>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    case EPI_BIAS_GELU_NEW: return launch_epi<EPI_BIAS_GELU_NEW>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    case EPI_BIAS_RELU: return launch_epi<EPI_BIAS_RELU>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    case EPI_BIAS_TANH: return launch_epi<EPI_BIAS_TANH>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    case EPI_BIAS_RESIDUAL: return launch_epi<EPI_BIAS_RESIDUAL>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace kjarni
