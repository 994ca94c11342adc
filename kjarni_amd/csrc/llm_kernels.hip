// Kernels of the decoder-only (Llama / Qwen2) path: a weight-streaming GEMV for 1-8 rows over f32 or bf16
// weights with RMSNorm folded in and the Q|K|V, SwiGLU and residual epilogues; RoPE; embedding gather;
// a two-stage argmax.  Attention over the KV cache reuses decode_attention_* (whisper_kernels.hip) with a
// grouped-query head mapping.
//
//   RMSNorm   crates/kjarni-transformers/src/cpu/normalization/rms_norm.rs:19-27
//   RoPE      cpu/rope/mod.rs:107-176
//   layer     cpu/decoder/rope_decoder_layer.rs:18-41, cpu/decoder/decoder_attention.rs:44-170,
//             cpu/feedforward/swiglu.rs:32-57
//   greedy    common/sampling.rs:83-88
#include <algorithm>
#include <atomic>

#include "device_utils.h"
#include "llm_kernels.h"
#include "kernels.h"

namespace kjarni {

namespace {

constexpr int LLM_MAX_ROWS = 8;
#ifdef KJARNI_TUNING
std::atomic<int> g_llm_gemv_variant{0};  // 1 = always the multi-row kernel; 3..6 columns per workgroup; 7 = narrow chunks -- tuning build only
#else
constexpr int g_llm_gemv_variant = 0;
#endif


struct F8 {
    float v[8];
};

// Eight consecutive weights starting at element 8*i of a row.
__device__ __forceinline__ F8 load8(const float* row, int i)
{
    const f32x4 a = *reinterpret_cast<const f32x4*>(row + i * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(row + i * 8 + 4);
    return F8{{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}};
}
__device__ __forceinline__ F8 load8(const uint16_t* row, int i)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 p = *reinterpret_cast<const u32x4*>(row + i * 8);
    F8 r;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        r.v[2 * c] = __uint_as_float(p[c] << 16);
        r.v[2 * c + 1] = __uint_as_float(p[c] & 0xFFFF0000u);
    }
    return r;
}

enum : int { LE_NONE = 0, LE_RESIDUAL = 1, LE_SWIGLU = 2 };

// Y[r, n] = epi(norm?(X[r, :]) . W[n, :] + bias[n]) for up to 8 rows; one wave per output column.
//  NORM: rows are RMS-normalised on the fly: (x / sqrt(mean(x^2) + eps)) * gamma, statistics recomputed per wave.
//  LE_SWIGLU: W2 is the `up` matrix; the output is silu(x.W[n]) * (x.W2[n]).
//  LE_RESIDUAL: + R[r, n].
//  seg > 0: columns [0,seg_q) -> Y0, then two segments of seg_kv columns -> Y1 / Y2 at row (*row_off_ptr | row_off) + r
//  (Q to scratch, K and V straight into the cache).
template <typename WT, int EPI, bool NORM>
__global__ __launch_bounds__(256) void llm_gemv_kernel(const float* __restrict__ X, int64_t ldx, int rows,
                                                       const float* __restrict__ gamma, float eps, const WT* __restrict__ W,
                                                       const WT* __restrict__ W2, const float* __restrict__ bias,
                                                       const float* __restrict__ R, int64_t ldr, int n_out, int k, int seg_q,
                                                       int seg_kv, float* __restrict__ Y0, int64_t ldy0, float* __restrict__ Y1,
                                                       float* __restrict__ Y2, int64_t ldy12, int row_off,
                                                       const int* __restrict__ row_off_ptr)
{
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_out) return;
    const int k8 = k >> 3;
    float scale[LLM_MAX_ROWS];
#pragma unroll
    for (int r = 0; r < LLM_MAX_ROWS; ++r) scale[r] = 1.0f;
    if (NORM) {
#pragma unroll
        for (int r = 0; r < LLM_MAX_ROWS; ++r) {
            if (r < rows) {
                float s = 0.0f;
                for (int i = lane; i < k8; i += 64) {
                    const F8 x = load8(X + r * ldx, i);
#pragma unroll
                    for (int c = 0; c < 8; ++c) s = fmaf(x.v[c], x.v[c], s);
                }
                scale[r] = sqrtf(wave_sum(s) / (float)k + eps);  // the rms; the reference divides by it
            }
        }
    }
    const WT* w_row = W + n * (int64_t)k;
    const WT* w2_row = EPI == LE_SWIGLU ? W2 + n * (int64_t)k : nullptr;
    float acc[LLM_MAX_ROWS], acc2[LLM_MAX_ROWS];
#pragma unroll
    for (int r = 0; r < LLM_MAX_ROWS; ++r) acc[r] = acc2[r] = 0.0f;
    for (int i = lane; i < k8; i += 64) {
        const F8 w = load8(w_row, i);
        F8 w2;
        if (EPI == LE_SWIGLU) w2 = load8(w2_row, i);
        F8 g;
        if (NORM) g = load8(gamma, i);
#pragma unroll
        for (int r = 0; r < LLM_MAX_ROWS; ++r) {
            if (r < rows) {
                F8 x = load8(X + r * ldx, i);
                if (NORM) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) x.v[c] = (x.v[c] / scale[r]) * g.v[c];
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    acc[r] = fmaf(x.v[c], w.v[c], acc[r]);
                    if (EPI == LE_SWIGLU) acc2[r] = fmaf(x.v[c], w2.v[c], acc2[r]);
                }
            }
        }
    }
    const float b = bias ? bias[n] : 0.0f;
    int which = 0;
    int64_t col = n;
    if (seg_q > 0 && n >= seg_q) {
        which = 1 + (int)((n - seg_q) / seg_kv);
        col = (n - seg_q) - (int64_t)(which - 1) * seg_kv;
    }
    float* Y = which == 0 ? Y0 : (which == 1 ? Y1 : Y2);
    const int64_t ldy = which == 0 ? ldy0 : ldy12;
    const int64_t r0 = which == 0 ? 0 : (row_off_ptr ? *row_off_ptr : row_off);
#pragma unroll
    for (int r = 0; r < LLM_MAX_ROWS; ++r) {
        if (r < rows) {
            float v = wave_sum(acc[r]) + b;
            if (EPI == LE_SWIGLU) {
                const float u = wave_sum(acc2[r]);
                v = (v / (1.0f + expf(-v))) * u;  // silu(gate) * up
            }
            if (EPI == LE_RESIDUAL) v += R[r * ldr + n];
            if (lane == 0) Y[(r0 + r) * ldy + col] = v;
        }
    }
}

// Single-row variant (the decode step).  The input row is normalised once per WORKGROUP into LDS (the
// multi-row kernel re-reads it from L2 for every output column, which costs more L2 bandwidth than the bf16
// weights cost HBM bandwidth); each wave then streams OPW weight rows, with all of a row's 16-byte loads in
// flight before the first FMA.
constexpr int G1_OPW = 2;        // outputs per wave
constexpr int G1_MAX_K = 16384;  // 64 KiB of LDS

template <typename WT, int EPI, bool NORM>
__global__ __launch_bounds__(256) void llm_gemv1_kernel(const float* __restrict__ X, const float* __restrict__ gamma, float eps,
                                                        const WT* __restrict__ W, const WT* __restrict__ W2,
                                                        const float* __restrict__ bias, const float* __restrict__ R, int n_out,
                                                        int k, int seg_q, int seg_kv, float* __restrict__ Y0,
                                                        float* __restrict__ Y1, float* __restrict__ Y2, int64_t ldy12, int row_off,
                                                        const int* __restrict__ row_off_ptr)
{
    extern __shared__ float xs[];  // [k]
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k8 = k >> 3;
    float rms = 1.0f;
    if (NORM) {
        float s = 0.0f;
        for (int i = tid; i < k8; i += 256) {
            const F8 x = load8(X, i);
#pragma unroll
            for (int c = 0; c < 8; ++c) s = fmaf(x.v[c], x.v[c], s);
        }
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        rms = sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)k + eps);
    }
    for (int i = tid; i < k8; i += 256) {
        F8 x = load8(X, i);
        if (NORM) {
            const F8 g = load8(gamma, i);
#pragma unroll
            for (int c = 0; c < 8; ++c) x.v[c] = (x.v[c] / rms) * g.v[c];
        }
        *reinterpret_cast<f32x4*>(xs + i * 8) = f32x4{x.v[0], x.v[1], x.v[2], x.v[3]};
        *reinterpret_cast<f32x4*>(xs + i * 8 + 4) = f32x4{x.v[4], x.v[5], x.v[6], x.v[7]};
    }
    __syncthreads();

    const int64_t n0 = ((int64_t)blockIdx.x * 4 + wave) * G1_OPW;
#pragma unroll
    for (int j = 0; j < G1_OPW; ++j) {
        const int64_t n = n0 + j;
        if (n >= n_out) break;
        const WT* w_row = W + n * (int64_t)k;
        const WT* w2_row = EPI == LE_SWIGLU ? W2 + n * (int64_t)k : nullptr;
        float acc = 0.0f, acc2 = 0.0f;
        for (int i0 = lane; i0 < k8; i0 += 256) {  // four chunks per lane per trip: 4 (8 with SwiGLU) loads in flight
            F8 w[4], u[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q * 64;
                if (i < k8) {
                    w[q] = load8(w_row, i);
                    if (EPI == LE_SWIGLU) u[q] = load8(w2_row, i);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q * 64;
                if (i < k8) {
                    const f32x4 xa = *reinterpret_cast<const f32x4*>(xs + i * 8);
                    const f32x4 xb = *reinterpret_cast<const f32x4*>(xs + i * 8 + 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc = fmaf(xa[c], w[q].v[c], acc);
                        acc = fmaf(xb[c], w[q].v[4 + c], acc);
                        if (EPI == LE_SWIGLU) {
                            acc2 = fmaf(xa[c], u[q].v[c], acc2);
                            acc2 = fmaf(xb[c], u[q].v[4 + c], acc2);
                        }
                    }
                }
            }
        }
        float v = wave_sum(acc) + (bias ? bias[n] : 0.0f);
        if (EPI == LE_SWIGLU) {
            const float up = wave_sum(acc2);
            v = (v / (1.0f + expf(-v))) * up;
        }
        if (EPI == LE_RESIDUAL) v += R[n];
        if (lane == 0) {
            int which = 0;
            int64_t col = n;
            if (seg_q > 0 && n >= seg_q) {
                which = 1 + (int)((n - seg_q) / seg_kv);
                col = (n - seg_q) - (int64_t)(which - 1) * seg_kv;
            }
            float* Y = which == 0 ? Y0 : (which == 1 ? Y1 : Y2);
            const int64_t r0 = which == 0 ? 0 : (row_off_ptr ? *row_off_ptr : row_off);
            Y[which == 0 ? col : r0 * ldy12 + col] = v;
        }
    }
}

// Single-row GEMV with the K dimension split over the four waves of a workgroup.  A workgroup owns OPW output
// columns; lane t of the block keeps chunks t, t+256, ... of the input row in registers (normalised in place when
// NORM — the sum of squares is reduced once through LDS), issues every weight load of its slice before the first
// FMA (OPW x chunks x 16 bytes per lane in flight) and the four per-wave partial sums meet in LDS.  Compared with
// one wave per column this puts 4x more workgroups on the chip for the narrow projections (o, down) and reads the
// input row once per OPW columns instead of once per column.
constexpr int SK_MAX_CHUNKS = 8;  // k <= 8 * 256 * 8 = 16384 with 4 waves; long rows (k >= 8192) use 16 waves per workgroup

template <typename WT, int EPI, bool NORM, int OPW, int CH, int NW>
__global__ __launch_bounds__(64 * NW) void llm_gemv_splitk_kernel(const float* __restrict__ X, const float* __restrict__ gamma, float eps,
                                                              const WT* __restrict__ W, const WT* __restrict__ W2,
                                                              const float* __restrict__ bias, const float* __restrict__ R,
                                                              int n_out, int k, float* __restrict__ Y)
{
    constexpr int NM = EPI == LE_SWIGLU ? 2 : 1;
    constexpr int THREADS = 64 * NW;
    __shared__ float red[NW];
    __shared__ float part[NW][NM * OPW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k8 = k >> 3;
    F8 x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = tid + c * THREADS;
        if (i < k8) x[c] = load8(X, i);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[c].v[e] = 0.0f;
        }
    }
    const int64_t n0 = (int64_t)blockIdx.x * OPW;
    // Weight loads do not depend on the statistics: issue them first so the reduction overlaps their latency.
    F8 w[OPW][CH], u[EPI == LE_SWIGLU ? OPW : 1][CH];
#pragma unroll
    for (int o = 0; o < OPW; ++o) {
        const int64_t n = n0 + o < n_out ? n0 + o : n_out - 1;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int i = tid + c * THREADS;
            if (i < k8) {
                w[o][c] = load8(W + n * (int64_t)k, i);
                if (EPI == LE_SWIGLU) u[o][c] = load8(W2 + n * (int64_t)k, i);
            }
        }
    }
    if (NORM) {
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(x[c].v[e], x[c].v[e], s);
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        float total = 0.0f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) total += red[wv];
        const float rms = sqrtf(total / (float)k + eps);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int i = tid + c * THREADS;
            if (i < k8) {
                const F8 g = load8(gamma, i);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[c].v[e] = (x[c].v[e] / rms) * g.v[e];
            }
        }
    }
#pragma unroll
    for (int o = 0; o < OPW; ++o) {
        float acc = 0.0f, acc2 = 0.0f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int i = tid + c * THREADS;
            if (i < k8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    acc = fmaf(x[c].v[e], w[o][c].v[e], acc);
                    if (EPI == LE_SWIGLU) acc2 = fmaf(x[c].v[e], u[o][c].v[e], acc2);
                }
            }
        }
        acc = wave_sum(acc);
        if (EPI == LE_SWIGLU) acc2 = wave_sum(acc2);
        if (lane == 0) {
            part[wave][o] = acc;
            if (EPI == LE_SWIGLU) part[wave][OPW + o] = acc2;
        }
    }
    __syncthreads();
    if (tid < OPW && n0 + tid < n_out) {
        const int64_t n = n0 + tid;
        float v = 0.0f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) v += part[wv][tid];
        v += bias ? bias[n] : 0.0f;
        if (EPI == LE_SWIGLU) {
            float up = 0.0f;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) up += part[wv][OPW + tid];
            v = (v / (1.0f + expf(-v))) * up;
        }
        if (EPI == LE_RESIDUAL) v += R[n];
        Y[n] = v;
    }
}

// Single-row GEMV as a weight STREAM (rows == 1, bf16 or f32 weights, K = 64 lanes x 8 elements x CH pieces): what a
// decode step is made of is reading every weight once, so the kernel is built around keeping 16-byte loads in flight.
// A few long-lived workgroups instead of thousands of two-column ones: the input row is normalised ONCE per workgroup
// into LDS, then every wave walks whole weight rows -- a batch is RB rows x CH pieces = 16 (32 with the SwiGLU pair)
// non-temporal 16-byte loads per lane, all issued before the first FMA (the weights are read once: nt keeps them
// from evicting what the other kernels of the step re-read) -- multiplies them with its LDS-resident slice of x and
// finishes each row with a wave reduction.  Rows are dealt batch by batch over all waves of the grid.
template <int N>
struct RawPieces {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v[N];
};

__device__ __forceinline__ RawPieces<1>::u32x4 load_nt16(const void* p)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
}

// dot of 8 consecutive weights (one 16-byte bf16 piece, or the pair of f32 pieces p0 / p1) with x[0..7]
__device__ __forceinline__ float dot8_bf16(RawPieces<1>::u32x4 p, const f32x4 xa, const f32x4 xb, float acc)
{
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float lo = __uint_as_float(p[c] << 16), hi = __uint_as_float(p[c] & 0xFFFF0000u);
        const float x0 = c < 2 ? xa[2 * c] : xb[2 * c - 4], x1 = c < 2 ? xa[2 * c + 1] : xb[2 * c - 3];
        acc = fmaf(x0, lo, acc);
        acc = fmaf(x1, hi, acc);
    }
    return acc;
}

constexpr int stream_rows_per_batch(int ch, int loads_per_piece, int nm, bool wide)
{
    return wide && 16 / (ch * loads_per_piece * nm) > 1 ? 16 / (ch * loads_per_piece * nm) : 1;
}

template <typename WT, int EPI, bool NORM, int CH, bool WIDE, bool LOOP, int THREADS, int ATT = 0>
__global__ __launch_bounds__(THREADS) void llm_gemv_stream_kernel(const float* __restrict__ X, const float* __restrict__ gamma, float eps,
                                                              const WT* __restrict__ W, const WT* __restrict__ W2,
                                                              const float* __restrict__ bias, const float* __restrict__ R,
                                                              int n_out, float* __restrict__ Y, int att_splits, int att_head_dim,
                                                              float* __restrict__ Xn)
{
    constexpr bool BF16 = sizeof(WT) == 2;
    constexpr int K = CH * 512;
    constexpr int NM = EPI == LE_SWIGLU ? 2 : 1;
    constexpr int LOADS_PER_PIECE = BF16 ? 1 : 2;                                // 8 weights = 16 or 32 bytes
    // rows per wave and batch: WIDE = 16 loads per lane in flight (large projections), else one row (a 2048-row projection
    // then still gives every CU 8 waves); LOOP = the grid does not cover the rows (the vocabulary head), else one batch per wave
    constexpr int RB = stream_rows_per_batch(CH, LOADS_PER_PIECE, NM, WIDE);
    __shared__ __attribute__((aligned(16))) float xs[K];
    constexpr int WAVES = THREADS / 64;  // 4 or 8: the larger workgroup halves the re-reads of the input row
    __shared__ float red[WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: row bases stay in SGPRs
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int XV = K / (4 * THREADS);  // f32x4 pieces of the input row per thread
    // Everything the workgroup reads is requested before anything is waited for: the input row (and gamma), then the first
    // batch of weight rows.  The row is then normalised once per workgroup into LDS (x / sqrt(mean(x^2) + eps)) * gamma
    // (rms_norm.rs:19-27) while the weights are still in flight; the counter waits are in issue order, so x must go first.
    // ATT > 0: the row is the context row of the decode attention, still in ATT-or-fewer slabs per head (max, sum of exp,
    // sum of exp * V: whisper_kernels.hip); their merge (the arithmetic of decode_attention_combine_kernel) happens here,
    // under the weight requests, instead of in a launch of its own.
    constexpr int AS = ATT > 0 ? ATT : 1;
    f32x4 xv[XV], gv[XV], hv[XV][AS], av[XV][AS];
    if (ATT > 0) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int col = (tid + i * THREADS) * 4, head = col / att_head_dim, j = col - head * att_head_dim;
            const float* slab = X + (int64_t)head * att_splits * (att_head_dim + 4);
#pragma unroll
            for (int sp = 0; sp < AS; ++sp) {
                const float* p = slab + (sp < att_splits ? sp : 0) * (att_head_dim + 4);
                hv[i][sp] = *reinterpret_cast<const f32x4*>(p);
                av[i][sp] = *reinterpret_cast<const f32x4*>(p + 4 + j);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < XV; ++i) xv[i] = *reinterpret_cast<const f32x4*>(X + (tid + i * THREADS) * 4);
    }
    if (NORM) {
#pragma unroll
        for (int i = 0; i < XV; ++i) gv[i] = *reinterpret_cast<const f32x4*>(gamma + (tid + i * THREADS) * 4);
    }
    const int waves_total = gridDim.x * WAVES;
    u32x4 raw[NM][RB][CH][LOADS_PER_PIECE];
    auto issue = [&](int n0) {
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int n = n0 + r < n_out ? n0 + r : n_out - 1;
                const char* row = reinterpret_cast<const char*>((m == 0 ? W : W2) + (int64_t)n * K);
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int l = 0; l < LOADS_PER_PIECE; ++l)
                        raw[m][r][c][l] = load_nt16(row + ((size_t)(c * 64 + lane) * 8) * sizeof(WT) + l * 16);
            }
    };
    const int first = (blockIdx.x * WAVES + wave) * RB;
    float bv[RB], rv[RB];  // the epilogue's bias / residual values, requested ahead of the batch's weights
    auto issue_tail = [&](int n0) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int n = n0 + r < n_out ? n0 + r : n_out - 1;
            bv[r] = bias ? bias[n] : 0.0f;
            rv[r] = EPI == LE_RESIDUAL ? R[n] : 0.0f;
        }
    };
    issue_tail(first);
    issue(first);  // unconditional (rows past the end clamp to the last one): a branch here would make the x wait drain the weights too
    if (ATT > 0) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            float M = -INFINITY;
#pragma unroll
            for (int sp = 0; sp < AS; ++sp)
                if (sp < att_splits) M = fmaxf(M, hv[i][sp][0]);
            float L = 0.0f;
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sp = 0; sp < AS; ++sp) {
                if (sp < att_splits) {
                    const float w = (hv[i][sp][0] == -INFINITY) ? 0.0f : expf(hv[i][sp][0] - M);
                    L = fmaf(hv[i][sp][1], w, L);
#pragma unroll
                    for (int c = 0; c < 4; ++c) a[c] = fmaf(av[i][sp][c], w, a[c]);
                }
            }
            const float inv = L > 0.0f ? 1.0f / L : 1.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) xv[i][c] = L > 0.0f ? a[c] * inv : a[c];
        }
    }
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < XV; ++i) ss += (xv[i][0] * xv[i][0] + xv[i][1] * xv[i][1]) + (xv[i][2] * xv[i][2] + xv[i][3] * xv[i][3]);
    if (NORM) {
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        float tot = (red[0] + red[1]) + (red[2] + red[3]);
        if (WAVES == 8) tot += (red[4] + red[5]) + (red[6] + red[7]);
        const float rms = sqrtf(tot / (float)K + eps);
#pragma unroll
        for (int i = 0; i < XV; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) xv[i][c] = (xv[i][c] / rms) * gv[i][c];
    }
#pragma unroll
    for (int i = 0; i < XV; ++i) *reinterpret_cast<f32x4*>(xs + (tid + i * THREADS) * 4) = xv[i];
    if (NORM && Xn && blockIdx.x == 0) {  // the normalised row is an output too (the model's last hidden state)
#pragma unroll
        for (int i = 0; i < XV; ++i) *reinterpret_cast<f32x4*>(Xn + (tid + i * THREADS) * 4) = xv[i];
    }
    __syncthreads();

    // Per batch of RB rows: the NEXT batch's bias / residual values and weights are requested before this batch's cross-lane
    // sums and stores, which need no loads, so nothing ever waits with fresh requests behind it (the counter retires in
    // issue order, and the tail values, older than their batch's weights, have landed once the products are done).
    for (int n0 = first;;) {
        float acc[NM][RB];
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int r = 0; r < RB; ++r) acc[m][r] = 0.0f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const f32x4 xa = *reinterpret_cast<const f32x4*>(xs + (c * 64 + lane) * 8);
            const f32x4 xb = *reinterpret_cast<const f32x4*>(xs + (c * 64 + lane) * 8 + 4);
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    if (BF16) {
                        acc[m][r] = dot8_bf16(raw[m][r][c][0], xa, xb, acc[m][r]);
                    } else {
                        const f32x4 wa = __builtin_bit_cast(f32x4, raw[m][r][c][0]);
                        const f32x4 wb = __builtin_bit_cast(f32x4, raw[m][r][c][LOADS_PER_PIECE - 1]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[m][r] = fmaf(xa[e], wa[e], acc[m][r]);
                            acc[m][r] = fmaf(xb[e], wb[e], acc[m][r]);
                        }
                    }
                }
        }
        float cb[RB], cr[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            cb[r] = bv[r];
            cr[r] = rv[r];
        }
        const int next = n0 + waves_total * RB;
        const bool more = LOOP && next < n_out;
        if (LOOP && more) {
            issue_tail(next);
            issue(next);
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            float v = wave_sum(acc[0][r]);
            float up = 0.0f;
            if (EPI == LE_SWIGLU) up = wave_sum(acc[NM - 1][r]);
            const int n = n0 + r;
            if (lane == 0 && n < n_out) {
                v += cb[r];
                if (EPI == LE_SWIGLU) v = (v / (1.0f + expf(-v))) * up;
                if (EPI == LE_RESIDUAL) v += cr[r];
                Y[n] = v;
            }
        }
        if (!more) break;
        n0 = next;
    }
}

// Decode-step fusion of RMSNorm + Q|K|V projection + RoPE (decoder_attention.rs:61-97 for one new token).  Same split-K
// layout as llm_gemv_splitk_kernel (the four waves of a workgroup share the K dimension, the row is normalised in
// registers); a workgroup owns the PAIR of output columns (i, i + d/2) of one head, so it rotates its own two outputs
// (rope/mod.rs:156-176) before Q goes to scratch and K to its cache row; V columns go in pairs of neighbours, unrotated.
template <typename WT, int CH>
__global__ __launch_bounds__(256) void llm_qkv_rope_kernel(const float* __restrict__ X, const float* __restrict__ gamma, float eps,
                                                           const WT* __restrict__ W, const float* __restrict__ bias, int k,
                                                           int n_heads, int n_kv_heads, int head_dim, const float* __restrict__ cos_t,
                                                           const float* __restrict__ sin_t, float* __restrict__ Q,
                                                           float* __restrict__ Kc, float* __restrict__ Vc, int pos,
                                                           const int* __restrict__ pos_ptr)
{
    __shared__ float red[4];
    __shared__ float part[4][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k8 = k >> 3, half = head_dim >> 1;
    const int q_dim = n_heads * head_dim, kv_dim = n_kv_heads * head_dim;
    const int q_tasks = q_dim / 2, k_tasks = kv_dim / 2;
    const int task = blockIdx.x;
    int n_a, n_b;  // the two output columns (rows of W) of this workgroup
    int kind;      // 0 = Q, 1 = K, 2 = V
    if (task < q_tasks + k_tasks) {
        kind = task < q_tasks ? 0 : 1;
        const int t = kind == 0 ? task : task - q_tasks;
        const int head = t / half, i = t - head * half;
        n_a = (kind == 0 ? 0 : q_dim) + head * head_dim + i;
        n_b = n_a + half;
    } else {
        kind = 2;
        n_a = q_dim + kv_dim + 2 * (task - q_tasks - k_tasks);
        n_b = n_a + 1;
    }
    F8 x[CH], wa[CH], wb[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = tid + c * 256;
        if (i < k8) {
            x[c] = load8(X, i);
            wa[c] = load8(W + (int64_t)n_a * k, i);
            wb[c] = load8(W + (int64_t)n_b * k, i);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[c].v[e] = wa[c].v[e] = wb[c].v[e] = 0.0f;
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(x[c].v[e], x[c].v[e], s);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float rms = sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)k + eps);
    float acc_a = 0.0f, acc_b = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = tid + c * 256;
        if (i < k8) {
            const F8 g = load8(gamma, i);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xn = (x[c].v[e] / rms) * g.v[e];
                acc_a = fmaf(xn, wa[c].v[e], acc_a);
                acc_b = fmaf(xn, wb[c].v[e], acc_b);
            }
        }
    }
    acc_a = wave_sum(acc_a);
    acc_b = wave_sum(acc_b);
    if (lane == 0) {
        part[wave][0] = acc_a;
        part[wave][1] = acc_b;
    }
    __syncthreads();
    if (tid != 0) return;
    const float va = ((part[0][0] + part[1][0]) + (part[2][0] + part[3][0])) + (bias ? bias[n_a] : 0.0f);
    const float vb = ((part[0][1] + part[1][1]) + (part[2][1] + part[3][1])) + (bias ? bias[n_b] : 0.0f);
    const int p = pos_ptr ? *pos_ptr : pos;
    if (kind == 2) {
        const int col = n_a - q_dim - kv_dim;
        Vc[(int64_t)p * kv_dim + col] = va;
        Vc[(int64_t)p * kv_dim + col + 1] = vb;
        return;
    }
    const int col = kind == 0 ? n_a : n_a - q_dim;
    const int i = col % head_dim;  // < half
    const float c = cos_t[(int64_t)p * half + i], sn = sin_t[(int64_t)p * half + i];
    float* out = kind == 0 ? Q : Kc + (int64_t)p * kv_dim;
    out[col] = va * c - vb * sn;
    out[col + half] = va * sn + vb * c;
}

// The same step in the weight-streaming form (K = 2048 / 4096 / 8192): one WAVE owns the pair of output columns, the row is
// normalised once per workgroup into LDS, and every request (x, gamma, position, the pair's two weight rows, bias, the
// rotation's cos / sin) is issued before anything is waited for.
// EMBED (the first layer of a one-token step): the input row is not read from X but gathered here from the embedding table, as
// llm_embed_kernel does (row *embed_ids, widened; an id >= vocab leaves zeros), by every workgroup for itself; workgroup 0 also
// stores it to x_raw_out, the residual stream the output projection adds to.  One launch fewer per token.
template <typename WT, int CH, bool EMBED = false>
__global__ __launch_bounds__(256) void llm_qkv_rope_stream_kernel(const float* __restrict__ X, const float* __restrict__ gamma, float eps,
                                                                  const WT* __restrict__ W, const float* __restrict__ bias,
                                                                  int n_heads, int n_kv_heads, int head_dim,
                                                                  const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                                  float* __restrict__ Q, float* __restrict__ Kc, float* __restrict__ Vc,
                                                                  int pos, const int* __restrict__ pos_ptr,
                                                                  const uint32_t* __restrict__ embed_ids, const WT* __restrict__ table,
                                                                  int vocab, float* __restrict__ x_raw_out)
{
    constexpr bool BF16 = sizeof(WT) == 2;
    constexpr int K = CH * 512;
    constexpr int LOADS_PER_PIECE = BF16 ? 1 : 2;
    constexpr int XV = K / 1024;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float xs[K];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = head_dim >> 1;
    const int q_dim = n_heads * head_dim, kv_dim = n_kv_heads * head_dim;
    const int q_tasks = q_dim / 2, k_tasks = kv_dim / 2, tasks = q_tasks + 2 * k_tasks;
    const int my = blockIdx.x * 4 + wave;
    const int task = my < tasks ? my : tasks - 1;  // surplus waves redo the last pair and store nothing
    int n_a, n_b, kind;  // the two output columns (rows of W) of this wave; 0 = Q, 1 = K, 2 = V
    if (task < q_tasks + k_tasks) {
        kind = task < q_tasks ? 0 : 1;
        const int t = kind == 0 ? task : task - q_tasks;
        const int head = t / half, i = t - head * half;
        n_a = (kind == 0 ? 0 : q_dim) + head * head_dim + i;
        n_b = n_a + half;
    } else {
        kind = 2;
        n_a = q_dim + kv_dim + 2 * (task - q_tasks - k_tasks);
        n_b = n_a + 1;
    }
    f32x4 xv[XV], gv[XV];
    if (EMBED) {
        const uint32_t id = embed_ids[0];
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            xv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (id < (uint32_t)vocab) {
                const WT* src = table + (int64_t)id * K + (tid + i * 256) * 4;
                if (BF16) {
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 r = *reinterpret_cast<const u32x2*>(src);
                    xv[i] = f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xFFFF0000u), __uint_as_float(r[1] << 16),
                                  __uint_as_float(r[1] & 0xFFFF0000u)};
                } else {
                    xv[i] = *reinterpret_cast<const f32x4*>(src);
                }
            }
        }
        if (blockIdx.x == 0 && x_raw_out) {
#pragma unroll
            for (int i = 0; i < XV; ++i) *reinterpret_cast<f32x4*>(x_raw_out + (tid + i * 256) * 4) = xv[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < XV; ++i) xv[i] = *reinterpret_cast<const f32x4*>(X + (tid + i * 256) * 4);
    }
#pragma unroll
    for (int i = 0; i < XV; ++i) gv[i] = *reinterpret_cast<const f32x4*>(gamma + (tid + i * 256) * 4);
    const int p = pos_ptr ? *pos_ptr : pos;
    const int col = kind == 0 ? n_a : (kind == 1 ? n_a - q_dim : n_a - q_dim - kv_dim);
    const int ri = kind == 2 ? 0 : col % head_dim;  // < half
    const float ba = bias ? bias[n_a] : 0.0f, bb = bias ? bias[n_b] : 0.0f;
    const float cs = cos_t[(int64_t)p * half + ri], sn = sin_t[(int64_t)p * half + ri];
    u32x4 raw[2][CH][LOADS_PER_PIECE];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const char* row = reinterpret_cast<const char*>(W + (int64_t)(r == 0 ? n_a : n_b) * K);
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int l = 0; l < LOADS_PER_PIECE; ++l)
                raw[r][c][l] = load_nt16(row + ((size_t)(c * 64 + lane) * 8) * sizeof(WT) + l * 16);
    }
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < XV; ++i) ss += (xv[i][0] * xv[i][0] + xv[i][1] * xv[i][1]) + (xv[i][2] * xv[i][2] + xv[i][3] * xv[i][3]);
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    const float rms = sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)K + eps);
#pragma unroll
    for (int i = 0; i < XV; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) xv[i][c] = (xv[i][c] / rms) * gv[i][c];
        *reinterpret_cast<f32x4*>(xs + (tid + i * 256) * 4) = xv[i];
    }
    __syncthreads();
    float acc[2] = {0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const f32x4 xa = *reinterpret_cast<const f32x4*>(xs + (c * 64 + lane) * 8);
        const f32x4 xb = *reinterpret_cast<const f32x4*>(xs + (c * 64 + lane) * 8 + 4);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (BF16) {
                acc[r] = dot8_bf16(raw[r][c][0], xa, xb, acc[r]);
            } else {
                const f32x4 wa = __builtin_bit_cast(f32x4, raw[r][c][0]);
                const f32x4 wb = __builtin_bit_cast(f32x4, raw[r][c][LOADS_PER_PIECE - 1]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[r] = fmaf(xa[e], wa[e], acc[r]);
                    acc[r] = fmaf(xb[e], wb[e], acc[r]);
                }
            }
        }
    }
    const float va = wave_sum(acc[0]) + ba, vb = wave_sum(acc[1]) + bb;
    if (lane != 0 || my >= tasks) return;
    if (kind == 2) {
        Vc[(int64_t)p * kv_dim + col] = va;
        Vc[(int64_t)p * kv_dim + col + 1] = vb;
        return;
    }
    float* out = kind == 0 ? Q : Kc + (int64_t)p * kv_dim;
    out[col] = va * cs - vb * sn;  // rope/mod.rs:156-176
    out[col + half] = va * sn + vb * cs;
}

// In-place rotation of `rows` rows of [n_heads * d] (rope/mod.rs:156-176): pairs (i, i + d/2), position =
// (*pos_ptr | pos) + row; x may start at row (*row_off_ptr | row_off) of a cache.
__global__ __launch_bounds__(256) void rope_kernel(float* __restrict__ x, int64_t ldx, int rows, int n_heads, int head_dim,
                                                   const float* __restrict__ cos_t, const float* __restrict__ sin_t, int pos,
                                                   const int* __restrict__ pos_ptr, int at_cache_row)
{
    const int half = head_dim >> 1;
    const int base = pos_ptr ? *pos_ptr : pos;
    const int total = rows * n_heads * half;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int i = idx % half;
        const int h = (idx / half) % n_heads;
        const int r = idx / (half * n_heads);
        const int p = base + r;
        float* row = x + (int64_t)(at_cache_row ? p : r) * ldx + h * head_dim;
        const float c = cos_t[(int64_t)p * half + i], s = sin_t[(int64_t)p * half + i];
        const float x0 = row[i], x1 = row[i + half];
        row[i] = x0 * c - x1 * s;
        row[i + half] = x0 * s + x1 * c;
    }
}

// out[r, :] = (x[r, :] / sqrt(mean(x^2) + eps)) * gamma   (rms_norm.rs:19-27); one wave per row.
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma, float eps,
                                                      int rows, int hidden, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* row = x + (int64_t)r * hidden;
    float s = 0.0f;
    for (int i = lane; i < hidden; i += 64) s = fmaf(row[i], row[i], s);
    const float rms = sqrtf(wave_sum(s) / (float)hidden + eps);
    for (int i = lane; i < hidden; i += 64) out[(int64_t)r * hidden + i] = (row[i] / rms) * gamma[i];
}

// The same with one 256-thread workgroup per row and 16-byte loads (hidden % 4 == 0).
__global__ __launch_bounds__(256) void rmsnorm_block_kernel(const float* __restrict__ x, const float* __restrict__ gamma, float eps,
                                                            int hidden, float* __restrict__ out)
{
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = x + (int64_t)blockIdx.x * hidden;
    const int h4 = hidden >> 2;
    float s = 0.0f;
    for (int i = tid; i < h4; i += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + i * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) s = fmaf(v[c], v[c], s);
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float rms = sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)hidden + eps);
    for (int i = tid; i < h4; i += 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(row + i * 4);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + i * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = (v[c] / rms) * g[c];
        *reinterpret_cast<f32x4*>(out + (int64_t)blockIdx.x * hidden + i * 4) = v;
    }
}

template <typename WT>
__global__ __launch_bounds__(256) void llm_embed_kernel(const uint32_t* __restrict__ ids, int hidden, int vocab,
                                                        const WT* __restrict__ table, float* __restrict__ out)
{
    const int s = blockIdx.x;
    const uint32_t id = ids[s];
    for (int i = threadIdx.x; i < hidden / 8; i += 256) {
        F8 v;
#pragma unroll
        for (int c = 0; c < 8; ++c) v.v[c] = 0.0f;
        if (id < (uint32_t)vocab) v = load8(table + (int64_t)id * hidden, i);
#pragma unroll
        for (int c = 0; c < 8; ++c) out[(int64_t)s * hidden + i * 8 + c] = v.v[c];
    }
}

__device__ __forceinline__ unsigned long long argmax_key(float v, int idx)
{
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // larger float -> larger uint
    if (v != v) u = 0;                               // NaN lowest
    return ((unsigned long long)u << 32) | (uint32_t)idx;  // equal values: the LAST index wins (Iterator::max_by)
}

__global__ __launch_bounds__(256) void argmax_partial_kernel(const float* __restrict__ logits, int vocab,
                                                             unsigned long long* __restrict__ best)
{
    __shared__ unsigned long long red[4];
    unsigned long long key = 0ull;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < vocab; i += gridDim.x * 256) {
        const unsigned long long k = argmax_key(logits[i], i);
        key = k > key ? k : key;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(key, off, kWave);
        key = o > key ? o : key;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) key = red[w] > key ? red[w] : key;
        atomicMax(best, key);
    }
}

// Publishes the winner, resets the accumulator, and (graph replay) appends the token and advances the counters.
__global__ void argmax_finalize_kernel(unsigned long long* __restrict__ best, int32_t* __restrict__ out,
                                       int32_t* __restrict__ history, int* __restrict__ count, int* __restrict__ pos)
{
    const int tok = (int)(uint32_t)(*best & 0xFFFFFFFFull);
    *best = 0ull;
    *out = tok;
    if (history) {
        history[*count] = tok;
        *count += 1;
        *pos += 1;
    }
}

// Prompt-row projections on the fp32 matrix cores: Y[M, N] = A[M, K] . W[N, K]^T (+ bias) (+ R), W in bf16 or f32 as
// stored.  The encoder's 128 x 128 GEMM (gemm.hip) is sized for 10^5 rows; a prompt has 10^2..10^3, where one
// 128 x 128 x 2048 tile keeps a CU busy for ~110 us while most CUs have no tile at all.  So: 64 x 64 block tiles
// (4 waves, one 32 x 32 MFMA tile each) -- 4x more workgroups of a quarter the work --, BK = 32, operands
// double-buffered in LDS with the next tile's global loads in flight during the MFMAs, bf16 weights widened when they
// are written to LDS (the same widening the GEMV kernels do).  v_mfma_f32_32x32x2_f32 is an exact k-ordered f32 fma
// chain, so the arithmetic class is the GEMV's; only the summation order differs.
constexpr int PG_BM = 64, PG_BN = 64, PG_BK = 32, PG_STRIDE = PG_BK + 4;

template <typename WT, bool RESIDUAL>
__global__ __launch_bounds__(256) void prefill_gemm_kernel(const float* __restrict__ A, int64_t lda, const WT* __restrict__ W,
                                                           const float* __restrict__ bias, const float* R, int64_t ldr, float* Y,
                                                           int64_t ldy, int M, int N, int K, int m_tiles, int ksplit,
                                                           float* __restrict__ P)
{
    __shared__ __attribute__((aligned(16))) float sA[2][PG_BM * PG_STRIDE];
    __shared__ __attribute__((aligned(16))) float sB[2][PG_BN * PG_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, l31 = lane & 31, half = lane >> 5;
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs; give every XCD one contiguous run of tiles
    // (row tiles fastest), so the tiles sharing a weight panel sit behind the same L2.
    const int64_t nwg = gridDim.x;
    const int64_t xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    const int64_t q8 = nwg / 8, r8 = nwg % 8;
    const int64_t bid0 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    // ksplit > 1 (short prompts: too few tiles for 256 CUs): slice ks of the K range, fastest index, so the slices of a
    // tile run side by side; the partial tiles go to P[ks][M][N] and prefill_splitk_reduce_kernel adds them in order.
    const int ks = (int)(bid0 % ksplit);
    const int64_t bid = bid0 / ksplit;
    const int m0 = (int)(bid % m_tiles) * PG_BM;
    const int n0 = (int)(bid / m_tiles) * PG_BN;
    const int k_len = K / ksplit, k_begin = ks * k_len;

    // A: 64 x 32 floats = 512 float4, two per thread.  W: f32 the same; bf16: 64 x 32 halves = 256 x 16 bytes, one per thread.
    const int a_row = tid >> 3, a_c4 = tid & 7;
    const float* a_ptr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a_ptr[i] = A + (int64_t)min(m0 + a_row + 32 * i, M - 1) * lda + a_c4 * 4;
    constexpr bool BF16 = sizeof(WT) == 2;
    const int b_row = BF16 ? (tid >> 2) : a_row, b_c = BF16 ? (tid & 3) * 8 : a_c4 * 4;
    const WT* b_ptr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) b_ptr[i] = W + (int64_t)min(n0 + b_row + 32 * i, N - 1) * K + b_c;
    f32x4 ga[2], gb[2];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) ga[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + k0);
        if (BF16) {
            const F8 w = load8(reinterpret_cast<const uint16_t*>(b_ptr[0]) + k0, 0);
            gb[0] = f32x4{w.v[0], w.v[1], w.v[2], w.v[3]};
            gb[1] = f32x4{w.v[4], w.v[5], w.v[6], w.v[7]};
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) gb[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(b_ptr[i]) + k0);
        }
    };
    auto store = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&sA[stage][(a_row + 32 * i) * PG_STRIDE + a_c4 * 4]) = ga[i];
        if (BF16) {
            *reinterpret_cast<f32x4*>(&sB[stage][b_row * PG_STRIDE + b_c]) = gb[0];
            *reinterpret_cast<f32x4*>(&sB[stage][b_row * PG_STRIDE + b_c + 4]) = gb[1];
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&sB[stage][(b_row + 32 * i) * PG_STRIDE + b_c]) = gb[i];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int nk = k_len / PG_BK;
    const int fa = (wr * 32 + l31) * PG_STRIDE + half * 4, fb = (wc * 32 + l31) * PG_STRIDE + half * 4;
    load(k_begin);
    store(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load(k_begin + (kt + 1) * PG_BK);
#pragma unroll
        for (int kk = 0; kk < PG_BK / 8; ++kk) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&sA[cur][fa + kk * 8]);
            const f32x4 b = *reinterpret_cast<const f32x4*>(&sB[cur][fb + kk * 8]);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c], b[c], acc, 0, 0, 0);
        }
        if (kt + 1 < nk) store(cur ^ 1);
        __syncthreads();
    }
    const int col = n0 + wc * 32 + l31;
    if (col < N) {
        if (ksplit > 1) {
            float* out = P + (int64_t)ks * M * N;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 32 + acc_row(r, half);
                if (row < M) out[(int64_t)row * N + col] = acc[r];
            }
            return;
        }
        const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wr * 32 + acc_row(r, half);
            if (row < M) {
                float v = acc[r] + bv;
                if (RESIDUAL) v += R[(int64_t)row * ldr + col];
                Y[(int64_t)row * ldy + col] = v;
            }
        }
    }
}

// Y = sum over the K slices (in slice order) + bias + residual: the epilogue of prefill_gemm_kernel for split launches.
__global__ __launch_bounds__(256) void prefill_splitk_reduce_kernel(const float* __restrict__ P, int ksplit, const float* __restrict__ bias,
                                                                    const float* R, int64_t ldr, float* Y, int64_t ldy, int M, int N,
                                                                    float* G)
{
    const int n4 = N >> 2;
    const int64_t total = (int64_t)M * n4, slab = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / n4;
        const int col = (int)(i - row * n4) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(P + row * N + col);
        for (int s = 1; s < ksplit; ++s) v += *reinterpret_cast<const f32x4*>(P + s * slab + row * N + col);
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + col);
        if (R) v += *reinterpret_cast<const f32x4*>(R + row * ldr + col);
        if (G) {  // this GEMM is the SwiGLU `up`: the gate buffer becomes silu(gate) * up (swiglu.rs:32-57)
            const f32x4 g = *reinterpret_cast<const f32x4*>(G + row * ldy + col);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (g[c] / (1.0f + expf(-g[c]))) * v[c];
            *reinterpret_cast<f32x4*>(G + row * ldy + col) = v;
        } else {
            *reinterpret_cast<f32x4*>(Y + row * ldy + col) = v;
        }
    }
}

// Causal grouped-query attention for a block of prompt rows against the KV cache (decoder_attention.rs:99-160 with the
// causal mask of utils/masks.rs:103-113), flash style: one workgroup per (32 query rows, head), 64-key tiles of K and V
// staged in LDS, running (max, sum, output) per query -- the [rows, keys] score matrix never exists.  8 threads per
// query: each scores 8 keys of the tile (keys kg, kg+8, ...: neighbouring threads read neighbouring K rows) and owns
// head_dim / 8 output columns.  Masked keys contribute exactly 0, as exp(-1e9 - max) does in the reference.
constexpr int PA_Q = 32, PA_K = 64;

template <int DPT>
__global__ __launch_bounds__(256) void prefill_attention_kernel(const float* __restrict__ q, int64_t ldq, int rows,
                                                                const float* __restrict__ K, int64_t ldk,
                                                                const float* __restrict__ V, int64_t ldv, int base, int kv_group,
                                                                float scale, float* __restrict__ ctx, int64_t ldc)
{
    constexpr int D = 8 * DPT, LD = D + 4;
    extern __shared__ __attribute__((aligned(16))) float pa_smem[];
    float* sQ = pa_smem;               // [PA_Q][LD]
    float* sK = sQ + PA_Q * LD;        // [PA_K][LD]
    float* sV = sK + PA_K * LD;        // [PA_K][LD]
    float* sP = sV + PA_K * LD;        // [PA_Q][PA_K + 4]
    const int tid = threadIdx.x, qi = tid >> 3, kg = tid & 7;
    const int h = blockIdx.y, hk = h / kv_group;
    const int q0 = blockIdx.x * PA_Q;
    const int q_row = q0 + qi;
    const bool valid = q_row < rows;
    const int limit = base + q_row;                                   // last visible key of this query
    const int last_key = base + min(rows, q0 + PA_Q) - 1;             // last key any query of the block sees
    for (int i = tid; i < PA_Q * (D / 4); i += 256) {
        const int r = i / (D / 4), c4 = i - r * (D / 4);
        const int row = min(q0 + r, rows - 1);
        *reinterpret_cast<f32x4*>(sQ + r * LD + c4 * 4) = *reinterpret_cast<const f32x4*>(q + (int64_t)row * ldq + h * D + c4 * 4);
    }
    float m_run = -INFINITY, l_run = 0.0f;
    float o[DPT];
#pragma unroll
    for (int c = 0; c < DPT; ++c) o[c] = 0.0f;
    for (int k0 = 0; k0 <= last_key; k0 += PA_K) {
        __syncthreads();  // previous tile fully consumed (and sQ written, first trip)
        for (int i = tid; i < PA_K * (D / 4); i += 256) {
            const int r = i / (D / 4), c4 = i - r * (D / 4);
            const int key = min(k0 + r, last_key);
            *reinterpret_cast<f32x4*>(sK + r * LD + c4 * 4) = *reinterpret_cast<const f32x4*>(K + (int64_t)key * ldk + hk * D + c4 * 4);
            *reinterpret_cast<f32x4*>(sV + r * LD + c4 * 4) = *reinterpret_cast<const f32x4*>(V + (int64_t)key * ldv + hk * D + c4 * 4);
        }
        __syncthreads();
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
        for (int c4 = 0; c4 < D / 4; ++c4) {
            const f32x4 qv = *reinterpret_cast<const f32x4*>(sQ + qi * LD + c4 * 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 kv = *reinterpret_cast<const f32x4*>(sK + (kg + 8 * j) * LD + c4 * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[j] = fmaf(qv[c], kv[c], acc[j]);
            }
        }
        float tile_max = -INFINITY;
        bool vis[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] *= scale;
            vis[j] = valid && (k0 + kg + 8 * j) <= limit;
            if (vis[j]) tile_max = fmaxf(tile_max, acc[j]);
        }
        tile_max = group8_max_asc(tile_max);  // (the query's 8 lanes)
        const float m_new = fmaxf(m_run, tile_max);
        float psum = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float p = vis[j] ? expf(acc[j] - m_new) : 0.0f;
            psum += p;
            sP[qi * (PA_K + 4) + kg + 8 * j] = p;
        }
        psum = group8_sum_asc(psum);
        const float alpha = (m_new == -INFINITY || m_run == -INFINITY) ? (m_run == -INFINITY ? 0.0f : 1.0f) : expf(m_run - m_new);
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int c = 0; c < DPT; ++c) o[c] *= alpha;
        m_run = m_new;
        __syncthreads();  // the tile's probabilities are in LDS
        for (int j = 0; j < PA_K; ++j) {
            const float p = sP[qi * (PA_K + 4) + j];
#pragma unroll
            for (int c = 0; c < DPT; ++c) o[c] = fmaf(p, sV[j * LD + kg * DPT + c], o[c]);
        }
    }
    if (valid) {
        const float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;
#pragma unroll
        for (int c = 0; c < DPT; ++c) ctx[(int64_t)q_row * ldc + h * D + kg * DPT + c] = o[c] * inv;
    }
}

// Prefill: silu(gate) * up over [rows, inter] (swiglu.rs:32-57).
__global__ __launch_bounds__(256) void swiglu_mul_kernel(float* __restrict__ gate, const float* __restrict__ up, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 g = *reinterpret_cast<const f32x4*>(gate + i * 4);
        const f32x4 u = *reinterpret_cast<const f32x4*>(up + i * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) g[c] = (g[c] / (1.0f + expf(-g[c]))) * u[c];
        *reinterpret_cast<f32x4*>(gate + i * 4) = g;
    }
}

}  // namespace

template <typename WT>
static hipError_t launch_gemv_t(const LlmGemvArgs& a, hipStream_t stream)
{
    const WT* W = static_cast<const WT*>(a.W);
    const WT* W2 = static_cast<const WT*>(a.W2);
    const bool norm = a.gamma != nullptr;
    // Staging the row in LDS pays when it also has to be normalised (otherwise every wave redoes the statistics); plain
    // projections keep the one-wave-per-column kernel, whose 4x larger grid hides latency better.
    // One row, K = 2048 / 4096 / 8192: the weight-streaming kernel.
    if (a.rows == 1 && a.seg_q == 0 && llm_gemv_streams(a.k, a.W, a.W2)) {
        constexpr bool BF = sizeof(WT) == 2;
        const int ch = a.k / 512;
        const int nm = a.swiglu ? 2 : 1, lpp = BF ? 1 : 2;
        // 16 loads per lane in flight where that still leaves >= 8 waves per CU, else one row per wave
        const bool wide = a.n_out / stream_rows_per_batch(ch, lpp, nm, true) >= 2048;
        const int rb = stream_rows_per_batch(ch, lpp, nm, wide);
        const int batches = (a.n_out + rb - 1) / rb;
        // 8-wave workgroups where they still cover every CU and staging the input row is the larger cost (a long row, or the
        // attention slabs to merge: each workgroup stages the whole row, so fewer, larger ones re-read it less; measured on
        // the 1B shape: down-proj 8.0 -> 7.6 us, output projection with the merge 6.4 -> 4.9 us, gate/up 12.8 -> 13.2 us);
        // the layer's projections get one batch per wave, the vocabulary head loops
        const bool big = batches >= 2048 && (ch >= 16 || a.att_splits > 0);
        const int wpw = big ? 8 : 4;
        const bool loop = (batches + wpw - 1) / wpw > 4096 / wpw;
        const int wgs = loop ? 2048 / wpw : std::max(1, (batches + wpw - 1) / wpw);
#define KJ_ST4(EPI, NORM, CH, WIDE, LOOP, THREADS)                                                                                   \
    hipLaunchKernelGGL((llm_gemv_stream_kernel<WT, EPI, NORM, CH, WIDE, LOOP, THREADS>), dim3((unsigned)wgs), dim3(THREADS), 0, stream, \
                       a.X, a.gamma, a.eps, W, W2, a.bias, a.R, a.n_out, a.Y0, 0, 0, a.norm_out)
#define KJ_ST3(EPI, NORM, CH, WIDE, LOOP)                                                                                            \
    do {                                                                                                                             \
        if (big) KJ_ST4(EPI, NORM, CH, WIDE, LOOP, 512);                                                                             \
        else KJ_ST4(EPI, NORM, CH, WIDE, LOOP, 256);                                                                                 \
    } while (0)
#define KJ_ST_ATT2(CH, WIDE, THREADS, ATT)                                                                                           \
    hipLaunchKernelGGL((llm_gemv_stream_kernel<WT, LE_RESIDUAL, false, CH, WIDE, false, THREADS, ATT>), dim3((unsigned)wgs),         \
                       dim3(THREADS), 0, stream, a.X, a.gamma, a.eps, W, W2, a.bias, a.R, a.n_out, a.Y0, a.att_splits, a.att_head_dim, \
                       nullptr)
#define KJ_ST_ATT(CH, ATT)                                                                                                           \
    do {                                                                                                                             \
        if (big && wide) KJ_ST_ATT2(CH, true, 512, ATT);                                                                             \
        else if (big) KJ_ST_ATT2(CH, false, 512, ATT);                                                                               \
        else if (wide) KJ_ST_ATT2(CH, true, 256, ATT);                                                                               \
        else KJ_ST_ATT2(CH, false, 256, ATT);                                                                                        \
    } while (0)
        if (a.att_splits > 0) {  // the context row arrives as attention slabs
            if (!llm_gemv_merges_attention(a.k, a.att_splits, a.att_head_dim) || !a.R || norm || a.swiglu || loop) return hipErrorInvalidValue;
            if (ch == 4) KJ_ST_ATT(4, 8);
            else KJ_ST_ATT(8, 4);
            return hipGetLastError();
        }
#define KJ_ST2(EPI, NORM, CH)                                                                                                        \
    do {                                                                                                                             \
        if (loop) KJ_ST3(EPI, NORM, CH, true, true);                                                                                 \
        else if (wide) KJ_ST3(EPI, NORM, CH, true, false);                                                                           \
        else KJ_ST3(EPI, NORM, CH, false, false);                                                                                    \
    } while (0)
#define KJ_ST1(EPI, NORM)                                                                                                            \
    do {                                                                                                                             \
        if (ch == 4) KJ_ST2(EPI, NORM, 4);                                                                                           \
        else if (ch == 8) KJ_ST2(EPI, NORM, 8);                                                                                      \
        else KJ_ST2(EPI, NORM, 16);                                                                                                  \
    } while (0)
        if (a.swiglu) {
            if (norm) KJ_ST1(LE_SWIGLU, true);
            else KJ_ST1(LE_SWIGLU, false);
        } else if (a.R) {
            if (norm) KJ_ST1(LE_RESIDUAL, true);
            else KJ_ST1(LE_RESIDUAL, false);
        } else {
            if (norm) KJ_ST1(LE_NONE, true);
            else KJ_ST1(LE_NONE, false);
        }
#undef KJ_ST1
#undef KJ_ST2
#undef KJ_ST3
#undef KJ_ST4
#undef KJ_ST_ATT
#undef KJ_ST_ATT2
        return hipGetLastError();
    }
    if (a.rows == 1 && a.seg_q == 0 && a.k <= SK_MAX_CHUNKS * 2048 && (g_llm_gemv_variant == 0 || g_llm_gemv_variant >= 3)) {
        // Long rows (down-proj: k = 8192 .. 14336) are spread over 16 waves per workgroup: a quarter of the loads per lane,
        // four times the waves in flight, at the same two columns per workgroup.
        const bool wide = a.k >= 8192 && g_llm_gemv_variant != 7;
        const int threads = wide ? 1024 : 256;
        const int chunks = (a.k / 8 + threads - 1) / threads;
        // Two columns per workgroup measured best on every projection of the 1B and 8B shapes (1, 4 and 8 were 2-18 % slower
        // end to end): the chip wants many small workgroups more than it wants deep per-lane load queues.
        int opw = 2;
#ifdef KJARNI_TUNING
        if (g_llm_gemv_variant >= 3 && g_llm_gemv_variant <= 6) opw = 1 << (g_llm_gemv_variant - 3);  // measurements: 3 -> 1 ... 6 -> 8
#endif
        if (opw > 8) opw = 8;
        if (chunks > 2 && opw > 4) opw = 4;
        if (chunks > 4 && opw > 2) opw = 2;
        if (a.swiglu && chunks > 2 && opw > 2) opw = 2;
        if (wide && opw > 2) opw = 2;  // the 16-wave kernel is instantiated for <= 2 columns
        const dim3 gridk((unsigned)((a.n_out + opw - 1) / opw));
#define KJ_SK4(EPI, NORM, OPW, CH, NW)                                                                                                \
    hipLaunchKernelGGL((llm_gemv_splitk_kernel<WT, EPI, NORM, OPW, CH, NW>), gridk, dim3(64 * NW), 0, stream, a.X, a.gamma, a.eps, W, \
                       W2, a.bias, a.R, a.n_out, a.k, a.Y0)
#define KJ_SK3(EPI, NORM, OPW, CH)                                                                                                    \
    do {                                                                                                                              \
        if (wide) KJ_SK4(EPI, NORM, (OPW > 2 ? 2 : OPW), (CH > 2 ? 2 : CH), 16);                                                      \
        else KJ_SK4(EPI, NORM, OPW, CH, 4);                                                                                           \
    } while (0)
#define KJ_SK2(EPI, NORM, OPW)                                                                                                        \
    do {                                                                                                                              \
        if (chunks <= 1) KJ_SK3(EPI, NORM, OPW, 1);                                                                                   \
        else if (chunks <= 2) KJ_SK3(EPI, NORM, OPW, 2);                                                                              \
        else if (chunks <= 4) KJ_SK3(EPI, NORM, (OPW > 4 ? 4 : OPW), 4);                                                              \
        else KJ_SK3(EPI, NORM, (OPW > 2 ? 2 : OPW), 8);                                                                               \
    } while (0)
#define KJ_SK1(EPI, NORM)                                                                                                             \
    do {                                                                                                                              \
        if (opw == 1) KJ_SK2(EPI, NORM, 1);                                                                                           \
        else if (opw == 2) KJ_SK2(EPI, NORM, 2);                                                                                      \
        else if (opw == 4) KJ_SK2(EPI, NORM, 4);                                                                                      \
        else KJ_SK2(EPI, NORM, 8);                                                                                                    \
    } while (0)
        if (a.swiglu) {
            if (norm) KJ_SK1(LE_SWIGLU, true);
            else KJ_SK1(LE_SWIGLU, false);
        } else if (a.R) {
            if (norm) KJ_SK1(LE_RESIDUAL, true);
            else KJ_SK1(LE_RESIDUAL, false);
        } else {
            if (norm) KJ_SK1(LE_NONE, true);
            else KJ_SK1(LE_NONE, false);
        }
#undef KJ_SK4
#undef KJ_SK1
#undef KJ_SK2
#undef KJ_SK3
        return hipGetLastError();
    }
    if (a.rows == 1 && norm && a.k <= G1_MAX_K && g_llm_gemv_variant != 1) {
        const dim3 grid1((unsigned)((a.n_out + 4 * G1_OPW - 1) / (4 * G1_OPW)));
        const size_t lds = (size_t)a.k * sizeof(float);
#define KJ_LLM1(EPI, NORM)                                                                                                            \
    do {                                                                                                                              \
        auto kern = llm_gemv1_kernel<WT, EPI, NORM>;                                                                                  \
        if (lds > 48 * 1024) {                                                                                                        \
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                     (int)lds);                                                                       \
            if (e != hipSuccess) return e;                                                                                            \
        }                                                                                                                             \
        hipLaunchKernelGGL(kern, grid1, dim3(256), lds, stream, a.X, a.gamma, a.eps, W, W2, a.bias, a.R, a.n_out, a.k, a.seg_q,       \
                           a.seg_kv, a.Y0, a.Y1, a.Y2, a.ldy12, a.row_off, a.row_off_ptr);                                            \
    } while (0)
        if (a.swiglu) {
            if (norm) KJ_LLM1(LE_SWIGLU, true);
            else KJ_LLM1(LE_SWIGLU, false);
        } else if (a.R) {
            if (norm) KJ_LLM1(LE_RESIDUAL, true);
            else KJ_LLM1(LE_RESIDUAL, false);
        } else {
            if (norm) KJ_LLM1(LE_NONE, true);
            else KJ_LLM1(LE_NONE, false);
        }
#undef KJ_LLM1
        return hipGetLastError();
    }
    const dim3 grid((unsigned)((a.n_out + 3) / 4));
#define KJ_LLM(EPI, NORM)                                                                                                       \
    hipLaunchKernelGGL((llm_gemv_kernel<WT, EPI, NORM>), grid, dim3(256), 0, stream, a.X, a.ldx, a.rows, a.gamma, a.eps, W, W2,  \
                       a.bias, a.R, a.ldr, a.n_out, a.k, a.seg_q, a.seg_kv, a.Y0, a.ldy0, a.Y1, a.Y2, a.ldy12, a.row_off,       \
                       a.row_off_ptr)
    if (a.swiglu) {
        if (norm) KJ_LLM(LE_SWIGLU, true);
        else KJ_LLM(LE_SWIGLU, false);
    } else if (a.R) {
        if (norm) KJ_LLM(LE_RESIDUAL, true);
        else KJ_LLM(LE_RESIDUAL, false);
    } else {
        if (norm) KJ_LLM(LE_NONE, true);
        else KJ_LLM(LE_NONE, false);
    }
#undef KJ_LLM
    return hipGetLastError();
}

#ifdef KJARNI_TUNING
void set_llm_gemv_variant(int v) { g_llm_gemv_variant = v; }
#endif

int prefill_gemm_ksplit(int M, int N, int K)
{
    const int tiles = ((M + PG_BM - 1) / PG_BM) * ((N + PG_BN - 1) / PG_BN);
    int s = 1;
    // up to 8 slices of >= 128 columns of K while the launch stays within ~4 workgroups per CU
    while (s < 8 && tiles * s * 2 <= 1024 && K % (s * 2 * PG_BK) == 0 && K / (s * 2) >= 128) s *= 2;
    return s;
}

size_t prefill_gemm_scratch_floats(int max_rows, int max_n)
{
    // ksplit * tiles <= 1024 tiles of 64 x 64, or one unsplit slab when there are more tiles than that
    (void)max_rows;
    (void)max_n;
    return (size_t)1024 * PG_BM * PG_BN;
}

namespace {
inline bool prefill_f32_mfma_for_bf16()
{
#ifdef KJARNI_TUNING
    static const bool v = [] { const char* e = std::getenv("KJARNI_HIP_LLM_WIDEN"); return e && e[0] == '1'; }();
    return v;
#else
    return false;
#endif
}
}  // namespace

hipError_t launch_prefill_gemm(const float* A, int64_t lda, const void* W, int bf16, const float* bias, const float* R, int64_t ldr, float* Y,
                               int64_t ldy, int M, int N, int K, hipStream_t stream, float* split_scratch, float* silu_gate)
{
    if (M <= 0 || N <= 0) return hipSuccess;
    if (K % PG_BK || lda % 4 || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(W) & 15)) return hipErrorInvalidValue;
    if (silu_gate && (ldy != N || (N & 3) || (reinterpret_cast<uintptr_t>(silu_gate) & 15))) return hipErrorInvalidValue;
    const int m_tiles = (M + PG_BM - 1) / PG_BM, n_tiles = (N + PG_BN - 1) / PG_BN;
    int ksplit = 1;
    if (split_scratch && (N & 3) == 0 && (ldy & 3) == 0 && (!R || (ldr & 3) == 0) && (reinterpret_cast<uintptr_t>(Y) & 15) == 0 &&
        (!R || (reinterpret_cast<uintptr_t>(R) & 15) == 0) && (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0))
        ksplit = prefill_gemm_ksplit(M, N, K);
    const dim3 grid((unsigned)(m_tiles * n_tiles * ksplit));
#define KJ_PG(WT, RES)                                                                                                              \
    hipLaunchKernelGGL((prefill_gemm_kernel<WT, RES>), grid, dim3(256), 0, stream, A, lda, static_cast<const WT*>(W), bias, R, ldr, Y, ldy, \
                       M, N, K, m_tiles, ksplit, split_scratch)
    if (bf16 && !prefill_f32_mfma_for_bf16()) {
        // bf16 weights: on the bf16 matrix cores, the activations as three exact bf16 pieces (gemm_split.hip)
        const hipError_t e = launch_prefill_tiles_bf16w((unsigned)(m_tiles * n_tiles * ksplit), A, lda, W, bias, R, ldr, Y, ldy, M, N, K, m_tiles,
                                                        ksplit, split_scratch, stream);
        if (e != hipSuccess) return e;
    } else if (bf16) {
        if (R) KJ_PG(uint16_t, true);
        else KJ_PG(uint16_t, false);
    } else {
        if (R) KJ_PG(float, true);
        else KJ_PG(float, false);
    }
#undef KJ_PG
    if (ksplit > 1) {
        const int64_t total = (int64_t)M * (N / 4);
        const unsigned blocks = (unsigned)std::min<int64_t>(2048, (total + 255) / 256);
        hipLaunchKernelGGL(prefill_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, split_scratch, ksplit, bias, R, ldr, Y, ldy, M, N,
                           silu_gate);
    } else if (silu_gate) {
        return launch_swiglu_mul(silu_gate, Y, (size_t)M * N, stream);
    }
    return hipGetLastError();
}

namespace {
// bf16 -> f32, eight values per thread (a long prompt's projections run the encoder's f32 tile GEMM on a widened copy)
__global__ __launch_bounds__(256) void widen_bf16_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst, size_t n8)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const F8 w = load8(src, (int64_t)i);
        *reinterpret_cast<f32x4*>(dst + i * 8) = f32x4{w.v[0], w.v[1], w.v[2], w.v[3]};
        *reinterpret_cast<f32x4*>(dst + i * 8 + 4) = f32x4{w.v[4], w.v[5], w.v[6], w.v[7]};
    }
}
}  // namespace

hipError_t launch_widen_bf16(const void* src, float* dst, size_t n, hipStream_t stream)
{
    if (n & 7) return hipErrorInvalidValue;
    const size_t n8 = n >> 3;
    hipLaunchKernelGGL(widen_bf16_kernel, dim3((unsigned)std::min<size_t>(4096, (n8 + 255) / 256)), dim3(256), 0, stream,
                       static_cast<const uint16_t*>(src), dst, n8);
    return hipGetLastError();
}

namespace {
// Causal grouped-query attention of a long prompt block on the fp32 matrix cores (the flash-style kernel above is a vector-ALU
// kernel: 27 TFLOP/s, a quarter of a 2 048-token prompt's time).  Structure of the encoder's attention_kernel (attention.hip):
// a workgroup owns 128 queries of one head (4 waves x 32), walks the keys in chunks of 128 staged in LDS (K row-major, V
// transposed), S^T = K Q^T and O += P V as 32 x 32 x 2 MFMA tiles, online softmax in the exp2 domain with the running
// (max, sum) on lane == query.  Causality: query row i of the block sees keys <= base + i (decoder_attention.rs:99-160,
// utils/masks.rs:103-113 overwrite masked scores with -1e9, whose exp is exactly 0 next to the unmasked diagonal score --
// here they are simply left out); chunks past a workgroup's last query are never visited, a wave skips the chunks past its
// own last query, and only the chunks that straddle a wave's diagonal pay for the per-element test.
constexpr int PM_Q = 128, PM_K = 128;
template <int D>
struct PmSmem {
    static constexpr int K_STRIDE = D + 4, VT_STRIDE = PM_K + 4;
    static constexpr int K_FLOATS = PM_K * K_STRIDE, VT_FLOATS = D * VT_STRIDE;
    static constexpr int BYTES = (K_FLOATS + VT_FLOATS) * 4;
};

template <int D>
__global__ __launch_bounds__(256, D <= 64 ? 2 : 1) void prefill_attention_mfma_kernel(const float* __restrict__ q, int64_t ldq, int rows,
                                                                                       const float* __restrict__ K, int64_t ldk,
                                                                                       const float* __restrict__ V, int64_t ldv, int base,
                                                                                       int kv_group, float scale, float* __restrict__ ctx,
                                                                                       int64_t ldc)
{
    using SM = PmSmem<D>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sK = smem;                  // [128][D + 4]
    float* sVt = smem + SM::K_FLOATS;  // [D][128 + 4]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    // A query block's work grows with its index (block i walks i + 1 key chunks when nothing is cached).  All workgroups of a
    // launch are resident at once, two per CU, and workgroups L and L + 256 tend to share one: the upper half of the heads
    // walks the blocks in reverse, so the pairs add up to the same work.
    const int h = blockIdx.y, kvh = h / kv_group;
    const int qb = (2 * (int)blockIdx.y >= (int)gridDim.y) ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const float* q_base = q + h * D;
    const float* k_base = K + kvh * D;
    const float* v_base = V + kvh * D;

    const int q_wave0 = qb * PM_Q + wid * 32;  // first query row of this wave
    const int q_row = q_wave0 + l31;
    f32x4 qf[D / 8];  // B operand of S^T = K Q^T: lane supplies Q[q][8 kk + 4 half + c]
#pragma unroll
    for (int kk = 0; kk < D / 8; ++kk)
        qf[kk] = q_row < rows ? *reinterpret_cast<const f32x4*>(q_base + (int64_t)q_row * ldq + kk * 8 + half * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x16 o[D / 32];
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.0f;
    float run_max = -INFINITY, run_sum = 0.0f;
    const int n_keys = base + rows;
    const int last_q_wg = min(rows, qb * PM_Q + PM_Q) - 1;     // last query row of the workgroup
    const int last_q_wave = min(rows - 1, q_wave0 + 31);        // (below q_wave0 when the wave has no query: it only helps staging)
    const int n_chunks = (base + last_q_wg) / PM_K + 1;
    const float c1 = scale * 1.4426950408889634f;

    for (int ch = 0; ch < n_chunks; ++ch) {
        const int key0 = ch * PM_K;
        if (ch > 0) __syncthreads();  // everyone done reading the previous chunk
        constexpr int V4_PER_ROW = D / 4;
        for (int f = tid; f < PM_K * V4_PER_ROW; f += 256) {
            const int r = f / V4_PER_ROW, c4 = f % V4_PER_ROW;
            const int key = key0 + r;
            f32x4 kv = f32x4{0.f, 0.f, 0.f, 0.f}, vv = kv;
            if (key < n_keys) {
                kv = *reinterpret_cast<const f32x4*>(k_base + (int64_t)key * ldk + c4 * 4);
                vv = *reinterpret_cast<const f32x4*>(v_base + (int64_t)key * ldv + c4 * 4);
            }
            *reinterpret_cast<f32x4*>(sK + r * SM::K_STRIDE + c4 * 4) = kv;
#pragma unroll
            for (int c = 0; c < 4; ++c) sVt[(c4 * 4 + c) * SM::VT_STRIDE + r] = vv[c];
        }
        __syncthreads();
        if (q_wave0 >= rows || key0 > base + last_q_wave) continue;  // no query here, or every key of the chunk is in this wave's future
        const bool plain = key0 + PM_K - 1 <= base + q_wave0;      // every key visible to every query of the wave

        f32x16 s[4];  // S^T tiles: 4 key tiles x 32 queries, K = D
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.0f;
            const float* pk = sK + (kt * 32 + l31) * SM::K_STRIDE + half * 4;
#pragma unroll
            for (int kk = 0; kk < D / 8; ++kk) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(pk + kk * 8);
#pragma unroll
                for (int c = 0; c < 4; ++c) s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c], qf[kk][c], s[kt], 0, 0, 0);
            }
        }
        float cmax = -INFINITY;
        if (plain) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[kt][r] *= c1;
                    cmax = fmaxf(cmax, s[kt][r]);
                }
        } else {
            const int limit = base + q_row;  // this lane's query sees keys <= limit
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + kt * 32 + acc_row(r, half);
                    const float v = key <= limit ? s[kt][r] * c1 : -INFINITY;
                    s[kt][r] = v;
                    cmax = fmaxf(cmax, v);
                }
        }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32, kWave));
        const float new_max = fmaxf(run_max, cmax);
        // (a lane without a query, or whose keys of this chunk are all in its future, keeps new_max == run_max)
        const float alpha = (run_max == -INFINITY) ? 0.0f : __builtin_amdgcn_exp2f(run_max - new_max);
        float csum = 0.0f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float e = __builtin_amdgcn_exp2f(s[kt][r] - new_max);
                if (new_max == -INFINITY) e = 0.0f;
                s[kt][r] = e;
                csum += e;
            }
        csum += __shfl_xor(csum, 32, kWave);
        run_sum = run_sum * alpha + csum;
        run_max = new_max;
#pragma unroll
        for (int r = 0; r < 16; ++r) {  // O's rows are queries indexed by (reg, half); alpha lives on lane == query
            const float a = __shfl(alpha, acc_row(r, half), kWave);
#pragma unroll
            for (int dt = 0; dt < D / 32; ++dt) o[dt][r] *= a;
        }
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) {  // O += P V: A = P (query l31, keys 8 g + 4 half + c), B = V^T[d][key]
            const float* pv = sVt + (dt * 32 + l31) * SM::VT_STRIDE + half * 4;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 vf = *reinterpret_cast<const f32x4*>(pv + kt * 32 + g * 8);
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[kt][g * 4 + c], vf[c], o[dt], 0, 0, 0);
                }
        }
    }
    const float inv = run_sum > 0.0f ? 1.0f / run_sum : 1.0f;
    float* out_base = ctx + h * D;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qq = q_wave0 + acc_row(r, half);
        const float is = __shfl(inv, acc_row(r, half), kWave);
        if (qq < rows) {
#pragma unroll
            for (int dt = 0; dt < D / 32; ++dt) out_base[(int64_t)qq * ldc + dt * 32 + l31] = o[dt][r] * is;
        }
    }
}
}  // namespace

bool prefill_attention_supported(int head_dim) { return head_dim == 16 || head_dim == 32 || head_dim == 64 || head_dim == 128; }

hipError_t launch_prefill_attention(const float* q, int64_t ldq, int rows, const float* K, int64_t ldk, const float* V, int64_t ldv, int base,
                                    int heads, int head_dim, int kv_group, float* ctx, int64_t ldc, hipStream_t stream)
{
    if (rows <= 0) return hipSuccess;
    if (!prefill_attention_supported(head_dim)) return hipErrorInvalidValue;
    // long blocks of 64- / 128-wide heads: the matrix-core kernel
    if (rows >= 256 && (head_dim == 64 || head_dim == 128) && g_llm_gemv_variant != 9 && (ldq & 3) == 0 && (ldk & 3) == 0 && (ldv & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(V)) & 15) == 0) {
        const dim3 mgrid((unsigned)((rows + PM_Q - 1) / PM_Q), (unsigned)heads);
        const float mscale = 1.0f / sqrtf((float)head_dim);
        const int g = kv_group < 1 ? 1 : kv_group;
        if (head_dim == 64) {
            auto kern = prefill_attention_mfma_kernel<64>;
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     PmSmem<64>::BYTES);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, mgrid, dim3(256), PmSmem<64>::BYTES, stream, q, ldq, rows, K, ldk, V, ldv, base, g, mscale, ctx, ldc);
        } else {
            auto kern = prefill_attention_mfma_kernel<128>;
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     PmSmem<128>::BYTES);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, mgrid, dim3(256), PmSmem<128>::BYTES, stream, q, ldq, rows, K, ldk, V, ldv, base, g, mscale, ctx, ldc);
        }
        return hipGetLastError();
    }
    const dim3 grid((unsigned)((rows + PA_Q - 1) / PA_Q), (unsigned)heads);
    const int LD = head_dim + 4;
    const size_t lds = ((size_t)(PA_Q + 2 * PA_K) * LD + (size_t)PA_Q * (PA_K + 4)) * sizeof(float);
    const float scale = 1.0f / sqrtf((float)head_dim);
#define KJ_PA(DPT)                                                                                                                      \
    do {                                                                                                                                \
        auto kern = prefill_attention_kernel<DPT>;                                                                                      \
        if (lds > 48 * 1024) {                                                                                                          \
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                                     (int)lds);                                                                         \
            if (e != hipSuccess) return e;                                                                                              \
        }                                                                                                                               \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, q, ldq, rows, K, ldk, V, ldv, base, kv_group < 1 ? 1 : kv_group, scale,  \
                           ctx, ldc);                                                                                                   \
    } while (0)
    switch (head_dim) {
    case 16: KJ_PA(2); break;
    case 32: KJ_PA(4); break;
    case 64: KJ_PA(8); break;
    default: KJ_PA(16); break;
    }
#undef KJ_PA
    return hipGetLastError();
}

hipError_t launch_swiglu_mul(float* gate, const float* up, size_t n, hipStream_t stream)
{
    if (n % 4) return hipErrorInvalidValue;
    const size_t n4 = n / 4;
    const unsigned grid = (unsigned)std::min<size_t>((n4 + 255) / 256, 8192);
    hipLaunchKernelGGL(swiglu_mul_kernel, dim3(grid), dim3(256), 0, stream, gate, up, n4);
    return hipGetLastError();
}

bool llm_qkv_rope_embeds(int k, const float* gamma, const void* W, const void* table)
{
    return (k == 2048 || k == 4096 || k == 8192) && gamma && (reinterpret_cast<uintptr_t>(W) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(table) & 15) == 0 && g_llm_gemv_variant != 8 && g_llm_gemv_variant != 1;
}

hipError_t launch_llm_qkv_rope(const float* X, const float* gamma, float eps, const void* W, int bf16, const float* bias, int k,
                               int n_heads, int n_kv_heads, int head_dim, const float* cos_t, const float* sin_t, float* Q, float* Kc,
                               float* Vc, int pos, const int* pos_ptr, hipStream_t stream, const uint32_t* embed_ids, const void* table,
                               int vocab, float* x_raw_out)
{
    if ((k & 7) || k > 8192 || (head_dim & 1)) return hipErrorInvalidValue;
    const int tasks = (n_heads * head_dim + 2 * n_kv_heads * head_dim) / 2;
    if (embed_ids && !llm_qkv_rope_embeds(k, gamma, W, table)) return hipErrorInvalidValue;
    if ((k == 2048 || k == 4096 || k == 8192) && gamma && (reinterpret_cast<uintptr_t>(W) & 15) == 0 && g_llm_gemv_variant != 8 &&
        g_llm_gemv_variant != 1) {
        const dim3 sgrid((unsigned)((tasks + 3) / 4));
#define KJ_QKVS(WT, CH)                                                                                                              \
    do {                                                                                                                             \
        if (embed_ids)                                                                                                               \
            hipLaunchKernelGGL((llm_qkv_rope_stream_kernel<WT, CH, true>), sgrid, dim3(256), 0, stream, X, gamma, eps,                \
                               static_cast<const WT*>(W), bias, n_heads, n_kv_heads, head_dim, cos_t, sin_t, Q, Kc, Vc, pos, pos_ptr, \
                               embed_ids, static_cast<const WT*>(table), vocab, x_raw_out);                                          \
        else                                                                                                                         \
            hipLaunchKernelGGL((llm_qkv_rope_stream_kernel<WT, CH, false>), sgrid, dim3(256), 0, stream, X, gamma, eps,               \
                               static_cast<const WT*>(W), bias, n_heads, n_kv_heads, head_dim, cos_t, sin_t, Q, Kc, Vc, pos, pos_ptr, \
                               nullptr, nullptr, 0, nullptr);                                                                        \
    } while (0)
        if (bf16) {
            if (k == 2048) KJ_QKVS(uint16_t, 4);
            else if (k == 4096) KJ_QKVS(uint16_t, 8);
            else KJ_QKVS(uint16_t, 16);
        } else {
            if (k == 2048) KJ_QKVS(float, 4);
            else if (k == 4096) KJ_QKVS(float, 8);
            else KJ_QKVS(float, 16);
        }
#undef KJ_QKVS
        return hipGetLastError();
    }
    const dim3 grid((unsigned)tasks);
    const int chunks = (k / 8 + 255) / 256;
#define KJ_QKV(WT, CH)                                                                                                              \
    hipLaunchKernelGGL((llm_qkv_rope_kernel<WT, CH>), grid, dim3(256), 0, stream, X, gamma, eps, static_cast<const WT*>(W), bias, k, \
                       n_heads, n_kv_heads, head_dim, cos_t, sin_t, Q, Kc, Vc, pos, pos_ptr)
    if (bf16) {
        if (chunks <= 1) KJ_QKV(uint16_t, 1);
        else if (chunks <= 2) KJ_QKV(uint16_t, 2);
        else KJ_QKV(uint16_t, 4);
    } else {
        if (chunks <= 1) KJ_QKV(float, 1);
        else if (chunks <= 2) KJ_QKV(float, 2);
        else KJ_QKV(float, 4);
    }
#undef KJ_QKV
    return hipGetLastError();
}

bool llm_gemv_streams(int k, const void* W, const void* W2)
{
    return (k == 2048 || k == 4096 || k == 8192) && g_llm_gemv_variant != 8 && g_llm_gemv_variant != 1 &&
           (reinterpret_cast<uintptr_t>(W) & 15) == 0 && (!W2 || (reinterpret_cast<uintptr_t>(W2) & 15) == 0);
}

bool llm_gemv_merges_attention(int k, int splits, int head_dim)
{
    if (g_llm_gemv_variant == 8 || g_llm_gemv_variant == 1) return false;  // measurements: the kernels before the streaming one
    if (head_dim < 4 || (head_dim & 3) || k % head_dim) return false;
    return (k == 2048 && splits >= 1 && splits <= 8) || (k == 4096 && splits >= 1 && splits <= 4);
}

hipError_t launch_llm_gemv(const LlmGemvArgs& a, hipStream_t stream)
{
    if (a.att_splits > 0 && (a.rows != 1 || a.seg_q != 0)) return hipErrorInvalidValue;
    if (a.norm_out && !(a.rows == 1 && a.seg_q == 0 && a.gamma && llm_gemv_streams(a.k, a.W, a.W2))) return hipErrorInvalidValue;
    if (a.rows <= 0 || a.n_out <= 0) return hipSuccess;
    if (a.rows > LLM_MAX_ROWS || (a.k & 7) || (a.ldx & 3) || (reinterpret_cast<uintptr_t>(a.X) & 15) ||
        (reinterpret_cast<uintptr_t>(a.W) & 15))
        return hipErrorInvalidValue;
    return a.bf16 ? launch_gemv_t<uint16_t>(a, stream) : launch_gemv_t<float>(a, stream);
}

hipError_t launch_rope(float* x, int64_t ldx, int rows, int n_heads, int head_dim, const float* cos_t, const float* sin_t, int pos,
                       const int* pos_ptr, int at_cache_row, hipStream_t stream)
{
    const int total = rows * n_heads * (head_dim / 2);
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, ldx, rows, n_heads, head_dim, cos_t,
                       sin_t, pos, pos_ptr, at_cache_row);
    return hipGetLastError();
}

hipError_t launch_rmsnorm(const float* x, const float* gamma, float eps, int rows, int hidden, float* out, hipStream_t stream)
{
    if (rows <= 0) return hipSuccess;
    if ((hidden & 3) == 0) {  // one workgroup per row, 16-byte loads (the wave-per-row kernel with scalar loads: 16 vs 5 us at 128 rows)
        hipLaunchKernelGGL(rmsnorm_block_kernel, dim3((unsigned)rows), dim3(256), 0, stream, x, gamma, eps, hidden, out);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(rmsnorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, gamma, eps, rows, hidden, out);
    return hipGetLastError();
}

#ifdef KJARNI_TUNING
namespace {
// Measurements only: pull `bytes` through the memory-side cache (plain loads, values discarded).
__global__ __launch_bounds__(256) void touch_kernel(const uint4* __restrict__ p, size_t n16, unsigned* __restrict__ sink)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9e3779b9u) *sink = acc;
}
}  // namespace

hipError_t launch_touch(const void* p, size_t bytes, unsigned* sink, hipStream_t stream)
{
    hipLaunchKernelGGL(touch_kernel, dim3(2048), dim3(256), 0, stream, static_cast<const uint4*>(p), bytes / 16, sink);
    return hipGetLastError();
}
#endif

hipError_t launch_llm_embed(const uint32_t* ids, int n, int hidden, int vocab, const void* table, int bf16, float* out,
                            hipStream_t stream)
{
    if (hidden & 7) return hipErrorInvalidValue;
    if (bf16)
        hipLaunchKernelGGL(llm_embed_kernel<uint16_t>, dim3((unsigned)n), dim3(256), 0, stream, ids, hidden, vocab,
                           static_cast<const uint16_t*>(table), out);
    else
        hipLaunchKernelGGL(llm_embed_kernel<float>, dim3((unsigned)n), dim3(256), 0, stream, ids, hidden, vocab,
                           static_cast<const float*>(table), out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Sampled decoding: what of sample_token (common/sampling.rs:81-114) and of the logits processors (:8-57) is O(vocab) runs
// here, on the logits where the vocabulary head left them; the host gets a few hundred (token, logit) pairs instead of
// 4 x vocab bytes and finishes the filters on them exactly (sampling.cpp, sampling_distribution_candidates).
//
//   logits processors: repetition penalty once per occurrence of every past token (counts per token + the list of distinct
//     tokens are kept on the device, updated by one tiny kernel per new token); no-repeat-n-gram bans by one thread per
//     window of the history.
//   sample_max:     per-workgroup maxima (the first launch also clears the histogram of the previous token);
//   sample_hist:    m = max; histogram of (m - logit) in bins of 1/8 (count and exp mass per bin); per-workgroup sums of
//                   exp(logit - m), each in a fixed order (the total is deterministic, but not the reference's index order:
//                   the host treats it as such);
//   sample_compact: every workgroup derives the cut `tau` from the histogram -- the distance below the maximum that holds
//                   top_k tokens and top_p of the mass, two bins of margin, and ln(1 / min_p) when min_p filters the whole
//                   vocabulary -- then appends every token with logit >= m - tau to the candidate list.
// A list longer than its capacity, or a cut the histogram cannot place, is reported in the header and the host falls back
// to fetching the logits.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int SAMPLE_BLOCKS = 64, SAMPLE_BINS = 512;

struct SampleScratch {                 // device memory, zero-initialised once
    float part_max[SAMPLE_BLOCKS];
    float part_sum[SAMPLE_BLOCKS];
    unsigned hist_count[SAMPLE_BINS + 1];
    float hist_mass[SAMPLE_BINS + 1];
};

__global__ __launch_bounds__(256) void sample_max_kernel(const float* __restrict__ logits, int vocab, SampleScratch* __restrict__ sc,
                                                         SampleHeader* __restrict__ header)
{
    __shared__ float red[4];
    float m = -INFINITY;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < vocab; i += gridDim.x * 256) m = fmaxf(m, logits[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) sc->part_max[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    // (the histogram and the candidate counter of the previous token are dead by now)
    for (int b = blockIdx.x * 256 + threadIdx.x; b <= SAMPLE_BINS; b += gridDim.x * 256) {
        sc->hist_count[b] = 0u;
        sc->hist_mass[b] = 0.0f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        header->count = 0u;
        header->overflow = 0u;
    }
}

__global__ __launch_bounds__(256) void sample_hist_kernel(const float* __restrict__ logits, int vocab, SampleScratch* __restrict__ sc)
{
    __shared__ unsigned h_count[SAMPLE_BINS + 1];
    __shared__ float h_mass[SAMPLE_BINS + 1];
    __shared__ float red[4];
    for (int b = threadIdx.x; b <= SAMPLE_BINS; b += 256) {
        h_count[b] = 0u;
        h_mass[b] = 0.0f;
    }
    float m = -INFINITY;
    for (int b = 0; b < SAMPLE_BLOCKS; ++b) m = fmaxf(m, sc->part_max[b]);
    __syncthreads();
    float sum = 0.0f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < vocab; i += gridDim.x * 256) {
        const float v = logits[i];
        const float e = expf(v - m);
        const float d = (m - v) * 8.0f;
        const int bin = d < (float)SAMPLE_BINS ? (int)d : SAMPLE_BINS;  // (NaN and -inf land in the last bin)
        sum += e;
        atomicAdd(&h_count[bin], 1u);
        atomicAdd(&h_mass[bin], e);
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) sc->part_sum[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    for (int b = threadIdx.x; b <= SAMPLE_BINS; b += 256) {
        if (h_count[b]) {
            atomicAdd(&sc->hist_count[b], h_count[b]);
            atomicAdd(&sc->hist_mass[b], h_mass[b]);
        }
    }
}

__global__ __launch_bounds__(256) void sample_compact_kernel(const float* __restrict__ logits, int vocab, long long top_k, float top_p,
                                                             float min_p, const SampleScratch* __restrict__ sc,
                                                             SampleHeader* __restrict__ header, SampleCandidate* __restrict__ cand, int cap)
{
    __shared__ float s_floor;
    __shared__ int s_all;
    if (threadIdx.x == 0) {
        float m = -INFINITY, sum = 0.0f;
        for (int b = 0; b < SAMPLE_BLOCKS; ++b) m = fmaxf(m, sc->part_max[b]);
        for (int b = 0; b < SAMPLE_BLOCKS; ++b) sum += sc->part_sum[b];
        // the cut: enough tokens for top-k, enough mass for top-p (the histogram's masses are summed in no fixed order:
        // a relative margin on top of the two bins), everything min-p can keep when it filters the whole vocabulary
        const bool k_on = top_k >= 0 && top_k < (long long)vocab;  // (top_k >= vocab filters nothing)
        const double need_count = k_on ? (double)top_k : 1.0;
        const double need_mass = top_p >= 0.0f ? (double)top_p * (double)sum * 1.001 : 0.0;
        double cnt = 0.0, mass = 0.0;
        float tau = INFINITY;
        for (int b = 0; b < SAMPLE_BINS; ++b) {
            cnt += (double)sc->hist_count[b];
            mass += (double)sc->hist_mass[b];
            if (cnt >= need_count && mass > need_mass) {
                tau = (float)(b + 2) / 8.0f;
                break;
            }
        }
        const bool minp_on_all = min_p >= 0.0f && !k_on && !(top_p >= 0.0f && top_p < 1.0f);
        if (minp_on_all) tau = min_p > 0.0f ? fmaxf(tau, -logf(min_p) + 0.25f) : INFINITY;
        s_all = !(tau < 1e30f) || !(m > -INFINITY) || !(m < INFINITY);
        s_floor = m - tau;
        if (blockIdx.x == 0) {
            header->mx = m;
            header->sum = sum;
            header->floor = s_all ? -INFINITY : m - tau;
        }
    }
    __syncthreads();
    if (s_all) {  // no usable cut: the host takes the logits
        if (blockIdx.x == 0 && threadIdx.x == 0) header->overflow = 1u;
        return;
    }
    const float floor = s_floor;
    const int lane = threadIdx.x & 63;
    for (int i0 = blockIdx.x * 256; i0 < vocab; i0 += gridDim.x * 256) {
        const int i = i0 + threadIdx.x;
        const float v = i < vocab ? logits[i] : -INFINITY;
        const bool keep = i < vocab && v >= floor;
        const unsigned long long bits = __ballot(keep);
        if (bits == 0ull) continue;
        unsigned base = 0u;
        if (lane == 0) base = atomicAdd(&header->count, (unsigned)__popcll(bits));
        base = __shfl(base, 0, kWave);
        if (keep) {
            const unsigned slot = base + (unsigned)__popcll(bits & ((1ull << lane) - 1ull));
            if (slot < (unsigned)cap) cand[slot] = SampleCandidate{(uint32_t)i, v};
            else header->overflow = 1u;
        }
    }
}

// counts[t] += 1 for each of the n tokens; a token seen for the first time joins the list of distinct tokens
__global__ __launch_bounds__(256) void token_counts_kernel(const int32_t* __restrict__ tokens, int n, int vocab, int* __restrict__ counts,
                                                           int32_t* __restrict__ distinct, int* __restrict__ n_distinct)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int t = tokens[i];
    if (t < 0 || t >= vocab) return;
    if (atomicAdd(&counts[t], 1) == 0) distinct[atomicAdd(n_distinct, 1)] = t;
}

// apply_repetition_penalty (sampling.rs:8-27 / generator.rs:331-337): s < 0 ? s * penalty : s / penalty, once per OCCURRENCE
// of the token in the history, in sequence (each application rounds)
__global__ __launch_bounds__(256) void repetition_penalty_kernel(float* __restrict__ logits, const int* __restrict__ counts,
                                                                 const int32_t* __restrict__ distinct,
                                                                 const int* __restrict__ n_distinct, float penalty)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= *n_distinct) return;
    const int t = distinct[j];
    float s = logits[t];
    for (int c = counts[t]; c > 0; --c) s = s < 0.0f ? __fmul_rn(s, penalty) : __fdiv_rn(s, penalty);
    logits[t] = s;
}

// apply_no_repeat_ngram (sampling.rs:29-57): every window of the history whose first n - 1 tokens equal the last n - 1 bans
// its n-th token
__global__ __launch_bounds__(256) void no_repeat_ngram_kernel(float* __restrict__ logits, int vocab, const int32_t* __restrict__ tokens,
                                                              int len, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i + n > len) return;
    const int32_t* tail = tokens + len - (n - 1);
    for (int k = 0; k < n - 1; ++k)
        if (tokens[i + k] != tail[k]) return;
    const int banned = tokens[i + n - 1];
    if (banned >= 0 && banned < vocab) logits[banned] = -INFINITY;
}

hipError_t launch_argmax(const float* logits, int vocab, unsigned long long* best_scratch, int32_t* out, int32_t* history, int* count,
                         int* pos, hipStream_t stream)
{
    int blocks = (vocab + 2047) / 2048;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(argmax_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, logits, vocab, best_scratch);
    hipLaunchKernelGGL(argmax_finalize_kernel, dim3(1), dim3(1), 0, stream, best_scratch, out, history, count, pos);
    return hipGetLastError();
}


size_t sample_scratch_bytes() { return sizeof(SampleScratch); }

hipError_t launch_sample_candidates(const float* logits, int vocab, int64_t top_k, float top_p, float min_p, void* scratch,
                                    SampleHeader* header, SampleCandidate* candidates, int capacity, hipStream_t stream)
{
    SampleScratch* sc = static_cast<SampleScratch*>(scratch);
    hipLaunchKernelGGL(sample_max_kernel, dim3(SAMPLE_BLOCKS), dim3(256), 0, stream, logits, vocab, sc, header);
    hipLaunchKernelGGL(sample_hist_kernel, dim3(SAMPLE_BLOCKS), dim3(256), 0, stream, logits, vocab, sc);
    hipLaunchKernelGGL(sample_compact_kernel, dim3(SAMPLE_BLOCKS), dim3(256), 0, stream, logits, vocab, (long long)top_k, top_p, min_p, sc,
                       header, candidates, capacity);
    return hipGetLastError();
}

hipError_t launch_token_counts(const int32_t* tokens, int n, int vocab, int* counts, int32_t* distinct, int* n_distinct, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(token_counts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tokens, n, vocab, counts, distinct,
                       n_distinct);
    return hipGetLastError();
}

hipError_t launch_logits_processors(float* logits, int vocab, const int32_t* tokens, int len, const int* counts, const int32_t* distinct,
                                    const int* n_distinct, float repetition_penalty, int no_repeat_ngram, hipStream_t stream)
{
    if (repetition_penalty != 1.0f && len > 0)
        hipLaunchKernelGGL(repetition_penalty_kernel, dim3((unsigned)((std::min(len, vocab) + 255) / 256)), dim3(256), 0, stream, logits,
                           counts, distinct, n_distinct, repetition_penalty);
    if (no_repeat_ngram > 0 && len + 1 >= no_repeat_ngram && len >= no_repeat_ngram)
        hipLaunchKernelGGL(no_repeat_ngram_kernel, dim3((unsigned)((len - no_repeat_ngram + 1 + 255) / 256)), dim3(256), 0, stream, logits,
                           vocab, tokens, len, no_repeat_ngram);
    return hipGetLastError();
}

}  // namespace kjarni
