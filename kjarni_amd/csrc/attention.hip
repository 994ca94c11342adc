// Fused encoder self-attention for all heads:
//   S = Q K^T * (1/sqrt(d)) -> padding mask (overwrite) -> softmax -> P V -> merge heads.
//
// Replaces EncoderSelfAttention::forward / forward_noalloc after the QKV
// projection (crates/kjarni-transformers/src/cpu/encoder/encoder_self_attention.rs
// :88-131 / :213-298), matmul_4d (utils/linear_algebra.rs:708-740),
// apply_padding_mask (utils/masks.rs:4-36, value -1e9) /
// apply_padding_mask_inplace (encoder_self_attention.rs:311-325, value -inf) and
// softmax_4d_inplace (activations.rs:223-303).  The [B,h,S,S] score tensor the
// reference materialises (cpu/encoder/buffers.rs:64) never leaves registers.
//
// Layout.  qkv is the fused projection output [tokens, 3H] (Q | K | V, head h at
// columns h*d inside each third).  A workgroup = 4 waves = 128 queries of one
// (sentence, head); each wave owns 32 queries.  Keys are processed in chunks of
// 128: K chunk staged in LDS row-major [128][d+4], V chunk staged TRANSPOSED
// [d][128+4].
//
// The scores are computed transposed, S^T = K Q^T, with v_mfma_f32_32x32x2_f32,
// so an accumulator register holds, for the lane's query (lane&31), the key
// (reg&3)+8*(reg>>2)+4*(lane>>5) of the tile: a query's row is lane-local
// (64 keys in this lane, the other 64 in lane^32) and max / sum need one
// cross-half shuffle.  The same registers are then the A operand of the PV
// product (P is [query x key], k index = lane half): four consecutive registers
// are four consecutive keys, which is one 16-byte read of the transposed V.
//
// One chunk (seq <= 128) follows the reference's op order exactly
// (max, exp(x-max), sum, p = e * (1/sum), then PV).  Longer sequences use the
// online-softmax recurrence over chunks and normalise at the end.
#include <atomic>

#include "device_utils.h"
#include "kernels.h"

namespace kjarni {

namespace {

#ifdef KJARNI_TUNING
std::atomic<int> g_attention_variant{0};  // 0: persistent pipelined kernel for seq <= 128 (default), 1: plain kernel -- tuning build only
#else
constexpr int g_attention_variant = 0;
#endif

constexpr int QBLK = 128;   // queries per workgroup
constexpr int KCHUNK = 128; // keys per LDS chunk

template <int D>
struct AttnSmem {
    static constexpr int K_STRIDE = D + 4;        // floats
    static constexpr int VT_STRIDE = KCHUNK + 4;  // floats
    static constexpr int K_FLOATS = KCHUNK * K_STRIDE;
    static constexpr int VT_FLOATS = D * VT_STRIDE;
    static constexpr int BYTES = (K_FLOATS + VT_FLOATS + KCHUNK) * 4;
};

// Two workgroups per CU (<= 256 VGPRs): one stages its next K / V chunk and runs its softmax while the other's MFMAs
// occupy the matrix pipe.  Left to itself the compiler took 276 registers for D = 64, i.e. one wave per SIMD and no overlap.
template <int D>
__global__ __launch_bounds__(256, 2) void attention_kernel(const float* __restrict__ qkv,
                                                        const uint32_t* __restrict__ mask, int seq,
                                                        int heads, float scale, float mask_value,
                                                        float* __restrict__ ctx)
{
    using SM = AttnSmem<D>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sK = smem;                    // [128][D+4]
    float* sVt = smem + SM::K_FLOATS;    // [D][128+4]
    float* sMask = sVt + SM::VT_FLOATS;  // [128] 1 = keep, 0 = masked, -1 = beyond seq

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;

    const int qb = blockIdx.x;  // query block inside the sentence
    const int h = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int hidden = heads * D;
    const int64_t row_stride = 3 * (int64_t)hidden;
    const float* base = qkv + b * seq * row_stride;
    const float* q_base = base + h * D;
    const float* k_base = base + hidden + h * D;
    const float* v_base = base + 2 * hidden + h * D;

    // Q fragment: B operand of S^T = K Q^T.  Lane supplies Q[q][8kk + 4half + c].
    const int q_row = qb * QBLK + wid * 32 + l31;
    f32x4 qf[D / 8];
#pragma unroll
    for (int kk = 0; kk < D / 8; ++kk) {
        if (q_row < seq)
            qf[kk] = *reinterpret_cast<const f32x4*>(q_base + q_row * row_stride + kk * 8 + half * 4);
        else
            qf[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    f32x16 o[D / 32];
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.0f;

    float run_max = -INFINITY;  // online-softmax state of this lane's query
    float run_sum = 0.0f;
    const int n_chunks = (seq + KCHUNK - 1) / KCHUNK;

    for (int ch = 0; ch < n_chunks; ++ch) {
        const int key0 = ch * KCHUNK;
        if (ch > 0) __syncthreads();  // everyone done reading the previous chunk

        // Stage K (row-major) and V (transposed) of this chunk.
        constexpr int V4_PER_ROW = D / 4;
        for (int f = tid; f < KCHUNK * V4_PER_ROW; f += 256) {
            const int r = f / V4_PER_ROW, c4 = f % V4_PER_ROW;
            const int key = key0 + r;
            f32x4 kv = f32x4{0.f, 0.f, 0.f, 0.f}, vv = kv;
            if (key < seq) {
                kv = *reinterpret_cast<const f32x4*>(k_base + key * row_stride + c4 * 4);
                vv = *reinterpret_cast<const f32x4*>(v_base + key * row_stride + c4 * 4);
            }
            *reinterpret_cast<f32x4*>(sK + r * SM::K_STRIDE + c4 * 4) = kv;
#pragma unroll
            for (int c = 0; c < 4; ++c) sVt[(c4 * 4 + c) * SM::VT_STRIDE + r] = vv[c];
        }
        int plain = 1;
        if (tid < KCHUNK) {
            const int key = key0 + tid;
            float mv = -1.0f;
            if (key < seq) mv = (mask == nullptr || mask[b * seq + key] != 0u) ? 1.0f : 0.0f;
            sMask[tid] = mv;
            plain = mv == 1.0f;
        }
        // also the staging barrier; all_plain: every key of the chunk exists and is kept, so the mask pass can be skipped
        const int all_plain = __syncthreads_and(plain);

        // S^T tiles: 4 key tiles x 32 queries, K = D.
        f32x16 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.0f;
            const float* pk = sK + (kt * 32 + l31) * SM::K_STRIDE + half * 4;
#pragma unroll
            for (int kk = 0; kk < D / 8; ++kk) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(pk + kk * 8);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c], qf[kk][c], s[kt], 0, 0, 0);
            }
        }

        // scale -> mask overwrite; keys beyond seq contribute exact zeros.  The softmax runs in the exp2 domain as in the
        // pipelined kernel: t = score * (scale * log2 e), p = exp2(t - max t), one v_exp_f32 per element (libm's expf is
        // eight instructions; the VALU work of this phase is what the matrix pipe waits for).
        const float c1 = scale * 1.4426950408889634f;
        const float masked_t = mask_value * 1.4426950408889634f;  // -1e9 -> -1.44e9, -inf -> -inf
        float cmax = -INFINITY;
        if (all_plain) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[kt][r] *= c1;
                    cmax = fmaxf(cmax, s[kt][r]);
                }
        } else {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float mk = sMask[kt * 32 + acc_row(r, half)];
                    float v = s[kt][r] * c1;
                    v = (mk == 0.0f) ? masked_t : v;
                    v = (mk < 0.0f) ? -INFINITY : v;
                    s[kt][r] = v;
                    cmax = fmaxf(cmax, v);
                }
        }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32, kWave));

        float alpha = 1.0f;
        float new_max = cmax;
        if (n_chunks > 1) {
            new_max = fmaxf(run_max, cmax);
            // exp2(-inf - finite) = 0 on the first chunk; guard -inf - -inf.
            alpha = (run_max == -INFINITY) ? 0.0f : __builtin_amdgcn_exp2f(run_max - new_max);
        }
        float csum = 0.0f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // One chunk: max, exp, sum, scale as the reference; a row that is entirely -inf gives NaN like the
                // reference's no-alloc path.  Several chunks: a chunk that is entirely -inf adds zeros.
                float e = __builtin_amdgcn_exp2f(s[kt][r] - new_max);
                if (n_chunks > 1 && new_max == -INFINITY) e = 0.0f;
                s[kt][r] = e;
                csum += e;
            }
        csum += __shfl_xor(csum, 32, kWave);

        if (n_chunks == 1) {
            // activations.rs:236-241: scale by 1/sum only when sum > 0
            if (csum > 0.0f) {
                const float inv = 1.0f / csum;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kt][r] *= inv;
            }
        } else {
            run_sum = run_sum * alpha + csum;
            run_max = new_max;
            // Rescale O: its rows are queries indexed by (reg, half); alpha lives on lane == query.
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a = __shfl(alpha, acc_row(r, half), kWave);
#pragma unroll
                for (int dt = 0; dt < D / 32; ++dt) o[dt][r] *= a;
            }
        }

        // O += P V.  A = P (this lane: query l31, k index = half <-> key 8g+4half+c),
        // B = V[key][d]: lane supplies V^T[d = dt*32 + l31][key], 4 consecutive keys per read.
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) {
            const float* pv = sVt + (dt * 32 + l31) * SM::VT_STRIDE + half * 4;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 vf = *reinterpret_cast<const f32x4*>(pv + kt * 32 + g * 8);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[kt][g * 4 + c], vf[c], o[dt], 0, 0, 0);
                }
        }
    }

    // Store: accumulator row = query (reg, half), column = d (lane&31): 128-byte segments.
    float inv_sum[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) inv_sum[r] = 1.0f;
    if (n_chunks > 1) {
        float inv = (run_sum > 0.0f) ? 1.0f / run_sum : 1.0f;
        if (run_max == -INFINITY) inv = __builtin_nanf("");  // every key -inf: NaN row, as the reference
#pragma unroll
        for (int r = 0; r < 16; ++r) inv_sum[r] = __shfl(inv, acc_row(r, half), kWave);
    }
    float* out_base = ctx + b * seq * (int64_t)hidden + h * D;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int q = qb * QBLK + wid * 32 + acc_row(r, half);
        if (q < seq) {
#pragma unroll
            for (int dt = 0; dt < D / 32; ++dt)
                out_base[(int64_t)q * hidden + dt * 32 + l31] = o[dt][r] * inv_sum[r];
        }
    }
}

// ---------------------------------------------------------------------------
// Persistent, software-pipelined form for seq <= 128 (the benchmark shape and
// the common case: the reference truncates to 512 tokens but sentence batches
// are short).  Same arithmetic as attention_kernel's single-chunk path.
//
// A workgroup walks (sentence, head) items; while it computes item i from LDS
// the K/V rows and Q fragments of item i+1 are already in flight into
// registers, so the global-memory latency that attention_kernel exposes once
// per workgroup is hidden behind 128 MFMAs + the softmax.  The padding mask of
// an item is two 64-bit ballots per wave (bit k = key k kept), tested with
// constant bit positions; an item without padding skips masking altogether.
// ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256, 2) void attention_pipe_kernel(const float* __restrict__ qkv,
                                                                const uint32_t* __restrict__ mask,
                                                                int64_t n_items, int seq, int heads,
                                                                float scale, float mask_value,
                                                                float* __restrict__ ctx)
{
    using SM = AttnSmem<D>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sK = smem;
    float* sVt = smem + SM::K_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int hidden = heads * D;
    const int64_t row_stride = 3 * (int64_t)hidden;
    constexpr int V4_PER_ROW = D / 4;
    constexpr int STAGE_ITERS = KCHUNK * V4_PER_ROW / 256;

    f32x4 kreg[STAGE_ITERS], vreg[STAGE_ITERS], qf[D / 8];
    unsigned long long keep_lo = ~0ull, keep_hi = ~0ull;
    const int q_row = wid * 32 + l31;

    auto prefetch_kv = [&](int64_t it_) {
        const int64_t b = it_ / heads;
        const int h = (int)(it_ % heads);
        const float* base = qkv + b * seq * row_stride + h * D;
#pragma unroll
        for (int it = 0; it < STAGE_ITERS; ++it) {
            const int f = tid + it * 256;
            const int r = f / V4_PER_ROW, c4 = f % V4_PER_ROW;
            kreg[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            vreg[it] = kreg[it];
            if (r < seq) {
                kreg[it] = *reinterpret_cast<const f32x4*>(base + hidden + r * row_stride + c4 * 4);
                vreg[it] = *reinterpret_cast<const f32x4*>(base + 2 * hidden + r * row_stride + c4 * 4);
            }
        }
    };
    // Q fragments (B operand of S^T = K Q^T) and the keep-bits of keys 0..63 / 64..127.
    auto prefetch_q = [&](int64_t it_) {
        const int64_t b = it_ / heads;
        const int h = (int)(it_ % heads);
        const float* base = qkv + b * seq * row_stride + h * D;
#pragma unroll
        for (int kk = 0; kk < D / 8; ++kk) {
            qf[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (q_row < seq) qf[kk] = *reinterpret_cast<const f32x4*>(base + q_row * row_stride + kk * 8 + half * 4);
        }
        bool k0 = lane < seq, k1 = lane + 64 < seq;
        if (mask != nullptr) {
            if (k0) k0 = mask[b * seq + lane] != 0u;
            if (k1) k1 = mask[b * seq + lane + 64] != 0u;
        }
        keep_lo = __ballot(k0);
        keep_hi = __ballot(k1);
    };

    int64_t item = blockIdx.x;
    if (item < n_items) {
        prefetch_kv(item);
        prefetch_q(item);
    }

    for (; item < n_items; item += gridDim.x) {
        // registers -> LDS (K row-major, V transposed)
#pragma unroll
        for (int it = 0; it < STAGE_ITERS; ++it) {
            const int f = tid + it * 256;
            const int r = f / V4_PER_ROW, c4 = f % V4_PER_ROW;
            *reinterpret_cast<f32x4*>(sK + r * SM::K_STRIDE + c4 * 4) = kreg[it];
#pragma unroll
            for (int c = 0; c < 4; ++c) sVt[(c4 * 4 + c) * SM::VT_STRIDE + r] = vreg[it][c];
        }
        const int64_t b = item / heads;
        const int h = (int)(item % heads);
        const int64_t next = item + gridDim.x;
        __syncthreads();
        if (next < n_items) prefetch_kv(next);  // in flight during this item's MFMAs + softmax

        f32x16 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.0f;
            const float* pk = sK + (kt * 32 + l31) * SM::K_STRIDE + half * 4;
#pragma unroll
            for (int kk = 0; kk < D / 8; ++kk) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(pk + kk * 8);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c], qf[kk][c], s[kt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep fragment live ranges short (register budget)
        }

        // Q fragments and mask bits of this item are consumed: fetch the next item's.
        const unsigned long long cur_lo = keep_lo, cur_hi = keep_hi;
        if (next < n_items) prefetch_q(next);

        // scale (after the dot product, as the reference) then mask overwrite.
        const unsigned long long valid_lo = seq >= 64 ? ~0ull : ((1ull << seq) - 1ull);
        const unsigned long long valid_hi = seq >= 128 ? ~0ull : (seq > 64 ? ((1ull << (seq - 64)) - 1ull) : 0ull);
        const bool no_mask = (cur_lo == ~0ull) && (cur_hi == ~0ull);  // wave-uniform
        // Softmax in the exp2 domain: t = score * log2(e), p = exp2(t - max t); one v_exp_f32 per
        // element.  c1 folds the reference's 1/sqrt(d) scaling (applied after the dot product) with
        // log2(e); a masked key's score is overwritten by mask_value (times log2 e).
        const float c1 = scale * 1.4426950408889634f;
        const float masked_t = mask_value * 1.4426950408889634f;  // -1e9 -> -1.44e9, -inf -> -inf
        float cmax = -INFINITY;
        if (no_mask) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[kt][r] *= c1;
                    cmax = fmaxf(cmax, s[kt][r]);
                }
        } else {
            // this lane's keys are bit (kt*32 + (r&3) + 8*(r>>2)) + 4*half of the 128-bit sets
            const unsigned sh = 4u * (unsigned)half;
            const unsigned keep_w[4] = {(unsigned)(cur_lo >> sh), (unsigned)(cur_lo >> (32 + sh)),
                                        (unsigned)(cur_hi >> sh), (unsigned)(cur_hi >> (32 + sh))};
            const unsigned val_w[4] = {(unsigned)(valid_lo >> sh), (unsigned)(valid_lo >> (32 + sh)),
                                       (unsigned)(valid_hi >> sh), (unsigned)(valid_hi >> (32 + sh))};
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned bit = 1u << ((r & 3) + 8 * (r >> 2));
                    float v = s[kt][r] * c1;
                    v = (keep_w[kt] & bit) ? v : masked_t;     // masked key: score overwritten
                    v = (val_w[kt] & bit) ? v : -INFINITY;     // key beyond seq: contributes exactly 0
                    s[kt][r] = v;
                    cmax = fmaxf(cmax, v);
                }
        }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32, kWave));
        float csum = 0.0f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(s[kt][r] - cmax);  // NaN for an all -inf row, as the reference
                s[kt][r] = e;
                csum += e;
            }
        csum += __shfl_xor(csum, 32, kWave);
        if (csum > 0.0f) {  // activations.rs:236-241
            const float inv = 1.0f / csum;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kt][r] *= inv;
        }

        f32x16 o[D / 32];
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.0f;
            const float* pv = sVt + (dt * 32 + l31) * SM::VT_STRIDE + half * 4;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 vf = *reinterpret_cast<const f32x4*>(pv + kt * 32 + g * 8);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[kt][g * 4 + c], vf[c], o[dt], 0, 0, 0);
                    if (g == 1 || g == 3) __builtin_amdgcn_sched_barrier(0);
                }
        }

        float* out_base = ctx + b * seq * (int64_t)hidden + h * D;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = wid * 32 + acc_row(r, half);
            if (q < seq) {
#pragma unroll
                for (int dt = 0; dt < D / 32; ++dt) out_base[(int64_t)q * hidden + dt * 32 + l31] = o[dt][r];
            }
        }
        __syncthreads();  // everyone is done with sK / sVt before the next item overwrites them
    }
}

// Any-head-dim fallback (head_dim not 32/64, e.g. toy models in tests): one
// wave per (sentence, head, query); scores for the row go through LDS.
__global__ __launch_bounds__(64) void attention_generic_kernel(const float* __restrict__ qkv,
                                                               const uint32_t* __restrict__ mask,
                                                               int seq, int heads, int head_dim,
                                                               float scale, float mask_value,
                                                               float* __restrict__ ctx)
{
    extern __shared__ float srow[];  // [seq]
    const int lane = threadIdx.x;
    const int q = blockIdx.x, h = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int hidden = heads * head_dim;
    const int64_t rs = 3 * (int64_t)hidden;
    const float* base = qkv + b * seq * rs;
    const float* qv = base + q * rs + h * head_dim;
    float mx = -INFINITY;
    for (int j = lane; j < seq; j += 64) {
        const float* kv = base + j * rs + hidden + h * head_dim;
        float s = 0.0f;
        for (int d = 0; d < head_dim; ++d) s = fmaf(qv[d], kv[d], s);
        s *= scale;
        if (mask && mask[b * seq + j] == 0u) s = mask_value;
        srow[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.0f;
    for (int j = lane; j < seq; j += 64) {
        const float e = expf(srow[j] - mx);
        srow[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = (sum > 0.0f) ? 1.0f / sum : 1.0f;
    __syncthreads();
    for (int d = lane; d < head_dim; d += 64) {
        float acc = 0.0f;
        for (int j = 0; j < seq; ++j)
            acc = fmaf(srow[j] * inv, base[j * rs + 2 * hidden + h * head_dim + d], acc);
        ctx[(b * seq + q) * (int64_t)hidden + h * head_dim + d] = acc;
    }
}

template <int D>
hipError_t launch_d(const float* qkv, const uint32_t* mask, int64_t batch, int seq, int heads,
                    float mask_value, float* ctx, hipStream_t stream)
{
    using SM = AttnSmem<D>;
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (SM::BYTES > 64 * 1024 && !attr_set[dev & 63]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<D>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, SM::BYTES);
        if (e != hipSuccess) return e;
        attr_set[dev & 63] = true;
    }
    const float scale = 1.0f / sqrtf((float)D);  // encoder_self_attention.rs:43
    if (D == 32 && seq <= KCHUNK && g_attention_variant == 0) {
        static bool pipe_attr[64] = {};
        if (SM::BYTES > 64 * 1024 && !pipe_attr[dev & 63]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_pipe_kernel<D>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, SM::BYTES);
            if (e != hipSuccess) return e;
            pipe_attr[dev & 63] = true;
        }
        const int64_t n_items = batch * heads;
        const int64_t max_blocks = 256 * 2;  // two resident workgroups per CU
        const unsigned grid = (unsigned)(n_items < max_blocks ? n_items : max_blocks);
        hipLaunchKernelGGL(attention_pipe_kernel<D>, dim3(grid), dim3(256), SM::BYTES, stream, qkv, mask, n_items,
                           seq, heads, scale, mask_value, ctx);
        return hipGetLastError();
    }
    // grid.z is limited to 65535 sentences per launch.
    for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
        const int64_t nb = (batch - b0 < 65535) ? (batch - b0) : 65535;
        dim3 grid((unsigned)((seq + QBLK - 1) / QBLK), (unsigned)heads, (unsigned)nb);
        hipLaunchKernelGGL(attention_kernel<D>, grid, dim3(256), SM::BYTES, stream,
                           qkv + b0 * seq * 3 * (int64_t)heads * D, mask ? mask + b0 * seq : nullptr,
                           seq, heads, scale, mask_value, ctx + b0 * seq * (int64_t)heads * D);
    }
    return hipGetLastError();
}

}  // namespace

#ifdef KJARNI_TUNING
void set_attention_variant(int v) { g_attention_variant = v; }
#endif

hipError_t launch_attention(const float* qkv, const uint32_t* mask, int64_t batch, int seq, int heads,
                            int head_dim, float mask_value, float* ctx, hipStream_t stream)
{
    if (batch <= 0 || seq <= 0) return hipSuccess;
    const bool aligned = ((heads * head_dim) % 4 == 0) && ((reinterpret_cast<uintptr_t>(qkv) & 15) == 0);
    if (head_dim == 32 && aligned) return launch_d<32>(qkv, mask, batch, seq, heads, mask_value, ctx, stream);
    if (head_dim == 64 && aligned) return launch_d<64>(qkv, mask, batch, seq, heads, mask_value, ctx, stream);
    const float scale = 1.0f / sqrtf((float)head_dim);
    for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
        const int64_t nb = (batch - b0 < 65535) ? (batch - b0) : 65535;
        dim3 grid((unsigned)seq, (unsigned)heads, (unsigned)nb);
        hipLaunchKernelGGL(attention_generic_kernel, grid, dim3(64), seq * sizeof(float), stream,
                           qkv + b0 * seq * 3 * (int64_t)heads * head_dim,
                           mask ? mask + b0 * seq : nullptr, seq, heads, head_dim, scale, mask_value,
                           ctx + b0 * seq * (int64_t)heads * head_dim);
    }
    return hipGetLastError();
}

}  // namespace kjarni
