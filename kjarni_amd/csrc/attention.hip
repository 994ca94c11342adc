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
#include "tuning.h"

namespace kjarni {

namespace {

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
                                                        const int32_t* __restrict__ cu, float* __restrict__ ctx)
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
    // packed rows: the sentence is rows cu[b] .. cu[b+1], all of them kept tokens; the grid covers the longest sentence
    int64_t row0 = b * seq;
    if (cu) {
        row0 = cu[b];
        seq = cu[b + 1] - cu[b];
        mask = nullptr;
        if (qb * QBLK >= seq) return;
    }
    const float* base = qkv + row0 * row_stride;
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
            if (key < seq) mv = (mask == nullptr || mask[row0 + key] != 0u) ? 1.0f : 0.0f;
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
    float* out_base = ctx + row0 * (int64_t)hidden + h * D;
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
// DIAG (tuning build only, tools/gemm_probe.py attn): 0 the kernel; knock-outs that show where an item's time goes --
// 1 no softmax arithmetic, 2 no LDS staging of K / V after the first item, 3 no output stores, 4 no K / V / Q prefetch,
// 5 no matrix work (memory only), 6 non-temporal K / V / Q loads.
// VARLEN (packed rows of a ragged batch): item (b, h) is rows cu[b] .. cu[b+1], every key a kept token -- no mask; a wave
// whose 32 queries lie past the sentence's end, and 32-key tiles past it, are skipped (wave-uniform branches), so an
// item costs ceil(len / 32)^2 / 16 of a 128-token one.
// (head_dim 64: K + V^T of an item are 69 KB of LDS, so two workgroups fit a CU whatever the registers say -- at three the
// 168-register budget spilled 109 VGPRs into the item loop; rounds 1-3 ran BERT-base shaped models that way at 66 TFLOP/s)
// MODE 2 (padded rows, seq <= 96: one sentence, short batches): the mask as MODE 0, the 32-key tiles and waves past `seq` skipped
// as MODE 1 -- a 28-token item is 32 MFMAs on one wave instead of 256 on each of four.
template <int D, int DIAG, int MODE>
__global__ __launch_bounds__(256, D >= 64 ? 2 : 3) void attention_pipe_kernel(const float* __restrict__ qkv,
                                                                const uint32_t* __restrict__ mask,
                                                                int64_t n_items, int seq, int heads,
                                                                float scale, float mask_value,
                                                                const int32_t* __restrict__ cu,
                                                                float* __restrict__ ctx)
{
    using SM = AttnSmem<D>;
    constexpr bool VARLEN = MODE == 1, TRIM = MODE != 0;
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

    // All global traffic of an item goes through buffer descriptors built from scalars (sentence, head): constant
    // per-lane offsets, no 64-bit vector address arithmetic and no per-lane predication in the item loop -- rows at
    // or past `seq` fall outside the descriptor and read as zeros / are not written (hardware bounds check).  The
    // vector ALU is the resource the f32 MFMAs run on here, so address math there is matrix time lost.
    // Staging map (key row, 16-byte column) of piece `it` of this thread, chosen for the LDS: a half-wave holds 16 consecutive
    // rows x two neighbouring columns.  V goes into LDS TRANSPOSED by 4-byte stores at [4 col + c][row] with a row stride of
    // 132 floats, i.e. bank (16 col + 4 c + row) mod 32: 16 rows x (an even and an odd column) are 32 different banks.  With the
    // row-major map (8 or 16 lanes along a row) the half-wave's eight columns fell on two banks per row -- 4-way conflicts on
    // every one of the 16 transposing stores per item, a third of the kernel's LDS cycles (SQ_LDS_BANK_CONFLICT 9.4e6 of
    // SQ_LDS_IDX_ACTIVE 2.9e7, profiles/r04m_pmc_pipeline_summary.txt).  K's 16-byte stores (8 lanes = 8 consecutive rows of one
    // column: banks 4 (row + col) .. + 3) and every fragment read stay conflict-free; a global load still covers 64
    // contiguous bytes of a row (lanes l, l + 16, l + 32, l + 48).
    constexpr int COL_QUADS = V4_PER_ROW / 4;  // groups of four 16-byte columns per row
    auto stage_row = [&](int it) { return 64 * (it / COL_QUADS) + 16 * wid + (lane & 15); };
    auto stage_col = [&](int it) { return 4 * (it % COL_QUADS) + 2 * (lane >> 5) + ((lane >> 4) & 1); };
    uint32_t off_kv[STAGE_ITERS];
#pragma unroll
    for (int it = 0; it < STAGE_ITERS; ++it)
        off_kv[it] = (uint32_t)(((int64_t)stage_row(it) * row_stride + stage_col(it) * 4) * 4);
    const uint32_t off_q = (uint32_t)(((int64_t)q_row * row_stride + half * 4) * 4);
    const uint32_t off_o = (uint32_t)(((int64_t)(wid * 32 + 4 * half) * hidden + l31) * 4);
    auto span_in_of = [&](int len) { return (int)((((int64_t)len - 1) * row_stride + D) * 4); };  // bytes of one head's rows of Q, K or V
    auto span_out_of = [&](int len) { return (int)((((int64_t)len - 1) * hidden + D) * 4); };
    // first row and length of an item's sentence (scalars)
    auto item_rows = [&](int b, int64_t& row0, int& len) {
        if (VARLEN) {
            const int c0 = __builtin_amdgcn_readfirstlane(cu[b]), c1 = __builtin_amdgcn_readfirstlane(cu[b + 1]);
            row0 = c0;
            len = c1 - c0;
        } else {
            row0 = (int64_t)b * seq;
            len = seq;
        }
    };
    auto rsrc = [](const float* p, int bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000);
    };
    auto ld16 = [](__amdgpu_buffer_rsrc_t r, uint32_t off, int soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, DIAG == 6 ? 2 : 0));
    };
    // (sentence, head) of an item advance with the grid stride without divisions
    const int step_b = (int)(gridDim.x / (unsigned)heads), step_h = (int)(gridDim.x % (unsigned)heads);
    // A workgroup per item (calls of up to a few hundred sentences): every XCD takes one contiguous run of items -- of sentences,
    // the rows the QKV projection's tiles left in this XCD's L2 and the rows the output projection's tiles will read here.
    const unsigned first = gridDim.x == n_items ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
    int cur_b = (int)(first / (unsigned)heads), cur_h = (int)(first % (unsigned)heads);
    auto advance = [&](int& bb, int& hh) {
        bb += step_b;
        hh += step_h;
        if (hh >= heads) {
            hh -= heads;
            ++bb;
        }
    };

    int64_t pre_row0 = 0;  // rows of the item whose K / V / Q were requested last
    int pre_len = seq;
    // live = false (a workgroup's last item has no successor): the same requests through descriptors without extent -- no
    // memory traffic, nothing to wait for at the end, and still no branch around a memory operation in the item loop
    auto prefetch_kv = [&](int b, int h, bool live) {
        item_rows(b, pre_row0, pre_len);
        const float* base = qkv + pre_row0 * row_stride + h * D;
        const int span_in = live ? span_in_of(pre_len) : 0;
        const __amdgpu_buffer_rsrc_t rk = rsrc(base + hidden, span_in), rv = rsrc(base + 2 * hidden, span_in);
#pragma unroll
        for (int it = 0; it < STAGE_ITERS; ++it) {
            kreg[it] = ld16(rk, off_kv[it], 0);
            vreg[it] = ld16(rv, off_kv[it], 0);
        }
    };
    // Q fragments (B operand of S^T = K Q^T) and the keep-bits of keys 0..63 / 64..127.
    // The mask words of an item's sentence (keys lane and lane + 64) are REQUESTED with its K / V rows, a whole item ahead, and
    // turned into the two 64-bit keep sets only at the end of the iteration: a ballot right behind the request would wait
    // for it -- and, the vector-memory counter being in order, for the K / V / Q prefetch in front of it -- in the middle
    // of the item (round 3: that wait sat between the score tiles and the softmax of every item of a masked call).
    uint32_t mask_w0 = 1u, mask_w1 = 1u;
    auto request_mask = [&](int b) {
        if (VARLEN || mask == nullptr) return;
        // rows past `seq` are never requested: their bits are cleared by the lane tests at the ballot
        const int64_t base = (int64_t)b * seq;
        mask_w0 = lane < seq ? __builtin_nontemporal_load(mask + base + lane) : 0u;
        mask_w1 = lane + 64 < seq ? __builtin_nontemporal_load(mask + base + lane + 64) : 0u;
    };
    auto ballot_mask = [&]() {
        if (VARLEN) return;
        keep_lo = __ballot(lane < seq && mask_w0 != 0u);
        keep_hi = __ballot(lane + 64 < seq && mask_w1 != 0u);
    };
    auto prefetch_q = [&](int b, int h, bool live) {  // (after prefetch_kv of the same item: pre_row0 / pre_len are its rows)
        const __amdgpu_buffer_rsrc_t rq = rsrc(qkv + pre_row0 * row_stride + h * D, live ? span_in_of(pre_len) : 0);
#pragma unroll
        for (int kk = 0; kk < D / 8; ++kk) qf[kk] = ld16(rq, off_q, kk * 32);
    };

    int64_t item = first;
    if (item < n_items) {
        prefetch_kv(cur_b, cur_h, true);
        request_mask(cur_b);
        prefetch_q(cur_b, cur_h, true);
        ballot_mask();
    }
    int64_t cur_row0 = pre_row0;  // rows of the item the loop body computes
    int cur_len = pre_len;

    // The finished item's output tile waits in registers and is stored right after the NEXT item's staging
    // barrier: its stores are then older than that item's prefetch loads, so the (conservative) wait for those
    // loads at the top of the following iteration no longer sits on a store issued a moment ago.
    f32x16 o_pend[D / 32];
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_pend[dt][r] = 0.0f;
    const float* pend_base = nullptr;
    int pend_len = seq;
    // (no branch around the stores: before the first item the descriptor has no extent and the hardware drops them.  The item
    // loop is kept free of data-dependent branches around memory operations on purpose -- the compiler counts outstanding
    // vector-memory operations exactly only along straight-line code; at a join it falls back to "wait for everything",
    // which made every item wait for the NEXT item's K / V prefetch before its score tiles were done, rounds 1-2.)
    auto flush = [&]() {
        const __amdgpu_buffer_rsrc_t ro = rsrc(pend_base != nullptr ? pend_base : ctx, pend_base != nullptr ? span_out_of(pend_len) : 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (DIAG != 3 || o_pend[0][r] == 123456.789f) {
#pragma unroll
                for (int dt = 0; dt < D / 32; ++dt) {
                    const float val = o_pend[dt][r];  // (bit_cast straight on the vector-element lvalue reads element 0)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), ro, off_o,
                                                          (((r & 3) + 8 * (r >> 2)) * hidden + dt * 32) * 4, 0);
                }
            }
        }
    };

    for (; item < n_items; item += gridDim.x) {
        // registers -> LDS (K row-major, V transposed)
        if (DIAG != 2 || item == (int64_t)first)
#pragma unroll
        for (int it = 0; it < STAGE_ITERS; ++it) {
            const int r = stage_row(it), c4 = stage_col(it);
            *reinterpret_cast<f32x4*>(sK + r * SM::K_STRIDE + c4 * 4) = kreg[it];
#pragma unroll
            for (int c = 0; c < 4; ++c) sVt[(c4 * 4 + c) * SM::VT_STRIDE + r] = vreg[it][c];
        }
        const int h = cur_h;
        const int64_t next = item + gridDim.x;
        int nb = cur_b, nh = cur_h;
        advance(nb, nh);
        const bool has_next = next < n_items;
        if (!has_next) {  // the last item of this workgroup "requests" itself (VARLEN: cu[b + 1] stays in range) -- without extent
            nb = cur_b;
            nh = cur_h;
        }
        __syncthreads();
        flush();                                               // the previous item's outputs
        if (DIAG != 4) {  // in flight during this item's MFMAs + softmax
            prefetch_kv(nb, nh, has_next);
            request_mask(nb);
        }

        const int len = cur_len;                                       // (VARLEN: this sentence's; else seq)
        const int nkt = TRIM ? ((len + 31) >> 5) : 4;                  // 32-key tiles that hold keys
        const bool wave_on = !TRIM || wid * 32 < len;                  // wave-uniform: do this wave's queries exist
        f32x16 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (TRIM && !(wave_on && kt < nkt)) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.0f;
            const float* pk = sK + (kt * 32 + l31) * SM::K_STRIDE + half * 4;
#pragma unroll
            for (int kk = 0; kk < D / 8; ++kk) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(pk + kk * 8);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (DIAG == 5) {
                        s[kt][c] += kf[c] * qf[kk][c];  // memory-only diagnostic: no matrix work
                    } else {
                        s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c], qf[kk][c], s[kt], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep fragment live ranges short (register budget)
        }

        // Q fragments and mask bits of this item are consumed: fetch the next item's.
        const unsigned long long cur_lo = keep_lo, cur_hi = keep_hi;
        if (DIAG != 4) prefetch_q(nb, nh, has_next);

        // scale (after the dot product, as the reference) then mask overwrite.
        const unsigned long long valid_lo = len >= 64 ? ~0ull : ((1ull << len) - 1ull);
        const unsigned long long valid_hi = len >= 128 ? ~0ull : (len > 64 ? ((1ull << (len - 64)) - 1ull) : 0ull);
        const bool no_mask = VARLEN ? (len & 31) == 0 : (cur_lo == ~0ull) && (cur_hi == ~0ull);  // wave-uniform
        // Softmax in the exp2 domain with as few vector-ALU instructions as it takes -- on this part the f32 MFMAs and
        // the VALU share the FP32 lanes, so every VALU instruction here is time the matrix pipe idles:
        //   max over the RAW dot products (c1 > 0, so max(c1 s) = c1 max(s); v_max3: two elements per instruction),
        //   e = exp2(s c1 - max c1) as ONE fma + one v_exp_f32 per element (fma on packed pairs),
        //   p = e (1/sum) on packed pairs (the query of a score is its LANE, of an output element its REGISTER, so
        //   the scaling stays on P).
        // c1 folds the reference's 1/sqrt(d) (applied after the dot product) with log2(e); a masked key's score is
        // overwritten by mask_value, here as mask_value / scale in the raw domain (-1e9 stays ~-1e9 log2 e after the
        // fma, -inf stays -inf).
        const float c1 = scale * 1.4426950408889634f;
        const float masked_raw = mask_value / scale;
        if (DIAG != 1 && DIAG != 5 && wave_on) {
        if (!no_mask) {
            // this lane's keys are bit (kt*32 + (r&3) + 8*(r>>2)) + 4*half of the 128-bit sets
            const unsigned sh = 4u * (unsigned)half;
            const unsigned keep_w[4] = {(unsigned)(cur_lo >> sh), (unsigned)(cur_lo >> (32 + sh)),
                                        (unsigned)(cur_hi >> sh), (unsigned)(cur_hi >> (32 + sh))};
            const unsigned val_w[4] = {(unsigned)(valid_lo >> sh), (unsigned)(valid_lo >> (32 + sh)),
                                       (unsigned)(valid_hi >> sh), (unsigned)(valid_hi >> (32 + sh))};
            const bool all_valid = len >= KCHUNK;  // wave-uniform
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (VARLEN ? kt != nkt - 1 : (TRIM && kt >= nkt)) continue;  // packed rows: only the last tile that holds keys is partial
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned bit = 1u << ((r & 3) + 8 * (r >> 2));
                    float v = (VARLEN || (keep_w[kt] & bit)) ? s[kt][r] : masked_raw;  // masked key: score overwritten
                    if (!all_valid) v = (val_w[kt] & bit) ? v : -INFINITY;  // key beyond seq: contributes exactly 0
                    s[kt][r] = v;
                }
            }
        }
        float cmax = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (TRIM && kt >= nkt) continue;
#pragma unroll
            for (int r = 0; r < 16; r += 2) cmax = fmaxf(fmaxf(cmax, s[kt][r]), s[kt][r + 1]);
        }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32, kWave));
        const float neg = -cmax * c1;  // (+inf for an all -inf row: exp2(-inf + inf) = NaN, as the reference)
        const f32x2 c1v = {c1, c1}, negv = {neg, neg};
        f32x2 sum2 = {0.0f, 0.0f};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (TRIM && kt >= nkt) continue;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 t = __builtin_elementwise_fma(f32x2{s[kt][r], s[kt][r + 1]}, c1v, negv);
                const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                s[kt][r] = e[0];
                s[kt][r + 1] = e[1];
                sum2 += e;
            }
        }
        float csum = sum2[0] + sum2[1];
        csum += __shfl_xor(csum, 32, kWave);
        if (csum > 0.0f) {  // activations.rs:236-241
            const float inv = 1.0f / csum;
            const f32x2 invv = {inv, inv};
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (TRIM && kt >= nkt) continue;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 pr = f32x2{s[kt][r], s[kt][r + 1]} * invv;
                    s[kt][r] = pr[0];
                    s[kt][r + 1] = pr[1];
                }
            }
        }
        }

        f32x16 o[D / 32];
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.0f;
            const float* pv = sVt + (dt * 32 + l31) * SM::VT_STRIDE + half * 4;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (TRIM && !(wave_on && kt < nkt)) continue;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 vf = *reinterpret_cast<const f32x4*>(pv + kt * 32 + g * 8);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (DIAG == 5) {
                            o[dt][c] += s[kt][g * 4 + c] * vf[c];
                        } else {
                            o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[kt][g * 4 + c], vf[c], o[dt], 0, 0, 0);
                        }
                    }
                    if (g == 1 || g == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

        pend_base = ctx + cur_row0 * hidden + h * D;
        pend_len = len;
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) o_pend[dt] = o[dt];
        if (DIAG != 4) ballot_mask();  // the next item's keep sets, from words requested an item ago
        cur_b = nb;
        cur_h = nh;
        cur_row0 = pre_row0;
        cur_len = pre_len;
        __syncthreads();  // everyone is done with sK / sVt before the next item overwrites them
    }
    flush();
}

// ---------------------------------------------------------------------------
// Small calls (one sentence, a handful: at most kSplitMaxItems (sentence, head) items), S <= 128.  The kernels above give an
// item to ONE workgroup, whose 4 x 256 MFMAs run on one CU: 6.8 us of matrix time for a 128-token item while 244 CUs idle --
// a one-sentence forward is a chain of dependent launches, so that is 6.8 us of latency per layer.  Here an item is up to
// four workgroups (one per block of 32 queries) and a workgroup's four waves take one 32-key tile each: 16 + 16 MFMAs per
// wave for d = 32.  A wave's K tile and its V rows are the MFMA operands as they lie in memory (K rows as 16-byte pieces, V
// rows as 128-byte row segments: a key's d values are the B operand's 32 lanes), so nothing is staged; the waves meet in LDS
// three times -- the per-query maxima of the four key tiles, the sums of exp, the four partial output tiles -- and the
// normalisation by 1 / sum happens on the way out.  Padded rows and packed rows (cu) alike; the mask semantics are the
// other kernels' (masked key: score overwritten by mask_value; key beyond the sentence: contributes exactly 0).
// ---------------------------------------------------------------------------
constexpr int kSplitMaxItems = 128;  // (96 items: -1.5 % per call against the item-per-workgroup kernel; 240: +2.5 %)

template <int D>
__global__ __launch_bounds__(256) void attention_split_kernel(const float* __restrict__ qkv, const uint32_t* __restrict__ mask, int seq,
                                                              int heads, float scale, float mask_value,
                                                              const int32_t* __restrict__ cu, float* __restrict__ ctx)
{
    __shared__ float s_max[4][32], s_sum[4][32];
    __shared__ __attribute__((aligned(16))) float s_o[4][32][D + 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave's key tile
    const int l31 = lane & 31, half = lane >> 5;
    const int qb = blockIdx.x, h = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int hidden = heads * D;
    const int64_t row_stride = 3 * (int64_t)hidden;
    int64_t row0 = b * seq;
    int len = seq;
    if (cu) {
        row0 = cu[b];
        len = cu[b + 1] - cu[b];
        mask = nullptr;
    }
    if (qb * 32 >= len) return;  // (the grid covers the longest sentence)
    const float* q_base = qkv + row0 * row_stride + h * D;
    const float* k_base = q_base + hidden;
    const float* v_base = q_base + 2 * hidden;
    const int key0 = wid * 32;
    const bool tile_on = key0 < len;  // wave-uniform: does this wave's key tile hold keys

    // Requests first: Q (B operand of S^T = K Q^T: query l31, k = 8 kk + 4 half + c), this wave's K rows (A operand: key l31, same k),
    // its V rows (B operand of O = P V: lane supplies V[key0 + 8 g + 4 half + c][32 dt + l31]) and its 32 mask words.
    const int q_row = qb * 32 + l31;
    const int k_row = key0 + l31;
    f32x4 qf[D / 8], kf[D / 8];
    float vf[D / 32][16];
    uint32_t mword = 1u;
#pragma unroll
    for (int kk = 0; kk < D / 8; ++kk) {
        qf[kk] = q_row < len ? *reinterpret_cast<const f32x4*>(q_base + q_row * row_stride + kk * 8 + half * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        kf[kk] = k_row < len ? *reinterpret_cast<const f32x4*>(k_base + k_row * row_stride + kk * 8 + half * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = key0 + 8 * (i >> 2) + 4 * half + (i & 3);
            vf[dt][i] = key < len ? v_base[key * row_stride + dt * 32 + l31] : 0.0f;
        }
    if (mask && k_row < len) mword = mask[b * seq + k_row];
    // this tile's keys: kept (mask != 0) and existing (< len), as 32-bit sets (both halves of the wave hold the same words)
    const unsigned keep = (unsigned)__ballot(mword != 0u && k_row < len);
    const unsigned exist = (unsigned)__ballot(k_row < len);

    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.0f;
    if (tile_on) {
#pragma unroll
        for (int kk = 0; kk < D / 8; ++kk)
#pragma unroll
            for (int c = 0; c < 4; ++c) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[kk][c], qf[kk][c], st, 0, 0, 0);
    }
    // scale after the dot product (as the reference), in the exp2 domain; masked key: overwritten; key beyond the sentence: -inf
    const float c1 = scale * 1.4426950408889634f;
    const float masked_raw = mask_value / scale;
    float cmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned bit = 1u << ((r & 3) + 8 * (r >> 2) + 4 * half);
        float v = (keep & bit) ? st[r] : masked_raw;
        v = (exist & bit) ? v : -INFINITY;
        st[r] = v;
        cmax = fmaxf(cmax, v);
    }
    cmax = fmaxf(cmax, __shfl_xor(cmax, 32, kWave));
    if (half == 0) s_max[wid][l31] = cmax;
    __syncthreads();
    const float qmax = fmaxf(fmaxf(s_max[0][l31], s_max[1][l31]), fmaxf(s_max[2][l31], s_max[3][l31]));
    const float neg = -qmax * c1;  // (+inf for a row whose every key is -inf: exp2(-inf + inf) = NaN, as the reference)
    float csum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(fmaf(st[r], c1, neg));
        st[r] = e;
        csum += e;
    }
    csum += __shfl_xor(csum, 32, kWave);
    if (half == 0) s_sum[wid][l31] = csum;

    // O_w = P_w V_w: A = P (query l31, k <-> key 8 g + c (+ 4 for the upper half) of MFMA (g, c)), B = the V values requested above
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt) {
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.0f;
        if (tile_on) {
#pragma unroll
            for (int i = 0; i < 16; ++i) o = __builtin_amdgcn_mfma_f32_32x32x2f32(st[i], vf[dt][i], o, 0, 0, 0);
        }
        // accumulator row = query (reg, half), column = d (l31)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_o[wid][acc_row(r, half)][dt * 32 + l31] = o[r];
    }
    __syncthreads();
    // 32 queries x D outputs: 16-byte pieces over the 256 threads, the four partial tiles added in tile order, then 1 / sum
    constexpr int PIECES = 32 * D / 4;
    for (int p = tid; p < PIECES; p += 256) {
        const int q = p / (D / 4), d4 = (p % (D / 4)) * 4;
        if (qb * 32 + q >= len) continue;
        f32x4 a = *reinterpret_cast<const f32x4*>(&s_o[0][q][d4]);
#pragma unroll
        for (int w = 1; w < 4; ++w) a += *reinterpret_cast<const f32x4*>(&s_o[w][q][d4]);
        const float total = (s_sum[0][q] + s_sum[1][q]) + (s_sum[2][q] + s_sum[3][q]);
        if (total > 0.0f) a = a * (1.0f / total);  // activations.rs:236-241: divide only when the sum is > 0
        *reinterpret_cast<f32x4*>(ctx + (row0 + qb * 32 + q) * hidden + h * D + d4) = a;
    }
}

// Any-head-dim fallback (head_dim not 32/64, e.g. toy models in tests): one
// wave per (sentence, head, query); scores for the row go through LDS.
__global__ __launch_bounds__(64) void attention_generic_kernel(const float* __restrict__ qkv,
                                                               const uint32_t* __restrict__ mask,
                                                               int seq, int heads, int head_dim,
                                                               float scale, float mask_value,
                                                               const int32_t* __restrict__ cu, float* __restrict__ ctx,
                                                               const float* __restrict__ pos_bias, int bias_seq)
{
    extern __shared__ float srow[];  // [seq]
    const int lane = threadIdx.x;
    const int q = blockIdx.x, h = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int hidden = heads * head_dim;
    const int64_t rs = 3 * (int64_t)hidden;
    int64_t row0 = b * seq;
    if (cu) {  // packed rows
        row0 = cu[b];
        seq = cu[b + 1] - cu[b];
        mask = nullptr;
        if (q >= seq) return;
    }
    const float* base = qkv + row0 * rs;
    const float* qv = base + q * rs + h * head_dim;
    float mx = -INFINITY;
    for (int j = lane; j < seq; j += 64) {
        const float* kv = base + j * rs + hidden + h * head_dim;
        float s = 0.0f;
        for (int d = 0; d < head_dim; ++d) s = fmaf(qv[d], kv[d], s);
        s *= scale;
        // additive position bias [heads, bias_seq, bias_seq], broadcast over sentences: after the scale, before the mask
        // (encoder_self_attention.rs:110-116, 250-257)
        if (pos_bias) s += pos_bias[((int64_t)h * bias_seq + q) * bias_seq + j];
        if (mask && mask[row0 + j] == 0u) s = mask_value;
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
        ctx[(row0 + q) * (int64_t)hidden + h * head_dim + d] = acc;
    }
}

template <int D>
hipError_t launch_d(const float* qkv, const uint32_t* mask, int64_t batch, int seq, int heads,
                    float mask_value, float* ctx, hipStream_t stream, const int32_t* cu)
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
    if (seq <= KCHUNK && batch * heads <= kSplitMaxItems && !tune::no_split_attention()) {
        // a few items: each over up to four workgroups, a key tile per wave (latency, not throughput)
        hipLaunchKernelGGL(attention_split_kernel<D>, dim3((unsigned)((seq + 31) / 32), (unsigned)heads, (unsigned)batch), dim3(256), 0, stream,
                           qkv, mask, seq, heads, scale, mask_value, cu, ctx);
        return hipGetLastError();
    }
    if (seq <= KCHUNK && !tune::no_pipelined_attention()) {
        const int64_t n_items = batch * heads;
        // head_dim 32: three resident workgroups per CU (35.8 KiB of LDS and <= 168 VGPRs each): the third hides what two
        // leave exposed of the K / V / Q streams' latency (measured 274 -> 264 us per launch of 12 288 items).
        // head_dim 64: 68.6 KiB of LDS per workgroup -- two per CU, and the kernel is compiled for that (<= 256 VGPRs).
        constexpr int kResident = D >= 64 ? 2 : 3;
        int64_t max_blocks = 256 * kResident;
        if (tune::attention_two_workgroups_per_cu()) max_blocks = 256 * 2;
        const unsigned grid = (unsigned)(n_items < max_blocks ? n_items : max_blocks);
        if (SM::BYTES > 64 * 1024) {  // the dynamic-LDS opt-in, once per device and kernel
            static bool pipe_attr_set[64] = {};
            if (!pipe_attr_set[dev & 63]) {
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_pipe_kernel<D, 0, 1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, SM::BYTES);
                if (e == hipSuccess)
                    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_pipe_kernel<D, 0, 0>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, SM::BYTES);
                if (e == hipSuccess)
                    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_pipe_kernel<D, 0, 2>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, SM::BYTES);
                if (e != hipSuccess) return e;
                pipe_attr_set[dev & 63] = true;
            }
        }
#ifdef KJARNI_TUNING
        switch (D == 32 ? tune::attention_knockout() + 10 : 0) {  // (the knock-outs are measured on the headline shape)
        case 11: hipLaunchKernelGGL((attention_pipe_kernel<D, 1, 0>), dim3(grid), dim3(256), SM::BYTES, stream, qkv, mask, n_items, seq, heads, scale, mask_value, nullptr, ctx); return hipGetLastError();
        case 12: hipLaunchKernelGGL((attention_pipe_kernel<D, 2, 0>), dim3(grid), dim3(256), SM::BYTES, stream, qkv, mask, n_items, seq, heads, scale, mask_value, nullptr, ctx); return hipGetLastError();
        case 13: hipLaunchKernelGGL((attention_pipe_kernel<D, 3, 0>), dim3(grid), dim3(256), SM::BYTES, stream, qkv, mask, n_items, seq, heads, scale, mask_value, nullptr, ctx); return hipGetLastError();
        case 16: hipLaunchKernelGGL((attention_pipe_kernel<D, 6, 0>), dim3(grid), dim3(256), SM::BYTES, stream, qkv, mask, n_items, seq, heads, scale, mask_value, nullptr, ctx); return hipGetLastError();
        case 15: hipLaunchKernelGGL((attention_pipe_kernel<D, 5, 0>), dim3(grid), dim3(256), SM::BYTES, stream, qkv, mask, n_items, seq, heads, scale, mask_value, nullptr, ctx); return hipGetLastError();
        case 14: hipLaunchKernelGGL((attention_pipe_kernel<D, 4, 0>), dim3(grid), dim3(256), SM::BYTES, stream, qkv, mask, n_items, seq, heads, scale, mask_value, nullptr, ctx); return hipGetLastError();
        default: break;
        }
#endif
        if (cu)
            hipLaunchKernelGGL((attention_pipe_kernel<D, 0, 1>), dim3(grid), dim3(256), SM::BYTES, stream, qkv, nullptr, n_items,
                               seq, heads, scale, mask_value, cu, ctx);
        else if (seq <= 96 && !tune::no_short_attention())  // at least one 32-key tile and one wave of every item are empty
            hipLaunchKernelGGL((attention_pipe_kernel<D, 0, 2>), dim3(grid), dim3(256), SM::BYTES, stream, qkv, mask, n_items,
                               seq, heads, scale, mask_value, nullptr, ctx);
        else
            hipLaunchKernelGGL((attention_pipe_kernel<D, 0, 0>), dim3(grid), dim3(256), SM::BYTES, stream, qkv, mask, n_items,
                               seq, heads, scale, mask_value, nullptr, ctx);
        return hipGetLastError();
    }
    // grid.z is limited to 65535 sentences per launch.
    for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
        const int64_t nb = (batch - b0 < 65535) ? (batch - b0) : 65535;
        dim3 grid((unsigned)((seq + QBLK - 1) / QBLK), (unsigned)heads, (unsigned)nb);
        if (cu)  // packed rows: absolute row offsets come from cu
            hipLaunchKernelGGL(attention_kernel<D>, grid, dim3(256), SM::BYTES, stream, qkv, nullptr, seq, heads, scale, mask_value,
                               cu + b0, ctx);
        else
            hipLaunchKernelGGL(attention_kernel<D>, grid, dim3(256), SM::BYTES, stream,
                               qkv + b0 * seq * 3 * (int64_t)heads * D, mask ? mask + b0 * seq : nullptr,
                               seq, heads, scale, mask_value, nullptr, ctx + b0 * seq * (int64_t)heads * D);
    }
    return hipGetLastError();
}

}  // namespace

// Calls of up to this many (sentence, head) items take the small-call kernel (another summation order than the item-per-workgroup
// kernels): callers that cut a batch in two keep both halves above it so that the cut does not change a bit of the result.
int attention_small_call_items() { return kSplitMaxItems; }

namespace {
hipError_t launch_generic(const float* qkv, const uint32_t* mask, int64_t batch, int seq, int heads, int head_dim, float mask_value,
                          float* ctx, hipStream_t stream, const int32_t* cu, const float* pos_bias, int bias_seq, bool scale_qk)
{
    const float scale = scale_qk ? 1.0f / sqrtf((float)head_dim) : 1.0f;
    for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
        const int64_t nb = (batch - b0 < 65535) ? (batch - b0) : 65535;
        dim3 grid((unsigned)seq, (unsigned)heads, (unsigned)nb);
        if (cu)
            hipLaunchKernelGGL(attention_generic_kernel, grid, dim3(64), seq * sizeof(float), stream, qkv, nullptr, seq, heads,
                               head_dim, scale, mask_value, cu + b0, ctx, pos_bias, bias_seq);
        else
            hipLaunchKernelGGL(attention_generic_kernel, grid, dim3(64), seq * sizeof(float), stream,
                               qkv + b0 * seq * 3 * (int64_t)heads * head_dim,
                               mask ? mask + b0 * seq : nullptr, seq, heads, head_dim, scale, mask_value, nullptr,
                               ctx + b0 * seq * (int64_t)heads * head_dim, pos_bias, bias_seq);
    }
    return hipGetLastError();
}
}  // namespace

hipError_t launch_attention(const float* qkv, const uint32_t* mask, int64_t batch, int seq, int heads,
                            int head_dim, float mask_value, float* ctx, hipStream_t stream, const int32_t* cu)
{
    if (batch <= 0 || seq <= 0) return hipSuccess;
    const bool aligned = ((heads * head_dim) % 4 == 0) && ((reinterpret_cast<uintptr_t>(qkv) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(ctx) & 15) == 0);  // (16-byte loads of qkv, 16-byte stores of ctx)
    if (head_dim == 32 && aligned) return launch_d<32>(qkv, mask, batch, seq, heads, mask_value, ctx, stream, cu);
    if (head_dim == 64 && aligned) return launch_d<64>(qkv, mask, batch, seq, heads, mask_value, ctx, stream, cu);
    return launch_generic(qkv, mask, batch, seq, heads, head_dim, mask_value, ctx, stream, cu, nullptr, 0, true);
}

hipError_t launch_attention_biased(const float* qkv, const uint32_t* mask, const float* pos_bias, int bias_seq, int64_t batch, int seq,
                                   int heads, int head_dim, bool scale_qk, float mask_value, float* ctx, hipStream_t stream)
{
    if (batch <= 0 || seq <= 0) return hipSuccess;
    if (pos_bias && bias_seq < seq) return hipErrorInvalidValue;
    return launch_generic(qkv, mask, batch, seq, heads, head_dim, mask_value, ctx, stream, nullptr, pos_bias, bias_seq, scale_qk);
}

}  // namespace kjarni
