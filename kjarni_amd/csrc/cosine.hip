// Cosine-similarity corpus scan + top-k selection (HBM-bound).
//
// Replaces VectorStore::cosine_similarity / search
// (crates/kjarni-search/src/vector.rs:131-166) and Segment::search_vectors +
// cosine_similarity_with_norm (crates/kjarni-rag/src/segment.rs:307-371): the
// reference walks the corpus on one thread with a scalar dot+norm per document
// and then sorts; here every document row (dim floats, raw little-endian f32 as
// in vectors.bin, segment.rs:240-262) is streamed once with 16-byte coalesced
// loads, one wave64 per row, the query held in registers, dot and squared norm
// reduced with wave shuffles.  Algorithmic traffic: dim*4 bytes per document
// per query group (+4 bytes of score written).
//
// Selection keeps the reference's order -- score descending, and for equal
// scores ascending document index (what the stable sort_by over an
// index-ordered Vec gives, vector.rs:162) -- by sorting 64-bit keys
// (orderable score bits << 32 | ~index) with LDS bitonic networks: each block
// reduces a segment of 2048..16384 candidates to its best KPAD, levels repeat until
// one block is left.
#include <algorithm>
#include <atomic>

#include <type_traits>

#include "device_utils.h"
#include "kernels.h"
#include "tuning.h"

namespace kjarni {

namespace {

constexpr int SCAN_NQ = 4;       // queries held in registers per pass
constexpr int SCAN_MAX_V4 = 4;   // dim <= 1024 on the float4 path

// mode 0: vector.rs:131-148   dot / max(sqrt(na)*sqrt(nb), 1e-9)
// mode 1: segment.rs:355-371  nb < 1e-9 ? 0 : dot / (qn * nb)
// mode 2 (internal): the row's squared norm itself
__device__ __forceinline__ float cosine_finish(float dot, float qn2, float dn2, int mode)
{
    if (mode == 2) return dn2;
    if (mode == 0) {
        const float den = fmaxf(sqrtf(qn2) * sqrtf(dn2), 1e-9f);
        return dot / den;
    }
    const float dn = sqrtf(dn2);
    if (dn < 1e-9f) return 0.0f;
    return dot / (sqrtf(qn2) * dn);
}

__global__ __launch_bounds__(256) void cosine_scores_kernel(const float* __restrict__ queries, int nq,
                                                            const float* __restrict__ corpus,
                                                            int64_t n_docs, int dim, int mode,
                                                            float* __restrict__ scores,
                                                            int64_t score_stride)
{
    const int lane = threadIdx.x & 63;
    const int nv4 = dim >> 2;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;

    f32x4 q[SCAN_NQ][SCAN_MAX_V4];
    float qn2[SCAN_NQ];
#pragma unroll
    for (int j = 0; j < SCAN_NQ; ++j) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < SCAN_MAX_V4; ++i) {
            const int c4 = lane + i * 64;
            q[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < nq && c4 < nv4) q[j][i] = *reinterpret_cast<const f32x4*>(queries + (int64_t)j * dim + c4 * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) s = fmaf(q[j][i][c], q[j][i][c], s);
        }
        qn2[j] = wave_sum(s);
    }

    for (int64_t d = wave; d < n_docs; d += n_waves) {
        const float* row = corpus + d * dim;
        f32x4 x[SCAN_MAX_V4];
#pragma unroll
        for (int i = 0; i < SCAN_MAX_V4; ++i) {
            const int c4 = lane + i * 64;
            x[i] = (c4 < nv4) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row + c4 * 4))
                              : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float dn2 = 0.0f;
        float dot[SCAN_NQ];
#pragma unroll
        for (int j = 0; j < SCAN_NQ; ++j) dot[j] = 0.0f;
#pragma unroll
        for (int i = 0; i < SCAN_MAX_V4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                dn2 = fmaf(x[i][c], x[i][c], dn2);
#pragma unroll
                for (int j = 0; j < SCAN_NQ; ++j) dot[j] = fmaf(q[j][i][c], x[i][c], dot[j]);
            }
        dn2 = wave_sum(dn2);
#pragma unroll
        for (int j = 0; j < SCAN_NQ; ++j) {
            if (j < nq) {  // wave-uniform
                const float dj = wave_sum(dot[j]);
                if (lane == 0) scores[(int64_t)j * score_stride + d] = cosine_finish(dj, qn2[j], dn2, mode);
            }
        }
    }
}

// Streaming variant for the common embedding widths.  Rows are contiguous, so a wave takes a GROUP of
// R rows as one flat run of R*NV4 float4s and issues all T = R*NV4/64 full-width 16-byte loads before
// touching any of them: T KiB in flight per wave instead of one (partly filled) row.  Which row a
// lane's element belongs to is compile-time per load except at the (at most NV4<64 ? many : one) row
// boundaries inside a load; the query is pre-permuted into the same lane order once per wave.
template <int NV4, int R, int NQ>
__global__ __launch_bounds__(256) void cosine_scores_stream_kernel(const float* __restrict__ queries,
                                                                   const float* __restrict__ corpus,
                                                                   int64_t n_groups, int mode,
                                                                   float* __restrict__ scores,
                                                                   int64_t score_stride)
{
    static_assert((R * NV4) % 64 == 0, "a group must be a whole number of wave-wide loads");
    constexpr int T = R * NV4 / 64;
    constexpr int DIM = NV4 * 4;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;

    f32x4 q[NQ][T];
    float qn2[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const f32x4* qv = reinterpret_cast<const f32x4*>(queries + (int64_t)j * DIM);
        float s = 0.0f;
        for (int c4 = lane; c4 < NV4; c4 += 64) {
            const f32x4 v = qv[c4];
#pragma unroll
            for (int c = 0; c < 4; ++c) s = fmaf(v[c], v[c], s);
        }
        qn2[j] = wave_sum(s);
#pragma unroll
        for (int t = 0; t < T; ++t) q[j][t] = qv[(t * 64 + lane) % NV4];
    }

    for (int64_t g = wave; g < n_groups; g += n_waves) {
        const f32x4* base = reinterpret_cast<const f32x4*>(corpus + g * (int64_t)(R * DIM));
        f32x4 x[T];
#pragma unroll
        for (int t = 0; t < T; ++t) x[t] = __builtin_nontemporal_load(base + t * 64 + lane);

        float dn[R], dt[NQ][R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            dn[r] = 0.0f;
#pragma unroll
            for (int j = 0; j < NQ; ++j) dt[j][r] = 0.0f;
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float s2 = 0.0f, sq[NQ];
#pragma unroll
            for (int c = 0; c < 4; ++c) s2 = fmaf(x[t][c], x[t][c], s2);
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                sq[j] = 0.0f;
#pragma unroll
                for (int c = 0; c < 4; ++c) sq[j] = fmaf(q[j][t][c], x[t][c], sq[j]);
            }
            const int r_lo = (t * 64) / NV4, r_hi = (t * 64 + 63) / NV4;  // constants after unrolling
            if (r_lo == r_hi) {
                dn[r_lo] += s2;
#pragma unroll
                for (int j = 0; j < NQ; ++j) dt[j][r_lo] += sq[j];
            } else {
                const int my = (t * 64 + lane) / NV4;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (r >= r_lo && r <= r_hi) {
                        const bool in = my == r;
                        dn[r] += in ? s2 : 0.0f;
#pragma unroll
                        for (int j = 0; j < NQ; ++j) dt[j][r] += in ? sq[j] : 0.0f;
                    }
                }
            }
        }
        // every lane ends up with all R results; lane r stores row r's score (one 4*R-byte store)
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            float mine = 0.0f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float d2 = (j == 0) ? (dn[r] = wave_sum(dn[r])) : dn[r];
                const float dj = wave_sum(dt[j][r]);
                const float sc = cosine_finish(dj, qn2[j], d2, mode);
                mine = (lane == r) ? sc : mine;
            }
            if (lane < R) scores[(int64_t)j * score_stride + g * R + lane] = mine;
        }
    }
}

// Any-dim fallback: one wave per row, scalar loads.
__global__ __launch_bounds__(256) void cosine_scores_generic_kernel(
    const float* __restrict__ queries, int nq, const float* __restrict__ corpus, int64_t n_docs,
    int dim, int mode, float* __restrict__ scores, int64_t score_stride)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int j = 0; j < nq; ++j) {
        const float* qv = queries + (int64_t)j * dim;
        float qs = 0.0f;
        for (int i = lane; i < dim; i += 64) qs = fmaf(qv[i], qv[i], qs);
        qs = wave_sum(qs);
        for (int64_t d = wave; d < n_docs; d += n_waves) {
            const float* row = corpus + d * dim;
            float dot = 0.0f, dn2 = 0.0f;
            for (int i = lane; i < dim; i += 64) {
                const float xv = row[i];
                dot = fmaf(qv[i], xv, dot);
                dn2 = fmaf(xv, xv, dn2);
            }
            dot = wave_sum(dot);
            dn2 = wave_sum(dn2);
            if (lane == 0) scores[(int64_t)j * score_stride + d] = cosine_finish(dot, qs, dn2, mode);
        }
    }
}

// ---------------------------------------------------------------------------
// Top-k
// ---------------------------------------------------------------------------

constexpr int TK_TILE = 2048;               // candidates examined between two barriers
constexpr int TK_PEND = 2 * TK_TILE;        // pending buffer (keys that beat the block's threshold)
constexpr int TK_MAX_TILES_PER_BLOCK = 64;  // a block folds up to 131072 candidates into its best KPAD

// The many-query searches' counters (word 0: overflow; word kCountStride (1 + q): entries in query q's candidate list) sit a
// 128-byte line apart: thousands of atomics on 64 adjacent words were one line's worth of serial work (profiles/r06z).
constexpr int kCountStride = 32;

// Larger float -> larger uint32 (total order, -0 < +0); NaN sorts lowest.
__device__ __forceinline__ uint32_t orderable(float f)
{
    if (f != f) return 0u;
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_orderable(uint32_t o)
{
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(u);
}
__device__ __forceinline__ uint64_t make_key(float score, uint32_t idx)
{
    return ((uint64_t)orderable(score) << 32) | (uint64_t)(~idx);
}

// Descending bitonic sort of n (power of two) keys in LDS by 256 threads.
__device__ __forceinline__ void bitonic_sort_desc(uint64_t* keys, int n, int tid)
{
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < (n >> 1); t += 256) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const uint64_t a = keys[lo], b = keys[hi];
                if ((a < b) == desc) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
        }
    }
    __syncthreads();
}

// Descending merge of a bitonic sequence of n keys.
__device__ __forceinline__ void bitonic_merge_desc(uint64_t* keys, int n, int tid)
{
    for (int stride = n >> 1; stride > 0; stride >>= 1) {
        __syncthreads();
        for (int t = tid; t < (n >> 1); t += 256) {
            const int lo = 2 * t - (t & (stride - 1));
            const int hi = lo + stride;
            const uint64_t a = keys[lo], b = keys[hi];
            if (a < b) {
                keys[lo] = b;
                keys[hi] = a;
            }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------
// One query, scan and selection in ONE pass: the scores never exist in memory.
//
// VectorStore::search / Segment::search_vectors score every document and then sort (kjarni-search/src/vector.rs:150-166,
// kjarni-rag/src/segment.rs:307-337); the two-launch form above writes 4 bytes per document and reads them back -- for one
// query over 10^5 documents the selection launches cost twice the scan.  Here every wave of the streaming kernel keeps the best
// 64 S keys it has seen IN REGISTERS: lane l, slot s holds the (64 s + l)-th best key of the wave, sorted descending, and the
// smallest kept key is a wave-uniform threshold.  A group's R scores are turned into keys (orderable score << 32 | ~index: score
// descending, ties by ascending index -- the reference's stable sort) and only a key above the threshold is inserted: a ballot
// finds its place, the tail shifts down by one lane.  Insertions are rare once the threshold has risen (about K ln(n / K) of
// a wave's n rows), and the loop has no barrier, so the stream runs at the rate of the score-only kernel.  At the end the four
// waves of a workgroup merge through LDS (one bitonic sort of 4 x 64 S keys) and the workgroup writes its best KOUT keys; a
// last small launch (topk_reduce_kernel on keys) merges the workgroups' lists and decodes.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

template <int S>
struct WaveTopK {
    uint64_t k[S];
    uint64_t thr;  // wave-uniform: the smallest kept key (slot S - 1 of lane 63); 0 while the list is not full
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int s = 0; s < S; ++s) k[s] = 0ull;
        thr = 0ull;
    }
    // c: wave-uniform, c > thr
    __device__ __forceinline__ void insert(uint64_t c, int lane)
    {
        int pos = 0;  // keys above c (all keys are distinct: the index is part of the key)
#pragma unroll
        for (int s = 0; s < S; ++s) pos += __popcll(__ballot(k[s] > c));
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            uint64_t up = __shfl_up(k[s], 1, kWave);
            if (s > 0) {
                const uint64_t carry = readlane64(k[s - 1], 63);
                up = lane == 0 ? carry : up;
            }
            const int idx = s * 64 + lane;
            k[s] = idx > pos ? up : (idx == pos ? c : k[s]);
        }
        thr = readlane64(k[S - 1], 63);
    }
};

template <int NV4, int R, int S>
__global__ __launch_bounds__(256) void cosine_search_stream_kernel(const float* __restrict__ query, const float* __restrict__ corpus,
                                                                   int64_t n_groups, int64_t n_docs, int mode, int kout,
                                                                   uint64_t* __restrict__ cand)
{
    static_assert((R * NV4) % 64 == 0, "a group must be a whole number of wave-wide loads");
    constexpr int T = R * NV4 / 64;
    constexpr int DIM = NV4 * 4;
    constexpr int KW = 64 * S;
    __shared__ uint64_t keys[4 * KW];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wid;
    const int64_t n_waves = (int64_t)gridDim.x * 4;

    f32x4 q[T];
    float qn2;
    {
        const f32x4* qv = reinterpret_cast<const f32x4*>(query);
        float s = 0.0f;
        for (int c4 = lane; c4 < NV4; c4 += 64) {
            const f32x4 v = qv[c4];
#pragma unroll
            for (int c = 0; c < 4; ++c) s = fmaf(v[c], v[c], s);
        }
        qn2 = wave_sum(s);
#pragma unroll
        for (int t = 0; t < T; ++t) q[t] = qv[(t * 64 + lane) % NV4];
    }
    WaveTopK<S> tk;
    tk.init();
    auto offer = [&](uint64_t key) {  // key: this lane's candidate (0: none)
        uint64_t m = __ballot(key > tk.thr);
        while (m) {
            const int l = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            const uint64_t c = readlane64(key, l);
            if (c > tk.thr) tk.insert(c, lane);  // (the threshold may have risen since the ballot)
        }
    };

    // A wave's next group is requested before its current one is reduced and offered (the loads of group g + n_waves are in
    // flight under the shuffles and the insertions of group g; past the last group the request repeats it: no branch around
    // a memory operation in the loop).
    f32x4 xn[T];
    {
        const int64_t g0 = wave < n_groups ? wave : (n_groups > 0 ? n_groups - 1 : 0);
        const f32x4* base = reinterpret_cast<const f32x4*>(corpus + g0 * (int64_t)(R * DIM));
#pragma unroll
        for (int t = 0; t < T; ++t) xn[t] = n_groups > 0 ? __builtin_nontemporal_load(base + t * 64 + lane) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t g = wave; g < n_groups; g += n_waves) {
        f32x4 x[T];
#pragma unroll
        for (int t = 0; t < T; ++t) x[t] = xn[t];
        {
            const int64_t gn = g + n_waves < n_groups ? g + n_waves : g;
            const f32x4* base = reinterpret_cast<const f32x4*>(corpus + gn * (int64_t)(R * DIM));
#pragma unroll
            for (int t = 0; t < T; ++t) xn[t] = __builtin_nontemporal_load(base + t * 64 + lane);
        }
        float dn[R], dt[R];
#pragma unroll
        for (int r = 0; r < R; ++r) dn[r] = dt[r] = 0.0f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float s2 = 0.0f, sq = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s2 = fmaf(x[t][c], x[t][c], s2);
                sq = fmaf(q[t][c], x[t][c], sq);
            }
            const int r_lo = (t * 64) / NV4, r_hi = (t * 64 + 63) / NV4;  // constants after unrolling
            if (r_lo == r_hi) {
                dn[r_lo] += s2;
                dt[r_lo] += sq;
            } else {
                const int my = (t * 64 + lane) / NV4;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (r >= r_lo && r <= r_hi) {
                        const bool in = my == r;
                        dn[r] += in ? s2 : 0.0f;
                        dt[r] += in ? sq : 0.0f;
                    }
                }
            }
        }
        // (the arithmetic of cosine_scores_stream_kernel<NV4, R, 1>, operation for operation: the same scores to the bit)
        float mine = 0.0f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float d2 = wave_sum(dn[r]);
            const float dj = wave_sum(dt[r]);
            const float sc = cosine_finish(dj, qn2, d2, mode);
            mine = (lane == r) ? sc : mine;
        }
        offer(lane < R ? make_key(mine, (uint32_t)(g * R + lane)) : 0ull);
    }
    // the (< R) rows past the last whole group: one wave of the launch, a row at a time
    if (wave == 0) {
        for (int64_t d = n_groups * R; d < n_docs; ++d) {
            const f32x4* row = reinterpret_cast<const f32x4*>(corpus + d * DIM);
            const f32x4* qv = reinterpret_cast<const f32x4*>(query);
            float s2 = 0.0f, sq = 0.0f;
            for (int c4 = lane; c4 < NV4; c4 += 64) {
                const f32x4 xv = row[c4], qq = qv[c4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    s2 = fmaf(xv[c], xv[c], s2);
                    sq = fmaf(qq[c], xv[c], sq);
                }
            }
            const float sc = cosine_finish(wave_sum(sq), qn2, wave_sum(s2), mode);
            offer(lane == 0 ? make_key(sc, (uint32_t)d) : 0ull);
        }
    }
    // workgroup merge: 4 x KW keys, sorted descending; the best kout leave
#pragma unroll
    for (int s = 0; s < S; ++s) keys[wid * KW + s * 64 + lane] = tk.k[s];
    bitonic_sort_desc(keys, 4 * KW, tid);
    for (int i = tid; i < kout; i += 256) cand[(int64_t)blockIdx.x * kout + i] = keys[i];
}

// One block (256 threads) folds candidates [seg0, seg0 + TK_TILE * tiles) of one query into its best KPAD keys, left sorted
// descending in best[].  Level 0 reads float scores (index = position), later levels read keys.  Keys >= `upper` are ignored
// (multi-pass k > 1024).
//
// Only candidates above the block's current KPAD-th best can matter, and on anything but adversarial
// input that threshold rises quickly: candidates that pass it are compacted into a pending buffer
// (wave-aggregated LDS append) and the bitonic sort + merge runs only when that buffer might overflow
// and once at the end -- typically twice per block instead of once per 2048 candidates.
template <int KPAD>
__device__ __forceinline__ void topk_block_reduce(const float* __restrict__ scores, const uint64_t* __restrict__ in_keys, int64_t n,
                                                  int64_t seg0, int tiles, uint64_t upper, uint64_t* pend, uint64_t* best, int* pend_count)
{
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < KPAD; i += 256) best[i] = 0ull;
    if (tid == 0) *pend_count = 0;
    __syncthreads();

    // sort the pending keys, fold them into best[], reset the buffer
    auto flush = [&]() {
        const int cnt = *pend_count;  // uniform: read after a barrier
        int p2 = KPAD;
        while (p2 < cnt) p2 <<= 1;
        __syncthreads();  // everyone has read pend_count
        for (int i = cnt + tid; i < p2; i += 256) pend[i] = 0ull;
        if (tid == 0) *pend_count = 0;
        bitonic_sort_desc(pend, p2, tid);
        // best (desc) and the reversed head of pend (asc) form a bitonic sequence whose element-wise
        // max holds the top KPAD of the union.
        for (int i = tid; i < KPAD; i += 256) {
            const uint64_t a = best[i], b2 = pend[KPAD - 1 - i];
            best[i] = a > b2 ? a : b2;
        }
        bitonic_merge_desc(best, KPAD, tid);
    };

    for (int tl = 0; tl < tiles; ++tl) {
        const int64_t t0 = seg0 + (int64_t)tl * TK_TILE;
        if (t0 >= n) break;
        if (*pend_count + TK_TILE > TK_PEND) flush();  // uniform branch (barrier at the end of the last tile)
        const uint64_t thr = best[KPAD - 1];
        for (int i = tid; i < TK_TILE; i += 256) {
            const int64_t p = t0 + i;
            uint64_t key = 0ull;
            if (p < n) {
                key = scores ? make_key(scores[p], (uint32_t)p) : in_keys[p];
                if (key >= upper) key = 0ull;
            }
            const bool take = key > thr;
            const uint64_t m = __ballot(take);
            if (m) {
                const int leader = __ffsll((unsigned long long)m) - 1;
                int base = 0;
                if (lane == leader) base = atomicAdd(pend_count, __popcll(m));
                base = __shfl(base, leader, kWave);
                if (take) pend[base + __popcll(m & ((1ull << lane) - 1ull))] = key;
            }
        }
        __syncthreads();
    }
    if (*pend_count > 0) flush();
    __syncthreads();
}

// One block reduces candidates [blk*seg, +seg), seg = TK_TILE*tiles_per_block, of query blockIdx.y to its
// best KPAD keys (descending).
template <int KPAD>
__global__ __launch_bounds__(256) void topk_reduce_kernel(const float* __restrict__ scores,
                                                          const uint64_t* __restrict__ in_keys,
                                                          int64_t n, int64_t in_stride,
                                                          const uint64_t* __restrict__ upper_ptr,
                                                          uint64_t* __restrict__ out_keys,
                                                          int64_t out_stride, int tiles_per_block,
                                                          const unsigned* __restrict__ run_flag = nullptr)
{
    __shared__ uint64_t pend[TK_PEND];
    __shared__ uint64_t best[KPAD];
    __shared__ int pend_count;
    if (run_flag != nullptr && *run_flag == 0u) return;  // (uniform: a fallback launch that is not needed)
    const int tid = threadIdx.x;
    const int qi = blockIdx.y;
    const uint64_t upper = upper_ptr ? upper_ptr[qi] : ~0ull;
    topk_block_reduce<KPAD>(scores ? scores + (int64_t)qi * in_stride : nullptr, in_keys ? in_keys + (int64_t)qi * in_stride : nullptr, n,
                            (int64_t)blockIdx.x * TK_TILE * tiles_per_block, tiles_per_block, upper, pend, best, &pend_count);
    for (int i = tid; i < KPAD; i += 256)
        out_keys[(int64_t)qi * out_stride + (int64_t)blockIdx.x * KPAD + i] = best[i];
}

// The last step of the fused one-query search: `lists` (<= 2 048) sorted lists of KPAD keys each (the scan's workgroups) -> the
// best k_take, decoded.  Any threshold T with at least k keys at or above it will do: the answer is the best k of the keys >= T.
// T = the k-th largest of a SAMPLE of up to 256 list heads (every ceil(lists / 256)-th list; a head is its list's largest key,
// so k sampled heads >= T are k keys >= T): one 256-key sort.  A list whose head is below T holds nothing of interest; the
// others are read from the top down while their keys stay >= T -- on anything but adversarial placements a few dozen keys in
// all -- sorted, decoded.  A few microseconds where the generic threshold-and-flush reduction over lists x KPAD keys took 120.
// Should the gathered set not fit the buffer (4 096 keys), the block runs the generic reduction over everything.
template <int KPAD>
__global__ __launch_bounds__(256) void topk_lists_final_kernel(const uint64_t* __restrict__ keys, int lists, int k_take,
                                                               int64_t* __restrict__ out_idx, float* __restrict__ out_score)
{
    __shared__ uint64_t buf[TK_PEND];
    __shared__ uint64_t best[KPAD];
    __shared__ int count;
    const int tid = threadIdx.x;
    constexpr int PER = 8;  // lists per thread (2 048 / 256)
    uint64_t head[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int l = tid + 256 * j;
        head[j] = l < lists ? keys[(int64_t)l * KPAD] : 0ull;
    }
    const int stride = (lists + 255) / 256, m = (lists + stride - 1) / stride;  // the sample: lists 0, stride, 2 stride, ...
    buf[tid] = tid < m ? keys[(int64_t)tid * stride * KPAD] : 0ull;
    if (tid == 0) count = 0;
    bitonic_sort_desc(buf, 256, tid);
    const uint64_t thr = k_take - 1 < m ? buf[k_take - 1] : 0ull;
    __syncthreads();  // everyone has read thr before the buffer is reused
    bool generic = lists > 256 * PER;
    if (!generic) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (head[j] == 0ull || head[j] < thr) continue;
            const uint64_t* lk = keys + (int64_t)(tid + 256 * j) * KPAD;
            for (int i = 0; i < KPAD; ++i) {
                const uint64_t key = i == 0 ? head[j] : lk[i];
                if (key == 0ull || key < thr) break;  // (sorted descending)
                const int at = atomicAdd(&count, 1);
                if (at < TK_PEND) buf[at] = key;
            }
        }
        __syncthreads();
        generic = count > TK_PEND;
    }
    if (!generic) {  // (uniform)
        const int cnt = count;
        int c2 = 1;
        while (c2 < cnt) c2 <<= 1;
        __syncthreads();
        for (int i = cnt + tid; i < c2; i += 256) buf[i] = 0ull;
        bitonic_sort_desc(buf, c2, tid);
        for (int i = tid; i < k_take; i += 256) {
            const uint64_t key = i < cnt ? buf[i] : 0ull;
            out_idx[i] = key == 0ull ? -1 : (int64_t)(uint32_t)(~(uint32_t)(key & 0xFFFFFFFFull));
            out_score[i] = key == 0ull ? -INFINITY : from_orderable((uint32_t)(key >> 32));
        }
        return;
    }
    __syncthreads();
    const int64_t n = (int64_t)lists * KPAD;
    topk_block_reduce<KPAD>(nullptr, keys, n, 0, (int)((n + TK_TILE - 1) / TK_TILE), ~0ull, buf, best, &count);
    for (int i = tid; i < k_take; i += 256) {
        const uint64_t key = i < KPAD ? best[i] : 0ull;
        out_idx[i] = key == 0ull ? -1 : (int64_t)(uint32_t)(~(uint32_t)(key & 0xFFFFFFFFull));
        out_score[i] = key == 0ull ? -INFINITY : from_orderable((uint32_t)(key >> 32));
    }
}

// The selection behind the fused many-query scan: one block per query folds that query's candidate list into its best
// k_take, decoded.  Returns at once when a list overflowed (the two-call form runs instead).
template <int KPAD>
__global__ __launch_bounds__(256) void topk_candidates_kernel(const uint64_t* __restrict__ cand_key, const unsigned* __restrict__ cand_count,
                                                              unsigned cap, int k_take, int64_t* __restrict__ out_idx,
                                                              float* __restrict__ out_score)
{
    __shared__ uint64_t pend[TK_PEND];
    __shared__ uint64_t best[KPAD];
    __shared__ int pend_count;
    if (cand_count[0] != 0u) return;
    const int tid = threadIdx.x, qi = blockIdx.x;
    const int64_t n = cand_count[kCountStride * (1 + qi)] < cap ? cand_count[kCountStride * (1 + qi)] : cap;
    topk_block_reduce<KPAD>(nullptr, cand_key + (size_t)qi * cap, n, 0, (int)((n + TK_TILE - 1) / TK_TILE), ~0ull, pend, best, &pend_count);
    for (int i = tid; i < k_take; i += 256) {
        const uint64_t key = i < KPAD ? best[i] : 0ull;
        const int64_t o = (int64_t)qi * k_take + i;
        out_idx[o] = key == 0ull ? -1 : (int64_t)(uint32_t)(~(uint32_t)(key & 0xFFFFFFFFull));
        out_score[o] = key == 0ull ? -INFINITY : from_orderable((uint32_t)(key >> 32));
    }
}

__global__ void topk_decode_kernel(const uint64_t* __restrict__ keys, int64_t key_stride, int k_take,
                                   int64_t* __restrict__ out_idx, float* __restrict__ out_score,
                                   int64_t out_stride, int64_t out_offset,
                                   uint64_t* __restrict__ last_key, const unsigned* __restrict__ run_flag = nullptr)
{
    if (run_flag != nullptr && *run_flag == 0u) return;
    const int qi = blockIdx.x;
    for (int i = threadIdx.x; i < k_take; i += blockDim.x) {
        const uint64_t key = keys[(int64_t)qi * key_stride + i];
        const int64_t o = (int64_t)qi * out_stride + out_offset + i;
        if (key == 0ull) {
            out_idx[o] = -1;
            out_score[o] = -INFINITY;
        } else {
            out_idx[o] = (int64_t)(uint32_t)(~(uint32_t)(key & 0xFFFFFFFFull));
            out_score[o] = from_orderable((uint32_t)(key >> 32));
        }
        if (i == k_take - 1 && last_key) last_key[qi] = key;
    }
}

int kpad_for(int k)
{
    int p = 16;
    while (p < k) p <<= 1;
    return p;
}

// Enough blocks to fill the chip (about 2048 across all queries), otherwise segments as long as
// possible: the longer a segment, the more of it is rejected by the threshold without sorting.
int tiles_per_block_for(int64_t n, int nq)
{
    const int64_t tiles = (n + TK_TILE - 1) / TK_TILE;
    const int64_t t = (tiles * nq + 2047) / 2048;
    return (int)(t < 1 ? 1 : (t > TK_MAX_TILES_PER_BLOCK ? TK_MAX_TILES_PER_BLOCK : t));
}

int64_t blocks_for(int64_t n, int nq)
{
    const int64_t seg = (int64_t)TK_TILE * tiles_per_block_for(n, nq);
    return (n + seg - 1) / seg;
}

template <int KPAD>
void launch_reduce(const float* scores, const uint64_t* in_keys, int64_t n, int64_t in_stride,
                   const uint64_t* upper, uint64_t* out_keys, int64_t out_stride, int nq,
                   hipStream_t stream, const unsigned* run_flag)
{
    dim3 grid((unsigned)blocks_for(n, nq), (unsigned)nq);
    hipLaunchKernelGGL(topk_reduce_kernel<KPAD>, grid, dim3(256), 0, stream, scores, in_keys, n,
                       in_stride, upper, out_keys, out_stride, tiles_per_block_for(n, nq), run_flag);
}

void dispatch_reduce(int kpad, const float* scores, const uint64_t* in_keys, int64_t n,
                     int64_t in_stride, const uint64_t* upper, uint64_t* out_keys, int64_t out_stride,
                     int nq, hipStream_t stream, const unsigned* run_flag = nullptr)
{
    switch (kpad) {
    case 16: launch_reduce<16>(scores, in_keys, n, in_stride, upper, out_keys, out_stride, nq, stream, run_flag); break;
    case 32: launch_reduce<32>(scores, in_keys, n, in_stride, upper, out_keys, out_stride, nq, stream, run_flag); break;
    case 64: launch_reduce<64>(scores, in_keys, n, in_stride, upper, out_keys, out_stride, nq, stream, run_flag); break;
    case 128: launch_reduce<128>(scores, in_keys, n, in_stride, upper, out_keys, out_stride, nq, stream, run_flag); break;
    case 256: launch_reduce<256>(scores, in_keys, n, in_stride, upper, out_keys, out_stride, nq, stream, run_flag); break;
    case 512: launch_reduce<512>(scores, in_keys, n, in_stride, upper, out_keys, out_stride, nq, stream, run_flag); break;
    default: launch_reduce<1024>(scores, in_keys, n, in_stride, upper, out_keys, out_stride, nq, stream, run_flag); break;
    }
}

}  // namespace

namespace {

template <int NV4, int R, int NQ>
void launch_stream(const float* queries, const float* corpus, int64_t n_groups, int mode, float* scores,
                   int64_t score_stride, hipStream_t stream)
{
    int64_t blocks = (n_groups + 3) / 4;
    const int64_t max_blocks = 256 * 8;
    if (blocks > max_blocks) blocks = max_blocks;
    hipLaunchKernelGGL((cosine_scores_stream_kernel<NV4, R, NQ>), dim3((unsigned)blocks), dim3(256), 0, stream, queries,
                       corpus, n_groups, mode, scores, score_stride);
}

// Rows per group for the streaming kernel (0 = width not specialised).
int stream_rows(int dim, bool multi)
{
    switch (dim) {
    case 384: return multi ? 2 : 4;
    case 768: return 2;
    case 1024: return multi ? 1 : 2;
    case 512: return multi ? 2 : 4;
    case 256: return multi ? 4 : 8;
    case 128: return multi ? 8 : 16;
    default: return 0;
    }
}

template <int NQ>
bool dispatch_stream(int dim, const float* queries, const float* corpus, int64_t n_groups, int mode, float* scores,
                     int64_t score_stride, hipStream_t stream)
{
    constexpr bool M = NQ > 1;
    switch (dim) {
    case 384: launch_stream<96, M ? 2 : 4, NQ>(queries, corpus, n_groups, mode, scores, score_stride, stream); return true;
    case 768: launch_stream<192, 2, NQ>(queries, corpus, n_groups, mode, scores, score_stride, stream); return true;
    case 1024: launch_stream<256, M ? 1 : 2, NQ>(queries, corpus, n_groups, mode, scores, score_stride, stream); return true;
    case 512: launch_stream<128, M ? 2 : 4, NQ>(queries, corpus, n_groups, mode, scores, score_stride, stream); return true;
    case 256: launch_stream<64, M ? 4 : 8, NQ>(queries, corpus, n_groups, mode, scores, score_stride, stream); return true;
    case 128: launch_stream<32, M ? 8 : 16, NQ>(queries, corpus, n_groups, mode, scores, score_stride, stream); return true;
    default: return false;
    }
}

}  // namespace

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// Many queries (20 or more): the fused matrix-core scan.
//
// dot(q_j, doc_d) for a block of up to 64 queries is a [64, dim] x [n_docs, dim]^T product, so the corpus is read ONCE
// for the block (a streaming pass serves 4 queries).  One launch does everything VectorStore::search /
// Segment::search_vectors do per document (kjarni-search/src/vector.rs:131-166, kjarni-rag/src/segment.rs:307-371):
// the dot products on the f32 matrix cores, ||doc||^2 from the very registers that stage the document rows (every
// document byte crosses HBM once and is touched twice in registers), and the cosine with the mode's zero-norm guard in
// the epilogue.  Both roofs are close for 64 queries: 32 flop per corpus byte = 4.9 TB/s at the f32 MFMA peak.
//
// Tile: 64 queries (rows past nq read as zeros) x 256 documents, BK = 16; four waves, wave w = documents 64 w ..
// 64 w + 63 against all 64 queries (2 x 2 MFMA tiles of 32 x 32).  LDS: two stages of [64 + 256][16 + 4] floats
// (51 KiB: three workgroups per CU); the staging map and the 80-byte row stride are those of gemm_nt_f32_mfma_ln
// (conflict-free 16-byte writes and fragment reads).  The kernel is persistent and its K-loop FLAT over (tile, K-step):
// registers hold step g + 1 while step g is multiplied and the loads of step g + 2 are in flight, across tile
// boundaries too, so a tile's first K-steps arrive under the previous tile's last ones and only the accumulator flush
// interrupts the matrix work.  Scores go out as 128-byte row segments straight from the accumulators (an accumulator
// register is 32 consecutive documents of one query).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int MQ_Q = 64, MQ_D = 256, MQ_BK = 16, MQ_STRIDE = MQ_BK + 4;
constexpr int MQ_STAGE_FLOATS = (MQ_Q + MQ_D) * MQ_STRIDE;
constexpr int MQ_LDS_BYTES = (2 * MQ_STAGE_FLOATS + 2 * MQ_D + 3 * MQ_Q) * 4;

// Selection inside the scan (FUSED).  VectorStore::search keeps the best k of every query (vector.rs:150-166); written out,
// 64 queries x 10^7 documents are 2.56 GB of scores that the selection reads back (2.1 of 7.5 ms).  With a per-query lower
// bound of the k-th best score -- the k-th best of a strided SAMPLE of the corpus' tiles, scanned first with this same kernel
// -- only scores at or above the bound can be in the answer, and the epilogue appends those (query, key) pairs to a candidate
// list instead of storing anything: about k x (tiles / sampled tiles) candidates per query.  The list is bounded; if an
// adversarial order overflows it the overflow word is raised and the launches of the two-call form, queued behind with that
// word as their run flag, do the work instead (they return at once otherwise).
struct ScanFuse {
    const float* thr_score;   // [nq, thr_k]: the sample's best thr_k scores per query (descending); bound = the last one
    const int64_t* thr_idx;   // [nq, thr_k]: -1 where the sample had fewer documents (bound = -inf)
    int thr_k;
    uint64_t* cand_key;       // [nq, cap]: a list per query (index within the whole call)
    unsigned* cand_count;     // [0]: overflow word, [kCountStride (1 + q)]: candidates appended to query q's list
    unsigned cap;             // entries per query
    int q_base;               // this launch's first query
};

// tile_stride > 1 (the sample pass, not FUSED): tiles 0, tile_stride, 2 tile_stride, ... are scanned and their scores written
// compactly ([nq, n_tiles * 256]).  run_flag != null: the launch runs only if *run_flag != 0.
// KIND: 0 scores out; 1 FUSED (selection inside the scan); 2 the sample pass of a fused search with a small k: instead of the
// sampled tiles' scores only the MAXIMUM of every (query, tile, wave) -- 64 documents -- goes out ([nq, n_tiles * 4]): the k-th
// largest of a query's maxima is a score that at least k documents reach, i.e. a valid bound, and for k far below the number of
// maxima it is as tight as the k-th best of all sampled scores (P(max of 64 >= T) ~ 64 P(score >= T)) -- at a 64th of the
// bytes and without the two-level selection over the sample (200 us of a 0.84 ms search of 10^6 documents).
constexpr int SCAN_SCORES = 0, SCAN_FUSED = 1, SCAN_SAMPLE_MAX = 2;

// dot / den as one v_rcp_f32 + a Newton step on the quotient (r = 1 / den to 1 ulp, v = dot r, v += (dot - den v) r): the IEEE
// division sequence without its range scaling, which den = ||q|| ||doc|| never needs; 6 vector instructions per score instead
// of 15.  ONE definition for the scan's epilogues and the filtered search's rescoring pass: the same bits from both.
template <int MODE>
__device__ __forceinline__ float mq_cosine(float dot, float qn, float dn)
{
    const float den = MODE == 0 ? fmaxf(qn * dn, 1e-9f) : qn * dn;   // vector.rs:131-148 | segment.rs:355-371
    const float rc = __builtin_amdgcn_rcpf(den);
    float v = dot * rc;
    v = fmaf(fmaf(-den, v, dot), rc, v);
    if (MODE == 1) v = dn < 1e-9f ? 0.0f : v;
    return v;
}

template <int MODE, int KIND>
__global__ __launch_bounds__(256, 2) void cosine_scan_mfma_kernel(const float* __restrict__ queries, int nq,
                                                                  const float* __restrict__ corpus, int64_t n_docs, int dim,
                                                                  const float* __restrict__ qn2, float* __restrict__ scores,
                                                                  int64_t stride, int64_t n_tiles, int tile_stride,
                                                                  const unsigned* __restrict__ run_flag, ScanFuse fuse)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sQ = smem;                                   // [2][64][20]   (stage s at + s * MQ_STAGE_FLOATS)
    float* sD = smem + MQ_Q * MQ_STRIDE;                // [2][256][20]
    float* sDn = smem + 2 * MQ_STAGE_FLOATS;            // [2][256]: ||doc||^2 of a tile, by tile parity
    float* sQn = sDn + 2 * MQ_D;                        // [64]: sqrt(||q||^2)
    float* sThr = sQn + MQ_Q;                           // [64]: FUSED: the query's score bound
    float* sTq = sThr + MQ_Q;                           // [64]: FUSED: the same bound for dot / ||doc|| (see the epilogue)
    constexpr bool FUSED = KIND == SCAN_FUSED;
    if (run_flag != nullptr && *run_flag == 0u) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int nk = dim / MQ_BK;
    // (tile and step counts fit 32 bits: a workgroup's share of at most 2^31 / 256 tiles; scalar registers are scarce here --
    // two buffer descriptors and the loop state -- and every spilled one is a v_readlane in the K-loop)
    const int nt = (int)n_tiles, grid = (int)gridDim.x;
    if ((int)blockIdx.x >= nt) return;
#ifdef KJARNI_TUNING
    // (diagnostics, cosine variants 3 / 4 / 5: -1 every tile reads the corpus' first 256 rows; -2 that and no epilogue;
    // -3 that and no norm arithmetic in the K-loop)
    // -4 / -5: the real corpus stream without the epilogue / without the norm arithmetic
    const bool same_tile = tile_stride < 0 && tile_stride >= -3, diag_no_epilogue = tile_stride == -2 || tile_stride == -4,
               diag_no_norms = tile_stride == -3 || tile_stride == -5;
    if (tile_stride < 0) tile_stride = 1;
#else
    constexpr bool diag_no_epilogue = false, diag_no_norms = false;
#endif
    const int my_tiles = (nt - 1 - (int)blockIdx.x) / grid + 1;

    // staging: thread -> (row, 16-byte column); lanes 0-3 take row r, lanes 4-7 row r + 4 (see gemm_nt_f32_mfma_ln)
    const int ld_grp = tid >> 3;
    const int ld_row = (ld_grp >> 2) * 8 + (ld_grp & 3) + 4 * ((tid >> 2) & 1), ld_c4 = tid & 3;
    const int st_off = ld_row * MQ_STRIDE + ld_c4 * 4;
    auto rsrc = [](const float* p, int64_t bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);
    };
    auto ld16 = [](__amdgpu_buffer_rsrc_t r, uint32_t off, int soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, 0));
    };
    const __amdgpu_buffer_rsrc_t rQ = rsrc(queries, (int64_t)nq * dim * 4);   // rows past nq: zeros
    const uint32_t offQ = (uint32_t)(((int64_t)ld_row * dim + ld_c4 * 4) * 4);
    uint32_t offD[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) offD[i] = (uint32_t)(((int64_t)(ld_row + 64 * i) * dim + ld_c4 * 4) * 4);

    // The staging side of the flat loop runs two steps ahead of the matrix side.  Everything a K-step does is STRAIGHT-LINE
    // code -- no branch around a load, a store or the tile bookkeeping: the instruction groups below (sched_group_barrier:
    // LDS writes and global loads dealt out between the MFMAs) only order instructions inside one basic block, and the
    // compiler counts outstanding loads exactly only along straight-line code (with `if (last step of the tile)` branches in
    // the body every K-step waited for ALL its loads, wrote its LDS stage and only then issued its first MFMA: 63 % of the
    // f32 MFMA peak; rounds 3-4).  So: the tile of the next request advances by scalar selects, a request past the
    // workgroup's last tile goes through a descriptor without extent (zeros, no traffic), and the once-per-tile work (norms,
    // epilogue) sits between the K-step loops, not inside them.
    int s_tile = (int)blockIdx.x;  // tile of the step to be requested next
    int s_k = 0, s_par = 0;        // its K-step; parity of the tile whose rows the store side is on
    auto tile_rsrc = [&](int tile) {
        int64_t d0 = (int64_t)tile * tile_stride * MQ_D;
#ifdef KJARNI_TUNING
        if (same_tile) d0 = 0;  // (diagnostic, cosine variant 3: every tile reads the corpus' first 256 rows -- no HBM stream)
#endif
        int64_t rows = n_docs - d0 < MQ_D ? n_docs - d0 : MQ_D;   // documents past n_docs: zeros
        rows = (tile < nt && rows > 0) ? rows : 0;                // past the workgroup's last tile: no extent at all

        return rsrc(corpus + d0 * dim, rows * dim * 4);
    };
    __amdgpu_buffer_rsrc_t rD = tile_rsrc(s_tile);
    f32x4 gq, gd[4];
    auto request = [&]() {  // the next step's rows -> registers; then on to the following step (branch-free)
        const int soff = s_k * MQ_BK * 4;
        gq = ld16(rQ, offQ, soff);
#pragma unroll
        for (int i = 0; i < 4; ++i) gd[i] = ld16(rD, offD[i], soff);
        const bool wrap = s_k + 1 == nk;
        s_k = wrap ? 0 : s_k + 1;
        s_tile = wrap ? s_tile + grid : s_tile;
        rD = tile_rsrc(s_tile);
    };
    // ||doc||^2 on packed pairs (v_pk_fma_f32: two of a piece's four squares per instruction -- the K-loop's only vector
    // arithmetic, and the f32 MFMAs do not hide it): even and odd elements accumulate apart and meet at the tile's end
    f32x2 sumsq[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    auto store = [&](int stage) {  // registers -> LDS stage; the documents' squared norms on the way
        *reinterpret_cast<f32x4*>(sQ + stage * MQ_STAGE_FLOATS + st_off) = gq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(sD + stage * MQ_STAGE_FLOATS + st_off + 64 * i * MQ_STRIDE) = gd[i];
            if (!diag_no_norms) {
                const f32x2 lo = {gd[i][0], gd[i][1]}, hi = {gd[i][2], gd[i][3]};
                sumsq[i] = __builtin_elementwise_fma(lo, lo, sumsq[i]);
                sumsq[i] = __builtin_elementwise_fma(hi, hi, sumsq[i]);
            }
        }
    };
    auto finish_norms = [&]() {  // after the store of a tile's LAST K-step: its norms are complete
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = sumsq[i][0] + sumsq[i][1];
            v += __shfl_xor(v, 1, kWave);
            v += __shfl_xor(v, 2, kWave);
            if (ld_c4 == 0) sDn[s_par * MQ_D + ld_row + 64 * i] = v;
            sumsq[i] = f32x2{0.0f, 0.0f};
        }
        s_par ^= 1;
    };

    if (tid < MQ_Q) {
        sQn[tid] = tid < nq ? sqrtf(qn2[tid]) : 0.0f;
        if (FUSED) {
            float t = -INFINITY;
            if (tid < nq && fuse.thr_idx[(int64_t)(fuse.q_base + tid) * fuse.thr_k + fuse.thr_k - 1] >= 0)
                t = fuse.thr_score[(int64_t)(fuse.q_base + tid) * fuse.thr_k + fuse.thr_k - 1];
            sThr[tid] = t;
            // The cheap test of the epilogue: cosine >= t  <=>  dot / ||doc|| >= t ||q||  (norms above 1e-4: no clamp applies),
            // relaxed by 4e-6 of its magnitude so that rounding can only ADMIT a score the exact comparison then rejects.  No
            // bound (or a query too small for the rule): -inf, everything goes to the exact comparison; rows past nq: +inf.
            float tq = INFINITY;
            if (tid < nq) {
                const float qn = sqrtf(qn2[tid]);
                const float b = t * qn;
                tq = (t == -INFINITY || !(qn >= 1e-4f && qn < INFINITY) || b != b) ? -INFINITY : b - 4e-6f * fabsf(b) - 1e-30f;
            }
            sTq[tid] = tq;
        }
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int a_off = l31 * MQ_STRIDE + half * 4;                 // query rows 0..31 (+32 for the second tile)
    const int b_off = (wid * 64 + l31) * MQ_STRIDE + half * 4;    // this wave's documents
    struct Frag {
        f32x4 a0, a1, b0, b1;
    };
    auto read_frag = [&](Frag& f, int stage, int kk) {
        const float* pa = sQ + stage * MQ_STAGE_FLOATS + a_off + kk * 8;
        const float* pb = sD + stage * MQ_STAGE_FLOATS + b_off + kk * 8;
        f.a0 = *reinterpret_cast<const f32x4*>(pa);
        f.a1 = *reinterpret_cast<const f32x4*>(pa + 32 * MQ_STRIDE);
        f.b0 = *reinterpret_cast<const f32x4*>(pb);
        f.b1 = *reinterpret_cast<const f32x4*>(pb + 32 * MQ_STRIDE);
    };
    auto mfma16 = [&](const Frag& f) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[c], f.b0[c], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[c], f.b1[c], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[c], f.b0[c], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[c], f.b1[c], acc[1][1], 0, 0, 0);
        }
    };

    // prologue: step 0 -> LDS stage 0, step 1 in flight in the registers, fragments kk = 0 of step 0
    request();
    store(0);
    request();
    __syncthreads();
    Frag fr[2];
    read_frag(fr[0], 0, 0);

    // FUSED: this lane's 32 bounds (query = accumulator register x lane half) in registers for the whole launch
    float tq[FUSED ? 32 : 1];
    if (FUSED) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) tq[i * 16 + r] = sTq[i * 32 + acc_row(r, half)];  // (written before the prologue's barrier)
    }
    // FUSED: does one of this lane-half's 32 queries hold a NaN or an infinity?  Its dot products are then NaN (or inf - inf once
    // the -inf bound is added), which the packed max below drops: such a lane takes the exact path for every tile, where a NaN
    // score is "not below the bound", is appended, overflows the list and leaves the query to the two-call form -- whose result
    // (k rows with NaN scores) is the reference's.
    bool odd_query = false;
    if (FUSED) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) odd_query = odd_query || !(sQn[i * 32 + acc_row(r, half)] < INFINITY);
    }

    int c_tile = (int)blockIdx.x;  // tile the matrix side is on
    int c_par = 0, cur = 0;        // parity of its norms; LDS stage of the step being multiplied
    // One K-step (straight-line).  NORMS: the rows stored in this step are a tile's last K-step.
    auto kstep = [&](auto norms_tag) {
        constexpr bool NORMS = decltype(norms_tag)::value;
        // phase 0: fragments of the second half of this step; the next step -> the other LDS stage; request the one after
        read_frag(fr[1], cur, 1);
        store(cur ^ 1);
        request();
        mfma16(fr[0]);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // the 4 fragment reads first
#pragma unroll
        for (int i = 0; i < 5; ++i) {                       // then a staging piece every few MFMAs
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA x 2
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
        }
        __builtin_amdgcn_sched_barrier(0);
        if (NORMS) finish_norms();
        // phase 1: everyone has read stage cur and written stage cur ^ 1
        __syncthreads();
        read_frag(fr[0], cur ^ 1, 0);
        mfma16(fr[1]);
        __builtin_amdgcn_sched_barrier(0);
        cur ^= 1;
    };
    for (int t = 0; t < my_tiles; ++t) {
        // K-steps 0 .. nk - 3; nk - 2 (whose store completes this tile's rows: norms); nk - 1 (stores the next tile's first step)
        for (int ks = 0; ks + 2 < nk; ++ks) kstep(std::false_type{});
        kstep(std::true_type{});
        kstep(std::false_type{});
        {
            // The tile is complete: cosines out, accumulators cleared.  dot / den as one v_rcp_f32 + a Newton step on the
            // quotient (r = 1 / den to 1 ulp, v = dot r, v += (dot - den v) r): the IEEE division sequence without its
            // range scaling, which den = ||q|| ||doc|| never needs; 6 vector instructions per score instead of 15 -- the
            // f32 MFMAs do not hide vector work.
            const int64_t d_base = (int64_t)c_tile * MQ_D + wid * 64 + l31;          // where the score goes (compact in a sample pass)
            const int64_t a_base = (int64_t)c_tile * tile_stride * MQ_D + wid * 64 + l31;  // the document's index
            const bool whole = nq == MQ_Q && ((int64_t)c_tile * tile_stride + 1) * MQ_D <= n_docs;  // no row or column of the tile is cut
            auto cosine_of = [&](float dot, float qn, float dn) { return mq_cosine<MODE>(dot, qn, dn); };
            if (diag_no_epilogue) {
                float keep = 0.0f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            keep += acc[i][j][r];
                            acc[i][j][r] = 0.0f;
                        }
                if (keep == 123456.789f) scores[0] = keep;
            } else if (KIND == SCAN_SCORES) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int64_t d = d_base + j * 32, ad = a_base + j * 32;
                    const float dn = sqrtf(sDn[c_par * MQ_D + wid * 64 + j * 32 + l31]);
                    float* out = scores + d;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int q = i * 32 + acc_row(r, half);
                            const float v = cosine_of(acc[i][j][r], sQn[q], dn);
                            if (whole || (q < nq && ad < n_docs)) out[(int64_t)q * stride] = v;
                            acc[i][j][r] = 0.0f;
                        }
                }
            } else if (KIND == SCAN_SAMPLE_MAX) {
                // the largest cosine of each of this lane-half's 32 queries over the wave's 64 documents (a NaN never wins)
                const float dn0 = sqrtf(sDn[c_par * MQ_D + wid * 64 + l31]), dn1 = sqrtf(sDn[c_par * MQ_D + wid * 64 + 32 + l31]);
                const bool ok0 = a_base < n_docs, ok1 = a_base + 32 < n_docs;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int q = i * 32 + acc_row(r, half);
                        const float qn = sQn[q];
                        const float v0 = cosine_of(acc[i][0][r], qn, dn0), v1 = cosine_of(acc[i][1][r], qn, dn1);
                        float m = fmaxf(ok0 ? v0 : -INFINITY, ok1 ? v1 : -INFINITY);
                        m = fmaxf(m, __shfl_xor(m, 1, kWave));
                        m = fmaxf(m, __shfl_xor(m, 2, kWave));
                        m = fmaxf(m, __shfl_xor(m, 4, kWave));
                        m = fmaxf(m, __shfl_xor(m, 8, kWave));
                        m = fmaxf(m, __shfl_xor(m, 16, kWave));
                        if (l31 == 0 && q < nq) scores[(int64_t)q * stride + (int64_t)c_tile * 4 + wid] = m;
                        acc[i][0][r] = 0.0f;
                        acc[i][1][r] = 0.0f;
                    }
            } else {
                // FUSED.  Nearly every score is below its query's bound, so the common path decides that with ONE multiply and
                // one compare per score -- dot x (1 / ||doc||) against the bound in that domain (tq, relaxed: it may admit, never
                // reject, what the exact comparison would keep; a NaN passes; a document too small for the rule passes) -- and
                // only a block of 32 x 64 scores in which some lane saw a pass computes cosines, compares exactly and appends:
                // the two half-waves of a register belong to two queries, each half appends to its query's list with one atomic.
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int64_t ad = a_base + j * 32;
                    const float dn2 = sDn[c_par * MQ_D + wid * 64 + j * 32 + l31];
                    const float dn = sqrtf(dn2);
                    const bool odd_doc = !(dn >= 1e-4f && dn < INFINITY);   // too small for the rule, or not finite: exact path
                    const float idn = __builtin_amdgcn_rcpf(dn);
                    // margin of every score over its bound, two per instruction (v_pk_fma_f32 / v_pk_max_f32); a pass is a
                    // margin that is not negative.  (A NaN margin is dropped by the max: a NaN dot product needs a non-finite
                    // document -- odd_doc -- or a non-finite query -- odd_query; both take the exact path.)
                    const f32x2 idn2 = {idn, idn};
                    f32x2 top = {-INFINITY, -INFINITY};
#pragma unroll
                    for (int e = 0; e < 32; e += 2) {
                        const f32x2 dots = {acc[e >> 4][j][e & 15], acc[e >> 4][j][(e & 15) + 1]};
                        const f32x2 margin = __builtin_elementwise_fma(dots, idn2, f32x2{-tq[e], -tq[e + 1]});
                        top = __builtin_elementwise_max(top, margin);
                    }
                    const bool any = odd_doc || odd_query || fmaxf(top[0], top[1]) >= 0.0f;
                    if (__ballot(any) != 0ull) {
#pragma unroll
                        for (int e = 0; e < 32; ++e) {
                            // (the cheap test again, per register: at ~1e-4 passes per score a block that has one has ONE, and
                            // the exact cosine + list bookkeeping below run for that register only, not for all 32)
                            const float margin = fmaf(acc[e >> 4][j][e & 15], idn, -tq[e]);
                            if (__ballot(odd_doc || !(margin < 0.0f)) == 0ull) continue;   // (a NaN margin passes)
                            int q = (e >> 4) * 32 + acc_row(e & 15, half);
                            // (opaque: otherwise the 32 list addresses below are hoisted out of the K-loop into 64 registers)
                            asm volatile("" : "+v"(q));
                            const float v = cosine_of(acc[e >> 4][j][e & 15], sQn[q], dn);
                            // (not below the bound: ties and NaN scores go to the exact comparison of the selection)
                            const bool hit = (whole || (q < nq && ad < n_docs)) && !(v < sThr[q]);
                            const uint64_t m = __ballot(hit);
                            if (m == 0ull) continue;
                            const uint32_t mh = half ? (uint32_t)(m >> 32) : (uint32_t)m;
                            const int leader = (half << 5) + (mh ? __ffs((int)mh) - 1 : 0);
                            unsigned base = 0;
                            if (mh != 0u && lane == leader) base = atomicAdd(fuse.cand_count + kCountStride * (1 + fuse.q_base + q), (unsigned)__popc(mh));
                            base = (unsigned)__shfl((int)base, leader, kWave);
                            if (hit) {
                                const unsigned at = base + (unsigned)__popc(mh & ((1u << l31) - 1u));
                                if (at < fuse.cap) fuse.cand_key[(size_t)(fuse.q_base + q) * fuse.cap + at] = make_key(v, (uint32_t)ad);
                                else fuse.cand_count[0] = 1u;  // overflow: the two-call form behind this launch takes over
                            }
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 32; ++e) acc[e >> 4][j][e & 15] = 0.0f;
                }
            }
            c_par ^= 1;
            c_tile += grid;
        }
    }
}

// The bound of a fused search from the sample pass's maxima (SCAN_SAMPLE_MAX): the k-th largest of a query's `n` maxima, or
// "no bound" (index -1) when fewer than k of them are finite.  One block per query, n <= 4 096: the maxima stay in registers as
// 32-bit orderable keys and the k-th largest is found bit by bit from the top -- the largest x with at least k keys >= x --
// 32 counting rounds (sixteen ballots + four LDS words each) instead of a sort.
// `minus` is taken off the result (the bf16 sample's error bound; 0 for the f32 sample).
// Also zeroes the search's candidate counters ([0] overflow word, [1 + q] list lengths): the launch in front of the fused scan.
template <int KPT = 16>   // keys per thread: n <= 256 KPT
__global__ __launch_bounds__(256) void sample_bound_kernel(const float* __restrict__ maxima, int n, int64_t stride, int k,
                                                           float* __restrict__ thr_score, int64_t* __restrict__ thr_idx,
                                                           unsigned* __restrict__ counters, float minus)
{
    __shared__ int wave_count[2][4];
    const int tid = threadIdx.x, q = blockIdx.x, lane = tid & 63, wid = tid >> 6;
    uint32_t key[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const int at = tid + 256 * i;
        const float v = at < n ? maxima[(int64_t)q * stride + at] : -INFINITY;
        key[i] = (v > -INFINITY) ? orderable(v) : 0u;   // (-inf, NaN and padding: 0; a finite score's key is never 0)
    }
    uint32_t found = 0u;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = found | (1u << bit);
        int c = 0;
#pragma unroll
        for (int i = 0; i < KPT; ++i) c += __popcll(__ballot(key[i] >= cand));   // (the wave's count: wave-uniform)
        if (lane == 0) wave_count[bit & 1][wid] = c;
        __syncthreads();   // (two alternating slots: one barrier per round)
        const int total = wave_count[bit & 1][0] + wave_count[bit & 1][1] + wave_count[bit & 1][2] + wave_count[bit & 1][3];
        if (total >= k) found = cand;
    }
    if (tid == 0) {
        thr_idx[q] = found == 0u ? -1 : 0;
        thr_score[q] = found == 0u ? -INFINITY : from_orderable(found) - minus;
        counters[kCountStride * (1 + q)] = 0u;
        if (q == 0) counters[0] = 0u;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Many queries over a large corpus: a bf16 FILTER pass in front of the exact arithmetic.
//
// With a per-query lower bound t of the k-th best score the search only has to find the documents whose cosine reaches t.  The
// f32 matrix cores make that pass MFMA-bound (32 flop per corpus byte); the bf16 matrix cores run at sixteen times their rate, so
// the same pass on bf16-rounded operands is bound by the corpus stream alone -- and its scores are off by a KNOWN amount:
// rounding both operands to 8 significant bits (round to nearest even, u = 2^-8) moves a product by at most (2u + u^2) |q_i d_i|,
// the dot product by at most (2u + u^2) ||q|| ||d|| (Cauchy-Schwarz), f32 accumulation of the bf16 x bf16 products (exact in f32)
// by another ~dim 2^-24.  So with eta = 0.0081 (2^-7 + 2^-16 + slack for the accumulation, the reciprocal and the square root;
// norms in f32 from the unrounded rows):   | dot_bf16 / (||q|| ||d||) - cosine | <= eta.
//  * KIND 1, the sample: the maximum of dot_bf16 / (||q|| ||d||) over each sampled unit (1 .. 16 tiles of 16 documents, about a
//    twentieth of a large corpus, an eighth at most of a small one: filter_plan); the k-th largest of a
//    query's maxima, minus eta, is a score at least k documents reach exactly: the bound t (sample_bound_kernel).
//  * KIND 0, the filter: every (query, document) with  dot_bf16 / ||d|| >= (t - eta) ||q||  -- a few hundred per query -- goes to
//    the list of the WAVE that found it (no atomics: the position is the wave's own count + the lane's rank among the passes);
//    cosine_rescore_kernel then computes the EXACT cosine of those pairs with the f32 MFMA sequence, norm arithmetic and division
//    of cosine_scan_mfma_kernel (bit-identical scores), and the pairs not below t go to the per-query candidate lists that
//    topk_candidates_kernel folds -- the lists the fused f32 scan would have produced.  What the filter cannot judge by the rule
//    -- a document or query with a norm below 1e-4 or not finite, a NaN anywhere -- passes and is decided exactly.
//
// No LDS staging of the corpus, no barrier in the loop.  v_mfma_f32_16x16x32_bf16 takes B as lane (n = lane % 16, g = lane / 16)
// -> 8 values of document n; with the K-step's 32 floats dealt as {4 g .. 4 g + 3} U {16 + 4 g .. 16 + 4 g + 3} the four lanes of
// a document read 64 contiguous bytes per 16-byte load instruction, and the A operand (the 64 queries, rounded once, 4 KB per
// K-step in the same dealing) comes from LDS by ds_read_b128.  A wave owns a tile of 16 documents: all its 2 NK row requests go
// out first (the whole tile in registers; the NEXT tile's as soon as this one's have been consumed, before its scores are looked
// at), then NK x (convert, squares, 4 MFMAs); three waves per SIMD overlap one another's waits.  An accumulator row of 16 lanes is ONE query x the tile's 16 documents.
constexpr float kFilterEta = 0.0081f;
constexpr int FILTER_LIST = 0, FILTER_SAMPLE_MAX = 1;
constexpr size_t kFilterListBytes = (size_t)16 << 20;   // the filter pass's (query, document) lists, all waves together
constexpr size_t kFilterMaxWaves = 4096;                // >= 4 x its largest grid (768 workgroups)
typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cbf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t cu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t bf16_pair(float a, float b)
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, cbf16x2));  // v_cvt_pk_bf16_f32, round to nearest even
}
__device__ __forceinline__ cbf16x8 bf16_eight(const f32x4 x0, const f32x4 x1)
{
    const cu32x4 u = {bf16_pair(x0[0], x0[1]), bf16_pair(x0[2], x0[3]), bf16_pair(x1[0], x1[1]), bf16_pair(x1[2], x1[3])};
    return __builtin_bit_cast(cbf16x8, u);
}

struct FilterOut {
    uint64_t* wave_list;    // FILTER_LIST: [waves, cap_w] entries (query of the whole call << 32 | document)
    unsigned* wave_count;   // [waves]: entries of each wave's list (written by every wave, also 0)
    unsigned cap_w;
    unsigned* overflow;     // raised when a list is full: the two-call form behind the search answers
    int q_base;             // this launch's first query
    float* maxima;          // FILTER_SAMPLE_MAX: [nq, max_stride]: per (query, sampled tile) the largest dot_bf16 / (||q|| ||d||)
    int64_t max_stride;
    int64_t units;          // FILTER_SAMPLE_MAX: sample units; unit u = unit_tiles consecutive tiles of 16 documents from tile
    int tile_stride;        //   u * tile_stride on
    int unit_tiles;
};

// PARTS > 1 (widths above 512: the tile's rows no longer fit the registers): the tile goes through in PARTS pieces of NK / PARTS
// K-steps -- a piece is requested when the piece before it has been consumed --, and the 96 / 128 KB of query fragments leave
// room for ONE workgroup per CU, which therefore has WAVES = 12 (8 at 1 024) waves.
template <int NK, int KIND, int PARTS = 1, int WAVES = 4>   // dim = 32 NK
__global__ __launch_bounds__(64 * WAVES, WAVES > 4 ? WAVES / 4 : ((NK / PARTS <= 12 && KIND == FILTER_LIST) ? 3 : 2)) void cosine_filter_bf16_kernel(const float* __restrict__ queries, int nq,
                                                                                  const float* __restrict__ corpus, int64_t n_docs,
                                                                                  const float* __restrict__ qn2,
                                                                                  const float* __restrict__ thr_score,
                                                                                  const int64_t* __restrict__ thr_idx, int thr_k,
                                                                                  FilterOut out)
{
    constexpr int dim = 32 * NK, NP = NK / PARTS;
    static_assert(NK % PARTS == 0, "whole K-steps per piece");
    extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
    cu32x4* sQf = reinterpret_cast<cu32x4*>(fsm);                 // [NK][4 query blocks][64 lanes]: A fragments, ready to use
    float* sTq = reinterpret_cast<float*>(fsm + NK * 4096);        // [64]: the bound in the dot / ||doc|| domain (sample: 1 / ||q||)
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, g = lane >> 4;
    for (int f = tid; f < NK * 256; f += 64 * WAVES) {
        const int s = f >> 8, b = (f >> 6) & 3, l = f & 63;
        const int q = 16 * b + (l & 15);
        f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = {0.f, 0.f, 0.f, 0.f};
        if (q < nq) {
            const float* p = queries + (int64_t)q * dim + 32 * s + 4 * (l >> 4);
            x0 = *reinterpret_cast<const f32x4*>(p);
            x1 = *reinterpret_cast<const f32x4*>(p + 16);
        }
        sQf[f] = __builtin_bit_cast(cu32x4, bf16_eight(x0, x1));
    }
    if (tid < 64) {
        float tq;
        if (KIND == FILTER_SAMPLE_MAX) {
            // 1 / ||q||; a query the rule does not cover (norm below 1e-4, not finite) or a row past nq: NaN -- its maxima come
            // out as "no bound" and every document of it goes through the exact pass
            const float qn = tid < nq ? sqrtf(qn2[tid]) : 0.0f;
            tq = (qn >= 1e-4f && qn < INFINITY) ? 1.0f / qn : __builtin_nanf("");
        } else {
            // rows past nq: +inf (nothing passes).  No bound, a query too small for the rule, not finite: -inf (everything passes
            // to the exact pass; the lists overflow and the two-call form answers, as for the f32 scan).
            tq = INFINITY;
            if (tid < nq) {
                float t = -INFINITY;
                if (thr_idx[(int64_t)tid * thr_k + thr_k - 1] >= 0) t = thr_score[(int64_t)tid * thr_k + thr_k - 1];
                const float qn = sqrtf(qn2[tid]);
                const float b = (t - kFilterEta) * qn;
                tq = (t == -INFINITY || !(qn >= 1e-4f && qn < INFINITY) || b != b) ? -INFINITY : b - 1e-5f * fabsf(b) - 1e-30f;
            }
        }
        sTq[tid] = tq;
    }
    __syncthreads();
    // this lane's 16 queries: block b, accumulator register r -> query 16 b + 4 g + r; their bounds as 16 consecutive floats per
    // lane group (read back per tile with four 16-byte LDS loads: 16 registers the tile's rows need more)
    float* sTl = sTq + 64;   // [4 groups][16]
    if (tid < 64) sTl[16 * (tid >> 4) + (tid & 15)] = sTq[16 * ((tid & 15) >> 2) + 4 * (tid >> 4) + (tid & 3)];
    __syncthreads();
    auto bounds = [&](float (&tq)[16]) {
        asm volatile("" ::: "memory");   // (read them here, every tile: hoisted out of the tile loop they are 16 live registers)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(sTl + 16 * g + 4 * b);
#pragma unroll
            for (int r = 0; r < 4; ++r) tq[4 * b + r] = v[r];
        }
    };

    const int64_t all_tiles = (n_docs + 15) >> 4;
    const int64_t waves = (int64_t)gridDim.x * WAVES, wave = (int64_t)blockIdx.x * WAVES + wid;
    const uint32_t voff = (uint32_t)((n16 * dim + 4 * g) * 4);
    // one piece (NP K-steps) of a tile of 16 documents: its rows requested (a tile past the corpus: a descriptor without extent
    // -- zeros, no traffic) ...
    auto request = [&](int64_t tile, int part, f32x4 (&x0)[NP], f32x4 (&x1)[NP]) {
        const int64_t d0 = tile << 4;
        int64_t rows = n_docs - d0 < 16 ? n_docs - d0 : 16;   // documents past n_docs: zeros
        rows = rows > 0 ? rows : 0;
        const __amdgpu_buffer_rsrc_t rD =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(corpus + (rows > 0 ? d0 : 0) * dim), 0, (int)(rows * dim * 4), 0x00020000);
        const uint32_t vo = voff + (uint32_t)(128 * NP * part);
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            // (plain loads: the two requests of a K-step take the two halves of the same 128-byte lines, the second finds them in
            // the L1 -- with the non-temporal bit it does not: 5.7 instead of 6.4 TB/s for this pattern alone,
            // tools/lab/stream_pattern_lab.hip)
            x0[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rD, vo + 128 * s, 0, 0));
            x1[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rD, vo + 128 * s + 64, 0, 0));
        }
    };
    // ... and its share of the approximate dot products of this lane's document with its 16 queries, and of ||doc||^2
    auto dots = [&](int part, const f32x4 (&x0)[NP], const f32x4 (&x1)[NP], f32x4 (&acc)[4], f32x2& sq) {
        const cu32x4* frag = sQf + (size_t)part * NP * 256 + lane;
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            const cbf16x8 bf = bf16_eight(x0[s], x1[s]);
            sq = __builtin_elementwise_fma(f32x2{x0[s][0], x0[s][1]}, f32x2{x0[s][0], x0[s][1]}, sq);
            sq = __builtin_elementwise_fma(f32x2{x0[s][2], x0[s][3]}, f32x2{x0[s][2], x0[s][3]}, sq);
            sq = __builtin_elementwise_fma(f32x2{x1[s][0], x1[s][1]}, f32x2{x1[s][0], x1[s][1]}, sq);
            sq = __builtin_elementwise_fma(f32x2{x1[s][2], x1[s][3]}, f32x2{x1[s][2], x1[s][3]}, sq);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const cbf16x8 af = __builtin_bit_cast(cbf16x8, frag[(s * 4 + b) * 64]);
                acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[b], 0, 0, 0);
            }
        }
        // (the squares are one dependent chain that the scheduler would finish long after the MFMAs, holding the piece's raw rows
        // -- past the request that refills their registers: spills; the chain ends HERE)
        if (PARTS > 1) asm volatile("" : "+v"(sq));
    };
    // A whole tile: its pieces in turn through ONE set of registers -- a piece is requested as soon as the piece before it has
    // been consumed, the LAST request is the first piece of `next` (the wave's next tile: on its way while this tile's scores
    // are looked at).  x0 / x1 hold the tile's first piece on entry and `next`'s on return.
    auto tile_dots = [&](int64_t tile, int64_t next, f32x4 (&x0)[NP], f32x4 (&x1)[NP], f32x4 (&acc)[4], float& idn, bool& odd_doc) {
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x2 sq = {0.f, 0.f};
#pragma unroll
        for (int part = 0; part < PARTS; ++part) {
            dots(part, x0, x1, acc, sq);
            // (a fence for the instruction scheduler: it would start the next piece's requests early, into registers of their own)
            if (PARTS > 1) __builtin_amdgcn_sched_barrier(0);
            if (part + 1 < PARTS) request(tile, part + 1, x0, x1);
            else request(next, 0, x0, x1);
            if (PARTS > 1) __builtin_amdgcn_sched_barrier(0);
        }
        // ||doc||^2: the four lanes of a document (g = 0 .. 3) hold a quarter each
        const float dn = sqrtf(sum_xor32(sum_xor16(sq[0] + sq[1])));
        odd_doc = !(dn >= 1e-4f && dn < INFINITY);   // too small for the rule, or not finite: the exact pass decides
        idn = __builtin_amdgcn_rcpf(dn);
    };
    if constexpr (KIND == FILTER_SAMPLE_MAX) {
        // unit `it` = the unit_tiles consecutive tiles from tile it * tile_stride on
        f32x4 x0[NP], x1[NP];
        request(wave < out.units ? wave * out.tile_stride : all_tiles, 0, x0, x1);
        for (int64_t it = wave; it < out.units; it += waves) {
            float run[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) run[e] = -INFINITY;
#pragma unroll 1
            for (int u = 0; u < out.unit_tiles; ++u) {
                const int64_t tile = it * out.tile_stride + u;   // (past the corpus: zeros, which give no bound)
                const int64_t next = u + 1 < out.unit_tiles ? tile + 1 : (it + waves < out.units ? (it + waves) * out.tile_stride : all_tiles);
                f32x4 acc[4];
                float idn;
                bool odd_doc;
                tile_dots(tile, next, x0, x1, acc, idn, odd_doc);
                const int64_t ad = (tile << 4) + n16;
                // documents the rule does not cover, lanes past n_docs and NaN dot products give no bound (fmaxf drops a NaN)
                const bool ok = ad < n_docs && !odd_doc;
                float tq[16];
                bounds(tq);
#pragma unroll
                for (int e = 0; e < 16; ++e) run[e] = fmaxf(run[e], ok ? acc[e >> 2][e & 3] * idn * tq[e] : -INFINITY);
            }
            // per query (a row of 16 lanes) the largest approximate cosine of the unit
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float m = row16_max_desc(run[e]);
                const int q = 16 * (e >> 2) + 4 * g + (e & 3);
                if (n16 == 0 && q < nq) out.maxima[(int64_t)q * out.max_stride + it] = m;
            }
        }
        return;
    }
    unsigned count = 0;   // entries in this wave's list (wave-uniform)
    f32x4 x0[NP], x1[NP];
    request(wave, 0, x0, x1);
    for (int64_t it = wave; it < all_tiles; it += waves) {
        f32x4 acc[4];
        float idn;
        bool odd_doc;
        tile_dots(it, it + waves, x0, x1, acc, idn, odd_doc);
        const int64_t ad = (it << 4) + n16;
        // bit e of `mask`: this lane's document passes for its query e (a NaN margin passes; a document the rule does not cover
        // passes for every real query).  Nearly every tile ends at the ballot; in one that does not, the lanes that have passes
        // -- one, as a rule -- take turns to write their entries at the wave's count.
        float tq[16];
        bounds(tq);
        uint32_t mask = 0u;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            mask |= (odd_doc || !(fmaf(acc[e >> 2][e & 3], idn, -tq[e]) < 0.0f)) && tq[e] < INFINITY ? 1u << e : 0u;
        if (ad >= n_docs) mask = 0u;
        uint64_t lanes = __ballot(mask != 0u);
        if (lanes == 0ull) continue;
        const unsigned mine = (unsigned)__popc(mask);
#pragma nounroll
        while (lanes != 0ull) {
            const int leader = __ffsll((long long)lanes) - 1;
            const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)mine, leader);
            if (lane == leader) {
                unsigned at = count;
#pragma nounroll
                for (uint32_t mm = mask; mm != 0u; mm &= mm - 1u, ++at) {
                    int e = __ffs((int)mm) - 1;
                    asm volatile("" : "+v"(e));   // (opaque: nothing of this path is prepared outside the tile loop)
                    const int q = out.q_base + 16 * (e >> 2) + 4 * g + (e & 3);
                    if (at < out.cap_w) out.wave_list[(size_t)wave * out.cap_w + at] = ((uint64_t)(uint32_t)q << 32) | (uint64_t)(uint32_t)ad;
                    else *out.overflow = 1u;   // the two-call form behind this search takes over
                }
            }
            count += c;
            lanes &= lanes - 1ull;
        }
    }
    if (lane == 0) out.wave_count[wave] = count < out.cap_w ? count : out.cap_w;
}

// The exact pass of the filtered search: the filter waves' lists taken as ONE sequence (an exclusive prefix of their lengths in
// LDS, a binary search per pair), 32 (query, document) pairs per wave and step -- full groups whatever the lists' lengths.  The dot
// products are taken with the f32 MFMA sequence of cosine_scan_mfma_kernel -- v_mfma_f32_32x32x2_f32 over k pairs (16 ks + 8 kk + c,
// + 4), c = 0 .. 3, kk = 0, 1, ks ascending -- with pair m's query as row m of A and pair n's document as column n of B: the
// DIAGONAL of the 32 x 32 block holds the pairs' dot products (an output element depends on its own row and column only), and
// ||doc||^2 comes from that kernel's partial sums: per 16-byte column c4 of a K-step an (even, odd) pair of fma chains over the
// K-steps, v_c4 = even + odd, then (v0 + v1) + (v2 + v3).  Same bits as the scan's scores; the pairs not below the query's bound go
// to its candidate list for topk_candidates_kernel.
template <int MODE, int NK>   // dim = 32 NK
__global__ __launch_bounds__(256) void cosine_rescore_kernel(const float* __restrict__ queries, const float* __restrict__ corpus,
                                                             const float* __restrict__ qn2, const float* __restrict__ thr_score,
                                                             const int64_t* __restrict__ thr_idx, int thr_k,
                                                             const uint64_t* __restrict__ wave_list,
                                                             const unsigned* __restrict__ wave_count, int n_lists, unsigned cap_w,
                                                             unsigned* __restrict__ counters, unsigned cap, uint64_t* __restrict__ cand_key)
{
    constexpr int dim = 32 * NK;
    __shared__ unsigned pre[kFilterMaxWaves + 1];   // exclusive prefix of the lists' lengths: pair p of the search is entry
    __shared__ unsigned wave_total[4];              //   p - pre[w] of list w, pre[w] <= p < pre[w + 1]
    if (counters[0] != 0u) return;   // a list overflowed: the two-call form answers
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    {
        // 16 consecutive lists per thread (n_lists <= 4 096), a shuffle scan over the wave, the four wave totals through LDS
        unsigned c[16], sum = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int w = 16 * tid + j;
            c[j] = w < n_lists ? (wave_count[w] < cap_w ? wave_count[w] : cap_w) : 0u;
            sum += c[j];
        }
        unsigned incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned up = (unsigned)__shfl_up((int)incl, off, kWave);
            if (lane >= off) incl += up;
        }
        if (lane == 63) wave_total[wid] = incl;
        __syncthreads();
        unsigned base = incl - sum;
        for (int w = 0; w < wid; ++w) base += wave_total[w];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            pre[16 * tid + j] = base;
            base += c[j];
        }
        if (tid == 255) pre[4096] = base;
        __syncthreads();
    }
    const unsigned total = pre[4096];
    const int l31 = lane & 31, half = lane >> 5;
    // the diagonal element of column l31 sits in the lane half (l31 >> 2) & 1, accumulator register (l31 & 3) + 4 (l31 >> 3)
    const bool holds_diag = half == ((l31 >> 2) & 1);
    const int diag_reg = (l31 & 3) + 4 * (l31 >> 3);
    for (unsigned g0 = ((unsigned)blockIdx.x * 4u + (unsigned)wid) * 32u; g0 < total; g0 += gridDim.x * 128u) {
        const bool valid = g0 + (unsigned)l31 < total;
        const unsigned pi = valid ? g0 + (unsigned)l31 : g0;   // (past the end: the group's first pair again)
        int lo = 0, hi = 4096;   // pre[lo] <= pi < pre[hi]
#pragma unroll 1
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= pi) lo = mid;
            else hi = mid;
        }
        const uint64_t pr = wave_list[(size_t)lo * cap_w + (pi - pre[lo])];
        const int q = (int)(pr >> 32);
        const uint32_t d = (uint32_t)pr;
        const float* qp = queries + (int64_t)q * dim + 4 * half;
        const float* dp = corpus + (int64_t)d * dim + 4 * half;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        float ev[2] = {0.f, 0.f}, od[2] = {0.f, 0.f};
        // (the rows are cold and scattered: a batch of K-steps is requested together -- step by step every K-step would be a
        // dependent round trip of its own)
        constexpr int CH = (dim / 16) % 12 == 0 ? 12 : 8;   // K-steps per batch of requests (384: two batches of 24 + 24 requests)
        static_assert((dim / 16) % CH == 0, "dim is a multiple of 128");
#pragma unroll 1
        for (int k0 = 0; k0 < dim / 16; k0 += CH) {
            f32x4 a[2 * CH], b[2 * CH];
#pragma unroll
            for (int j = 0; j < 2 * CH; ++j) {
                a[j] = *reinterpret_cast<const f32x4*>(qp + 16 * k0 + 8 * j);
                b[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dp + 16 * k0 + 8 * j));
            }
#pragma unroll
            for (int j = 0; j < 2 * CH; ++j) {   // j = 2 (ks - k0) + kk
                const int kk = j & 1;
                ev[kk] = fmaf(b[j][0], b[j][0], ev[kk]);
                od[kk] = fmaf(b[j][1], b[j][1], od[kk]);
                ev[kk] = fmaf(b[j][2], b[j][2], ev[kk]);
                od[kk] = fmaf(b[j][3], b[j][3], od[kk]);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][cc], b[j][cc], acc, 0, 0, 0);
            }
        }
        const float v01 = sum_xor32(ev[0] + od[0]), v23 = sum_xor32(ev[1] + od[1]);   // columns (0, 1) and (2, 3) of the K-step
        const float dn = sqrtf(v01 + v23);
        float dot = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) dot = r == diag_reg ? acc[r] : dot;
        const float v = mq_cosine<MODE>(dot, sqrtf(qn2[q]), dn);
        float t = -INFINITY;
        if (thr_idx[(int64_t)q * thr_k + thr_k - 1] >= 0) t = thr_score[(int64_t)q * thr_k + thr_k - 1];
        if (valid && holds_diag && !(v < t)) {   // (not below the bound: ties and NaN scores go to the selection)
            const unsigned at = atomicAdd(counters + kCountStride * (1 + q), 1u);
            if (at < cap) cand_key[(size_t)q * cap + at] = make_key(v, d);
            else counters[0] = 1u;
        }
    }
}

// The same bound by ONE WAVE per query (four queries per workgroup) -- no barrier between the 32 counting rounds, which are what
// sample_bound_kernel's 14 us consist of: the wave holds all n <= 64 KPL maxima (KPL keys per lane), a round is KPL compare-and-
// counts per lane and one wave sum.  Used for the filtered search's sample when n <= 1 024 (KPL = 16; with 64 keys per lane the
// block kernel wins: 37 against 14 us).
template <int KPL>
__global__ __launch_bounds__(256) void sample_bound_wave_kernel(const float* __restrict__ maxima, int n, int64_t stride, int k, int nq,
                                                                float* __restrict__ thr_score, int64_t* __restrict__ thr_idx,
                                                                unsigned* __restrict__ counters, float minus)
{
    const int lane = threadIdx.x & 63, q = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (q >= nq) return;
    uint32_t key[KPL];
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        const int at = lane + 64 * i;
        const float v = at < n ? maxima[(int64_t)q * stride + at] : -INFINITY;
        key[i] = (v > -INFINITY) ? orderable(v) : 0u;   // (-inf, NaN and padding: 0; a finite score's key is never 0)
    }
    uint32_t found = 0u;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = found | (1u << bit);
        // (every lane counts its own keys on the vector ALU, one DPP sum per round: a ballot + scalar count per key is a chain
        // of vector-to-scalar hand-offs -- 58 us for 64 keys per lane; counts up to 4 096 are exact in f32)
        float c[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // (four short chains instead of one of KPL dependent adds)
#pragma unroll
        for (int i = 0; i < KPL; ++i) c[i & 3] += key[i] >= cand ? 1.0f : 0.0f;
        if (wave_sum((c[0] + c[1]) + (c[2] + c[3])) >= (float)k) found = cand;
    }
    if (lane == 0) {
        thr_idx[q] = found == 0u ? -1 : 0;
        thr_score[q] = found == 0u ? -INFINITY : from_orderable(found) - minus;
        counters[kCountStride * (1 + q)] = 0u;
        if (q == 0) counters[0] = 0u;
    }
}

// One streaming pass over the corpus per group of SCAN_NQ queries.
hipError_t scan_passes(const float* queries, int nq, const float* corpus, int64_t n_docs, int dim, int mode,
                       float* scores, int64_t score_stride, hipStream_t stream);

// Many queries: blocks of 64 through the fused matrix-core scan (one corpus read per block).
//   tile_stride > 1: the sample pass (scores of every tile_stride-th tile, compact, row length `score_stride`)
//   fuse != null:    selection inside the scan (no scores): candidates into fuse's list
//   run_flag:        the launches run only if *run_flag != 0
//   sample_max:      (with tile_stride) per-(query, tile, wave) maxima [nq, n_tiles * 4] into `scores` instead of the scores
//   qn2_given:       the queries' squared norms [nq] (query_sqnorms below), computed once by a caller that makes several passes
// One wave per query, 16-byte pieces dealt lane by lane, an fma chain per lane, one wave sum (the arithmetic of the streaming
// kernels' own query norm): a query's norm -- and with it its scores -- does not depend on how many queries share its call (as a
// pass of the streaming scan over the query rows, rows of a last incomplete group were summed in another order).
__global__ __launch_bounds__(256) void query_sqnorm_kernel(const float* __restrict__ queries, int nq, int dim, float* __restrict__ qn2)
{
    const int lane = threadIdx.x & 63, q = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (q >= nq) return;
    const float* row = queries + (int64_t)q * dim;
    float s = 0.0f;
    if ((dim & 3) == 0 && (reinterpret_cast<uintptr_t>(queries) & 15) == 0) {
        for (int c4 = lane; c4 < (dim >> 2); c4 += 64) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * c4);
#pragma unroll
            for (int c = 0; c < 4; ++c) s = fmaf(v[c], v[c], s);
        }
    } else {
        for (int c = lane; c < dim; c += 64) s = fmaf(row[c], row[c], s);
    }
    s = wave_sum(s);
    if (lane == 0) qn2[q] = s;
}

hipError_t query_sqnorms(const float* queries, int nq, int dim, float* qn2, hipStream_t stream)
{
    hipLaunchKernelGGL(query_sqnorm_kernel, dim3((unsigned)(nq + 3) / 4), dim3(256), 0, stream, queries, nq, dim, qn2);
    return hipGetLastError();
}

hipError_t scan_mfma(const float* queries, int nq, const float* corpus, int64_t n_docs, int dim, int mode,
                     float* scores, hipStream_t stream, int tile_stride = 1, int64_t score_stride = -1, const ScanFuse* fuse = nullptr,
                     const unsigned* run_flag = nullptr, bool sample_max = false, const float* qn2_given = nullptr)
{
    float* qn2 = const_cast<float*>(qn2_given);
    hipError_t e = hipSuccess;
    if (!qn2_given) {
        e = hipMallocAsync(reinterpret_cast<void**>(&qn2), (size_t)(nq + 8) * sizeof(float), stream);
        if (e != hipSuccess) return e;
        e = query_sqnorms(queries, nq, dim, qn2, stream);
    }
    const int64_t all_tiles = (n_docs + MQ_D - 1) / MQ_D;
    const int64_t n_tiles = (all_tiles + tile_stride - 1) / tile_stride;
    if (score_stride < 0) score_stride = n_docs;
    const unsigned grid = (unsigned)std::min<int64_t>(n_tiles, 256 * 2);  // the workgroups the chip holds at once (two per CU)
    for (int q0 = 0; q0 < nq && e == hipSuccess; q0 += MQ_Q) {
        const int m = nq - q0 < MQ_Q ? nq - q0 : MQ_Q;
        ScanFuse f = fuse ? *fuse : ScanFuse{};
        f.q_base = q0;
        float* sc = scores ? scores + (int64_t)q0 * score_stride : nullptr;
#define KJ_SCAN(MODE_, KIND_)                                                                                                       \
    hipLaunchKernelGGL((cosine_scan_mfma_kernel<MODE_, KIND_>), dim3(grid), dim3(256), MQ_LDS_BYTES, stream, queries + (int64_t)q0 * dim, \
                       m, corpus, n_docs, dim, qn2 + q0, sc, score_stride, n_tiles, (tune::scan_diag() && tile_stride == 1) ? -tune::scan_diag() : tile_stride, run_flag, f)
        if (fuse) {
            if (mode == 0) KJ_SCAN(0, SCAN_FUSED);
            else KJ_SCAN(1, SCAN_FUSED);
        } else if (sample_max) {
            if (mode == 0) KJ_SCAN(0, SCAN_SAMPLE_MAX);
            else KJ_SCAN(1, SCAN_SAMPLE_MAX);
        } else {
            if (mode == 0) KJ_SCAN(0, SCAN_SCORES);
            else KJ_SCAN(1, SCAN_SCORES);
        }
#undef KJ_SCAN
        e = hipGetLastError();
    }
    const hipError_t fe = qn2_given ? hipSuccess : hipFreeAsync(qn2, stream);
    return e != hipSuccess ? e : fe;
}

// The bf16 passes of a many-query search: blocks of 64 queries, 3 workgroups per CU (dim <= 384; 2 at 512).
//   sample (maxima != null): per (query, unit) maxima [nq, units]
//   filter: per-wave lists, then the exact pass over them (rescore), query block by query block
struct FilterPlan {
    int nk;
    int waves;           // per workgroup: 4; widths above 512 (one workgroup per CU: the query fragments take 96 / 128 KB) 12 / 8
    size_t lds;
    unsigned grid;       // workgroups of the filter launch (the lists are per wave: waves x grid of them)
    unsigned cap_w;      // entries per wave list
    int64_t units;       // the sample's units, their first tiles tile_stride apart, unit_tiles tiles of 16 documents each
    int tile_stride, unit_tiles;
};
inline bool filter_width(int dim) { return dim == 128 || dim == 256 || dim == 384 || dim == 512 || dim == 768 || dim == 1024; }
FilterPlan filter_plan(int64_t n_docs, int dim, size_t list_bytes)
{
    FilterPlan p{};
    p.nk = dim / 32;
    p.waves = p.nk <= 16 ? 4 : (p.nk == 24 ? 12 : 8);
    p.lds = (size_t)p.nk * 4096 + 512;
    const int64_t tiles = (n_docs + 15) / 16;
    const int per_cu = p.nk <= 12 ? 3 : (p.nk <= 16 ? 2 : 1);   // workgroups a CU holds
    p.grid = (unsigned)std::min<int64_t>((tiles + p.waves - 1) / p.waves, 256 * per_cu);
#ifdef KJARNI_TUNING
    if (const char* e = getenv("KJARNI_HIP_FILTER_GRID")) p.grid = (unsigned)std::min<int64_t>((tiles + p.waves - 1) / p.waves, std::min(4096 / p.waves, atoi(e)));
#endif
    p.cap_w = (unsigned)(list_bytes / 8 / ((size_t)p.grid * p.waves));
    // ~1 / 20 of a large corpus (10^7 documents: 8 tiles per unit 2.68 ms, 4: 2.73, 16: 2.75) and at least 4 096 tiles (fewer only
    // when the corpus has fewer); at most 4 096 units (sample_bound_kernel)
    p.unit_tiles = (int)std::min<int64_t>(16, std::max<int64_t>(1, (tiles + 4096 * 20 - 1) / (4096 * 20)));
    // (4 096 units is also where the search is fastest: 2 048 / 1 024 / 512 units cost 10^6 documents 0.399 / 0.459 / 0.576 ms against
    // 0.360 -- more candidates --, 6 144 / 8 192 0.379 / 0.390 against 0.366 -- a longer sample; docs/history/r06.md 4b)
    p.units = std::min<int64_t>(4096, std::max<int64_t>(1, tiles / p.unit_tiles / 8));   // (and at most 1 / 8 of a small corpus)
    p.tile_stride = (int)std::max<int64_t>(p.unit_tiles, tiles / p.units);
    return p;
}

template <int KIND>
hipError_t filter_launch(const FilterPlan& p, const float* queries, int m, const float* corpus, int64_t n_docs, const float* qn2,
                         const float* thr_score, const int64_t* thr_idx, int thr_k, const FilterOut& out, hipStream_t stream)
{
    // (the sample: one round of the workgroups a CU holds at its register budget -- two of four waves, one of 12 / 8)
    const unsigned grid = KIND == FILTER_SAMPLE_MAX ? (unsigned)std::min<int64_t>((p.units + p.waves - 1) / p.waves, p.waves == 4 ? 512 : 256)
                                                    : p.grid;
#define KJ_FILTER(NK_, PARTS_, WAVES_)                                                                                           \
    do {                                                                                                                         \
        auto kern = cosine_filter_bf16_kernel<NK_, KIND, PARTS_, WAVES_>;                                                        \
        if (p.lds > 48 * 1024) {                                                                                                 \
            const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                      (int)p.lds);                                                               \
            if (ea != hipSuccess) return ea;                                                                                     \
        }                                                                                                                        \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES_), p.lds, stream, queries, m, corpus, n_docs, qn2, thr_score, thr_idx, thr_k, \
                           out);                                                                                                 \
    } while (0)
    switch (p.nk) {
    case 4: KJ_FILTER(4, 1, 4); break;
    case 8: KJ_FILTER(8, 1, 4); break;
    case 12: KJ_FILTER(12, 1, 4); break;
    case 16: KJ_FILTER(16, 1, 4); break;
    case 24:   // (768: two pieces of 12 K-steps for the filter; the sample, whose epilogue holds 16 running maxima more, four of 6)
        if (KIND == FILTER_SAMPLE_MAX) KJ_FILTER(24, 4, 12);
        else KJ_FILTER(24, 2, 12);
        break;
    case 32:
        if (KIND == FILTER_SAMPLE_MAX) KJ_FILTER(32, 4, 8);
        else KJ_FILTER(32, 2, 8);
        break;
    default: return hipErrorInvalidValue;
    }
#undef KJ_FILTER
    return hipGetLastError();
}

hipError_t filter_sample(const FilterPlan& p, const float* queries, int nq, const float* corpus, int64_t n_docs, int dim, const float* qn2,
                         float* maxima, hipStream_t stream)
{
    for (int q0 = 0; q0 < nq; q0 += MQ_Q) {
        FilterOut out{};
        out.maxima = maxima + (int64_t)q0 * p.units;
        out.max_stride = p.units;
        out.units = p.units;
        out.tile_stride = p.tile_stride;
        out.unit_tiles = p.unit_tiles;
        const hipError_t e = filter_launch<FILTER_SAMPLE_MAX>(p, queries + (int64_t)q0 * dim, std::min(MQ_Q, nq - q0), corpus, n_docs, qn2 + q0,
                                                              nullptr, nullptr, 1, out, stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t filter_and_rescore(const FilterPlan& p, const float* queries, int nq, const float* corpus, int64_t n_docs, int dim, int mode,
                              const float* qn2, const float* thr_score, const int64_t* thr_idx, int thr_k, uint64_t* wave_list,
                              unsigned* wave_count, unsigned* counters, unsigned cap_q, uint64_t* cand_key, hipStream_t stream)
{
    for (int q0 = 0; q0 < nq; q0 += MQ_Q) {
        FilterOut out{};
        out.wave_list = wave_list;
        out.wave_count = wave_count;
        out.cap_w = p.cap_w;
        out.overflow = counters;
        out.q_base = q0;
        hipError_t e = filter_launch<FILTER_LIST>(p, queries + (int64_t)q0 * dim, std::min(MQ_Q, nq - q0), corpus, n_docs, qn2 + q0,
                                                  thr_score + (int64_t)q0 * thr_k, thr_idx + (int64_t)q0 * thr_k, thr_k, out, stream);
        if (e != hipSuccess) return e;
        // (queries, bounds and lists of the WHOLE call: the list entries carry the call's query index)
#define KJ_RESCORE(MODE_, NK_)                                                                                                   \
    hipLaunchKernelGGL((cosine_rescore_kernel<MODE_, NK_>), dim3(256), dim3(256), 0, stream, queries, corpus, qn2, thr_score, thr_idx, \
                       thr_k, wave_list, wave_count, (int)p.grid * p.waves, p.cap_w, counters, cap_q, cand_key)
#define KJ_RESCORE_M(NK_)                                                                                                        \
    do {                                                                                                                         \
        if (mode == 0) KJ_RESCORE(0, NK_);                                                                                       \
        else KJ_RESCORE(1, NK_);                                                                                                 \
    } while (0)
        switch (p.nk) {
        case 4: KJ_RESCORE_M(4); break;
        case 8: KJ_RESCORE_M(8); break;
        case 12: KJ_RESCORE_M(12); break;
        case 16: KJ_RESCORE_M(16); break;
        case 24: KJ_RESCORE_M(24); break;
        default: KJ_RESCORE_M(32); break;
        }
#undef KJ_RESCORE_M
#undef KJ_RESCORE
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace

hipError_t launch_cosine_scores(const float* queries, int nq, const float* corpus, int64_t n_docs,
                                int dim, int mode, float* scores, hipStream_t stream)
{
    if (nq <= 0 || n_docs <= 0) return hipSuccess;
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(corpus) & 15) == 0) &&
                           ((reinterpret_cast<uintptr_t>(queries) & 15) == 0) &&
                           ((reinterpret_cast<uintptr_t>(scores) & 15) == 0);
    // Crossover: a streaming pass serves 4 queries, the matrix-core scan 64 at about the cost of two passes.
    // (dim >= 2 K-steps: with a single K-step per tile the double-buffered row norms of tile T + 2 would be stored while slower
    // waves still read tile T's in their epilogue)
    if (nq >= 20 && aligned16 && dim % MQ_BK == 0 && dim >= 2 * MQ_BK && (int64_t)MQ_D * dim * 4 < ((int64_t)1 << 31) && (mode == 0 || mode == 1) &&
        !tune::scan_streaming_only())
        return scan_mfma(queries, nq, corpus, n_docs, dim, mode, scores, stream);
    return scan_passes(queries, nq, corpus, n_docs, dim, mode, scores, n_docs, stream);
}

namespace {

hipError_t scan_passes(const float* queries, int nq, const float* corpus, int64_t n_docs, int dim, int mode,
                       float* scores, int64_t score_stride, hipStream_t stream)
{
    const bool aligned = ((reinterpret_cast<uintptr_t>(corpus) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(queries) & 15) == 0);
    const bool fast = (dim % 4 == 0) && dim <= 256 * SCAN_MAX_V4 && aligned;
    if (!fast) {
        int64_t blocks = (n_docs + 3) / 4;
        if (blocks > 256 * 8) blocks = 256 * 8;
        hipLaunchKernelGGL(cosine_scores_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                           queries, nq, corpus, n_docs, dim, mode, scores, score_stride);
        return hipGetLastError();
    }
    for (int q0 = 0; q0 < nq; q0 += SCAN_NQ) {
        const int n = (nq - q0 < SCAN_NQ) ? (nq - q0) : SCAN_NQ;
        const float* qp = queries + (int64_t)q0 * dim;
        float* sp = scores + (int64_t)q0 * score_stride;
        // whole groups through the streaming kernel, the (< R) leftover rows through the row kernel
        int64_t done = 0;
        const int R = (n == 1 || n == SCAN_NQ) ? stream_rows(dim, n > 1) : 0;
        if (R > 0 && n_docs >= R) {
            const int64_t groups = n_docs / R;
            const bool ok = (n == 1) ? dispatch_stream<1>(dim, qp, corpus, groups, mode, sp, score_stride, stream)
                                     : dispatch_stream<SCAN_NQ>(dim, qp, corpus, groups, mode, sp, score_stride, stream);
            if (ok) done = groups * R;
        }
        if (done < n_docs) {
            const int64_t rest = n_docs - done;
            int64_t blocks = (rest + 3) / 4;
            if (blocks > 256 * 8) blocks = 256 * 8;
            hipLaunchKernelGGL(cosine_scores_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, qp, n,
                               corpus + done * dim, rest, dim, mode, sp + done, score_stride);
        }
    }
    return hipGetLastError();
}

}  // namespace

// Workspace: two ping-pong key buffers sized for the first level's output, plus
// one `upper` key per query.
size_t cosine_topk_workspace_bytes(int nq, int64_t n_docs, int k)
{
    const int kpad = kpad_for(k < 1024 ? k : 1024);
    const int64_t per_q = blocks_for(n_docs, nq) * kpad;
    return (size_t)(2 * per_q * nq + nq) * sizeof(uint64_t) + 256;
}

namespace {
hipError_t cosine_topk_impl(const float* scores, int nq, int64_t n_docs, int k, void* workspace, int64_t* out_idx, float* out_score,
                            hipStream_t stream, const unsigned* run_flag);
}

hipError_t launch_cosine_topk(const float* scores, int nq, int64_t n_docs, int k, void* workspace,
                              int64_t* out_idx, float* out_score, hipStream_t stream)
{
    return cosine_topk_impl(scores, nq, n_docs, k, workspace, out_idx, out_score, stream, nullptr);
}

namespace {
// run_flag != null: every launch returns at once unless *run_flag != 0 (the fallback behind a fused scan)
hipError_t cosine_topk_impl(const float* scores, int nq, int64_t n_docs, int k, void* workspace, int64_t* out_idx, float* out_score,
                            hipStream_t stream, const unsigned* run_flag)
{
    if (nq <= 0 || k <= 0) return hipSuccess;
    if (n_docs <= 0 || n_docs >= (int64_t)0xFFFFFFFF) return hipErrorInvalidValue;
    const int kpad = kpad_for(k < 1024 ? k : 1024);
    const int64_t per_q = blocks_for(n_docs, nq) * kpad;
    uint64_t* buf_a = reinterpret_cast<uint64_t*>(workspace);
    uint64_t* buf_b = buf_a + per_q * nq;
    uint64_t* upper = buf_b + per_q * nq;

    // k > 1024: repeated selections, each restricted to keys below the previous pass's last key.
    for (int done = 0; done < k; done += 1024) {
        const int take = (k - done < 1024) ? (k - done) : 1024;
        dispatch_reduce(kpad, scores, nullptr, n_docs, n_docs, done ? upper : nullptr, buf_a, per_q, nq,
                        stream, run_flag);
        int64_t n = blocks_for(n_docs, nq) * kpad;
        uint64_t *src = buf_a, *dst = buf_b;
        while (n > kpad) {
            dispatch_reduce(kpad, nullptr, src, n, per_q, nullptr, dst, per_q, nq, stream, run_flag);
            n = blocks_for(n, nq) * kpad;
            uint64_t* t = src;
            src = dst;
            dst = t;
        }
        hipLaunchKernelGGL(topk_decode_kernel, dim3((unsigned)nq), dim3(256), 0, stream, src, per_q, take,
                           out_idx, out_score, (int64_t)k, (int64_t)done, upper, run_flag);
    }
    return hipGetLastError();
}
}  // namespace

namespace {

// Workgroups of the fused one-query pass: every wave gets at least ~4 groups (a wave walks its groups one after the other, a
// group ahead in flight: with 16 groups per wave a 10^5-row corpus was a 40 us latency chain), at most the 2 048 the streaming
// kernels use.
inline int64_t search_blocks(int64_t n_groups)
{
    const int64_t b = n_groups / (4 * 4);
    return b < 1 ? 1 : (b > 2048 ? 2048 : b);
}

template <int NV4, int R>
void launch_search_stream(int kw, const float* query, const float* corpus, int64_t n_groups, int64_t n_docs, int mode, int kout,
                          uint64_t* cand, unsigned blocks, hipStream_t stream)
{
#define KJ_SRCH(S_)                                                                                                                 \
    hipLaunchKernelGGL((cosine_search_stream_kernel<NV4, R, S_>), dim3(blocks), dim3(256), 0, stream, query, corpus, n_groups, n_docs, \
                       mode, kout, cand)
    if (kw <= 64) KJ_SRCH(1);
    else if (kw <= 128) KJ_SRCH(2);
    else KJ_SRCH(4);
#undef KJ_SRCH
}

// The one-query fused pass: widths the streaming kernel is specialised for, k <= 256, 16-byte aligned rows.
inline bool search_fused_ok(int nq, int dim, int mode, int k, const float* queries, const float* corpus, int64_t n_docs)
{
    const int R = stream_rows(dim, false);
    return nq == 1 && k <= 256 && R > 0 && n_docs >= R && (mode == 0 || mode == 1) && (reinterpret_cast<uintptr_t>(corpus) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(queries) & 15) == 0 && !tune::scan_two_launches();
}

// The last step over the scan's `lists` sorted lists of kpad keys: one block, outputs decoded.
void final_lists(int kpad, const uint64_t* keys, int lists, int k, int64_t* out_idx, float* out_score, hipStream_t stream)
{
#define KJ_FIN(KP_) hipLaunchKernelGGL(topk_lists_final_kernel<KP_>, dim3(1), dim3(256), 0, stream, keys, lists, k, out_idx, out_score)
    switch (kpad) {
    case 16: KJ_FIN(16); break;
    case 32: KJ_FIN(32); break;
    case 64: KJ_FIN(64); break;
    case 128: KJ_FIN(128); break;
    default: KJ_FIN(256); break;
    }
#undef KJ_FIN
}

}  // namespace

// Workspace of launch_cosine_search: the workgroups' candidate lists (+ one intermediate level) for the fused one-query pass,
// otherwise the scores [nq, n_docs] and the selection's buffers.
namespace {
constexpr size_t kManyCandCap = (size_t)4 << 20;       // candidate keys of the fused many-query scan over all its queries (32 MB)
constexpr int64_t kManyFusedMinDocs = 400000;           // below: the two-call form (its selection is a small share there)
constexpr int kFilterMinQueries = 2;                    // queries from which the widths with the filter pass take it: a streaming pass
                                                        // serves 4 queries, the filter pass 64 at 1.3 x the cost of ONE (profiles/r06zc)
constexpr int64_t kFilterMinDocs = 20000;               // the widths with the bf16 filter pass: 64 queries x 10^5 documents 0.30 -> 0.11 ms (profiles/r06zb)
// (a streaming pass serves 4 queries; the filter pass 64 at 1.3 x the cost of one)
inline int many_min_queries(int dim)
{
#ifdef KJARNI_TUNING
    if (const char* e = getenv("KJARNI_HIP_FILTER_MIN_QUERIES")) return filter_width(dim) ? atoi(e) : 20;
#endif
    return filter_width(dim) && !tune::scan_f32_select() ? kFilterMinQueries : 20;
}
inline int64_t many_min_docs(int dim)
{
#ifdef KJARNI_TUNING
    if (const char* e = getenv("KJARNI_HIP_FILTER_MIN_DOCS")) return filter_width(dim) ? atoll(e) : kManyFusedMinDocs;
#endif
    return filter_width(dim) && !tune::scan_f32_select() ? kFilterMinDocs : kManyFusedMinDocs;
}
constexpr int kSampleMaxK = 128;                        // up to this k the sample pass hands over per-wave maxima, not scores
inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }
}  // namespace

size_t cosine_search_workspace_bytes(int nq, int64_t n_docs, int dim, int k)
{
    (void)dim;
    const size_t fused = (size_t)(2048 * 256 + 2 * 2048) * sizeof(uint64_t);
    size_t two = pad256((size_t)nq * (size_t)n_docs * sizeof(float)) + pad256(cosine_topk_workspace_bytes(nq, n_docs, k));
    // the many-query searches (from kFilterMinQueries queries on): candidate list, counters, the sample's best k per query
    if (nq >= kFilterMinQueries)
        two += pad256(kManyCandCap * 8) + pad256((size_t)(nq + 1) * kCountStride * 4) + pad256((size_t)nq * k * 8) + pad256((size_t)nq * k * 4) +
               pad256((size_t)(nq + 8) * 4) +   // (+ the queries' squared norms, shared by the passes of one search)
               pad256(kFilterListBytes) + pad256(kFilterMaxWaves * 4);   // (+ the bf16 filter pass's per-wave lists and their lengths)
    return fused > two ? fused : two;
}

// For callers whose query and corpus pointers are 16-byte aligned by construction (the Searcher's device image and staging
// buffers): ONE query over a width the fused pass covers needs the workgroups' candidate lists only (4 MB) -- not the
// [1, n_docs] score array and the two-call selection buffers the general bound above includes.
size_t cosine_search_one_query_workspace_bytes(int64_t n_docs, int dim, int k)
{
    const int R = stream_rows(dim, false);
    if (k <= 256 && R > 0 && n_docs >= R && !tune::scan_two_launches()) return (size_t)(2048 * 256 + 2 * 2048) * sizeof(uint64_t);
    return cosine_search_workspace_bytes(1, n_docs, dim, k);
}

hipError_t launch_cosine_search(const float* queries, int nq, const float* corpus, int64_t n_docs, int dim, int mode, int k,
                                void* workspace, int64_t* out_idx, float* out_score, hipStream_t stream)
{
    if (nq <= 0 || k <= 0) return hipSuccess;
    if (n_docs <= 0 || n_docs >= (int64_t)0xFFFFFFFF) return hipErrorInvalidValue;
    if (search_fused_ok(nq, dim, mode, k, queries, corpus, n_docs)) {
        const int R = stream_rows(dim, false);
        const int64_t groups = n_docs / R;
        const unsigned blocks = (unsigned)search_blocks(groups);
        const int kout = kpad_for(k), kw = kout < 64 ? 64 : kout;
        uint64_t* cand = reinterpret_cast<uint64_t*>(workspace);
        switch (dim) {
        case 384: launch_search_stream<96, 4>(kw, queries, corpus, groups, n_docs, mode, kout, cand, blocks, stream); break;
        case 768: launch_search_stream<192, 2>(kw, queries, corpus, groups, n_docs, mode, kout, cand, blocks, stream); break;
        case 1024: launch_search_stream<256, 2>(kw, queries, corpus, groups, n_docs, mode, kout, cand, blocks, stream); break;
        case 512: launch_search_stream<128, 4>(kw, queries, corpus, groups, n_docs, mode, kout, cand, blocks, stream); break;
        case 256: launch_search_stream<64, 8>(kw, queries, corpus, groups, n_docs, mode, kout, cand, blocks, stream); break;
        case 128: launch_search_stream<32, 16>(kw, queries, corpus, groups, n_docs, mode, kout, cand, blocks, stream); break;
        default: return hipErrorInvalidValue;
        }
        final_lists(kout, cand, (int)blocks, k, out_idx, out_score, stream);
        return hipGetLastError();
    }
    float* scores = reinterpret_cast<float*>(workspace);
    const size_t s_bytes = pad256((size_t)nq * (size_t)n_docs * sizeof(float));
    uint8_t* topk_ws = static_cast<uint8_t*>(workspace) + s_bytes;
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(corpus) & 15) == 0) && ((reinterpret_cast<uintptr_t>(queries) & 15) == 0);
    if (nq >= many_min_queries(dim) && nq <= 1024 && n_docs >= many_min_docs(dim) && k <= 1024 && aligned16 && dim % MQ_BK == 0 && dim >= 2 * MQ_BK &&
        (int64_t)MQ_D * dim * 4 < ((int64_t)1 << 31) && (mode == 0 || mode == 1) && !tune::scan_streaming_only() &&
        !tune::scan_two_launches()) {
        // Many queries, selection inside the scan (ScanFuse): sample pass -> per-query bounds -> fused scan -> per-query selection
        // of the candidates; the two-call form is queued behind with the overflow word as its run flag.
        uint8_t* p = topk_ws + pad256(cosine_topk_workspace_bytes(nq, n_docs, k));
        uint64_t* cand_key = reinterpret_cast<uint64_t*>(p);
        p += pad256(kManyCandCap * 8);
        unsigned* counters = reinterpret_cast<unsigned*>(p);  // [0] overflow, [1 + q] list lengths
        const size_t counter_bytes = pad256((size_t)(nq + 1) * kCountStride * 4);
        p += counter_bytes;
        int64_t* thr_idx = reinterpret_cast<int64_t*>(p);
        p += pad256((size_t)nq * k * 8);
        float* thr_score = reinterpret_cast<float*>(p);
        p += pad256((size_t)nq * k * 4);
        float* qn2 = reinterpret_cast<float*>(p);   // the queries' squared norms: ONE launch for the sample, the scan and the fallback
        p += pad256((size_t)(nq + 8) * 4);
        uint64_t* wave_list = reinterpret_cast<uint64_t*>(p);   // the bf16 filter pass's per-wave (query, document) lists
        p += pad256(kFilterListBytes);
        unsigned* wave_count = reinterpret_cast<unsigned*>(p);
        // Six widths (filter_width) have the bf16 filter pass + exact rescoring (HBM-bound); the others the f32 matrix-core scan
        // with the selection inside (MFMA-bound).
        const bool filtered = filter_width(dim) && !tune::scan_f32_select();
        const FilterPlan fp = filtered ? filter_plan(n_docs, dim, kFilterListBytes) : FilterPlan{};
        const unsigned cap_q = (unsigned)(kManyCandCap / (size_t)nq);
        hipError_t e = query_sqnorms(queries, nq, dim, qn2, stream);
        if (e != hipSuccess) return e;
        const int64_t all_tiles = (n_docs + MQ_D - 1) / MQ_D;
        // 128-256 sampled tiles (up to 512 for large corpora, below), spread over the corpus: ONE tile per CU, so the sample pass is a single round of workgroups that
        // each have a CU's four matrix pipes to themselves (with 256-511 tiles it was two per CU and took twice as long: 63 us of a
        // 0.59 ms search over 10^6 documents); ~50 000 sampled documents still bound k = 10 at ~160 candidates per query there.
        // (from ~2 M documents on the sample pass is < 2 % of the search and the candidates it leaves are what costs: two tiles per CU)
        const int64_t sample_tiles = all_tiles >= 8192 ? 512 : 256;
        const int ts = (int)std::max<int64_t>(1, (all_tiles + sample_tiles - 1) / sample_tiles);
        const int64_t ns = (all_tiles + ts - 1) / ts;
        const int64_t last_rows = std::min<int64_t>(MQ_D, n_docs - (ns - 1) * ts * (int64_t)MQ_D);
        const int64_t n_sample = (ns - 1) * MQ_D + last_rows;
        int thr_k = k;
        if (filtered && k <= kSampleMaxK && fp.units >= 8 * (int64_t)k) {
            // the bf16 sample: per (query, unit of 16 unit_tiles documents) maxima; bound = their k-th largest - eta
            e = filter_sample(fp, queries, nq, corpus, n_docs, dim, qn2, scores, stream);
            if (e != hipSuccess) return e;
            // (and zeroes the candidate counters)
            if (fp.units <= 1024)   // (a small sample: a wave per query without barriers; above, the block per query is faster)
                hipLaunchKernelGGL(sample_bound_wave_kernel<16>, dim3((unsigned)(nq + 3) / 4), dim3(256), 0, stream, scores, (int)fp.units, fp.units,
                                   k, nq, thr_score, thr_idx, counters, kFilterEta);
            else
                hipLaunchKernelGGL(sample_bound_kernel<16>, dim3((unsigned)nq), dim3(256), 0, stream, scores, (int)fp.units, fp.units, k, thr_score,
                                   thr_idx, counters, kFilterEta);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            thr_k = 1;
        } else if (k <= kSampleMaxK && ns * 4 <= 4096 && ns * 4 >= 8 * (int64_t)k) {
            // a small k: the bound is the k-th largest of the sampled tiles' per-wave maxima (SCAN_SAMPLE_MAX) -- no sample scores
            e = scan_mfma(queries, nq, corpus, n_docs, dim, mode, scores, stream, ts, ns * 4, nullptr, nullptr, true, qn2);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(sample_bound_kernel<16>, dim3((unsigned)nq), dim3(256), 0, stream, scores, (int)(ns * 4), ns * 4, k, thr_score, thr_idx,
                               counters, 0.0f);   // (and zeroes the candidate counters)
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            thr_k = 1;
        } else {
            e = hipMemsetAsync(counters, 0, counter_bytes, stream);
            if (e != hipSuccess) return e;
            e = scan_mfma(queries, nq, corpus, n_docs, dim, mode, scores, stream, ts, n_sample, nullptr, nullptr, false, qn2);
            if (e != hipSuccess) return e;
            e = cosine_topk_impl(scores, nq, n_sample, k, topk_ws, thr_idx, thr_score, stream, nullptr);
            if (e != hipSuccess) return e;
        }
        if (filtered) {
            // the bf16 filter pass over the corpus (HBM-bound), then the exact cosines of what it lets through
            e = filter_and_rescore(fp, queries, nq, corpus, n_docs, dim, mode, qn2, thr_score, thr_idx, thr_k, wave_list, wave_count, counters,
                                   cap_q, cand_key, stream);
            if (e != hipSuccess) return e;
        } else {
            ScanFuse f{thr_score, thr_idx, thr_k, cand_key, counters, cap_q, 0};
            e = scan_mfma(queries, nq, corpus, n_docs, dim, mode, nullptr, stream, 1, -1, &f, nullptr, false, qn2);
            if (e != hipSuccess) return e;
        }
        const int kpad = kpad_for(k);
#define KJ_CAND(KP_)                                                                                                          \
    hipLaunchKernelGGL(topk_candidates_kernel<KP_>, dim3((unsigned)nq), dim3(256), 0, stream, cand_key, counters, cap_q, k, out_idx, \
                       out_score)
        switch (kpad) {
        case 16: KJ_CAND(16); break;
        case 32: KJ_CAND(32); break;
        case 64: KJ_CAND(64); break;
        case 128: KJ_CAND(128); break;
        case 256: KJ_CAND(256); break;
        case 512: KJ_CAND(512); break;
        default: KJ_CAND(1024); break;
        }
#undef KJ_CAND
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        // the two-call form, run only if the candidate list overflowed
        e = scan_mfma(queries, nq, corpus, n_docs, dim, mode, scores, stream, 1, -1, nullptr, counters, false, qn2);
        if (e != hipSuccess) return e;
        return cosine_topk_impl(scores, nq, n_docs, k, topk_ws, out_idx, out_score, stream, counters);
    }
    const hipError_t e = launch_cosine_scores(queries, nq, corpus, n_docs, dim, mode, scores, stream);
    if (e != hipSuccess) return e;
    return launch_cosine_topk(scores, nq, n_docs, k, topk_ws, out_idx, out_score, stream);
}

}  // namespace kjarni
