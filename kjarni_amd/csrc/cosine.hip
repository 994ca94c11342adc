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
// reduces a segment of 16384 candidates to its best KPAD, levels repeat until
// one block is left.
#include "device_utils.h"
#include "kernels.h"

namespace kjarni {

namespace {

constexpr int SCAN_NQ = 4;       // queries held in registers per pass
constexpr int SCAN_MAX_V4 = 4;   // dim <= 1024 on the float4 path

// mode 0: vector.rs:131-148   dot / max(sqrt(na)*sqrt(nb), 1e-9)
// mode 1: segment.rs:355-371  nb < 1e-9 ? 0 : dot / (qn * nb)
__device__ __forceinline__ float cosine_finish(float dot, float qn2, float dn2, int mode)
{
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

constexpr int TK_TILE = 2048;          // keys sorted at once in LDS
constexpr int TK_TILES_PER_BLOCK = 8;  // segment = 16384 candidates
constexpr int TK_SEG = TK_TILE * TK_TILES_PER_BLOCK;

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

// One block reduces candidates [blk*TK_SEG, +TK_SEG) of query blockIdx.y to its
// best KPAD keys (descending).  Level 0 reads float scores (index = position),
// later levels read keys.  Keys >= `upper` are ignored (multi-pass k > 1024).
template <int KPAD>
__global__ __launch_bounds__(256) void topk_reduce_kernel(const float* __restrict__ scores,
                                                          const uint64_t* __restrict__ in_keys,
                                                          int64_t n, int64_t in_stride,
                                                          const uint64_t* __restrict__ upper_ptr,
                                                          uint64_t* __restrict__ out_keys,
                                                          int64_t out_stride)
{
    __shared__ uint64_t tile[TK_TILE];
    __shared__ uint64_t best[KPAD];
    __shared__ int any_flag;
    const int tid = threadIdx.x;
    const int qi = blockIdx.y;
    const uint64_t upper = upper_ptr ? upper_ptr[qi] : ~0ull;
    const int64_t seg0 = (int64_t)blockIdx.x * TK_SEG;
    for (int i = tid; i < KPAD; i += 256) best[i] = 0ull;
    __syncthreads();

    for (int tl = 0; tl < TK_TILES_PER_BLOCK; ++tl) {
        const int64_t t0 = seg0 + (int64_t)tl * TK_TILE;
        if (t0 >= n) break;
        const uint64_t thr = best[KPAD - 1];
        if (tid == 0) any_flag = 0;
        __syncthreads();
        bool any = false;
        for (int i = tid; i < TK_TILE; i += 256) {
            const int64_t p = t0 + i;
            uint64_t key = 0ull;
            if (p < n) {
                key = scores ? make_key(scores[(int64_t)qi * in_stride + p], (uint32_t)p)
                             : in_keys[(int64_t)qi * in_stride + p];
                if (key >= upper) key = 0ull;
            }
            tile[i] = key;
            any |= key > thr;
        }
        if (any) any_flag = 1;
        __syncthreads();
        const int go = any_flag;
        __syncthreads();  // everyone has read the flag before thread 0 clears it again
        if (!go) continue;  // nothing in this tile can enter the block's best KPAD
        bitonic_sort_desc(tile, TK_TILE, tid);
        // best (desc) and reversed tile head (asc) form a bitonic sequence whose
        // element-wise max holds the top KPAD of the union.
        for (int i = tid; i < KPAD; i += 256) {
            const uint64_t a = best[i], b = tile[KPAD - 1 - i];
            best[i] = a > b ? a : b;
        }
        bitonic_merge_desc(best, KPAD, tid);
    }
    __syncthreads();
    for (int i = tid; i < KPAD; i += 256)
        out_keys[(int64_t)qi * out_stride + (int64_t)blockIdx.x * KPAD + i] = best[i];
}

__global__ void topk_decode_kernel(const uint64_t* __restrict__ keys, int64_t key_stride, int k_take,
                                   int64_t* __restrict__ out_idx, float* __restrict__ out_score,
                                   int64_t out_stride, int64_t out_offset,
                                   uint64_t* __restrict__ last_key)
{
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

int64_t blocks_for(int64_t n) { return (n + TK_SEG - 1) / TK_SEG; }

template <int KPAD>
void launch_reduce(const float* scores, const uint64_t* in_keys, int64_t n, int64_t in_stride,
                   const uint64_t* upper, uint64_t* out_keys, int64_t out_stride, int nq,
                   hipStream_t stream)
{
    dim3 grid((unsigned)blocks_for(n), (unsigned)nq);
    hipLaunchKernelGGL(topk_reduce_kernel<KPAD>, grid, dim3(256), 0, stream, scores, in_keys, n,
                       in_stride, upper, out_keys, out_stride);
}

void dispatch_reduce(int kpad, const float* scores, const uint64_t* in_keys, int64_t n,
                     int64_t in_stride, const uint64_t* upper, uint64_t* out_keys, int64_t out_stride,
                     int nq, hipStream_t stream)
{
    switch (kpad) {
    case 16: launch_reduce<16>(scores, in_keys, n, in_stride, upper, out_keys, out_stride, nq, stream); break;
    case 32: launch_reduce<32>(scores, in_keys, n, in_stride, upper, out_keys, out_stride, nq, stream); break;
    case 64: launch_reduce<64>(scores, in_keys, n, in_stride, upper, out_keys, out_stride, nq, stream); break;
    case 128: launch_reduce<128>(scores, in_keys, n, in_stride, upper, out_keys, out_stride, nq, stream); break;
    case 256: launch_reduce<256>(scores, in_keys, n, in_stride, upper, out_keys, out_stride, nq, stream); break;
    case 512: launch_reduce<512>(scores, in_keys, n, in_stride, upper, out_keys, out_stride, nq, stream); break;
    default: launch_reduce<1024>(scores, in_keys, n, in_stride, upper, out_keys, out_stride, nq, stream); break;
    }
}

}  // namespace

hipError_t launch_cosine_scores(const float* queries, int nq, const float* corpus, int64_t n_docs,
                                int dim, int mode, float* scores, hipStream_t stream)
{
    if (nq <= 0 || n_docs <= 0) return hipSuccess;
    int64_t waves = n_docs;
    const int64_t max_blocks = 256 * 8;  // 8 workgroups per CU, grid-stride the rest
    int64_t blocks = (waves + 3) / 4;
    if (blocks > max_blocks) blocks = max_blocks;
    const bool fast = (dim % 4 == 0) && dim <= 256 * SCAN_MAX_V4 &&
                      ((reinterpret_cast<uintptr_t>(corpus) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(queries) & 15) == 0);
    if (fast) {
        for (int q0 = 0; q0 < nq; q0 += SCAN_NQ) {
            const int n = (nq - q0 < SCAN_NQ) ? (nq - q0) : SCAN_NQ;
            hipLaunchKernelGGL(cosine_scores_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                               queries + (int64_t)q0 * dim, n, corpus, n_docs, dim, mode,
                               scores + (int64_t)q0 * n_docs, n_docs);
        }
    } else {
        hipLaunchKernelGGL(cosine_scores_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                           queries, nq, corpus, n_docs, dim, mode, scores, n_docs);
    }
    return hipGetLastError();
}

// Workspace: two ping-pong key buffers sized for the first level's output, plus
// one `upper` key per query.
size_t cosine_topk_workspace_bytes(int nq, int64_t n_docs, int k)
{
    const int kpad = kpad_for(k < 1024 ? k : 1024);
    const int64_t per_q = blocks_for(n_docs) * kpad;
    return (size_t)(2 * per_q * nq + nq) * sizeof(uint64_t) + 256;
}

hipError_t launch_cosine_topk(const float* scores, int nq, int64_t n_docs, int k, void* workspace,
                              int64_t* out_idx, float* out_score, hipStream_t stream)
{
    if (nq <= 0 || k <= 0) return hipSuccess;
    if (n_docs <= 0 || n_docs >= (int64_t)0xFFFFFFFF) return hipErrorInvalidValue;
    const int kpad = kpad_for(k < 1024 ? k : 1024);
    const int64_t per_q = blocks_for(n_docs) * kpad;
    uint64_t* buf_a = reinterpret_cast<uint64_t*>(workspace);
    uint64_t* buf_b = buf_a + per_q * nq;
    uint64_t* upper = buf_b + per_q * nq;

    // k > 1024: repeated selections, each restricted to keys below the previous pass's last key.
    for (int done = 0; done < k; done += 1024) {
        const int take = (k - done < 1024) ? (k - done) : 1024;
        dispatch_reduce(kpad, scores, nullptr, n_docs, n_docs, done ? upper : nullptr, buf_a, per_q, nq,
                        stream);
        int64_t n = blocks_for(n_docs) * kpad;
        uint64_t *src = buf_a, *dst = buf_b;
        while (n > kpad) {
            dispatch_reduce(kpad, nullptr, src, n, per_q, nullptr, dst, per_q, nq, stream);
            n = blocks_for(n) * kpad;
            uint64_t* t = src;
            src = dst;
            dst = t;
        }
        hipLaunchKernelGGL(topk_decode_kernel, dim3((unsigned)nq), dim3(256), 0, stream, src, per_q, take,
                           out_idx, out_score, (int64_t)k, (int64_t)done, upper);
    }
    return hipGetLastError();
}

}  // namespace kjarni
