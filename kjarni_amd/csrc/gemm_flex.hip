// Projection GEMM for calls of a few hundred to a few thousand rows (the reference's default batch is 32 sentences:
// kjarni-ffi/src/indexer.rs:132):   Y[M,N] = X[M,K] * W[N,K]^T + b (+ epilogue)      -- LinearLayer::matmul,
// crates/kjarni-transformers/src/linear_layer/linear_layer.rs:160-282, cpu/ops/matmul.rs:571-686.
//
// Why its own kernel.  The large-batch tiles (gemm.hip: 128 x 128, two workgroups per CU) are sized for 10^5 rows.  At
// 4 096 rows a projection is 100-400 such tiles on 256 CUs, and what a call of this size costs is the BUSIEST CU's share:
// 288 tiles on 256 CUs take two tile times.  Here the tile is chosen per call so that the tiles of a call are one per CU
// (or a whole number per CU): a workgroup of four waves owns BM = 64 * RA rows x BN = 16 * CB columns (RA = 1, 2; CB = 3 .. 12,
// the 16 x 16 x 4 f32 MFMA gives 16-column steps), one workgroup per CU (it claims more than half of the LDS), and the
// launcher picks (RA, CB) with the fewest rounds x the longest tile -- 4 096 x 1 152 x 384 (QKV at 32 sentences): 32 x 8 =
// 256 tiles of 128 x 144; 4 096 x 1 536 (FC1): 256 tiles of 128 x 192.
//
// Wave w owns rows w*16*RA .. of the tile and ALL its columns (RA x CB accumulator tiles of 16 x 16): every wave reads the
// whole W tile from LDS, nobody shares an A row.  K-step 32, operand tiles in LDS as plain 128-byte rows whose eight 16-byte
// chunks are XOR-swizzled by (row >> 1) & 7: a ds_read_b128 lane group (16 rows, two k-chunks 4 apart) and a staging
// ds_write_b128 group (the 8 chunks of one row) both touch every bank once.  One ds_read_b128 feeds four MFMAs (the k order
// inside a K-step is free as long as A and B agree: chunk g of a phase gives k = 4g + q to MFMA q), so a K-step is two
// phases of 4 * RA * CB MFMAs with RA + CB fragment reads each.
//
// Software pipeline (the large tiles' scheme): tile t in LDS[cur], tile t+1 in registers, tile t+2 requested.
//   phase 0: fragments of phase 1 | MFMAs of phase 0, and spread under them the staging pieces: registers -> LDS[cur^1],
//            then the same registers reloaded with tile t+2 (buffer descriptors: scalar K offset, constant lane offsets, rows
//            past M / N read as zeros)
//   phase 1: barrier, fragments of phase 0 of LDS[cur^1] | MFMAs of phase 1
//
// Arithmetic: v_mfma_f32_16x16x4_f32 is a k-ordered f32 fma chain (exact f32 products, one rounding per accumulate), the same
// class as the reference's AVX2 FMA loop; a row's k order does not depend on the tile shape, so a row's result does not depend
// on how many rows the call has or which (RA, CB) the launcher picked.  K slices (narrow outputs with a long K: FC2): the
// reduce kernels add the slices' partial sums in slice order; a workgroup that owns several slices (VSLICES: calls with
// enough rows to fill the chip without cutting K) adds them in that same order -- bit-identical either way.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "device_utils.h"
#include "gemm_epilogue.h"
#include "kernels.h"
#include "tuning.h"

namespace kjarni {

namespace {

constexpr int FBK = 32;   // K-step
constexpr int FROW = 32;  // floats per LDS row

template <int RA, int CB>
struct Flex {
    static constexpr int BM = 64 * RA, BN = 16 * CB;
    static constexpr int NA = BM / 32;         // 16-byte staging pieces per thread and K-step, A tile
    static constexpr int NB = (BN + 31) / 32;  // ... W tile (rows BN .. NB*32 are staged and never read)
    static constexpr int B_ROWS = NB * 32;
    static constexpr int STAGE_FLOATS = (BM + B_ROWS) * FROW;
    static constexpr int ES = BN + 4;  // epilogue staging: floats per row (4 mod 8: the 4-row accumulator stores hit 32 banks)
    static constexpr int EPI_FLOATS = 4 * 16 * RA * ES;
    static constexpr int OP_FLOATS = 2 * STAGE_FLOATS;
    static constexpr int MAIN_FLOATS = OP_FLOATS > EPI_FLOATS ? OP_FLOATS : EPI_FLOATS;
    static constexpr int LDS_FLOATS = MAIN_FLOATS + BN;  // + the tile's bias slice
    static constexpr int PIECES = NA + NB;
    static constexpr int MFMAS = 4 * RA * CB;  // per phase
};

// Registers: the smaller tiles are compiled for two waves per SIMD (<= 256 registers), so that two workgroups share a CU where
// their LDS fits and one's epilogue runs under the other's K-loop; the larger ones take the whole file (one wave per SIMD).
constexpr int flex_waves_per_simd(int ra, int cb, bool vslices)
{
    return vslices ? (ra * cb <= 8 ? 2 : 1) : ((ra == 1 && cb <= 9) || (ra == 2 && cb <= 6) ? 2 : 1);
}
constexpr int kFlexLdsBytes = 104 * 1024;  // (tuning build: the claim that forces one workgroup per CU)

template <int RA, int CB>
struct FlexFrag {
    f32x4 a[RA], b[CB];
};

// DIAG (tuning build, tools/flex_probe.py): knock-outs that show where a tile's time goes -- tuning.h
template <int RA, int CB, bool VSLICES, int DIAG = 0>
__global__ __launch_bounds__(256, flex_waves_per_simd(RA, CB, VSLICES)) void gemm_nt_f32_flex(const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
                                                          int64_t ldw, const float* __restrict__ bias, const float* R, int64_t ldr,
                                                          float* Y, int64_t ldy, int M, int N, int k_len, int ksplit, int vslices,
                                                          int epi, int out_mode, int n_tiles, int total)
// R and Y are not __restrict__: the residual epilogue runs in place (every element is read by the lane that stores it).
{
    using T = Flex<RA, CB>;
    constexpr int BM = T::BM, BN = T::BN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sBias = smem + T::MAIN_FLOATS;
    const int partial = out_mode & 1;          // raw sums to slabs
    const bool write_through = out_mode & 2;   // output stores with the sc1 policy (below)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;

    // XCD-aware order (workgroups are dealt round-robin over the 8 XCDs): every XCD takes one contiguous run of
    // (row tile, column tile, K slice) triples, slices fastest, then columns: the workgroups of an XCD share A row panels.
    const unsigned nwg = (unsigned)total;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7u;
    for (unsigned wg = blockIdx.x; wg < nwg; wg += gridDim.x) {
    const unsigned xcd = wg & 7u, slot = wg >> 3;
    const unsigned bid0 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const unsigned bid = bid0 / (unsigned)ksplit;
    const int ks = (int)(bid0 - bid * (unsigned)ksplit);
    const unsigned m_tile = bid / (unsigned)n_tiles;
    const int m0 = (int)m_tile * BM;
    const int n0 = (int)(bid - m_tile * (unsigned)n_tiles) * BN;

    // staging: thread -> (row r0 + 32 i, chunk c8) of the [rows][32] operand tiles
    const int c8 = tid & 7, r0 = tid >> 3;
    const int rows_a = (M - m0 < BM) ? (M - m0) : BM;
    const int rows_w = (N - n0 < T::B_ROWS) ? (N - n0) : T::B_ROWS;
    const float* Ab = A + (int64_t)m0 * lda + (int64_t)ks * k_len;
    const float* Wb = W + (int64_t)n0 * ldw + (int64_t)ks * k_len;
    const __amdgpu_buffer_rsrc_t rsrcA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ab), 0, (int)((((int64_t)rows_a - 1) * lda + k_len) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wb), 0, (int)((((int64_t)rows_w - 1) * ldw + k_len) * 4), 0x00020000);
    uint32_t offA[T::NA], offW[T::NB];
#pragma unroll
    for (int i = 0; i < T::NA; ++i) offA[i] = (uint32_t)(((int64_t)(r0 + 32 * i) * lda + c8 * 4) * 4);
#pragma unroll
    for (int i = 0; i < T::NB; ++i) offW[i] = (uint32_t)(((int64_t)(r0 + 32 * i) * ldw + c8 * 4) * 4);
    f32x4 ga[T::NA], gb[T::NB];
    auto ld16 = [](__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, int k0) {
        // (DIAG 6 / 7 / 8: the cache-policy bits of the staging loads -- sc1 = 16, sc0 = 1, nt = 2)
        constexpr int AUX = DIAG == 6 ? 16 : DIAG == 7 ? 1 : DIAG == 8 ? 2 : 0;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, k0 * 4, AUX));
    };
    // (a row's 16-byte chunk c sits at chunk c ^ ((row >> 1) & 7); rows 32 apart share the term)
    const int st_off = r0 * FROW + ((c8 ^ ((r0 >> 1) & 7)) << 2);
    auto load_piece = [&](int i, int k0) {
        if (i < T::NA) ga[i] = ld16(rsrcA, offA[i], k0);
        else gb[i - T::NA] = ld16(rsrcW, offW[i - T::NA], k0);
    };
    auto store_piece = [&](int stage, int i) {
        float* base = smem + stage * T::STAGE_FLOATS + st_off;
        if (i < T::NA) *reinterpret_cast<f32x4*>(base + 32 * i * FROW) = ga[i];
        else *reinterpret_cast<f32x4*>(base + (BM + 32 * (i - T::NA)) * FROW) = gb[i - T::NA];
    };

    // the epilogue's bias slice waits in LDS from here (the K-loop's barriers order it before the epilogue)
    if (tid < BN / 4) {
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        const int n = n0 + tid * 4;
        if (bias != nullptr && !partial && n < N) bv = *reinterpret_cast<const f32x4*>(bias + n);
        *reinterpret_cast<f32x4*>(sBias + tid * 4) = bv;
    }

    f32x4 acc[RA][CB], sum[VSLICES ? RA : 1][VSLICES ? CB : 1];
#pragma unroll
    for (int i = 0; i < RA; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragments: lane (r16, g) reads row r16 of a 16-row block, chunk 4 p + g of phase p
    const int sw = (r16 >> 1) & 7;
    const int fa0 = (wid * 16 * RA + r16) * FROW + ((g ^ sw) << 2);
    const int fa1 = (wid * 16 * RA + r16) * FROW + (((4 + g) ^ sw) << 2);
    const int fb0 = (BM + r16) * FROW + ((g ^ sw) << 2);
    const int fb1 = (BM + r16) * FROW + (((4 + g) ^ sw) << 2);
    using Frag = FlexFrag<RA, CB>;
    auto read_frag = [&](Frag& f, int stage, int p) {
        const float* base = smem + stage * T::STAGE_FLOATS;
#pragma unroll
        for (int i = 0; i < RA; ++i) f.a[i] = *reinterpret_cast<const f32x4*>(base + (p ? fa1 : fa0) + i * 16 * FROW);
#pragma unroll
        for (int j = 0; j < CB; ++j) f.b[j] = *reinterpret_cast<const f32x4*>(base + (p ? fb1 : fb0) + j * 16 * FROW);
    };
    auto mfma_phase = [&](const Frag& f) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < RA; ++i)
#pragma unroll
                for (int j = 0; j < CB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[i][q], f.b[j][q], acc[i][j], 0, 0, 0);
    };

    const int nk = k_len / FBK;
#pragma unroll
    for (int i = 0; i < T::PIECES; ++i) load_piece(i, 0);
#pragma unroll
    for (int i = 0; i < T::PIECES; ++i) store_piece(0, i);
    if (nk > 1) {
#pragma unroll
        for (int i = 0; i < T::PIECES; ++i) load_piece(i, FBK);
    }
    __syncthreads();
    Frag fr0, fr1;
    read_frag(fr0, 0, 0);

    auto step = [&](auto store_tag, auto load_tag, int kt) {
        constexpr bool STORE = decltype(store_tag)::value;
        constexpr bool LOAD = decltype(load_tag)::value;
        const int cur = kt & 1;
        if (DIAG != 3) read_frag(fr1, cur, 1);
#pragma unroll
        for (int i = 0; i < T::PIECES; ++i) {
            if (STORE && DIAG != 3 && DIAG != 5) store_piece(cur ^ 1, i);
            if (LOAD && DIAG != 2 && DIAG != 3 && DIAG != 5) load_piece(i, (kt + 2) * FBK);
        }
        mfma_phase(fr0);
        __builtin_amdgcn_sched_group_barrier(0x100, RA + CB, 0);  // DS reads first
        if (STORE && DIAG != 3 && DIAG != 5) {
            constexpr int GAP = T::MFMAS / T::PIECES;  // MFMAs per staging piece
            constexpr int G1 = GAP > 2 ? GAP - 1 : 1;
#pragma unroll
            for (int i = 0; i < T::PIECES; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, G1, 0);  // MFMA
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
                if (LOAD && DIAG != 2) {
                    __builtin_amdgcn_sched_group_barrier(0x008, GAP - G1 > 0 ? GAP - G1 : 1, 0);  // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                          // VMEM read
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (DIAG != 3 && DIAG != 4) __syncthreads();
        if (STORE && DIAG != 3) read_frag(fr0, cur ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_phase(fr1);
        __builtin_amdgcn_sched_barrier(0);
    };
    using TT = std::true_type;
    using FF = std::false_type;
    if (VSLICES) {
        const int sps = nk / vslices;  // K-steps per slice (>= 2)
        int kt = 0;
        for (int s = 0; s < vslices; ++s) {
            const int end = (s + 1) * sps;
            for (; kt < end && kt + 2 < nk; ++kt) step(TT{}, TT{}, kt);
            if (s == vslices - 1) {
                if (kt + 1 < nk) step(TT{}, FF{}, kt++);
                step(FF{}, FF{}, kt++);
            }
#pragma unroll
            for (int i = 0; i < RA; ++i)
#pragma unroll
                for (int j = 0; j < CB; ++j) {
                    // (slice 0: sum = acc exactly -- the slab form reads P0 and adds P1, P2, ...)
                    if (s == 0) sum[VSLICES ? i : 0][VSLICES ? j : 0] = acc[i][j];
                    else sum[VSLICES ? i : 0][VSLICES ? j : 0] += acc[i][j];
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
        }
#pragma unroll
        for (int i = 0; i < RA; ++i)
#pragma unroll
            for (int j = 0; j < CB; ++j) acc[i][j] = sum[VSLICES ? i : 0][VSLICES ? j : 0];
    } else {
        int kt = 0;
        for (; kt + 2 < nk; ++kt) step(TT{}, TT{}, kt);
        if (kt + 1 < nk) step(TT{}, FF{}, kt++);
        step(FF{}, FF{}, kt);
    }

    // ---- epilogue: the wave's 16 RA x BN block goes through its own LDS region (the operand stages are dead after the
    // barrier below) and leaves as 16-byte pieces of whole row segments ----
    __syncthreads();
    float* sw_ = smem + wid * (16 * RA * T::ES);
#pragma unroll
    for (int i = 0; i < RA; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) sw_[(i * 16 + g * 4 + r) * T::ES + j * 16 + r16] = acc[i][j][r];
    constexpr int V4_PER_ROW = BN / 4;  // 16 RA rows x BN / 4 pieces = 64 x RA x CB, in RA rounds of CB pieces per lane
    float* Yb = partial ? Y + (int64_t)ks * M * N : Y;
    const int64_t ldo = partial ? N : ldy;
    const bool need_r = !partial && (epi == EPI_BIAS_RESIDUAL || epi == EPI_BIAS_MUL_SILU);
#pragma unroll
    for (int rd = 0; rd < RA; ++rd) {
        // (all residual loads of a round are issued back to back, in front of its arithmetic)
        f32x4 res[CB];
        if (need_r) {
#pragma unroll
            for (int it = 0; it < CB; ++it) {
                const int idx = lane + 64 * (rd * CB + it);
                const int row = idx / V4_PER_ROW, c4 = idx - row * V4_PER_ROW;
                int m = m0 + wid * 16 * RA + row;
                m = m < M ? m : M - 1;
                int n = n0 + c4 * 4;
                n = n < N ? n : N - 4;
                res[it] = *reinterpret_cast<const f32x4*>(R + (int64_t)m * ldr + n);
            }
        }
#pragma unroll
        for (int it = 0; it < CB; ++it) {
            const int idx = lane + 64 * (rd * CB + it);
            const int row = idx / V4_PER_ROW, c4 = idx - row * V4_PER_ROW;
            const int m = m0 + wid * 16 * RA + row, n = n0 + c4 * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(sw_ + row * T::ES + c4 * 4);
            if (!partial) {
                v += *reinterpret_cast<const f32x4*>(sBias + c4 * 4);
                if (epi == EPI_BIAS_GELU) {
                    const f32x2 lo = gelu_erf_fast2(f32x2{v[0], v[1]}), hi = gelu_erf_fast2(f32x2{v[2], v[3]});
                    v = f32x4{lo[0], lo[1], hi[0], hi[1]};
                } else if (epi == EPI_BIAS_RESIDUAL) {
                    v += res[it];
                } else if (epi == EPI_BIAS_MUL_SILU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] *= silu_ref(res[it][c]);
                } else if (epi == EPI_BIAS_GELU_NEW) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = gelu_tanh(v[c]);
                } else if (epi == EPI_BIAS_RELU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.0f);
                } else if (epi == EPI_BIAS_TANH) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = tanhf(v[c]);
                }
            }
            if (m < M && n < N && (DIAG != 1 || v[0] == 123456.789f)) {
                // Where the tile goes.  Plain stores park it dirty in this XCD's L2 and the kernel ENDS with the write-back of all of
                // it at once, nothing left to overlap (FC1 + GELU at 4 096 rows: 5 of its 50 us, tools/flex_knockout.py).  Streaming
                // (nt) stores remove that and lose the call: the XCD-aware tile order gives the same rows to the same XCD in the
                // consuming launch, which then reads from memory (16 x 128 tokens 0.646 -> 0.675 ms).  Write-through stores (sc1) do
                // both -- the line goes to memory at once AND stays in L2: 8 / 16 / 32 x 128 tokens 0.458 / 0.627 / 0.90 -> 0.440 /
                // 0.602 / 0.889 ms.  Calls of more than kWriteThroughMaxRows rows keep plain stores (64 x 128: no difference;
                // their halves run on two streams, and the other half's work covers the burst).
                f32x4* dst = reinterpret_cast<f32x4*>(Yb + (int64_t)m * ldo + n);
                if (DIAG == 9) __builtin_nontemporal_store(v, dst);
                else if (DIAG >= 10 && DIAG <= 12) {
                    // (measurement: write-through stores -- cache-policy bits sc1 = 16 / sc0 = 1 / both; the tile's rows through a descriptor)
                    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(Yb + (int64_t)m0 * ldo, 0, 0x7fffffff, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), ry,
                                                           (unsigned)(((int64_t)(m - m0) * ldo + n) * 4), 0,
                                                           DIAG == 10 ? 16 : DIAG == 11 ? 1 : 17);
                }
                else if (write_through) {
                    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(Yb + (int64_t)m0 * ldo, 0, 0x7fffffff, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), ry,
                                                           (unsigned)(((int64_t)(m - m0) * ldo + n) * 4), 0, 16);
                } else *dst = v;
            }
        }
    }
    __syncthreads();  // (the epilogue regions overlap the operand stages of a next tile)
    }
}

constexpr int kWriteThroughMaxRows = 3072;

struct FlexChoice {
    int ra = 0, cb = 0;
    double cost = 0.0;
};

template <int RA, int CB, bool VS, int DIAG = 0>
hipError_t flex_launch_one(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, const float* R, int64_t ldr,
                           float* Y, int64_t ldy, int M, int N, int k_len, int ksplit, int vslices, int epi, int partial,
                           hipStream_t stream)
{
    using T = Flex<RA, CB>;
    static_assert(T::LDS_FLOATS * 4 <= kFlexLdsBytes, "tile does not fit the per-launch LDS claim");
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!attr_set[dev & 63]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_f32_flex<RA, CB, VS, DIAG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kFlexLdsBytes);
        if (e != hipSuccess) return e;
        attr_set[dev & 63] = true;
    }
    const int n_tiles = (N + T::BN - 1) / T::BN;
    const int64_t total = (int64_t)((M + T::BM - 1) / T::BM) * n_tiles * ksplit;
    if (total > 0x7fffffff) return hipErrorInvalidValue;
    // (the LDS a tile needs, so that the smaller tiles share a CU; the tuning build can claim more than half a CU instead)
    const int lds = tune::flex_one_workgroup_per_cu() ? kFlexLdsBytes : T::LDS_FLOATS * 4;
    hipLaunchKernelGGL((gemm_nt_f32_flex<RA, CB, VS, DIAG>), dim3((unsigned)total), dim3(256), lds, stream, A, lda, W, ldw, bias, R, ldr,
                       Y, ldy, M, N, k_len, ksplit, vslices, epi, partial, n_tiles, (int)total);
    return hipGetLastError();
}

constexpr int kFlexCb[] = {3, 4, 6, 8, 9, 12};
constexpr int kFlexConfigs = 12;

// Workgroups of a tile kernel one CU holds at once (registers, LDS): asked of the runtime once per process and kernel.
template <int RA, int CB, bool VS>
int flex_residency_of()
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&gemm_nt_f32_flex<RA, CB, VS, 0>), 256,
                                                     Flex<RA, CB>::LDS_FLOATS * 4) != hipSuccess) {
        (void)hipGetLastError();
        n = 1;
    }
    return n < 1 ? 1 : (n > 4 ? 4 : n);
}

struct FlexResidency {
    int wg[2][kFlexConfigs];  // [vslices][config]
    FlexResidency()
    {
#define KJ_RES(I_, RA_, CB_)                               \
    wg[0][I_] = flex_residency_of<RA_, CB_, false>();      \
    wg[1][I_] = flex_residency_of<RA_, CB_, true>();
        KJ_RES(0, 1, 3) KJ_RES(1, 1, 4) KJ_RES(2, 1, 6) KJ_RES(3, 1, 8) KJ_RES(4, 1, 9) KJ_RES(5, 1, 12)
        KJ_RES(6, 2, 3) KJ_RES(7, 2, 4) KJ_RES(8, 2, 6) KJ_RES(9, 2, 8) KJ_RES(10, 2, 9) KJ_RES(11, 2, 12)
#undef KJ_RES
    }
};

inline int flex_residency(int ra, int cb, bool vs)
{
    static const FlexResidency table;  // (one device kind per process: every GPU of a node is the same part)
    int ci = 0;
    while (ci < 6 && kFlexCb[ci] != cb) ++ci;
    return table.wg[vs ? 1 : 0][(ra - 1) * 6 + (ci < 6 ? ci : 0)];
}

// Estimated time of a call in shader cycles (fitted to tools/flex_probe.py on the MiniLM shapes, 1 024 .. 8 192 rows): the
// busiest CU's tiles at the K-loop's measured efficiency (1.2 x the MFMA issue time; 1.3 x for the one-wave-per-SIMD tiles) --
// not below what staging the operands through L2 allows --, a fixed cost per round of resident workgroups (launch ramp,
// prologue, the epilogue's own instructions) and the last round's output draining at ~3 TB/s with nothing to hide behind.
inline double flex_cost(int M, int N, int k_len, int ksplit, int ra, int cb, bool vs)
{
    const int bm = 64 * ra, bn = 16 * cb;
    const int64_t tiles = (int64_t)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * ksplit;
    const int wg = tune::flex_one_workgroup_per_cu() ? 1 : flex_residency(ra, cb, vs);
    const int64_t per_cu = (tiles + 255) / 256, rounds = (tiles + 256 * wg - 1) / (256 * wg);
    const double mfma = (double)bm * bn * k_len / 128.0 * (flex_waves_per_simd(ra, cb, vs) == 1 ? 1.3 : 1.2);
    const double stage = (double)(bm + ((bn + 31) / 32) * 32) * k_len * 4.0 / 14.0;
    const int64_t last = tiles - (rounds - 1) * 256 * wg;
    const double tail = (double)std::min<int64_t>(last, 256 * wg) * bm * bn * 4.0 / 1300.0;
    return (double)per_cu * std::max(mfma, stage) + (double)rounds * 7000.0 + tail;
}

inline FlexChoice flex_choose(int M, int N, int k_len, int ksplit, bool vs)
{
    FlexChoice best;
#ifdef KJARNI_TUNING
    if (tune::flex_config_override() > 0) {
        best.ra = tune::flex_config_override() / 100;
        best.cb = tune::flex_config_override() % 100;
        best.cost = flex_cost(M, N, k_len, ksplit, best.ra, best.cb, vs);
        return best;
    }
#endif
    for (int ra = 1; ra <= 2; ++ra)
        for (int cb : kFlexCb) {
            if (16 * cb > N && cb != kFlexCb[0]) continue;
            if (vs && ra == 2 && cb == 12) continue;  // (two register sets of 96 accumulators: spills)
            const double c = flex_cost(M, N, k_len, ksplit, ra, cb, vs);
            if (best.ra == 0 || c < best.cost) best = FlexChoice{ra, cb, c};
        }
    return best;
}

}  // namespace

bool gemm_flex_shape_ok(int64_t M, int N, int K, int64_t lda, int64_t ldy, int64_t ldr, const float* A, const float* W, const float* Y,
                        const float* bias, const float* R)
{
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return M >= 1 && M <= 65536 * 8 && N % 4 == 0 && N >= 16 && K % FBK == 0 && lda % 4 == 0 && ldy % 4 == 0 && (!R || ldr % 4 == 0) && al16(A) &&
           al16(W) && al16(Y) && al16(bias) && al16(R) && (int64_t)128 * lda * 4 < ((int64_t)1 << 31) &&
           (int64_t)224 * K * 4 < ((int64_t)1 << 31);
}

double gemm_flex_cost(int M, int N, int K, int ksplit, int logical_slices)
{
    return flex_choose(M, N, K / ksplit, ksplit, logical_slices / ksplit > 1).cost;
}

// ksplit: physical K slices (1, or the logical count); logical_slices: how many slices the result's summation order has
// (1 or 4: mid_ksplit(N, K)).  partial: raw sums to slabs P[ksplit][M][N] (no bias / epilogue) for mid_reduce_*.
hipError_t launch_gemm_flex(const float* A, int64_t lda, const float* W, const float* bias, const float* R, int64_t ldr, float* Y,
                            int64_t ldy, int M, int N, int K, GemmEpilogue epi, int ksplit, int logical_slices, float* partials,
                            hipStream_t stream)
{
    if (M <= 0) return hipSuccess;
    const int k_len = K / ksplit;
    const int vslices = logical_slices / ksplit;  // slices a workgroup adds up itself
    const FlexChoice ch = flex_choose(M, N, k_len, ksplit, vslices > 1);
    const int partial = (partials != nullptr ? 1 : 0) | (M <= kWriteThroughMaxRows ? 2 : 0);  // (the kernel's out_mode)
    float* out = partials != nullptr ? partials : Y;
#ifdef KJARNI_TUNING
    // (knock-out 9 with KJARNI_HIP_FLEX_STORE = 10 / 11 / 12 in the environment: the write-through store policies instead)
    static const int store_policy = [] { const char* e = std::getenv("KJARNI_HIP_FLEX_STORE"); return e ? std::atoi(e) : 9; }();
    const int knock = tune::flex_knockout() == 9 ? store_policy : tune::flex_knockout();
#define KJ_FLEX_DIAG(RA_, CB_, D_)                                                                                                     \
    if (ch.ra == RA_ && ch.cb == CB_ && knock == D_ && vslices <= 1)                                                    \
        return flex_launch_one<RA_, CB_, false, D_>(A, lda, W, K, bias, R, ldr, out, ldy, M, N, k_len, ksplit, 1, (int)epi, partial, stream);
    KJ_FLEX_DIAG(2, 9, 1) KJ_FLEX_DIAG(2, 9, 2) KJ_FLEX_DIAG(2, 9, 3) KJ_FLEX_DIAG(2, 9, 4) KJ_FLEX_DIAG(2, 9, 5) KJ_FLEX_DIAG(2, 9, 6) KJ_FLEX_DIAG(2, 9, 7) KJ_FLEX_DIAG(2, 9, 8) KJ_FLEX_DIAG(2, 9, 9) KJ_FLEX_DIAG(2, 9, 10) KJ_FLEX_DIAG(2, 9, 11) KJ_FLEX_DIAG(2, 9, 12)
    KJ_FLEX_DIAG(2, 12, 1) KJ_FLEX_DIAG(2, 12, 2) KJ_FLEX_DIAG(2, 12, 3) KJ_FLEX_DIAG(2, 12, 4) KJ_FLEX_DIAG(2, 12, 5) KJ_FLEX_DIAG(2, 12, 6) KJ_FLEX_DIAG(2, 12, 7) KJ_FLEX_DIAG(2, 12, 8) KJ_FLEX_DIAG(2, 12, 9) KJ_FLEX_DIAG(2, 12, 10) KJ_FLEX_DIAG(2, 12, 11) KJ_FLEX_DIAG(2, 12, 12)
#undef KJ_FLEX_DIAG
#endif
#define KJ_FLEX(RA_, CB_)                                                                                                             \
    if (ch.ra == RA_ && ch.cb == CB_) {                                                                                               \
        if (vslices > 1)                                                                                                              \
            return flex_launch_one<RA_, CB_, true>(A, lda, W, K, bias, R, ldr, out, ldy, M, N, k_len, ksplit, vslices, (int)epi, partial, \
                                                   stream);                                                                          \
        return flex_launch_one<RA_, CB_, false>(A, lda, W, K, bias, R, ldr, out, ldy, M, N, k_len, ksplit, 1, (int)epi, partial, stream); \
    }
    KJ_FLEX(1, 3) KJ_FLEX(1, 4) KJ_FLEX(1, 6) KJ_FLEX(1, 8) KJ_FLEX(1, 9) KJ_FLEX(1, 12)
    KJ_FLEX(2, 3) KJ_FLEX(2, 4) KJ_FLEX(2, 6) KJ_FLEX(2, 8) KJ_FLEX(2, 9) KJ_FLEX(2, 12)
#undef KJ_FLEX
    return hipErrorInvalidValue;
}

}  // namespace kjarni
