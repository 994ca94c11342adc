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
// Tiling: 128x128 block tile, 4 waves in a 2x2 grid, each wave a 64x64 sub-tile
// = 2x2 MFMA tiles of 32x32.  Operand tiles live in LDS as [128 rows][BK k] with
// a (BK+4)-float row stride, which makes the 16-byte fragment reads (one
// ds_read_b128 feeds four MFMAs: the k order inside an MFMA is free as long as A
// and B agree) bank-conflict free.
//
// Software pipeline.  A K-step is BK/8 = 4 phases of 16 MFMAs and every memory
// operation is issued under the MFMAs of an earlier phase:
//   phase p < last : read fragments kk = p+1                           | MFMA kk = p
//                    + 3, 3, 2 of the 8 staging pieces: write the piece of tile t+1 from its registers to the
//                      other LDS buffer, then reload the same registers with the piece of tile t+2 --
//                      one piece per 5 MFMAs
//   last phase     : barrier, read fragments kk = 0 of tile t+1        | MFMA (from registers)
// Spreading the pieces matters: with all eight ds_write_b128 of a wave in one phase (and the eight waves of
// a CU in near lockstep) the LDS store path, ~77 B/clk per CU, backs up past the 64-cycle MFMA shadow and the
// in-order waves stall on issue; one store per five MFMAs measured +4 % (K = 1536) to +6 % (K = 384) on the
// projection shapes (tools/lab/mfma_lab.hip).  An LDS-DMA variant (global_load_lds, XOR-swizzled unpadded
// tiles) was bit-identical and slower (87 % vs 92 %): with two buffers its data has three phases to arrive,
// the register-staged tile has a whole K-step.
//
// Epilogue: the wave's tile is transposed through its own LDS region (the operand
// buffers are free after the last barrier) so that bias / activation / residual /
// store run on 16 bytes per lane: 1 KiB per memory instruction instead of two
// 128-byte row segments.
//
// Workgroup order: XCD-aware (every XCD walks one contiguous run of tiles, N
// fastest) so the tiles that share an X row panel hit the same L2.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "device_utils.h"
#include "gemm_epilogue.h"
#include "kernels.h"
#include "tuning.h"

namespace kjarni {

namespace {

// (the tune:: predicates are the A/B switches of the tuning build, tuning.h: constant false in the shipped library)

constexpr int EPI_GELU_LIBM = 100;
constexpr int BM = 128, BN = 128;
constexpr int EPI_STRIDE = 68;  // floats per staged output row (64 + 4: conflict-light b128 reads)

template <int BKT>
struct Tile {
    static constexpr int BK = BKT;
    static constexpr int NKK = BKT / 8;            // phases (groups of 16 MFMAs) per K-step
    static constexpr int STRIDE = BKT + 4;         // floats
    static constexpr int TILE_FLOATS = BM * STRIDE;
    static constexpr int LDS_BYTES = 2 * 2 * TILE_FLOATS * 4;  // 2 operands x 2 stages
    static constexpr int V4_PER_ROW = BKT / 4;
    static constexpr int LOADS = BM * V4_PER_ROW / 256;  // float4 per thread per operand
    static constexpr int ROWS_PER_PASS = 256 / V4_PER_ROW;
    static constexpr int PIECES = 2 * LOADS;                  // 16-byte loads per thread per K-step (A then W)
    static constexpr int PIECES_PER_PHASE = (PIECES + NKK - 2) / (NKK - 1);  // spread over the phases before the barrier
    static constexpr int WAVES_PER_SIMD = 2;  // workgroups per CU
};

struct Frag {
    f32x4 a0, a1, b0, b1;
};

template <int STRIDE>
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

template <int EPI, int BKT, int DIAG, int OUT_POLICY = 0>  // OUT_POLICY: 0 plain output stores, 1 streaming
__global__ __launch_bounds__(256, Tile<BKT>::WAVES_PER_SIMD) void gemm_nt_f32_mfma(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ W, const float* __restrict__ bias,
    const float* R, int64_t ldr, float* Y, int64_t ldy, int64_t M, int N, int K, int n_tiles, int64_t total_tiles)
// R and Y are NOT __restrict__: the encoder's residual GEMMs run in place (R == Y, encoder.cpp); every
// element is read by the one lane that later stores it, and all of a round's residual loads precede its stores.
{
    using T = Tile<BKT>;
    constexpr int BK = T::BK, NKK = T::NKK, STRIDE = T::STRIDE, TILE_FLOATS = T::TILE_FLOATS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                    // [2][128][STRIDE]
    float* sB = smem + 2 * TILE_FLOATS;  // [2][128][STRIDE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int l31 = lane & 31, half = lane >> 5;

    // Workgroups are dealt round-robin over the 8 XCDs (private L2 each).  Remap so that every
    // XCD walks one contiguous run of tiles (N fastest).  Bijective for any tile count.
    // The grid is the workgroups the chip holds at once; workgroup w takes tiles w, w + gridDim.x, ... (gridDim.x is a
    // multiple of 8 whenever it is smaller than the tile count): a finished workgroup's successor does not wait for a dispatch.
    // (32-bit unsigned arithmetic -- the launcher holds the tile count below 2^31: a 64-bit division here is ~200 scalar
    // instructions between a persistent workgroup's tiles)
    const unsigned nwg = (unsigned)total_tiles;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7u;
    for (unsigned wg = blockIdx.x; wg < nwg; wg += gridDim.x) {
    const unsigned xcd = wg & 7u, slot = wg >> 3;
    const unsigned bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const unsigned m_tile = bid / (unsigned)n_tiles;
    const int64_t m0 = (int64_t)m_tile * BM;
    const int n0 = (int)(bid - m_tile * (unsigned)n_tiles) * BN;

    // Global staging pointers; rows past M are clamped (their results are never stored).
    const int ld_row = tid / T::V4_PER_ROW, ld_c4 = tid % T::V4_PER_ROW;
    // Staging loads go through buffer descriptors: a wave-uniform 128-bit resource over this tile's rows, a 32-bit
    // per-lane byte offset that never changes, and the K offset as a scalar operand.  The K-loop then carries no
    // vector address arithmetic at all -- on this part the f32 MFMAs execute on the same FP32 lanes as the VALU,
    // so every vector ALU instruction in the loop is time the matrix work does not get -- and rows past M read as
    // zeros by the hardware bounds check (their results are never stored).
    const int64_t rows_a = (M - m0 < BM) ? (M - m0) : BM;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(A + m0 * lda), 0, (int)(((rows_a - 1) * lda + K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(W + (int64_t)n0 * K), 0, (int)((int64_t)BN * K * 4), 0x00020000);
    uint32_t offA[T::LOADS], offW[T::LOADS];
#pragma unroll
    for (int i = 0; i < T::LOADS; ++i) {
        const int64_t r = ld_row + T::ROWS_PER_PASS * i;
        offA[i] = (uint32_t)((r * lda + ld_c4 * 4) * 4);
        offW[i] = (uint32_t)((r * K + ld_c4 * 4) * 4);
    }
    f32x4 ga[T::LOADS], gb[T::LOADS];
    auto ld16 = [](__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, int k0) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, k0 * 4, 0));
    };
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < T::LOADS; ++i) {
            ga[i] = ld16(rsrcA, offA[i], k0);
            gb[i] = ld16(rsrcW, offW[i], k0);
        }
    };
    const int st_off = ld_row * STRIDE + ld_c4 * 4;
    auto store_tiles = [&](int stage) {
#pragma unroll
        for (int i = 0; i < T::LOADS; ++i) {
            *reinterpret_cast<f32x4*>(sA + stage * TILE_FLOATS + st_off + T::ROWS_PER_PASS * i * STRIDE) = ga[i];
            *reinterpret_cast<f32x4*>(sB + stage * TILE_FLOATS + st_off + T::ROWS_PER_PASS * i * STRIDE) = gb[i];
        }
    };
    // piece i < LOADS: A rows, else W rows
    auto store_piece = [&](int stage, int i) {
        if (i < T::LOADS)
            *reinterpret_cast<f32x4*>(sA + stage * TILE_FLOATS + st_off + T::ROWS_PER_PASS * i * STRIDE) = ga[i];
        else
            *reinterpret_cast<f32x4*>(sB + stage * TILE_FLOATS + st_off + T::ROWS_PER_PASS * (i - T::LOADS) * STRIDE) =
                gb[i - T::LOADS];
    };
    auto load_piece = [&](int i, int k0) {
        if (i < T::LOADS)
            ga[i] = ld16(rsrcA, offA[i], k0);
        else
            gb[i - T::LOADS] = ld16(rsrcW, offW[i - T::LOADS], k0);
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
    // The epilogue's bias slice, requested now: after the K-loop its latency would be exposed once per tile.
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + n0 + wc * 64 + (lane & 15) * 4);

    // Prologue: tile 0 -> LDS[0], tile 1 in flight in registers, fragments kk=0 of tile 0.
    load_tiles(0);
    store_tiles(0);
    if (nk > 1) load_tiles(BK);
    __syncthreads();
    Frag fr[2];
    read_frag<STRIDE>(fr[0], sA + a_off, sB + b_off, 0);

    // One K-step.  STORE: tile kt+1 exists (registers -> other LDS buffer, and its first
    // fragments are fetched after the barrier); LOAD: tile kt+2 exists (global -> registers).
    auto step = [&](auto store_tag, auto load_tag, int kt) {
        constexpr bool STORE = decltype(store_tag)::value;
        constexpr bool LOAD = decltype(load_tag)::value;
        const int cur = kt & 1;
        const float* pa = sA + cur * TILE_FLOATS + a_off;
        const float* pb = sB + cur * TILE_FLOATS + b_off;
#pragma unroll
        for (int p = 0; p < NKK; ++p) {
            Frag& use = fr[p & 1];
            Frag& nxt = fr[(p + 1) & 1];
            if (p + 1 < NKK) {
                read_frag<STRIDE>(nxt, pa, pb, p + 1);
            } else {
                // Everyone is done reading LDS[cur] (fragments are in registers) and has written
                // LDS[cur^1]: cross the barrier, fetch the next tile's first fragments, then run the
                // last 16 MFMAs of this tile from registers.
                __syncthreads();
                if (STORE)
                    read_frag<STRIDE>(nxt, sA + (cur ^ 1) * TILE_FLOATS + a_off, sB + (cur ^ 1) * TILE_FLOATS + b_off, 0);
            }
            // This phase's share of the staging pieces: LDS write of tile kt+1, then the same registers take tile kt+2.
            constexpr int PP = T::PIECES_PER_PHASE;
#pragma unroll
            for (int i = 0; i < PP; ++i) {
                const int piece = p * PP + i;
                if (p + 1 < NKK && piece < T::PIECES) {
                    if (STORE) store_piece(cur ^ 1, piece);
                    if (LOAD) load_piece(piece, (kt + 2) * BK);
                }
            }
            mfma16(acc, use);
            if (p + 1 < NKK) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // 4 x DS read first
#pragma unroll
            for (int i = 0; i < PP; ++i) {
                const int piece = p * PP + i;
                if (p + 1 < NKK && piece < T::PIECES && STORE) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // MFMA x 4
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
                    if (LOAD) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    static_assert((NKK & 1) == 0, "fragment double-buffer parity must repeat every K-step");
    using TT = std::true_type;
    using FF = std::false_type;
    int kt = 0;
    for (; kt + 2 < nk; ++kt) step(TT{}, TT{}, kt);
    if (kt + 1 < nk) step(TT{}, FF{}, kt++);
    step(FF{}, FF{}, kt);

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
        __syncthreads();
        continue;
    }

    // Epilogue through LDS (wave-private region, LDS operations of one wave execute in order), 32 rows per round.  The regions
    // lie in the SECOND stage of the operand tiles (waves 0 / 1 in A's, waves 2 / 3 in B's: 2 x 32 x 68 floats fit a 128 x 36
    // stage), which nobody reads after the barrier of the last K-step and which the next tile does not write before its
    // own prologue barrier: a wave that is done goes straight on to request and stage the next tile's first operands (into
    // the FIRST stage) while slower waves are still in their epilogue -- no barrier closes a tile.
    // That is the plain epilogue's scheme (the persistent launch: QKV +0.5 %).  Epilogues with arithmetic (GELU, ...) run one
    // workgroup per tile and keep ONE round of 64 rows over the whole operand area: sixteen independent 16-byte pieces per
    // thread in flight through the activation instead of eight (FC1 + GELU measured 1.7 % slower in two rounds).
    // (Round 5 measured the alternative without the transpose -- operands swapped so that a lane holds one output row and four
    // registers a 16-byte piece of it, stores straight from the accumulators: bit-identical, 10 % SLOWER; a store instruction
    // then covers 32 rows x 32 bytes instead of 4 rows x 256.  Where a K = 384 tile's epilogue goes (tuning build, gemm variants
    // 9 / 52): the global stores 4.5-5 % of the kernel, the LDS transpose + bias 1.5 %, the GELU arithmetic 5 % more.)
    constexpr bool kSplitRounds = EPI == EPI_BIAS;
    constexpr int EROWS = kSplitRounds ? 32 : 64;
    static_assert(2 * 32 * EPI_STRIDE <= TILE_FLOATS, "two waves' epilogue regions must fit one operand stage");
    static_assert(4 * 64 * EPI_STRIDE * 4 <= T::LDS_BYTES, "the one-round epilogue needs the whole operand area");
    float* sw = kSplitRounds ? (wid < 2 ? sA : sB) + TILE_FLOATS + (wid & 1) * (EROWS * EPI_STRIDE) : smem + wid * (EROWS * EPI_STRIDE);
    const int e_row = lane >> 4, e_c4 = lane & 15;
    const int n = n0 + wc * 64 + e_c4 * 4;
#pragma unroll
    for (int round = 0; round < 64 / EROWS; ++round) {
#pragma unroll
        for (int ii = 0; ii < EROWS / 32; ++ii) {
            const int i = round * (EROWS / 32) + ii;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sw[(ii * 32 + acc_row(r, half)) * EPI_STRIDE + j * 32 + l31] = acc[i][j][r];
        }
        const int64_t m_base = m0 + wr * 64 + round * EROWS + e_row;
        // Fully unrolled: all residual loads of the round are issued back to back (the accumulator
        // registers are free once the tile sits in LDS), instead of a few exposed round trips.
        f32x4 res[EROWS / 4];
        if (EPI == EPI_BIAS_RESIDUAL || EPI == EPI_BIAS_MUL_SILU) {
#pragma unroll
            for (int it = 0; it < EROWS / 4; ++it) {
                int64_t m = m_base + it * 4;
                m = m < M ? m : M - 1;
                res[it] = *reinterpret_cast<const f32x4*>(R + m * ldr + n);
            }
        }
#pragma unroll
        for (int it = 0; it < EROWS / 4; ++it) {
            const int64_t m = m_base + it * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(sw + (it * 4 + e_row) * EPI_STRIDE + e_c4 * 4);
            v += bv;
            if (EPI == EPI_BIAS_RESIDUAL) v += res[it];
            if (EPI == EPI_BIAS_MUL_SILU) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] *= silu_ref(res[it][c]);
            }
            if (EPI == EPI_BIAS_GELU) {
                const f32x2 lo = gelu_erf_fast2(f32x2{v[0], v[1]}), hi = gelu_erf_fast2(f32x2{v[2], v[3]});
                v = f32x4{lo[0], lo[1], hi[0], hi[1]};
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = epilogue<EPI>(v[c]);
            }
            if (DIAG == 2) {  // (diagnostic: the epilogue's LDS transpose and arithmetic without its global stores)
                if (v[0] + v[1] + v[2] + v[3] == 123456.789f) Y[0] = v[0];
            } else if (m < M) {
                // OUT_POLICY 1 (outputs larger than the memory-side cache): streaming stores.  Plain ones leave the tile's 64 KB in
                // the XCD's L2, where they evict the A panels and weights the neighbouring tiles are still reading -- FC1 + GELU
                // at 262 144 rows fetched 1.31 GB for 0.41 GB of operands (PMC, profiles/r04n_traffic_ab.log: 0.43 GB with streaming
                // stores, QKV 0.96 -> 0.50); the time is the same within 0.3 %, the HBM traffic of the layer 14 % lower.
                // (A template parameter, not a kernel argument: the compiler merges the two stores of a run-time choice into one
                // plain store.)
                f32x4* dst = reinterpret_cast<f32x4*>(Y + m * ldy + n);
                if (OUT_POLICY == 1) {
                    __builtin_nontemporal_store(v, dst);
                } else {
                    *dst = v;
                }
            }
        }
    }
    if (!kSplitRounds) __syncthreads();  // (the one-round staging overlaps the first operand stage of a next tile)
    }
}

// ---------------------------------------------------------------------------------------------------
// The same 128 x 128 tiles as ONE continuous K-stream per workgroup (the headline's K = 384 projections: QKV and FC1 + GELU
// at >= 10^5 rows, where a tile is only 12 K-steps and its edges are what the kernel above loses -- r05 knock-outs: a tile
// without an epilogue runs at the K = 1536 rate, the output stores alone cost 4.5-5 %).
//
// Vector-memory operations of a wave complete in issue order (loads and stores share `vmcnt`), so the first operand load
// a wave issues AFTER its sixteen output stores cannot be consumed before those stores are acknowledged by the L2
// (600-3 000 cycles under load).  In the kernel above that load is the next tile's prologue: the wave sits out the store
// round trip AND its own load latency once per tile.  Here a workgroup's tiles form one stream of K-blocks: the last two
// K-steps of a tile request and stage the first two blocks of the NEXT tile exactly as any other step does (another buffer
// descriptor, nothing else), so there is no prologue after the first tile, the blocks consumed right after an epilogue were
// requested before its stores, and the first load issued behind the stores is not needed until a whole K-step
// (4 096 matrix cycles) later.  Bias and activation run on the accumulators in place (64 independent values per lane, before
// the transpose); the transpose uses the second operand stage, free from the last K-step's barrier until the next tile's
// first K-step stages into it -- the one barrier that closes a tile.
// Same K order per output as the kernel above: bit-identical results.
template <int EPI, int DIAG, int OUT_POLICY>
__global__ __launch_bounds__(256, 2) void gemm_nt_f32_stream(const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
                                                             const float* __restrict__ bias, float* __restrict__ Y, int64_t ldy, int64_t M,
                                                             int K, int n_tiles, int64_t total_tiles)
{
    static_assert(EPI == EPI_BIAS || EPI == EPI_BIAS_GELU, "epilogues without a residual operand");
    using T = Tile<32>;
    constexpr int BK = T::BK, NKK = T::NKK, STRIDE = T::STRIDE, TILE_FLOATS = T::TILE_FLOATS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                    // [2][128][STRIDE]
    float* sB = smem + 2 * TILE_FLOATS;  // [2][128][STRIDE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int l31 = lane & 31, half = lane >> 5;

    // tile order: as in gemm_nt_f32_mfma (every XCD walks one contiguous run of tiles, N fastest)
    const unsigned nwg = (unsigned)total_tiles;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7u;
    struct Where {
        int64_t m0;
        int n0;
    };
    auto tile_of = [&](unsigned wg) {
        const unsigned xcd = wg & 7u, slot = wg >> 3;
        const unsigned bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
        const unsigned m_tile = bid / (unsigned)n_tiles;
        return Where{(int64_t)m_tile * BM, (int)(bid - m_tile * (unsigned)n_tiles) * BN};
    };
    // (descriptors of a tile that does not exist have no extent: its loads return zeros and move nothing)
    auto rsrc_a = [&](Where w, bool live) {
        const int64_t rows = (M - w.m0 < BM) ? (M - w.m0) : BM;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + w.m0 * lda), 0, live ? (int)(((rows - 1) * lda + K) * 4) : 0, 0x00020000);
    };
    auto rsrc_w = [&](Where w, bool live) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W + (int64_t)w.n0 * K), 0, live ? (int)((int64_t)BN * K * 4) : 0, 0x00020000);
    };

    const int ld_row = tid / T::V4_PER_ROW, ld_c4 = tid % T::V4_PER_ROW;
    uint32_t offA[T::LOADS], offW[T::LOADS];
#pragma unroll
    for (int i = 0; i < T::LOADS; ++i) {
        const int64_t r = ld_row + T::ROWS_PER_PASS * i;
        offA[i] = (uint32_t)((r * lda + ld_c4 * 4) * 4);
        offW[i] = (uint32_t)((r * K + ld_c4 * 4) * 4);
    }
    f32x4 ga[T::LOADS], gb[T::LOADS];
    auto ld16 = [](__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, int k0) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, k0 * 4, 0));
    };
    const int st_off = ld_row * STRIDE + ld_c4 * 4;
    auto store_piece = [&](int stage, int i) {
        if (i < T::LOADS)
            *reinterpret_cast<f32x4*>(sA + stage * TILE_FLOATS + st_off + T::ROWS_PER_PASS * i * STRIDE) = ga[i];
        else
            *reinterpret_cast<f32x4*>(sB + stage * TILE_FLOATS + st_off + T::ROWS_PER_PASS * (i - T::LOADS) * STRIDE) = gb[i - T::LOADS];
    };
    auto load_piece = [&](__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rw, int i, int k0) {
        if (i < T::LOADS)
            ga[i] = ld16(ra, offA[i], k0);
        else
            gb[i - T::LOADS] = ld16(rw, offW[i - T::LOADS], k0);
    };

    const int nk = K / BK;  // (even and >= 4: the launcher)
    const int a_off = (wr * 64 + l31) * STRIDE + half * 4;
    const int b_off = (wc * 64 + l31) * STRIDE + half * 4;

    unsigned wg = blockIdx.x;
    Where here = tile_of(wg);
    __amdgpu_buffer_rsrc_t rA = rsrc_a(here, true), rW = rsrc_w(here, true);
    // the one prologue of the workgroup: block 0 -> LDS[0], block 1 in registers, the first fragments
#pragma unroll
    for (int i = 0; i < T::PIECES; ++i) load_piece(rA, rW, i, 0);
#pragma unroll
    for (int i = 0; i < T::PIECES; ++i) store_piece(0, i);
#pragma unroll
    for (int i = 0; i < T::PIECES; ++i) load_piece(rA, rW, i, BK);
    __syncthreads();
    Frag fr[2];
    read_frag<STRIDE>(fr[0], sA + a_off, sB + b_off, 0);

    f32x16 acc[2][2];
    float* sw = (wid < 2 ? sA : sB) + TILE_FLOATS + (wid & 1) * (32 * EPI_STRIDE);  // this wave's transpose region, second stage
    static_assert(2 * 32 * EPI_STRIDE <= TILE_FLOATS, "two waves' epilogue regions must fit one operand stage");
    const int e_row = lane >> 4, e_c4 = lane & 15;
    // Outputs and bias through descriptors as well: rows past M are dropped by the bounds check and a missing bias reads as
    // zeros, so that NO vector-memory instruction of the tile loop sits under a branch -- the compiler then counts `vmcnt`
    // exactly across the tile edge (a store under `if (m < M)` makes it assume the stores were not issued, and the waits of the
    // K-steps behind them would then include them after all).
    const uint32_t off_y = (uint32_t)((((int64_t)wr * 64 + e_row) * ldy + wc * 64 + e_c4 * 4) * 4);
    // K-step kt of a tile: matrix work on block kt from LDS[kt & 1]; the block in the registers (kt + 1, or the next tile's
    // first) goes to the other stage and the registers take the block after that, described by (ra, rw, k0).
    // DRAIN (the first K-step after an epilogue): the previous tile's sixteen output pieces leave from `hold`, four per phase,
    // dealt out between the MFMAs like the staging pieces.
    f32x4 hold[16];
    __amdgpu_buffer_rsrc_t rY_prev = __builtin_amdgcn_make_buffer_rsrc(Y, 0, 0, 0x00020000);
    auto put = [&](int piece_of_tile) {
        // piece 8 i + it of a tile: rows i * 32 + it * 4 + (lane >> 4) of the wave's 64, 16 bytes of each
        // (OUT_POLICY 1: streaming stores, as in the kernel above -- cache-policy bit 1 = nt)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, hold[piece_of_tile]), rY_prev,
                                               off_y, (int)((int64_t)(piece_of_tile * 4) * ldy * 4), OUT_POLICY == 1 && DIAG != 5 ? 2 : 0);
    };
    auto step = [&](auto drain_tag, int kt, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rw, int k0) {
        constexpr bool DRAIN = decltype(drain_tag)::value && (DIAG == 0 || DIAG >= 3);
        const int cur = kt & 1;
        const float* pa = sA + cur * TILE_FLOATS + a_off;
        const float* pb = sB + cur * TILE_FLOATS + b_off;
#pragma unroll
        for (int p = 0; p < NKK; ++p) {
            Frag& use = fr[p & 1];
            Frag& nxt = fr[(p + 1) & 1];
            if (p + 1 < NKK) {
                read_frag<STRIDE>(nxt, pa, pb, p + 1);
            } else {
                __syncthreads();
                read_frag<STRIDE>(nxt, sA + (cur ^ 1) * TILE_FLOATS + a_off, sB + (cur ^ 1) * TILE_FLOATS + b_off, 0);
            }
            constexpr int PP = T::PIECES_PER_PHASE;
#pragma unroll
            for (int i = 0; i < PP; ++i) {
                const int piece = p * PP + i;
                if (p + 1 < NKK && piece < T::PIECES) {
                    store_piece(cur ^ 1, piece);
                    load_piece(ra, rw, piece, k0);
                }
            }
            // (all sixteen in the phases before the barrier, 6 + 5 + 5: the waits of the NEXT K-step are the loop's static counts and
            // take in every store issued before them -- the later a store, the less time its acknowledgement has had)
            const int first_out = p == 0 ? 0 : 1 + 5 * p, n_out = p + 1 < NKK ? (p == 0 ? 6 : 5) : 0;
            if (DRAIN) {
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    if (q < n_out) put(first_out + q);
            }
            mfma16(acc, use);
            if (p + 1 < NKK) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // 4 x DS read first
            int drained = 0;
#pragma unroll
            for (int i = 0; i < PP; ++i) {
                const int piece = p * PP + i;
                if (p + 1 < NKK && piece < T::PIECES) {
                    __builtin_amdgcn_sched_group_barrier(0x008, DRAIN ? 2 : 4, 0);  // MFMA x 4
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);              // DS write
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);              // VMEM read
                    if (DRAIN) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);  // VMEM write
                        ++drained;
                    }
                }
            }
            if (DRAIN) {
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    if (q >= drained && q < n_out) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);  // VMEM write
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    static_assert((NKK & 1) == 0, "fragment double-buffer parity must repeat every K-step");
    static_assert(NKK == 4, "the drain deals 6 + 5 + 5 output pieces over three phases");
    using TT = std::true_type;
    using FF = std::false_type;

    const uint32_t off_bias = (uint32_t)((wc * 64 + l31) * 4);
    const __amdgpu_buffer_rsrc_t r_bias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, bias ? n_tiles * BN * 4 : 0, 0x00020000);
    auto rsrc_y = [&](Where w) {
        const int64_t rows = (M - w.m0 < BM) ? (M - w.m0) : BM;
        return __builtin_amdgcn_make_buffer_rsrc(Y + w.m0 * ldy + w.n0, 0, (int)(((rows - 1) * ldy + BN) * 4), 0x00020000);
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    };
    auto bias_of = [&](Where w, int j) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_bias, off_bias + j * 128, w.n0 * 4, 0));
    };

    unsigned wg_next = wg + gridDim.x;
    bool more = wg_next < nwg;
    Where next = tile_of(more ? wg_next : wg);
    __amdgpu_buffer_rsrc_t rAn = rsrc_a(next, more), rWn = rsrc_w(next, more);
    float b0 = bias_of(here, 0), b1 = bias_of(here, 1);
    zero_acc();
    // (DIAG 3, tuning build: shader cycles this wave spends in the K-steps / between them, summed over its tiles)
    uint64_t t_loop = 0, t_edge = 0, t_mark = DIAG == 3 ? __builtin_amdgcn_s_memtime() : 0;
    unsigned tiles_done = 0;
    step(FF{}, 0, rA, rW, 2 * BK);
    for (;;) {
        // (K-step 1 apart from the loop: its waits are then counted exactly behind the drained stores, and the loop's static
        // counts -- which take in every older store -- start a whole K-step after the last of them)
        step(FF{}, 1, rA, rW, 3 * BK);
        int kt = 2;
        for (; kt + 2 < nk; ++kt) step(FF{}, kt, rA, rW, (kt + 2) * BK);
        step(FF{}, kt++, rAn, rWn, 0);
        step(FF{}, kt, rAn, rWn, BK);
        if (DIAG == 3) {
            const uint64_t t = __builtin_amdgcn_s_memtime();
            t_loop += t - t_mark;
            t_mark = t;
            ++tiles_done;
        }

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
        } else {
            // bias (+ activation) on the accumulators, then the transpose into `hold`: 16 bytes of four rows per register quad
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        f32x2 v = f32x2{acc[i][j][r], acc[i][j][r + 1]} + (j ? b1 : b0);
                        if (EPI == EPI_BIAS_GELU) v = gelu_erf_fast2(v);
                        acc[i][j][r] = v[0];
                        acc[i][j][r + 1] = v[1];
                    }
            rY_prev = rsrc_y(DIAG == 5 ? Where{(int64_t)(blockIdx.x / (unsigned)n_tiles) * BM, (int)(blockIdx.x % (unsigned)n_tiles) * BN} : here);  // (5: every tile of a workgroup to the same place)
            if (DIAG == 4) {
                // (diagnostic: the same stores without the LDS transpose and without the barrier -- the values land in the wrong places)
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    hold[q] = f32x4{acc[q >> 3][(q >> 2) & 1][(q & 3) * 4], acc[q >> 3][(q >> 2) & 1][(q & 3) * 4 + 1],
                                    acc[q >> 3][(q >> 2) & 1][(q & 3) * 4 + 2], acc[q >> 3][(q >> 2) & 1][(q & 3) * 4 + 3]};
            } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sw[acc_row(r, half) * EPI_STRIDE + j * 32 + l31] = acc[i][j][r];
#pragma unroll
                for (int it = 0; it < 8; ++it)
                    hold[i * 8 + it] = *reinterpret_cast<const f32x4*>(sw + (it * 4 + e_row) * EPI_STRIDE + e_c4 * 4);
            }
            }
            if (DIAG == 2) {  // (diagnostic: everything but the global stores)
                f32x4 t = hold[0];
#pragma unroll
                for (int q = 1; q < 16; ++q) t += hold[q];
                if (t[0] + t[1] + t[2] + t[3] == 123456.789f) Y[0] = t[0];
            }
        }
        if (!more) {
            if (DIAG == 0 || DIAG >= 3) {
#pragma unroll
                for (int q = 0; q < 16; ++q) put(q);
            }
            if (DIAG == 3) {
                __builtin_amdgcn_s_waitcnt(0);
                __syncthreads();
                if (tid == 0) {
                    float* o = Y + (int64_t)blockIdx.x * 4;
                    o[0] = (float)t_loop;
                    o[1] = (float)t_edge;
                    o[2] = (float)tiles_done;
                }
            }
            break;
        }
        if (DIAG != 4) __syncthreads();  // every wave has read its transpose region: the next K-step stages into that LDS
        if (DIAG == 3) {
            const uint64_t t = __builtin_amdgcn_s_memtime();
            t_edge += t - t_mark;
            t_mark = t;
        }
        wg = wg_next;
        here = next;
        rA = rAn;
        rW = rWn;
        wg_next = wg + gridDim.x;
        more = wg_next < nwg;
        next = tile_of(more ? wg_next : wg);
        rAn = rsrc_a(next, more);
        rWn = rsrc_w(next, more);
        b0 = bias_of(here, 0);
        b1 = bias_of(here, 1);
        zero_acc();
        step(TT{}, 0, rA, rW, 2 * BK);
    }
}

// ---------------------------------------------------------------------------------------------------
// Residual GEMM with the LayerNorm folded into its epilogue:
//     Y = LayerNorm(A W^T + bias + R) * gamma + beta          (R == Y allowed)
// replaces out-proj / FC2 + residual add + LayerNorm of the post-norm layer
// (cpu/encoder/encoder_layer.rs:129-147, 155-176; cpu/normalization/layer_norm.rs:37-131): one launch and
// one pass over the [T, H] residual stream instead of two launches and two passes.
//
// LayerNorm needs whole rows, so a workgroup owns BM = 64 complete rows: block tile 64 x N with
// N = 128 * NT (NT = 3: the 384-wide MiniLM rows), 4 waves side by side, each 64 rows x (32 * NT) columns =
// 2 x NT accumulator tiles.  BK = 16 keeps the double-buffered operand tiles at (64 + N) * 20 * 4 * 2 bytes
// = 70 KiB for N = 384, i.e. two workgroups per CU, so one workgroup's epilogue (no MFMA work) runs under the
// other's K-loop.  Row stride BK + 4 = 20 floats: the 16 lanes of a ds_read_b128 group cover all 16 16-byte
// slots of the 256-byte bank row.  Same software pipeline as gemm_nt_f32_mfma (2 phases of 8 * NT MFMAs per
// K-step, tile t+1 written to LDS and tile t+2 requested under the first phase's MFMAs).
//
// Epilogue, two rounds of 32 rows: the four waves drop their accumulators into one [32][N + 32] LDS tile (row
// stride = 8 sixteen-byte slots mod 16: conflict-free 16-byte reads for the 8-lanes-per-row pattern below),
// then 8 lanes own a row: v = acc + bias + R, two-pass statistics in registers (sum, then sum of squared
// deviations; 3 butterfly steps across the 8 lanes), normalise, 16-byte stores.  Population variance, eps
// inside the sqrt, as the reference.
template <int NT>
struct LnTile {
    static constexpr int BM = 64, BK = 16, NKK = 2, STRIDE = BK + 4;
    static constexpr int BN = 128 * NT;
    static constexpr int STAGE_FLOATS = (BM + BN) * STRIDE;
    static constexpr int OUT_STRIDE = BN + 32;
    static constexpr int OUT_FLOATS = 32 * OUT_STRIDE;
    static constexpr int LDS_FLOATS = 2 * STAGE_FLOATS > OUT_FLOATS ? 2 * STAGE_FLOATS : OUT_FLOATS;
    // + bias | gamma | beta: the epilogue's per-column vectors wait in LDS from the start of the tile (3 x 1.5 KiB for 384 columns;
    // two workgroups still fit a CU) instead of being 72 global loads per thread per tile in front of the statistics
    static constexpr int PARAM_FLOATS = 3 * BN;
    static constexpr int LDS_BYTES = (LDS_FLOATS + PARAM_FLOATS) * 4;
    static constexpr int B_LOADS = BN * (BK / 4) / 256;  // 16-byte loads per thread per K-step for W (A: one)
    static constexpr int V4_PER_THREAD = BN / 4 / 8;     // epilogue: 8 lanes per row
};

template <int NT, bool PARAMS_IN_LDS>
__global__ __launch_bounds__(256, 2) void gemm_nt_f32_mfma_ln(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ W, const float* __restrict__ bias,
    const float* R, int64_t ldr, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    float* Y, int64_t ldy, int64_t M, int K, int64_t total_tiles)
{
    using T = LnTile<NT>;
    constexpr int BM = T::BM, BK = T::BK, STRIDE = T::STRIDE, BN = T::BN;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;

    // (persistent over the row tiles, as gemm_nt_f32_mfma)
    const int64_t nwg = total_tiles;
    const int64_t q8 = nwg / 8, r8 = nwg % 8;
    for (int64_t wg = blockIdx.x; wg < total_tiles; wg += gridDim.x) {
    const int64_t xcd = wg % 8, slot = wg / 8;
    const int64_t bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int64_t m0 = bid * BM;
    float* sP = smem + T::LDS_FLOATS;  // [bias | gamma | beta][BN]
    if (PARAMS_IN_LDS) {
        // (its own region: no hazard with the operand stages; the K-loop's barriers order it before the epilogue)
        for (int q = tid; q < 3 * BN / 4; q += 256) {
            const int which = q / (BN / 4), c = (q - which * (BN / 4)) * 4;
            const float* src = which == 0 ? bias : which == 1 ? gamma : beta;
            *reinterpret_cast<f32x4*>(sP + which * BN + c) = src ? *reinterpret_cast<const f32x4*>(src + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    // staging: thread -> (row, 16-byte column) of the [rows][BK] operand tiles, through buffer descriptors (scalar
    // K offset, constant 32-bit lane offsets: no vector address arithmetic in the K-loop; rows past M read as zeros)
    // A row of a tile is 64 bytes here, so the 8 lanes of one ds_write_b128 group cover two rows; with the
    // 80-byte row stride rows r and r + 4 start 16 banks apart (320 B mod 128 B = 64 B): lanes 0-3 take row r,
    // lanes 4-7 row r + 4 and the group touches every bank once (r and r + 1 would overlap in four banks: the
    // PMC pass showed a third of this kernel's LDS cycles as bank conflicts with that map).
    const int ld_grp = tid >> 3;
    const int ld_row = (ld_grp >> 2) * 8 + (ld_grp & 3) + 4 * ((tid >> 2) & 1), ld_c4 = tid & 3;
    const int64_t rows_a = (M - m0 < BM) ? (M - m0) : BM;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(A + m0 * lda), 0, (int)(((rows_a - 1) * lda + K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, (int)((int64_t)BN * K * 4), 0x00020000);
    const uint32_t offA = (uint32_t)(((int64_t)ld_row * lda + ld_c4 * 4) * 4);
    uint32_t offW[T::B_LOADS];
#pragma unroll
    for (int i = 0; i < T::B_LOADS; ++i) offW[i] = (uint32_t)(((int64_t)(ld_row + 64 * i) * K + ld_c4 * 4) * 4);
    f32x4 ga, gb[T::B_LOADS];
    auto ld16 = [](__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, int k0) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, k0 * 4, 0));
    };
    auto load_tiles = [&](int k0) {
        ga = ld16(rsrcA, offA, k0);
#pragma unroll
        for (int i = 0; i < T::B_LOADS; ++i) gb[i] = ld16(rsrcW, offW[i], k0);
    };
    const int st_off = ld_row * STRIDE + ld_c4 * 4;
    auto store_tiles = [&](int stage) {
        float* base = smem + stage * T::STAGE_FLOATS;
        *reinterpret_cast<f32x4*>(base + st_off) = ga;
#pragma unroll
        for (int i = 0; i < T::B_LOADS; ++i)
            *reinterpret_cast<f32x4*>(base + BM * STRIDE + st_off + 64 * i * STRIDE) = gb[i];
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    struct LnFrag {
        f32x4 a[2], b[NT];
    };
    const int a_off = l31 * STRIDE + half * 4;
    const int b_off = BM * STRIDE + (wid * 32 * NT + l31) * STRIDE + half * 4;
    auto read_frag = [&](LnFrag& f, int stage, int kk) {
        const float* base = smem + stage * T::STAGE_FLOATS + kk * 8;
        f.a[0] = *reinterpret_cast<const f32x4*>(base + a_off);
        f.a[1] = *reinterpret_cast<const f32x4*>(base + a_off + 32 * STRIDE);
#pragma unroll
        for (int j = 0; j < NT; ++j) f.b[j] = *reinterpret_cast<const f32x4*>(base + b_off + j * 32 * STRIDE);
    };
    auto mfma_phase = [&](const LnFrag& f) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][c], f.b[j][c], acc[i][j], 0, 0, 0);
    };

    const int nk = K / BK;
    load_tiles(0);
    store_tiles(0);
    if (nk > 1) load_tiles(BK);
    __syncthreads();
    LnFrag fr[2];
    read_frag(fr[0], 0, 0);

    auto step = [&](auto store_tag, auto load_tag, int kt) {
        constexpr bool STORE = decltype(store_tag)::value;
        constexpr bool LOAD = decltype(load_tag)::value;
        const int cur = kt & 1;
        // phase 0: fragments kk = 1 | MFMAs kk = 0, with the 1 + B_LOADS staging pieces spread under them: piece =
        // LDS write of tile kt+1 from its registers, then the same registers reloaded with tile kt+2
        read_frag(fr[1], cur, 1);
        float* wbase = smem + (cur ^ 1) * T::STAGE_FLOATS;
        if (STORE) *reinterpret_cast<f32x4*>(wbase + st_off) = ga;
        if (LOAD) ga = ld16(rsrcA, offA, (kt + 2) * BK);
#pragma unroll
        for (int i = 0; i < T::B_LOADS; ++i) {
            if (STORE) *reinterpret_cast<f32x4*>(wbase + BM * STRIDE + st_off + 64 * i * STRIDE) = gb[i];
            if (LOAD) gb[i] = ld16(rsrcW, offW[i], (kt + 2) * BK);
        }
        mfma_phase(fr[0]);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 + NT, 0);  // DS reads first
        if (STORE) {
#pragma unroll
            for (int i = 0; i < 1 + T::B_LOADS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, NT == 3 ? 2 : 1, 0);  // MFMA
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);               // DS write
                if (LOAD) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // phase 1: everyone has read LDS[cur] and written LDS[cur^1]: barrier, fetch the next tile's first
        // fragments, then the last MFMAs of this tile from registers
        __syncthreads();
        if (STORE) read_frag(fr[0], cur ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_phase(fr[1]);
        __builtin_amdgcn_sched_barrier(0);
    };
    using TT = std::true_type;
    using FF = std::false_type;
    int kt = 0;
    for (; kt + 2 < nk; ++kt) step(TT{}, TT{}, kt);
    if (kt + 1 < nk) step(TT{}, FF{}, kt++);

    // The epilogue's residual rows.  Round 0's are requested HERE, before the last K-step (whose staging registers are
    // idle: nothing is left to stage), so their round trip runs under that step's MFMAs instead of in front of the
    // epilogue; round 1's are requested once round 0's accumulators have left their registers for LDS.
    constexpr int OS = T::OUT_STRIDE, NV = T::V4_PER_THREAD;
    const int e_row = tid >> 3, e_t8 = tid & 7;
    f32x4 res[NV];
    auto load_residual = [&](int i) {
        int64_t m = m0 + i * 32 + e_row;
        m = m < M ? m : M - 1;
#pragma unroll
        for (int q = 0; q < NV; ++q) res[q] = *reinterpret_cast<const f32x4*>(R + m * ldr + (e_t8 + 8 * q) * 4);
    };
    load_residual(0);
    __builtin_amdgcn_sched_barrier(0);
    step(FF{}, FF{}, kt);

    // ---- epilogue ----
    const float inv_n = 1.0f / (float)BN;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        __syncthreads();  // round 0: the operand tiles are dead; round 1: round 0's reads are done
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                smem[acc_row(r, half) * OS + wid * 32 * NT + j * 32 + l31] = acc[i][j][r];
        __syncthreads();
        f32x4 x[NV];
        float s = 0.0f;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int c = (e_t8 + 8 * q) * 4;
            x[q] = *reinterpret_cast<const f32x4*>(smem + e_row * OS + c);
            if (PARAMS_IN_LDS) x[q] += *reinterpret_cast<const f32x4*>(sP + c);
            else if (bias) x[q] += *reinterpret_cast<const f32x4*>(bias + c);
            x[q] += res[q];
            s += (x[q][0] + x[q][1]) + (x[q][2] + x[q][3]);
        }
        if (i == 0) load_residual(1);  // (R == Y in place: round 1's rows are not written before the second round's stores)
        s += __shfl_xor(s, 1, kWave);
        s += __shfl_xor(s, 2, kWave);
        s += __shfl_xor(s, 4, kWave);
        const float mean = s * inv_n;
        float v = 0.0f;
#pragma unroll
        for (int q = 0; q < NV; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float d = x[q][c] - mean;
                v = fmaf(d, d, v);
            }
        v += __shfl_xor(v, 1, kWave);
        v += __shfl_xor(v, 2, kWave);
        v += __shfl_xor(v, 4, kWave);
        const float inv_std = 1.0f / sqrtf(v * inv_n + eps);
        const int64_t m = m0 + i * 32 + e_row;
        if (m < M) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int c = (e_t8 + 8 * q) * 4;
                const f32x4 g = *reinterpret_cast<const f32x4*>((PARAMS_IN_LDS ? sP + BN : gamma) + c);
                const f32x4 b = *reinterpret_cast<const f32x4*>((PARAMS_IN_LDS ? sP + 2 * BN : beta) + c);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (x[q][e] - mean) * inv_std * g[e] + b[e];
                *reinterpret_cast<f32x4*>(Y + m * ldy + c) = o;
            }
        }
    }
    __syncthreads();  // the epilogue tile overlaps the operand tiles of the next row tile
    }
}

template <int NT>
hipError_t launch_ln_tiled(const float* A, int64_t lda, const float* W, const float* bias, const float* R, int64_t ldr,
                           const float* gamma, const float* beta, float eps, float* Y, int64_t ldy, int64_t M, int K,
                           hipStream_t stream)
{
    using T = LnTile<NT>;
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (T::LDS_BYTES > 64 * 1024 && !attr_set[dev & 63]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_f32_mfma_ln<NT, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
        if (e != hipSuccess) return e;
#ifdef KJARNI_TUNING
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_f32_mfma_ln<NT, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
        if (e != hipSuccess) return e;
#endif
        attr_set[dev & 63] = true;
    }
    const int64_t total = (M + T::BM - 1) / T::BM;
    // (a grid of the 512 resident workgroups measured +0.6 % on the K = 384 shape and -0.6 % on K = 1536: one workgroup per tile)
    dim3 grid((unsigned)(tune::persistent_layernorm_tiles() ? std::min<int64_t>(total, 256 * 2) : total));
#ifdef KJARNI_TUNING
    if (tune::layernorm_params_from_global()) {
        hipLaunchKernelGGL((gemm_nt_f32_mfma_ln<NT, false>), grid, dim3(256), T::LDS_BYTES, stream, A, lda, W, bias, R, ldr,
                           gamma, beta, eps, Y, ldy, M, K, total);
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((gemm_nt_f32_mfma_ln<NT, true>), grid, dim3(256), T::LDS_BYTES, stream, A, lda, W, bias, R, ldr,
                       gamma, beta, eps, Y, ldy, M, K, total);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Few rows (M <= kFewRowsMax = 256: one sentence to classify or embed, a handful -- BASELINE.json configs[0]).  The tiled kernels
// above are sized for 10^5 rows; at M = 28 they run N / 128 workgroups that each walk the whole K dimension alone
// (25-124 us per launch, 3-12 of the 256 CUs busy).  Here a workgroup of 16 waves owns 32 output columns of a group of 32
// (or 64) rows and splits K sixteen ways; a call is N / 32 x ceil(M / 32) such workgroups (blockIdx.z = the row group).  No
// operand staging: a wave's MFMA fragments come straight from global memory through buffer descriptors (rows >= M read
// as zeros) -- within a row group every weight element is needed by exactly one wave.  Per 32-row tile the sixteen partial
// tiles meet in LDS (64 KiB) and are summed in wave order, then bias / activation / residual and 128-byte row-segment
// stores.  An output's arithmetic -- a k-ordered MFMA chain per K slice, then the sixteen partials in order -- does not
// depend on M or on the row's place in the call, so a row's result is the same in any call of up to kFewRowsMax rows.
// Why rows over the grid and not more rows per workgroup: a workgroup's MFMAs run on ONE CU however its K is dealt over the
// waves (32 columns x 128 rows x K = 384: 5.1 us of matrix time), and a call this small leaves most of the 256 CUs idle --
// one 128-token sentence went 0.445 -> 0.333 ms when its four row tiles became four workgroups; from ~400 rows the
// 64 x 64-tile route below is faster (512 rows: 0.455 against 0.492 ms; measured with tools/few_rows_sweep.sh).
// SLICED (the residual + LayerNorm projections with a long K, FC2): blockIdx.y is one of gridDim.y K slices of K floats each
// (W rows ldw = gridDim.y * K apart), and the workgroup leaves its 32 columns of that slice's partial sums in slab blockIdx.y of
// Y ([gridDim.y][M][N], no epilogue) for mid_reduce_ln_kernel: 4 x N / 32 workgroups instead of N / 32 -- a workgroup's MFMAs
// run on ONE CU however its K is dealt over the waves, and FC2 on 12 CUs was 13 us of matrix time and round trips.
template <int EPI, int MT, bool SLICED = false>
__global__ __launch_bounds__(1024) void gemm_nt_f32_skinny(const float* __restrict__ A, int64_t lda,
                                                            const float* __restrict__ W, const float* __restrict__ bias,
                                                            const float* R, int64_t ldr, float* Y, int64_t ldy, int M, int N,
                                                            int K)
{
    constexpr int WAVES = 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [WAVES][32][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int ks = K / WAVES;  // this wave's K slice (a multiple of 8)
    const int64_t ldw = SLICED ? (int64_t)K * gridDim.y : K;
    if (SLICED) {
        A += (int64_t)blockIdx.y * K;
        W += (int64_t)blockIdx.y * K;
        Y += (int64_t)blockIdx.y * M * N;
        ldy = N;
    }
    // blockIdx.z: which group of MT * 32 rows (a call of a few hundred rows is that many independent few-rows problems: every
    // workgroup still has one CU's matrix pipes for its 32 columns x MT * 32 rows, and there are CUs to spare)
    {
        const int m0 = blockIdx.z * (MT * 32);
        A += (int64_t)m0 * lda;
        if (R) R += (int64_t)m0 * ldr;
        Y += (int64_t)m0 * ldy;
        M = M - m0 < MT * 32 ? M - m0 : MT * 32;
    }
    const __amdgpu_buffer_rsrc_t rA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (int)((((int64_t)M - 1) * lda + K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W + (int64_t)n0 * ldw), 0,
                                                                        (int)(((int64_t)31 * ldw + K) * 4), 0x00020000);
    uint32_t offA[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) offA[mt] = (uint32_t)(((int64_t)(mt * 32 + l31) * lda + half * 4) * 4);
    const uint32_t offW = (uint32_t)(((int64_t)l31 * ldw + half * 4) * 4);
    auto ld16 = [](__amdgpu_buffer_rsrc_t r, uint32_t off, int soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, 0));
    };
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
    const int k_begin = wid * ks;
    // Up to DEPTH K-steps of 8 in flight per wave: with K = 384 (24 per wave) ALL of the wave's fragments are requested before the
    // first MFMA -- one trip to memory per launch instead of three (a call this small is a chain of latencies, not bandwidth:
    // one 128-token sentence 0.290 -> 0.272 ms).  Steps past the slice are not requested; rows past the matrix read as zeros
    // through the descriptor.
    constexpr int DEPTH = 4;
    f32x4 b[DEPTH], a[DEPTH][MT];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        if (d * 8 < ks) {
            b[d] = ld16(rW, offW, (k_begin + d * 8) * 4);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[d][mt] = ld16(rA, offA[mt], (k_begin + d * 8) * 4);
        }
    }
    for (int k = 0; k < ks; k += 8 * DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (k + d * 8 < ks) {
                const f32x4 cb = b[d];
                f32x4 ca[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) ca[mt] = a[d][mt];
                if (k + (d + DEPTH) * 8 < ks) {
                    b[d] = ld16(rW, offW, (k_begin + k + (d + DEPTH) * 8) * 4);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) a[d][mt] = ld16(rA, offA[mt], (k_begin + k + (d + DEPTH) * 8) * 4);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[mt][c], cb[c], acc[mt], 0, 0, 0);
            }
        }
    }
    float* mine = smem + wid * (32 * 32);
    const int n = n0 + (tid & 31);
    const float bv = (!SLICED && bias) ? bias[n] : 0.0f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (mt > 0) __syncthreads();  // the previous tile's partials have been summed
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[acc_row(r, half) * 32 + l31] = acc[mt][r];
        __syncthreads();
        // 1024 outputs, one per thread, column fastest
        const int m = mt * 32 + (tid >> 5);
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) v += smem[w * (32 * 32) + tid];
        if (m < M) {
            if (SLICED) {
                Y[(int64_t)m * ldy + n] = v;
            } else {
                v += bv;
                if (EPI == EPI_BIAS_RESIDUAL) v += R[(int64_t)m * ldr + n];
                if (EPI == EPI_BIAS_MUL_SILU) v *= silu_ref(R[(int64_t)m * ldr + n]);
                Y[(int64_t)m * ldy + n] = epilogue<EPI>(v);
            }
        }
    }
}

template <int EPI>
hipError_t launch_skinny(const float* A, int64_t lda, const float* W, const float* bias, const float* R, int64_t ldr,
                         float* Y, int64_t ldy, int M, int N, int K, hipStream_t stream)
{
    constexpr int LDS = 16 * 32 * 32 * 4;  // 64 KiB
    // one 32-row tile per workgroup while that leaves the chip room (<= 2 workgroups per CU), else two
    const int z1 = (M + 31) / 32;
    if ((int64_t)z1 * (N / 32) <= 512 || M <= 32) {
        hipLaunchKernelGGL((gemm_nt_f32_skinny<EPI, 1>), dim3((unsigned)(N / 32), 1, (unsigned)z1), dim3(1024), LDS, stream, A, lda, W, bias,
                           R, ldr, Y, ldy, M, N, K);
    } else {
        hipLaunchKernelGGL((gemm_nt_f32_skinny<EPI, 2>), dim3((unsigned)(N / 32), 1, (unsigned)((M + 63) / 64)), dim3(1024), LDS, stream, A,
                           lda, W, bias, R, ldr, Y, ldy, M, N, K);
    }
    return hipGetLastError();
}

// Any-shape fallback (odd hidden sizes in tests, tiny heads): 32x32 LDS tiles,
// plain FMA.  Not on the MiniLM/BERT hot path.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_f32_generic(const float* __restrict__ A, int64_t lda,
                                                           const float* __restrict__ W,
                                                           const float* __restrict__ bias,
                                                           const float* R, int64_t ldr, float* Y, int64_t ldy,
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
                if (EPI == EPI_BIAS_MUL_SILU) v *= silu_ref(R[m * ldr + n]);
                Y[m * ldy + n] = epilogue<EPI>(v);
            }
        }
    }
}

// Outputs from this size leave through streaming stores: nothing the size of the memory-side cache (256 MB) is found there again.
constexpr int64_t kStreamOutBytes = (int64_t)256 << 20;
// K-steps (of 32) up to which a tile's edges are worth the continuous-stream kernel (gemm_nt_f32_stream): K <= 768
constexpr int kStreamMaxKSteps = 24;

template <int EPI, int BKT, int DIAG, int OUT_POLICY>
hipError_t launch_tiled_as(const float* A, int64_t lda, const float* W, const float* bias, const float* R,
                           int64_t ldr, float* Y, int64_t ldy, int64_t M, int N, int K, hipStream_t stream)
{
    using T = Tile<BKT>;
    if (T::LDS_BYTES > 64 * 1024) {
        // > 64 KiB of dynamic LDS needs the opt-in once per device.
        static bool attr_set[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (!attr_set[dev & 63]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_f32_mfma<EPI, BKT, DIAG, OUT_POLICY>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
            if (e != hipSuccess) return e;
            attr_set[dev & 63] = true;
        }
    }
    const int n_tiles = N / BN;
    const int64_t m_tiles = (M + BM - 1) / BM;
    const int64_t total = m_tiles * n_tiles;
    if (total > 0x7fffffff) return hipErrorInvalidValue;  // (the kernel's tile index is 32-bit; a chunk is <= 262 144 rows)
    // Short K and many tiles per workgroup (the headline's QKV and FC1): the tiles as one continuous K-stream.
    if constexpr ((EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) && BKT == 32) {
        const int nk = K / 32;
        if (R == nullptr && nk >= 6 && (nk & 1) == 0 && nk <= kStreamMaxKSteps && (int64_t)BM * ldy * 4 < ((int64_t)1 << 31) && total >= 4 * (int64_t)256 * T::WAVES_PER_SIMD &&
            !tune::no_k_stream_tiles()) {
            static bool attr_set[64] = {};
            int dev = 0;
            hipError_t e = hipGetDevice(&dev);
            if (e != hipSuccess) return e;
            if (!attr_set[dev & 63]) {
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_f32_stream<EPI, DIAG, OUT_POLICY>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
                if (e != hipSuccess) return e;
                attr_set[dev & 63] = true;
            }
            hipLaunchKernelGGL((gemm_nt_f32_stream<EPI, DIAG, OUT_POLICY>), dim3((unsigned)((int64_t)256 * T::WAVES_PER_SIMD)), dim3(256),
                               T::LDS_BYTES, stream, A, lda, W, bias, Y, ldy, M, K, n_tiles, total);
            return hipGetLastError();
        }
    }
    // The kernel walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...  With the plain epilogue a grid of the workgroups the
    // chip holds at once measured +2.2 % (QKV shape: no dispatch between a workgroup's tiles); with the GELU epilogue -1.1 %
    // (the barrier that ends a tile waits for the slowest wave's epilogue), so those launch one workgroup per tile.
    const int64_t resident = (int64_t)256 * T::WAVES_PER_SIMD;
    const bool persistent = EPI == EPI_BIAS && !tune::no_persistent_tile_loop();  // (GELU in the pipeline: 1.7 % slower as a persistent launch, r03l)
    dim3 grid((unsigned)(persistent ? std::min(total, resident) : total));
    hipLaunchKernelGGL((gemm_nt_f32_mfma<EPI, BKT, DIAG, OUT_POLICY>), grid, dim3(256), T::LDS_BYTES, stream, A, lda, W, bias,
                       R, ldr, Y, ldy, M, N, K, n_tiles, total);
    return hipGetLastError();
}

template <int EPI, int BKT, int DIAG>
hipError_t launch_tiled(const float* A, int64_t lda, const float* W, const float* bias, const float* R,
                        int64_t ldr, float* Y, int64_t ldy, int64_t M, int N, int K, hipStream_t stream)
{
    // (in place -- R == Y -- the stores stay plain: the rows are read again as the next projection's residual)
    if (DIAG == 0 && (int64_t)M * N * 4 >= kStreamOutBytes && R != Y && !tune::no_streaming_output_stores())
        return launch_tiled_as<EPI, BKT, 0, 1>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    return launch_tiled_as<EPI, BKT, DIAG, 0>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
}

// ---- Calls of 257 .. 8192 rows (the reference's default batch is 32 sentences) -------------------------------------------
// The large-batch tiles (128 x 128, and 64 x 384 for the fused LayerNorm) are sized for 10^5 rows: at 4 096 rows the
// 384-wide GEMMs are 64 workgroups on 256 CUs, each with the whole K-loop serial (FC2 + LN: 150 us of a 320 us layer).
// Here: 64 x 64 tiles, four waves of one 32 x 32 MFMA tile, BK = 32, LDS double buffer; narrow outputs with a long K
// (N <= 1024, K >= 1024) cut K into 4 slices.  Sliced launches, and every LayerNorm one, leave partial tiles in a scratch
// slab per slice; mid_reduce_* adds the slabs in slice order and applies the epilogue (bias, activation, residual,
// LayerNorm with a wave per row).  The slice count depends on (N, K) only, so a row's arithmetic does not depend on how
// many rows the call has.
constexpr int MID_BM = 64, MID_BN = 64, MID_BK = 32, MID_STRIDE = MID_BK + 4;

inline int mid_ksplit(int N, int K) { return (N <= 1024 && K >= 1024 && K % (4 * MID_BK) == 0) ? 4 : 1; }

// DIAG (tuning build only, tools/mid_probe.py with GEMM_VARIANT 21 / 22): knock-outs that show where a tile's time goes --
// 1 no global loads in the K-loop (stale operands), 2 no MFMAs, 3 no barrier in the K-loop (and no loads), 4 no LDS stores
// (and no loads), 5 no LDS fragment reads (and no loads), 6 nothing but the MFMAs.
template <int EPI, bool PARTIAL, int DIAG = 0>
__global__ __launch_bounds__(256) void gemm_nt_f32_mid(const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
                                                       const float* __restrict__ bias, const float* R, int64_t ldr, float* Y,
                                                       int64_t ldy, int M, int N, int K, int m_tiles, int ksplit,
                                                       float* __restrict__ P, int n_tiles, int total)
{
    __shared__ __attribute__((aligned(16))) float sA[2][MID_BM * MID_STRIDE];
    __shared__ __attribute__((aligned(16))) float sB[2][MID_BN * MID_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, l31 = lane & 31, half = lane >> 5;
    const int t_row = tid >> 3, t_c4 = tid & 7;  // staging: 64 x 32 floats = 512 float4 per operand tile, two per thread
    const int k_len = K / ksplit, nk = k_len / MID_BK;
    const int fa = (wr * 32 + l31) * MID_STRIDE + half * 4, fb = (wc * 32 + l31) * MID_STRIDE + half * 4;

    // The `total` (tile, K-slice) pairs are walked by the gridDim.x workgroups the chip holds at once: workgroup w takes
    // pairs w, w + gridDim.x, ... (a finished workgroup's successor would otherwise wait for a dispatch: 10 % at 4 096
    // rows), and requests the first operand tiles of its next pair before it stores the results of the current one.
    // Order of the pairs, XCD-aware (workgroups are dealt round-robin over the 8 XCDs; gridDim.x is a multiple of 8 whenever
    // a workgroup takes more than one pair): every XCD gets one contiguous run, COLUMN tiles fastest -- an XCD then works on
    // a block of rows against the whole weight matrix, and both (a few hundred rows of A, all of W: 1-3 MB) stay in its
    // 4 MB L2.
    // (32-bit unsigned arithmetic: the 64-bit divisions this started with were ~1 us of every tile's fixed cost)
    const unsigned q8 = (unsigned)total >> 3, r8 = (unsigned)total & 7u;
    struct Pair {
        int m0, n0, ks;
    };
    auto pair_of = [&](unsigned w) {
        const unsigned xcd = w & 7u, slot = w >> 3;
        const unsigned bid0 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
        const unsigned bid = bid0 / (unsigned)ksplit;
        const unsigned mt = bid / (unsigned)n_tiles;
        return Pair{(int)mt * MID_BM, (int)(bid - mt * (unsigned)n_tiles) * MID_BN, (int)(bid0 - bid * (unsigned)ksplit)};
    };
    const float *a_ptr[2], *b_ptr[2];
    f32x4 ga[2], gb[2];
    auto point_at = [&](const Pair& p) {  // (rows past the end repeat the last one)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a_ptr[i] = A + (int64_t)min(p.m0 + t_row + 32 * i, M - 1) * lda + t_c4 * 4 + p.ks * k_len;
            b_ptr[i] = W + (int64_t)min(p.n0 + t_row + 32 * i, N - 1) * K + t_c4 * 4 + p.ks * k_len;
        }
    };
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ga[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + k0);
            gb[i] = *reinterpret_cast<const f32x4*>(b_ptr[i] + k0);
        }
    };
    auto store = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4*>(&sA[stage][(t_row + 32 * i) * MID_STRIDE + t_c4 * 4]) = ga[i];
            *reinterpret_cast<f32x4*>(&sB[stage][(t_row + 32 * i) * MID_STRIDE + t_c4 * 4]) = gb[i];
        }
    };

    Pair cur_pair = pair_of(blockIdx.x);
    point_at(cur_pair);
    load(0);
    for (unsigned w = blockIdx.x; w < (unsigned)total; w += gridDim.x) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        // The epilogue's bias value is requested HERE, in front of the K-loop: requested in the epilogue it sits behind the next
        // pair's operand tiles (already in flight by then) in the in-order vector-memory counter, and the first store of every
        // tile waited for that whole prefetch.
        float bv = 0.0f;
        if (!PARTIAL && bias != nullptr) {
            const int bcol = cur_pair.n0 + wc * 32 + l31;
            bv = bias[bcol < N ? bcol : N - 1];
        }
        store(0);
        __syncthreads();
        // (a second K-step of global loads in flight measured no faster at 4 096 rows and 7 % slower at 1 024)
        // Within a K-step the wave pipelines itself: the fragments of group kk + 1 are requested from LDS before the four MFMAs of
        // group kk, the staging stores of the next step sit between the third and the fourth group.  (Left to the compiler the
        // step was reads -> wait -> 8 MFMAs -> reads -> wait -> 8 MFMAs -> stores -> barrier, and knock-outs showed that LDS / barrier
        // time ADDED to the matrix time -- 17.7 us without MFMAs + 23 us of MFMAs = the 43 us of the QKV launch at 4 096 rows: the
        // other resident workgroups' waves do not fill those gaps.)
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            const bool more = kt + 1 < nk;
            if (more && (DIAG == 0 || DIAG == 2)) load((kt + 1) * MID_BK);
            f32x4 a[2], b[2];
            if (DIAG < 5 || kt == 0) {
                a[0] = *reinterpret_cast<const f32x4*>(&sA[cur][fa]);
                b[0] = *reinterpret_cast<const f32x4*>(&sB[cur][fb]);
            }
#pragma unroll
            for (int kk = 0; kk < MID_BK / 8; ++kk) {
                if (kk + 1 < MID_BK / 8 && (DIAG < 5 || kt == 0)) {
                    a[(kk + 1) & 1] = *reinterpret_cast<const f32x4*>(&sA[cur][fa + (kk + 1) * 8]);
                    b[(kk + 1) & 1] = *reinterpret_cast<const f32x4*>(&sB[cur][fb + (kk + 1) * 8]);
                }
                if (kk == MID_BK / 8 - 1 && more && DIAG != 4 && DIAG != 6) store(cur ^ 1);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (DIAG == 2) acc[c] += a[kk & 1][c] * b[kk & 1][c];
                    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk & 1][c], b[kk & 1][c], acc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);  // (keeps the groups in this order)
            }
            if (DIAG != 3 && DIAG != 6) __syncthreads();
        }
        const Pair done = cur_pair;
        if (w + gridDim.x < (unsigned)total) {
            cur_pair = pair_of(w + gridDim.x);
            point_at(cur_pair);
            load(0);
        }
        const int col = done.n0 + wc * 32 + l31;
        if (col < N) {
            if (PARTIAL) {
                float* out = P + (int64_t)done.ks * M * N;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = done.m0 + wr * 32 + acc_row(r, half);
                    if (row < M) out[(int64_t)row * N + col] = acc[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = done.m0 + wr * 32 + acc_row(r, half);
                    if (row < M) {
                        float v = acc[r] + bv;
                        if (EPI == EPI_BIAS_RESIDUAL) v += R[(int64_t)row * ldr + col];
                        else if (EPI == EPI_BIAS_MUL_SILU) v *= silu_ref(R[(int64_t)row * ldr + col]);
                        else v = epilogue<EPI>(v);
                        Y[(int64_t)row * ldy + col] = v;
                    }
                }
            }
        }
    }
}

// Y = epilogue(sum of the K-slice slabs, in slice order): four columns per thread.
template <int EPI>
__global__ __launch_bounds__(256) void mid_reduce_kernel(const float* __restrict__ P, int ksplit, const float* __restrict__ bias,
                                                         const float* R, int64_t ldr, float* Y, int64_t ldy, int M, int N)
{
    const int n4 = N >> 2;
    const int64_t total = (int64_t)M * n4, slab = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / n4;
        const int col = (int)(i - row * n4) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(P + row * N + col);
        for (int s = 1; s < ksplit; ++s) v += *reinterpret_cast<const f32x4*>(P + s * slab + row * N + col);
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + col);
        if (EPI == EPI_BIAS_RESIDUAL) {
            v += *reinterpret_cast<const f32x4*>(R + row * ldr + col);
        } else if (EPI == EPI_BIAS_MUL_SILU) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(R + row * ldr + col);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] *= silu_ref(g[c]);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = epilogue<EPI>(v[c]);
        }
        *reinterpret_cast<f32x4*>(Y + row * ldy + col) = v;
    }
}

// Y = LayerNorm(sum of the slabs + bias + R) * gamma + beta, one wave per row (N <= 256 * NCH): two-pass statistics as
// normalization/layer_norm.rs (mean, then the mean of squared deviations), R == Y allowed.
template <int NCH>
__global__ __launch_bounds__(256) void mid_reduce_ln_kernel(const float* __restrict__ P, int ksplit, const float* __restrict__ bias,
                                                            const float* R, int64_t ldr, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, float* Y, int64_t ldy,
                                                            int M, int N)
{
    const int lane = threadIdx.x & 63;
    // (four rows per workgroup, the XCDs' runs of rows as the projections' tiles deal them: the slabs and the residual this
    // workgroup reads were written, and the rows it writes will be read, by tiles of its own XCD)
    const int64_t row = (int64_t)xcd_contiguous(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int64_t slab = (int64_t)M * N;
    f32x4 v[NCH];
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (col < N) {
            v[c] = *reinterpret_cast<const f32x4*>(P + row * N + col);
            for (int k = 1; k < ksplit; ++k) v[c] += *reinterpret_cast<const f32x4*>(P + k * slab + row * N + col);
            if (bias) v[c] += *reinterpret_cast<const f32x4*>(bias + col);
            v[c] += *reinterpret_cast<const f32x4*>(R + row * ldr + col);
            s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
        }
    }
    const float mean = wave_sum(s) / (float)N;
    float q = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c * 256 + lane * 4 < N) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[c][e] - mean;
                q = fmaf(d, d, q);
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)N + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < N) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + col), b = *reinterpret_cast<const f32x4*>(beta + col);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[c][e] - mean) * rstd * g[e] + b[e];
            *reinterpret_cast<f32x4*>(Y + row * ldy + col) = o;
        }
    }
}

constexpr int64_t kFewRowsMax = 256, kMidMaxRows = 8192;
// ... for models up to 512 wide; wider ones have more column tiles per row group and reach two workgroups per CU sooner: 128 rows
// for <= 1 024 (a 768-wide model: 128 rows 0.529 against 0.626 ms on the 64 x 64 tiles, 192 rows even, 256 rows 0.681 against
// 0.639; tools/few_rows_sweep_768.sh), 64 beyond.  The width is min(N, K): the same for all four projections of a layer (QKV,
// out-proj and FC1 have K = hidden, FC2 has N = hidden), so a call takes one route throughout.
// (8 192 rows: 1.94 -> 1.58 ms per call in the mode; 4 096 rows: 1.14 against 1.11 on the 64 x 64 f32 tiles; 2 048: 0.92 against
// 0.69 -- tools/split_min_rows_sweep.sh)
inline int64_t split_min_rows() { return tune::split_min_rows_override() > 0 ? tune::split_min_rows_override() : 6144; }

inline int64_t few_rows_max(int N, int K)
{
    if (tune::few_rows_max_override() > 0) return tune::few_rows_max_override();
    const int width = N < K ? N : K;
    return width <= 512 ? kFewRowsMax : (width <= 1024 ? 128 : 64);
}
constexpr int kMidResident = 256 * 4;  // workgroups of gemm_nt_f32_mid the chip holds at once (36 KiB of LDS each)

inline bool mid_shape_ok(int64_t M, int N, int K, int64_t lda, int64_t ldy, int64_t ldr, const float* A, const float* W, const float* Y,
                         const float* bias, const float* R, int64_t min_rows = -1)
{
    if (min_rows < 0) min_rows = few_rows_max(N, K) + 1;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return M >= min_rows && (M <= kMidMaxRows || (tune::flex_any_rows() && M <= 65536 * 8)) && N % 4 == 0 && K % MID_BK == 0 && lda % 4 == 0 && ldy % 4 == 0 && (!R || ldr % 4 == 0) &&
           al16(A) && al16(W) && al16(Y) && al16(bias) && al16(R) && !tune::no_mid_route();
}

// gemm_flex.hip takes the mid-size calls (EPI_GELU_LIBM, a tuning-build epilogue, stays on the 64 x 64 tiles)
inline bool flex_route_ok(int64_t M, int N, int K, int64_t lda, int64_t ldy, int64_t ldr, const float* A, const float* W, const float* Y,
                          const float* bias, const float* R)
{
    return !tune::no_flex_route() && gemm_flex_shape_ok(M, N, K, lda, ldy, ldr, A, W, Y, bias, R);
}

// Physical K slices of a call whose result is DEFINED as the in-order sum of `logical` slices: all of them as separate workgroups
// (slabs + a reduce launch) or none (the workgroup adds its slices itself), whichever the cost estimate says is faster; the
// slabs cost their bytes twice (written, then read by the reduce kernel).
inline int flex_physical_split(int M, int N, int K, int logical)
{
    if (logical <= 1) return 1;
    const double slab_cycles = 2.0 * (logical - 1) * (double)M * N * 4.0 / 2000.0 + 4000.0;
    return gemm_flex_cost(M, N, K, logical, logical) + slab_cycles < gemm_flex_cost(M, N, K, 1, logical) ? logical : 1;
}

template <int EPI>
hipError_t launch_mid(const float* A, int64_t lda, const float* W, const float* bias, const float* R, int64_t ldr, float* Y,
                      int64_t ldy, int M, int N, int K, hipStream_t stream, const GemmScratch& scratch)
{
    const int m_tiles = (M + MID_BM - 1) / MID_BM, n_tiles = (N + MID_BN - 1) / MID_BN;
    int ksplit = mid_ksplit(N, K);
    if (ksplit > 1 && (size_t)ksplit * M * N > scratch.floats) ksplit = 1;
    const int total = m_tiles * n_tiles * ksplit;
    // the opt-in mode: the same tiles and slices on the bf16 matrix cores (gemm_split.hip) -- 32 x 128 tokens 1.12 -> 0.94 ms per
    // call, 8 x 128: 0.57 -> 0.49 (512 rows: 0.43 against 0.41 on the f32 tiles; the whole route moves, so that within the mode a
    // row's result still does not depend on how many rows share its call)
    if (get_f32_on_bf16() && !tune::mid_split_off()) {
        if (ksplit == 1) return launch_gemm_mid_split(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, 1, nullptr, (GemmEpilogue)EPI, stream);
        const hipError_t e = launch_gemm_mid_split(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, ksplit, scratch.p, (GemmEpilogue)EPI, stream);
        if (e != hipSuccess) return e;
        const int64_t quads = (int64_t)M * (N / 4);
        hipLaunchKernelGGL(mid_reduce_kernel<EPI>, dim3((unsigned)std::min<int64_t>(2048, (quads + 255) / 256)), dim3(256), 0, stream, scratch.p,
                           ksplit, bias, R, ldr, Y, ldy, M, N);
        return hipGetLastError();
    }
    // one workgroup per CU on a tile chosen for this call (gemm_flex.hip); K slices only while the call is too small to fill the
    // chip without them -- a workgroup that owns all the slices adds them in the reduce kernel's order, so the result is the same
    if (EPI != EPI_GELU_LIBM && flex_route_ok(M, N, K, lda, ldy, ldr, A, W, Y, bias, R)) {
        const int phys = flex_physical_split(M, N, K, ksplit);
        if (phys == 1) return launch_gemm_flex(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, (GemmEpilogue)EPI, 1, ksplit, nullptr, stream);
        const hipError_t e = launch_gemm_flex(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, (GemmEpilogue)EPI, phys, ksplit, scratch.p, stream);
        if (e != hipSuccess) return e;
        const int64_t quads = (int64_t)M * (N / 4);
        hipLaunchKernelGGL(mid_reduce_kernel<EPI>, dim3((unsigned)std::min<int64_t>(2048, (quads + 255) / 256)), dim3(256), 0, stream, scratch.p,
                           phys, bias, R, ldr, Y, ldy, M, N);
        return hipGetLastError();
    }
    const dim3 grid((unsigned)std::min(total, tune::mid_one_workgroup_per_tile() ? total : (tune::mid_grid_override() > 0 ? tune::mid_grid_override() : kMidResident)));
    if (ksplit == 1) {
#ifdef KJARNI_TUNING
        if (tune::mid_knockout() == 1) {
            hipLaunchKernelGGL((gemm_nt_f32_mid<EPI, false, 1>), grid, dim3(256), 0, stream, A, lda, W, bias, R, ldr, Y, ldy, M, N, K, m_tiles, 1,
                               nullptr, n_tiles, total);
            return hipGetLastError();
        }
#define KJ_MID_DIAG(D_)                                                                                                            \
    if (tune::mid_knockout() == D_) {                                                                                               \
        hipLaunchKernelGGL((gemm_nt_f32_mid<EPI, false, D_>), grid, dim3(256), 0, stream, A, lda, W, bias, R, ldr, Y, ldy, M, N, K, m_tiles, \
                           1, nullptr, n_tiles, total);                                                                             \
        return hipGetLastError();                                                                                                   \
    }
        KJ_MID_DIAG(2) KJ_MID_DIAG(3) KJ_MID_DIAG(4) KJ_MID_DIAG(5) KJ_MID_DIAG(6)
#undef KJ_MID_DIAG
#endif
        hipLaunchKernelGGL((gemm_nt_f32_mid<EPI, false>), grid, dim3(256), 0, stream, A, lda, W, bias, R, ldr, Y, ldy, M, N, K, m_tiles, 1,
                           nullptr, n_tiles, total);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((gemm_nt_f32_mid<EPI, true>), grid, dim3(256), 0, stream, A, lda, W, bias, R, ldr, Y, ldy, M, N, K, m_tiles, ksplit,
                       scratch.p, n_tiles, total);
    const int64_t quads = (int64_t)M * (N / 4);
    hipLaunchKernelGGL(mid_reduce_kernel<EPI>, dim3((unsigned)std::min<int64_t>(2048, (quads + 255) / 256)), dim3(256), 0, stream, scratch.p,
                       ksplit, bias, R, ldr, Y, ldy, M, N);
    return hipGetLastError();
}

// Shapes the few-rows kernel takes: M <= 128 (64 when the caller lends the mid-size route its scratch), whole 32-column tiles, K divisible by 8 x 16 waves, 16-byte rows.
inline bool aligned6(int64_t M, int N, int K, int64_t lda, int64_t ldy, int64_t ldr, const float* A, const float* W,
                     const float* Y, const float* bias, const float* R)
{
    (void)ldy;
    (void)ldr;
    (void)Y;
    (void)bias;
    (void)R;
    return M <= few_rows_max(N, K) && N % 32 == 0 && K % 128 == 0 && lda % 4 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(W) & 15) == 0 && (int64_t)64 * lda * 4 < ((int64_t)1 << 31) &&
           (int64_t)32 * K * 4 < ((int64_t)1 << 31);
}

template <int EPI>
hipError_t launch_epi(const float* A, int64_t lda, const float* W, const float* bias, const float* R,
                      int64_t ldr, float* Y, int64_t ldy, int64_t M, int N, int K,
                      hipStream_t stream, const GemmScratch& scratch)
{
    // (a tile's rows are addressed through 32-bit buffer offsets: 128 rows of A must span less than 2 GiB)
    const bool aligned = (N % BN == 0) && (K % 32 == 0) && (lda % 4 == 0) && (ldy % 4 == 0) &&
                         ((int64_t)BM * lda * 4 < (int64_t)1 << 31) && ((int64_t)BN * K * 4 < (int64_t)1 << 31) &&
                         ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(W) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(Y) & 15) == 0) &&
                         (bias == nullptr || (reinterpret_cast<uintptr_t>(bias) & 15) == 0) &&
                         (R == nullptr || ((ldr % 4 == 0) && (reinterpret_cast<uintptr_t>(R) & 15) == 0));
    // few rows: K split over the waves of a workgroup instead of a serial K-loop in N / 128 workgroups
    const bool mid = scratch.p && mid_shape_ok(M, N, K, lda, ldy, ldr, A, W, Y, bias, R);
    // (not when the 128 x 128 tiles fill the chip anyway -- 64 queries against a 10^7-row corpus: 11.5 ms tiled, 13.7 ms here)
    const bool tiles_fill_chip = aligned && ((M + BM - 1) / BM) * (int64_t)(N / BN) >= 768;
    if (aligned6(M, N, K, lda, ldy, ldr, A, W, Y, bias, R) && !tune::no_few_rows_route() && !mid && !tiles_fill_chip)
        return launch_skinny<EPI>(A, lda, W, bias, R, ldr, Y, ldy, (int)M, N, K, stream);
    // f32-on-bf16 mode: from split_min_rows() the split kernel's 128 x 128 tiles also replace the 64 x 64-tile route
    if (mid && aligned && get_f32_on_bf16() && K % 64 == 0 && M >= split_min_rows())
        return launch_gemm_split(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, (GemmEpilogue)EPI, stream);
    // a few hundred to a few thousand rows (and a caller that lends a scratch slab): quarter-size tiles, K slices
    if (mid)
        return launch_mid<EPI>(A, lda, W, bias, R, ldr, Y, ldy, (int)M, N, K, stream, scratch);
    if (aligned) {
#ifdef KJARNI_TUNING
        if (tune::tiles_without_epilogue()) return launch_tiled<EPI, 32, 1>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
        if (tune::tiles_without_stores()) return launch_tiled<EPI, 32, 2>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
        if (tune::tiles_with_cycle_counts()) return launch_tiled<EPI, 32, 3>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
        if (tune::tiles_without_transpose()) return launch_tiled<EPI, 32, 4>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
        if (tune::tiles_stored_in_place()) return launch_tiled<EPI, 32, 5>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
#endif
        if (get_f32_on_bf16() && K % 64 == 0) return launch_gemm_split(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, (GemmEpilogue)EPI, stream);
        return launch_tiled<EPI, 32, 0>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    }
    dim3 grid((unsigned)((N + 31) / 32), (unsigned)((M + 31) / 32));
    hipLaunchKernelGGL(gemm_nt_f32_generic<EPI>, grid, dim3(256), 0, stream, A, lda, W, bias, R, ldr, Y, ldy, M, N,
                       K);
    return hipGetLastError();
}

}  // namespace

// Rows from which the residual + LayerNorm GEMMs take the K-sliced route: with a long K (>= 1024: the FC2 shape) from the
// first row -- unsliced, the few-rows kernel has N / 32 = 12 workgroups there, and a workgroup's MFMAs run on one CU however
// its K is dealt over the waves (13 us for one sentence, + 5 us of LayerNorm launch; four slices of the same kernel on
// 48 CUs + the LayerNorm reduce: 10.5 us) -- otherwise from kFewRowsMax + 1.
inline int64_t mid_ln_min_rows(int N, int K) { return mid_ksplit(N, K) > 1 ? 1 : few_rows_max(N, K) + 1; }

int64_t gemm_few_rows_max(int hidden) { return few_rows_max(hidden, hidden); }

// The largest call whose projections all take the mid-size route (where a row's result does not depend on the rows beside it): 8 192
// rows; in the f32-on-bf16 mode the 128 x 128 split tiles take over from split_min_rows().
int64_t gemm_mid_route_max_rows() { return get_f32_on_bf16() ? std::min<int64_t>(kMidMaxRows, split_min_rows() - 1) : kMidMaxRows; }

bool gemm_mid_layernorm_supported(int64_t M, int N, int K)
{
    return M >= mid_ln_min_rows(N, K) && M <= kMidMaxRows && N <= 1024 && N % 4 == 0 && K % MID_BK == 0 && !tune::no_mid_route();
}

size_t gemm_scratch_floats(int64_t max_rows, int max_narrow_n)
{
    // 4 K-slice slabs of the narrow (N <= 1024) outputs, for calls of up to kMidMaxRows rows
    return (size_t)4 * (size_t)std::min<int64_t>(max_rows, kMidMaxRows) * (size_t)std::min(max_narrow_n, 1024);
}

bool gemm_residual_layernorm_supported(int N, int K)
{
    if (tune::no_fused_layernorm()) return false;
    return (N == 384 || N == 256) && K >= 16 && K % 16 == 0;  // (rows are addressed through 32-bit buffer offsets)
}

hipError_t launch_gemm_residual_layernorm(const float* A, int64_t lda, const float* W, const float* bias,
                                          const float* R, int64_t ldr, const float* gamma, const float* beta, float eps,
                                          float* Y, int64_t ldy, int64_t M, int N, int K, hipStream_t stream, GemmScratch scratch)
{
    if (M <= 0) return hipSuccess;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    // f32-on-bf16 mode, large calls with a long K (FC2): the split kernel with the residual epilogue + the LayerNorm kernel (the
    // fused LayerNorm tile has no split form yet; at K = 384 -- out-proj -- the fused f32 tile is the faster of the two: 0.62
    // against 0.60 + 0.17 ms)
    if (R && gamma && beta && get_f32_on_bf16() && M >= split_min_rows() && N % BN == 0 && K % 64 == 0 && K >= 1024 && gemm_residual_layernorm_supported(N, K) && ldy == N && lda % 4 == 0 && ldr % 4 == 0 && al16(A) && al16(W) && al16(R) && al16(Y) && al16(bias) &&
        (int64_t)BM * lda * 4 < ((int64_t)1 << 31) && (int64_t)BN * K * 4 < ((int64_t)1 << 31)) {
        const hipError_t e = launch_gemm_split(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, EPI_BIAS_RESIDUAL, stream);
        if (e != hipSuccess) return e;
        return launch_layernorm(Y, gamma, beta, eps, M, N, Y, stream);
    }
    if (scratch.p && R && gamma && beta && N <= 1024 && al16(gamma) && al16(beta) &&
        mid_shape_ok(M, N, K, lda, ldy, ldr, A, W, Y, bias, R, mid_ln_min_rows(N, K))) {
        const int ksplit = mid_ksplit(N, K);
        if ((size_t)ksplit * M * N <= scratch.floats) {
            const int m_tiles = ((int)M + MID_BM - 1) / MID_BM, n_tiles = (N + MID_BN - 1) / MID_BN;
            const int total = m_tiles * n_tiles * ksplit;
            // up to few_rows_max() rows (256 for models up to 512 wide): the slices' partial tiles from the few-rows kernel (K over the sixteen waves of a workgroup: a
            // 3-step chain per wave instead of the tile kernel's 12-step one -- FC2 + LayerNorm of one sentence 15.5 -> 10.5 us)
            const int k_len = K / ksplit;
            int slabs = ksplit;  // slabs the reduce kernel adds
            if (M <= few_rows_max(N, K) && ksplit > 1 && N % 32 == 0 && k_len % 128 == 0 && (int64_t)64 * lda * 4 < ((int64_t)1 << 31) &&
                (int64_t)32 * K * 4 < ((int64_t)1 << 31) && !tune::no_few_rows_route() && !tune::no_few_rows_k_slices()) {
                constexpr int LDS = 16 * 32 * 32 * 4;
                const int z1 = ((int)M + 31) / 32;
                if ((int64_t)z1 * (N / 32) * ksplit <= 512 || M <= 32)
                    hipLaunchKernelGGL((gemm_nt_f32_skinny<EPI_BIAS, 1, true>), dim3((unsigned)(N / 32), (unsigned)ksplit, (unsigned)z1),
                                       dim3(1024), LDS, stream, A, lda, W, nullptr, nullptr, 0, scratch.p, N, (int)M, N, k_len);
                else
                    hipLaunchKernelGGL((gemm_nt_f32_skinny<EPI_BIAS, 2, true>),
                                       dim3((unsigned)(N / 32), (unsigned)ksplit, (unsigned)(((int)M + 63) / 64)), dim3(1024), LDS, stream, A, lda,
                                       W, nullptr, nullptr, 0, scratch.p, N, (int)M, N, k_len);
            } else if (get_f32_on_bf16() && !tune::mid_split_off()) {
                const hipError_t e = launch_gemm_mid_split(A, lda, W, bias, R, ldr, Y, ldy, (int)M, N, K, ksplit, scratch.p, EPI_BIAS, stream);
                if (e != hipSuccess) return e;
            } else if (flex_route_ok(M, N, K, lda, ldy, ldr, A, W, Y, bias, R)) {
                // (gemm_flex.hip: slabs of the physical slices; a workgroup that owns all the slices has added them in slice order)
                slabs = flex_physical_split((int)M, N, K, ksplit);
                const hipError_t e = launch_gemm_flex(A, lda, W, nullptr, nullptr, 0, Y, ldy, (int)M, N, K, EPI_BIAS, slabs, ksplit, scratch.p, stream);
                if (e != hipSuccess) return e;
            } else
            hipLaunchKernelGGL((gemm_nt_f32_mid<EPI_BIAS, true>), dim3((unsigned)std::min(total, tune::mid_one_workgroup_per_tile() ? total : kMidResident)),
                               dim3(256), 0, stream, A, lda, W, bias, R, ldr, Y, ldy, (int)M, N, K, m_tiles, ksplit, scratch.p, n_tiles, total);
            const dim3 rgrid((unsigned)((M + 3) / 4));
#define KJ_MID_LN(NCH)                                                                                                              \
    hipLaunchKernelGGL(mid_reduce_ln_kernel<NCH>, rgrid, dim3(256), 0, stream, scratch.p, slabs, bias, R, ldr, gamma, beta, eps, Y, \
                       ldy, (int)M, N)
            if (N <= 256) KJ_MID_LN(1);
            else if (N <= 512) KJ_MID_LN(2);
            else if (N <= 768) KJ_MID_LN(3);
            else KJ_MID_LN(4);
#undef KJ_MID_LN
            return hipGetLastError();
        }
    }
    if (!R || !gamma || !beta) return hipErrorInvalidValue;
    // a handful of rows and a short K: the few-rows projection + a LayerNorm launch (the 64-row tiles would be one workgroup
    // walking all of K alone: 38 us for out-proj at 28 rows against 11)
    const bool few_rows_pair = M <= few_rows_max(N, K) && ldy == N && aligned6(M, N, K, lda, ldy, ldr, A, W, Y, bias, R) && !tune::no_few_rows_route();
    if (few_rows_pair || !gemm_residual_layernorm_supported(N, K) || lda % 4 || ldr % 4 || ldy % 4 ||
        (int64_t)64 * lda * 4 >= (int64_t)1 << 31 || (int64_t)N * K * 4 >= (int64_t)1 << 31 ||
        !al16(A) || !al16(W) || !al16(bias) || !al16(R) || !al16(gamma) || !al16(beta) || !al16(Y)) {
        // Neither fused form takes this call (an output pointer that is 4- but not 16-byte aligned, a row width the
        // tile kernel does not cover with a call outside the K-sliced range, ...): the same result in two launches.
        if (ldy != N) return hipErrorInvalidValue;  // (launch_layernorm works on contiguous rows)
        const hipError_t e = launch_gemm(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, EPI_BIAS_RESIDUAL, stream, scratch);
        if (e != hipSuccess) return e;
        return launch_layernorm(Y, gamma, beta, eps, M, N, Y, stream);
    }
    if (N == 384) return launch_ln_tiled<3>(A, lda, W, bias, R, ldr, gamma, beta, eps, Y, ldy, M, K, stream);
    return launch_ln_tiled<2>(A, lda, W, bias, R, ldr, gamma, beta, eps, Y, ldy, M, K, stream);
}

hipError_t launch_gemm(const float* A, int64_t lda, const float* W, const float* bias, const float* R,
                       int64_t ldr, float* Y, int64_t ldy, int64_t M, int N, int K, GemmEpilogue epi,
                       hipStream_t stream, GemmScratch scratch)
{
    if (M <= 0 || N <= 0 || K <= 0) return hipSuccess;
    switch (epi) {
    case EPI_BIAS: return launch_epi<EPI_BIAS>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream, scratch);
    case EPI_BIAS_GELU:
#ifdef KJARNI_TUNING
        if (tune::gelu_through_libm()) return launch_epi<EPI_GELU_LIBM>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream, scratch);
#endif
        return launch_epi<EPI_BIAS_GELU>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream, scratch);
    case EPI_BIAS_GELU_NEW: return launch_epi<EPI_BIAS_GELU_NEW>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream, scratch);
    case EPI_BIAS_RELU: return launch_epi<EPI_BIAS_RELU>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream, scratch);
    case EPI_BIAS_TANH: return launch_epi<EPI_BIAS_TANH>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream, scratch);
    case EPI_BIAS_RESIDUAL: return launch_epi<EPI_BIAS_RESIDUAL>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream, scratch);
    case EPI_BIAS_MUL_SILU: return launch_epi<EPI_BIAS_MUL_SILU>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream, scratch);
    }
    return hipErrorInvalidValue;
}

}  // namespace kjarni
