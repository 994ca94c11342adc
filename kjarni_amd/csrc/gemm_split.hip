// f32 projections on the bf16 matrix cores: the opt-in mode of the encoder (both operands f32: six products per f32 product) and
// the decoder's prompt projections over bf16 weights (always on: three products, nothing dropped).  Its own translation unit
// because it is compiled with -fno-slp-vectorize: the SLP vectoriser packs the split's f32 subtractions into v_pk_add_f32, which
// costs ~26 issue cycles beside MFMAs against 4 + 4 for two plain subtractions (MI355X_MICROARCH.md, "packed f32 VALU ... an
// anti-lever beside MFMAs").
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "device_utils.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace kjarni {

namespace {

constexpr int BM = 128, BN = 128;  // block tile (as gemm.hip's large-batch tiles)
constexpr int EPI_STRIDE = 68;     // floats per staged output row (64 + 4), as gemm.hip

// ---- fp32 products on the bf16 matrix cores (opt-in: kjarni_hip_set_f32_on_bf16 / KJARNI_HIP_F32_ON_BF16=1) -----------------
// gfx950's bf16 MFMA rate is 16x its f32 MFMA rate (2.5 PFLOP/s against 157 TFLOP/s), and an f32 number is EXACTLY the sum of
// three bf16 numbers (8 + 8 + 8 significand bits; bf16 has f32's exponent range): x = x1 + x2 + x3 with x1 = bf16(x),
// x2 = bf16(x - x1), x3 = bf16(x - x1 - x2), every subtraction exact in f32.  A product a b is then the nine cross products
// a_i b_j, each exact in the MFMA's f32 accumulation; the three with i + j >= 5 are below 2^-24 of the product -- f32's own
// rounding -- and are dropped: SIX bf16 MFMAs (32 x 32 x 16, 32 cycles each) do the work of eight f32 ones (32 x 32 x 2, 64 cycles
// each), 2.67x the f32 matrix peak at the accuracy of f32 arithmetic (the dropped terms and the f32 accumulation are the only
// roundings: |error| <= ~3 x 2^-24 per product, against 2^-24 for a chain of f32 FMAs; measured against the oracle in
// tests/test_gpu_split.py).  The operands stay f32 in memory: a tile's rows are split on their way into LDS (4.5 vector
// instructions per element, which issue in the gaps of the bf16 MFMAs -- unlike beside the f32 MFMAs), three bf16 planes per
// operand (layout below).  Not the default: the reference computes in f32, and whether an
// f32 result assembled from bf16 pieces counts as "the reference's precision" is the integrator's call; non-finite inputs
// (inf - inf in the split) come out as NaN where the f32 path keeps an infinity.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int SP_BK = 16;                              // K per step: one 32 x 32 x 16 MFMA block
// LDS rows are 16 bf16 = 32 bytes with NO padding; the two 16-byte halves of a row swap places in rows 8..15 of every 16
// (half ^ ((row >> 3) & 1)).  A fragment read (16 lanes x 16 bytes: rows r .. r + 15, one logical half) then covers all 64
// banks once, and so does a staging store (32 lanes x 8 bytes: 8 whole rows) -- with padded rows one of the two always
// collides (48-byte rows: a third of the LDS cycles were bank conflicts).
constexpr int SP_ST = 16;                              // bf16 per LDS row
constexpr int SP_PLANE = 128 * SP_ST;                  // bf16 per piece plane of a 128-row tile
constexpr int SP_STAGE = 2 * 3 * SP_PLANE;             // A and W, three planes each (bf16 elements)
constexpr int SP_LDS_BYTES = 2 * SP_STAGE * 2;         // two stages: 49 152 bytes
static_assert(4 * 32 * EPI_STRIDE * 4 <= SP_LDS_BYTES, "four wave-private epilogue regions must fit the operand planes");

__device__ __forceinline__ uint32_t sp_pack(float a, float b)
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2v{a, b}, bf16x2));  // v_cvt_pk_bf16_f32, round to nearest even
}
__device__ __forceinline__ void sp_split(const f32x4 x, u32x2& p1, u32x2& p2, u32x2& p3)
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float x0 = x[2 * i], x1 = x[2 * i + 1];
        const uint32_t a = sp_pack(x0, x1);
        const float r0 = x0 - __builtin_bit_cast(float, a << 16), r1 = x1 - __builtin_bit_cast(float, a & 0xffff0000u);
        const uint32_t b = sp_pack(r0, r1);
        const float s0 = r0 - __builtin_bit_cast(float, b << 16), s1 = r1 - __builtin_bit_cast(float, b & 0xffff0000u);
        p1[i] = a;
        p2[i] = b;
        p3[i] = sp_pack(s0, s1);
    }
}

// Epilogue through LDS, as gemm_nt_f32_mfma's: the accumulators of a 32-row tile go to a wave-private [32][68] region (the
// operand planes are free after the barrier), then 16 lanes own a row: 16-byte row-contiguous loads of the residual and
// stores (64 four-byte stores per lane straight from the accumulators made the store issue, not the MFMAs, the tile's time).
template <int EPI, bool STREAM_OUT = false>
__device__ __forceinline__ void sp_epilogue(f32x16 (&acc)[2][2], float* lds, const float* __restrict__ bias, const float* R, int64_t ldr,
                                            float* Y, int64_t ldy, int64_t M, int64_t m0, int n0, int wid, int wr, int wc, int lane, int l31,
                                            int half)
{
    __syncthreads();
    float* sw = lds + wid * (32 * EPI_STRIDE);
    const int e_row = lane >> 4, e_c4 = lane & 15;
    const int n = n0 + wc * 64 + e_c4 * 4;
    const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) sw[acc_row(r, half) * EPI_STRIDE + j * 32 + l31] = acc[i][j][r];
        const int64_t m_base = m0 + wr * 64 + i * 32 + e_row;
        f32x4 res[8];
        if (EPI == EPI_BIAS_RESIDUAL || EPI == EPI_BIAS_MUL_SILU) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                int64_t m = m_base + it * 4;
                m = m < M ? m : M - 1;
                res[it] = *reinterpret_cast<const f32x4*>(R + m * ldr + n);
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
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
            if (m < M) {
                // (STREAM_OUT: outputs of 256 MB and more, as gemm_nt_f32_mfma -- plain stores evict the operand panels from the L2)
                f32x4* dst = reinterpret_cast<f32x4*>(Y + m * ldy + n);
                if (STREAM_OUT) __builtin_nontemporal_store(v, dst);
                else *dst = v;
            }
        }
    }
}

// One K-step = 16 of K: 24 MFMAs per wave (2 x 2 tiles x 6 products).  Software pipeline inside the wave: while step k is
// multiplied from LDS stage k & 1, the raw f32 rows of step k + 1 (in registers since the previous step) are split and stored to
// the other stage, and the rows of step k + 2 are requested into the registers step k's rows left -- the split's vector
// instructions issue in the gaps of the bf16 MFMAs (an MFMA holds the vector issue port 8 of its 32 cycles).  One barrier per step.
template <int EPI, bool STREAM_OUT = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_f32_split(const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
                                                            const float* __restrict__ bias, const float* R, int64_t ldr, float* Y,
                                                            int64_t ldy, int64_t M, int N, int K, int n_tiles, int64_t total_tiles)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t sp_smem[];  // [2 stages][A | W][3][128][SP_ST]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, l31 = lane & 31, half = lane >> 5;
    // XCD-aware tile order, as gemm_nt_f32_mfma (one workgroup per tile)
    const unsigned nwg = (unsigned)total_tiles, q8 = nwg >> 3, r8 = nwg & 7u;
    const unsigned wg = blockIdx.x, xcd = wg & 7u, slot = wg >> 3;
    const unsigned bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const unsigned m_tile = bid / (unsigned)n_tiles;
    const int64_t m0 = (int64_t)m_tile * BM;
    const int n0 = (int)(bid - m_tile * (unsigned)n_tiles) * BN;

    // staging: 128 rows x 16 floats per operand = 512 sixteen-byte pieces, two per thread; buffer descriptors (rows past M: zeros)
    const int ld_row = tid >> 2, ld_c4 = tid & 3;
    const int64_t rows_a = (M - m0 < BM) ? (M - m0) : BM;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + m0 * lda), 0,
                                                                           (int)(((rows_a - 1) * lda + K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W + (int64_t)n0 * K), 0,
                                                                           (int)((int64_t)BN * K * 4), 0x00020000);
    uint32_t offA[2], offW[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t r = ld_row + 64 * i;
        offA[i] = (uint32_t)((r * lda + ld_c4 * 4) * 4);
        offW[i] = (uint32_t)((r * K + ld_c4 * 4) * 4);
    }
    // Raw rows are requested a GROUP (two K-steps = 32 floats = one 128-byte line per row) at a time, the two 64-byte halves of
    // every line by two back-to-back instructions; two register sets of a group each.  (Measured the same as one half-line per
    // row and step with the requests three steps ahead, 1.53 ms for the QKV launch either way -- like every other change to the
    // staging: under this load the chip is at its power limit (1.5 GHz), and only moving fewer bytes per product buys time.)
    f32x4 ga[2][2][2], gb[2][2][2];  // [group set][row pass][half line]
    const int st_off = ld_row * SP_ST + (((ld_c4 >> 1) ^ ((ld_row >> 3) & 1)) * 2 + (ld_c4 & 1)) * 4;  // (bf16 elements; halves swapped in rows 8..15)
    const int sw_half = half ^ ((l31 >> 3) & 1);
    const int fa = (wr * 64 + l31) * SP_ST + sw_half * 8, fb = 3 * SP_PLANE + (wc * 64 + l31) * SP_ST + sw_half * 8;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    auto load_group = [&](auto set, int k0) {   // k0 = the group's first k (a multiple of 32)
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                ga[S][i][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcA, offA[i], (k0 + 16 * h) * 4, 0));
                gb[S][i][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcW, offW[i], (k0 + 16 * h) * 4, 0));
            }
    };
    auto split_store = [&](auto set, auto hline, int stage) {
        constexpr int S = decltype(set)::value, HL = decltype(hline)::value;
        uint16_t* base = sp_smem + stage * SP_STAGE + st_off;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x2 p1, p2, p3;
            sp_split(ga[S][i][HL], p1, p2, p3);
            *reinterpret_cast<u32x2*>(base + 64 * i * SP_ST) = p1;
            *reinterpret_cast<u32x2*>(base + SP_PLANE + 64 * i * SP_ST) = p2;
            *reinterpret_cast<u32x2*>(base + 2 * SP_PLANE + 64 * i * SP_ST) = p3;
            sp_split(gb[S][i][HL], p1, p2, p3);
            *reinterpret_cast<u32x2*>(base + 3 * SP_PLANE + 64 * i * SP_ST) = p1;
            *reinterpret_cast<u32x2*>(base + 4 * SP_PLANE + 64 * i * SP_ST) = p2;
            *reinterpret_cast<u32x2*>(base + 5 * SP_PLANE + 64 * i * SP_ST) = p3;
        }
    };
    const int nk = K / SP_BK;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // step kt: MFMAs from stage kt & 1, while the rows of step kt + 1 -- half (kt + 1) & 1 of group (kt + 1) / 2 -- are split into
    // stage (kt + 1) & 1; on odd steps the set whose second half was split during the previous step receives group kt / 2 + 2.
    auto step = [&](auto split_set, auto split_half, auto load, auto load_set, int kt) {
        const uint16_t* st = sp_smem + (kt & 1) * SP_STAGE;
        bf16x8 a[2][3], b[2][3];
        // (in the order the products below need them: the first four MFMAs wait for four reads, not twelve)
        constexpr int RA[3] = {2, 1, 0}, RB[3] = {0, 1, 2};
#pragma unroll
        for (int o = 0; o < 3; ++o)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t][RA[o]] = *reinterpret_cast<const bf16x8*>(st + RA[o] * SP_PLANE + fa + t * 32 * SP_ST);
                b[t][RB[o]] = *reinterpret_cast<const bf16x8*>(st + RB[o] * SP_PLANE + fb + t * 32 * SP_ST);
            }
        // (unconditional: the last steps request and split rows nobody multiplies -- past K a row's bytes are the next row's or,
        // past the descriptor, zeros -- so that the step stays one straight-line block the MFMAs can be interleaved with)
        if (decltype(load)::value) load_group(load_set, (kt / 2 + 2) * 32);
        split_store(split_set, split_half, (kt + 1) & 1);
        // the six products with i + j <= 4, the smallest first
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[q]], b[j][PB[q]], acc[i][j], 0, 0, 0);
        // spread the split (vector ALU), its LDS stores and the requests over the MFMAs: ~3 vector instructions per MFMA gap
#pragma unroll
        for (int g = 0; g < 24; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // VALU
            if (g % 2 == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
            if (g < 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // VMEM read
        }
        __syncthreads();
    };
    load_group(I0{}, 0);
    load_group(I1{}, 32);
    split_store(I0{}, I0{}, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 4) {   // (K % 64 == 0)
        step(I0{}, I1{}, I0{}, I0{}, kt);       // split group 2j half 1;   no request
        step(I1{}, I0{}, I1{}, I0{}, kt + 1);   // split group 2j+1 half 0; set 0 <- group 2j+2
        step(I1{}, I1{}, I0{}, I0{}, kt + 2);   // split group 2j+1 half 1
        step(I0{}, I0{}, I1{}, I1{}, kt + 3);   // split group 2j+2 half 0; set 1 <- group 2j+3
    }

    sp_epilogue<EPI, STREAM_OUT>(acc, reinterpret_cast<float*>(sp_smem), bias, R, ldr, Y, ldy, M, m0, n0, wid, wr, wc, lane, l31, half);
}

// ---- f32 activations x bf16 weights (the decoder's prompt projections; always on for bf16 checkpoints) ---------------------------
// The weights ARE bf16, so only the activations need pieces: a w = a1 w + a2 w + a3 w, every term exact in the MFMA's f32
// accumulation and none dropped -- the same products as widening w to f32 and multiplying in f32, in another order of summation.
// Three bf16 MFMAs (32 cycles) per 32 x 32 x 16 block instead of eight f32 ones (64 cycles), and no f32 copy of the weights.
// K-step = 32: A as three planes + W as one, 64-byte rows whose four 16-byte chunks are XOR-swizzled with (row >> 2) & 3
// (fragment reads of 16 rows x one chunk, and staging stores of 4 rows x 64 bytes, both cover all 64 banks once); two LDS stages
// of 32 KiB, two workgroups per CU; pipeline as gemm_nt_f32_split.
constexpr int BW_ST = 32;                       // bf16 per LDS row
constexpr int BW_PLANE = 128 * BW_ST;           // bf16 per plane of a 128-row tile
constexpr int BW_STAGE = 4 * BW_PLANE;          // A1 | A2 | A3 | W
constexpr int BW_LDS_BYTES = 2 * BW_STAGE * 2;  // 65 536
static_assert(4 * 32 * EPI_STRIDE * 4 <= BW_LDS_BYTES, "four wave-private epilogue regions must fit the operand planes");

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_f32_bf16w(const float* __restrict__ A, int64_t lda, const uint16_t* __restrict__ W,
                                                            const float* __restrict__ bias, const float* R, int64_t ldr, float* Y,
                                                            int64_t ldy, int64_t M, int N, int K, int n_tiles, int64_t total_tiles)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t sp_smem[];  // [2 stages][A1 | A2 | A3 | W][128][BW_ST]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, l31 = lane & 31, half = lane >> 5;
    const unsigned nwg = (unsigned)total_tiles, q8 = nwg >> 3, r8 = nwg & 7u;
    const unsigned wg = blockIdx.x, xcd = wg & 7u, slot = wg >> 3;
    const unsigned bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const unsigned m_tile = bid / (unsigned)n_tiles;
    const int64_t m0 = (int64_t)m_tile * BM;
    const int n0 = (int)(bid - m_tile * (unsigned)n_tiles) * BN;

    // staging per step: A 128 rows x 32 floats = 1 024 sixteen-byte pieces (four per thread: row = tid / 8 + 32 i, floats 4 (tid % 8) ..);
    // W 128 rows x 32 bf16 = 512 pieces (two per thread: row = tid / 4 + 64 i, bf16 8 (tid % 4) ..)
    const int a_row = tid >> 3, a_c4 = tid & 7, w_row = tid >> 2, w_c = tid & 3;
    const int64_t rows_a = (M - m0 < BM) ? (M - m0) : BM;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + m0 * lda), 0,
                                                                           (int)(((rows_a - 1) * lda + K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(W + (int64_t)n0 * K), 0,
                                                                           (int)((int64_t)BN * K * 2), 0x00020000);
    uint32_t offA[4], offW[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) offA[i] = (uint32_t)((((int64_t)a_row + 32 * i) * lda + a_c4 * 4) * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) offW[i] = (uint32_t)((((int64_t)w_row + 64 * i) * K + w_c * 8) * 2);
    f32x4 ga[2][4];
    u32x4 gw[2][2];
    // LDS offsets (bf16 elements); chunk c of a row sits at chunk c ^ ((row >> 2) & 3)
    int stA[4], stW[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = a_row + 32 * i;
        stA[i] = row * BW_ST + (((a_c4 >> 1) ^ ((row >> 2) & 3)) * 2 + (a_c4 & 1)) * 4;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = w_row + 64 * i;
        stW[i] = 3 * BW_PLANE + row * BW_ST + (w_c ^ ((row >> 2) & 3)) * 8;
    }
    const int fsw = (l31 >> 2) & 3;
    const int fa = (wr * 64 + l31) * BW_ST, fb = 3 * BW_PLANE + (wc * 64 + l31) * BW_ST;  // (+ 32 rows for the second tile: same swizzle)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    auto load_set = [&](auto set, int k0) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) ga[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcA, offA[i], k0 * 4, 0));
#pragma unroll
        for (int i = 0; i < 2; ++i) gw[S][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcW, offW[i], k0 * 2, 0));
    };
    auto split_store = [&](auto set, int stage) {
        constexpr int S = decltype(set)::value;
        uint16_t* base = sp_smem + stage * BW_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x2 p1, p2, p3;
            sp_split(ga[S][i], p1, p2, p3);
            *reinterpret_cast<u32x2*>(base + stA[i]) = p1;
            *reinterpret_cast<u32x2*>(base + BW_PLANE + stA[i]) = p2;
            *reinterpret_cast<u32x2*>(base + 2 * BW_PLANE + stA[i]) = p3;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(base + stW[i]) = gw[S][i];
    };
    const int nk = K / 32;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // step kt: MFMAs from stage kt & 1; set (kt + 1) & 1 -> split -> stage (kt + 1) & 1; set kt & 1 <- the rows of step kt + 2
    auto step = [&](auto cur_set, auto next_set, int kt) {
        const uint16_t* st = sp_smem + (kt & 1) * BW_STAGE;
        load_set(cur_set, (kt + 2) * 32);          // (unconditional, as gemm_nt_f32_split: the step stays one straight-line block)
        split_store(next_set, (kt + 1) & 1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int ch = ((kb * 2 + half) ^ fsw) * 8;
            bf16x8 a[2][3], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                b[t] = *reinterpret_cast<const bf16x8*>(st + fb + t * 32 * BW_ST + ch);
#pragma unroll
                for (int p = 2; p >= 0; --p) a[t][p] = *reinterpret_cast<const bf16x8*>(st + p * BW_PLANE + fa + t * 32 * BW_ST + ch);
            }
#pragma unroll
            for (int p = 2; p >= 0; --p)   // the smallest piece first
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][p], b[j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 24; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // VALU
            if (g % 2 == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
            if (g < 6) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // VMEM read
        }
        __syncthreads();
    };
    load_set(S0{}, 0);
    load_set(S1{}, 32);
    split_store(S0{}, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {   // (K % 64 == 0)
        step(S0{}, S1{}, kt);
        step(S1{}, S0{}, kt + 1);
    }
    sp_epilogue<EPI>(acc, reinterpret_cast<float*>(sp_smem), bias, R, ldr, Y, ldy, M, m0, n0, wid, wr, wc, lane, l31, half);
}

template <int EPI>
hipError_t launch_bf16w(const float* A, int64_t lda, const uint16_t* W, const float* bias, const float* R, int64_t ldr, float* Y,
                        int64_t ldy, int64_t M, int N, int K, hipStream_t stream)
{
    static_assert(BW_LDS_BYTES <= 64 * 1024, "more dynamic LDS than a launch gets without the opt-in attribute");
    const int n_tiles = N / BN;
    const int64_t total = ((M + BM - 1) / BM) * n_tiles;
    if (total > 0x7fffffff) return hipErrorInvalidValue;
    hipLaunchKernelGGL((gemm_nt_f32_bf16w<EPI>), dim3((unsigned)total), dim3(256), BW_LDS_BYTES, stream, A, lda, W, bias, R, ldr, Y, ldy, M, N,
                       K, n_tiles, total);
    return hipGetLastError();
}

// The same for the decoder's SHORT prompt blocks (llm_kernels.hip's prefill_gemm_kernel: 64 x 64 tiles, optional K slices whose
// partial tiles prefill_splitk_reduce_kernel adds): with bf16 weights, six bf16 MFMAs per wave and K-step of 32 instead of sixteen
// f32 ones.  Same tile order, slices and partial layout as that kernel, so its launcher and reduce kernel serve both.
constexpr int PGW_BM = 64, PGW_BN = 64, PGW_BK = 32;
constexpr int PGW_PLANE = 64 * 32;  // bf16 per plane (64 rows x 64 bytes, chunks swizzled as above)

template <bool RESIDUAL>
__global__ __launch_bounds__(256) void prefill_gemm_bf16w_kernel(const float* __restrict__ A, int64_t lda, const uint16_t* __restrict__ W,
                                                                 const float* __restrict__ bias, const float* R, int64_t ldr, float* Y,
                                                                 int64_t ldy, int M, int N, int K, int m_tiles, int ksplit,
                                                                 float* __restrict__ P)
{
    __shared__ __attribute__((aligned(16))) uint16_t sp[2][4 * PGW_PLANE];  // [stage][A1 | A2 | A3 | W]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, l31 = lane & 31, half = lane >> 5;
    const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7u;
    const unsigned bid0 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int ks = (int)(bid0 % (unsigned)ksplit);
    const unsigned bid = bid0 / (unsigned)ksplit;
    const int m0 = (int)(bid % (unsigned)m_tiles) * PGW_BM;
    const int n0 = (int)(bid / (unsigned)m_tiles) * PGW_BN;
    const int k_len = K / ksplit, k_begin = ks * k_len;

    const int a_row = tid >> 3, a_c4 = tid & 7, b_row = tid >> 2, b_c = tid & 3;
    const float* a_ptr[2];
    int stA[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = a_row + 32 * i;
        a_ptr[i] = A + (int64_t)min(m0 + row, M - 1) * lda + a_c4 * 4 + k_begin;
        stA[i] = row * 32 + (((a_c4 >> 1) ^ ((row >> 2) & 3)) * 2 + (a_c4 & 1)) * 4;
    }
    const uint16_t* b_ptr = W + (int64_t)min(n0 + b_row, N - 1) * K + b_c * 8 + k_begin;
    const int stW = 3 * PGW_PLANE + b_row * 32 + (b_c ^ ((b_row >> 2) & 3)) * 8;
    f32x4 ga[2];
    u32x4 gw;
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) ga[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + k0);
        gw = *reinterpret_cast<const u32x4*>(b_ptr + k0);
    };
    auto store = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x2 p1, p2, p3;
            sp_split(ga[i], p1, p2, p3);
            *reinterpret_cast<u32x2*>(&sp[stage][stA[i]]) = p1;
            *reinterpret_cast<u32x2*>(&sp[stage][PGW_PLANE + stA[i]]) = p2;
            *reinterpret_cast<u32x2*>(&sp[stage][2 * PGW_PLANE + stA[i]]) = p3;
        }
        *reinterpret_cast<u32x4*>(&sp[stage][stW]) = gw;
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int nk = k_len / PGW_BK;
    const int fsw = (l31 >> 2) & 3;
    const int fa = (wr * 32 + l31) * 32, fb = 3 * PGW_PLANE + (wc * 32 + l31) * 32;
    load(0);
    store(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load((kt + 1) * PGW_BK);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int ch = ((kb * 2 + half) ^ fsw) * 8;
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(&sp[cur][fb + ch]);
#pragma unroll
            for (int p = 2; p >= 0; --p) {   // the smallest piece first
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&sp[cur][p * PGW_PLANE + fa + ch]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            }
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

// The opt-in mode's form of gemm.hip's 64 x 64-tile route (calls of 257 .. 6 143 rows): both operands f32, three planes each, six
// products -- twelve bf16 MFMAs per wave and K-step of 32 instead of sixteen f32 ones at twice the cycles.  One workgroup per
// (tile, K slice), gemm_nt_f32_mid's tile order and partial-slab layout (its reduce kernels serve both).
template <int EPI, bool PARTIAL>
__global__ __launch_bounds__(256) void gemm_nt_f32_mid_split(const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
                                                             const float* __restrict__ bias, const float* R, int64_t ldr, float* Y,
                                                             int64_t ldy, int M, int N, int K, int n_tiles, int ksplit, float* __restrict__ P)
{
    __shared__ __attribute__((aligned(16))) uint16_t sp[2][6 * PGW_PLANE];  // [stage][A1 | A2 | A3 | W1 | W2 | W3]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, l31 = lane & 31, half = lane >> 5;
    const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7u;
    const unsigned bid0 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const unsigned bid = bid0 / (unsigned)ksplit;
    const int ks = (int)(bid0 - bid * (unsigned)ksplit);
    const unsigned mt = bid / (unsigned)n_tiles;
    const int m0 = (int)mt * 64, n0 = (int)(bid - mt * (unsigned)n_tiles) * 64;
    const int k_len = K / ksplit, k_begin = ks * k_len;

    const int s_row = tid >> 3, s_c4 = tid & 7;
    const float *a_ptr[2], *b_ptr[2];
    int st[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = s_row + 32 * i;
        a_ptr[i] = A + (int64_t)min(m0 + row, M - 1) * lda + s_c4 * 4 + k_begin;
        b_ptr[i] = W + (int64_t)min(n0 + row, N - 1) * K + s_c4 * 4 + k_begin;
        st[i] = row * 32 + (((s_c4 >> 1) ^ ((row >> 2) & 3)) * 2 + (s_c4 & 1)) * 4;
    }
    f32x4 ga[2], gb[2];
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
            u32x2 p1, p2, p3;
            sp_split(ga[i], p1, p2, p3);
            *reinterpret_cast<u32x2*>(&sp[stage][st[i]]) = p1;
            *reinterpret_cast<u32x2*>(&sp[stage][PGW_PLANE + st[i]]) = p2;
            *reinterpret_cast<u32x2*>(&sp[stage][2 * PGW_PLANE + st[i]]) = p3;
            sp_split(gb[i], p1, p2, p3);
            *reinterpret_cast<u32x2*>(&sp[stage][3 * PGW_PLANE + st[i]]) = p1;
            *reinterpret_cast<u32x2*>(&sp[stage][4 * PGW_PLANE + st[i]]) = p2;
            *reinterpret_cast<u32x2*>(&sp[stage][5 * PGW_PLANE + st[i]]) = p3;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int nk = k_len / 32;
    const int fsw = (l31 >> 2) & 3;
    const int fa = (wr * 32 + l31) * 32, fb = 3 * PGW_PLANE + (wc * 32 + l31) * 32;
    float bv = 0.0f;
    if (!PARTIAL && bias != nullptr) {
        const int bcol = n0 + wc * 32 + l31;
        bv = bias[bcol < N ? bcol : N - 1];
    }
    load(0);
    store(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load((kt + 1) * 32);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int ch = ((kb * 2 + half) ^ fsw) * 8;
            bf16x8 a[3], b[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[p] = *reinterpret_cast<const bf16x8*>(&sp[cur][p * PGW_PLANE + fa + ch]);
                b[p] = *reinterpret_cast<const bf16x8*>(&sp[cur][p * PGW_PLANE + fb + ch]);
            }
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};  // i + j <= 4, the smallest first
#pragma unroll
            for (int q = 0; q < 6; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]], b[PB[q]], acc, 0, 0, 0);
        }
        if (kt + 1 < nk) store(cur ^ 1);
        __syncthreads();
    }
    const int col = n0 + wc * 32 + l31;
    if (col < N) {
        if (PARTIAL) {
            float* out = P + (int64_t)ks * M * N;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 32 + acc_row(r, half);
                if (row < M) out[(int64_t)row * N + col] = acc[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 32 + acc_row(r, half);
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

std::atomic<int> g_f32_on_bf16{-1};  // -1: not decided yet (KJARNI_HIP_F32_ON_BF16 is read at the first launch)
inline bool f32_on_bf16()
{
    int v = g_f32_on_bf16.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = std::getenv("KJARNI_HIP_F32_ON_BF16");
        v = (e && e[0] == '1') ? 1 : 0;
        g_f32_on_bf16.store(v, std::memory_order_relaxed);
    }
    return v == 1;
}

template <int EPI>
hipError_t launch_split(const float* A, int64_t lda, const float* W, const float* bias, const float* R, int64_t ldr, float* Y,
                        int64_t ldy, int64_t M, int N, int K, hipStream_t stream)
{
    static_assert(SP_LDS_BYTES <= 64 * 1024, "more dynamic LDS than a launch gets without the opt-in attribute");
    const int n_tiles = N / BN;
    const int64_t total = ((M + BM - 1) / BM) * n_tiles;
    if (total > 0x7fffffff) return hipErrorInvalidValue;
    if ((int64_t)M * N * 4 >= ((int64_t)256 << 20) && R != Y)  // (not in place: those rows are the next projection's residual)
        hipLaunchKernelGGL((gemm_nt_f32_split<EPI, true>), dim3((unsigned)total), dim3(256), SP_LDS_BYTES, stream, A, lda, W, bias, R, ldr, Y, ldy,
                           M, N, K, n_tiles, total);
    else
        hipLaunchKernelGGL((gemm_nt_f32_split<EPI, false>), dim3((unsigned)total), dim3(256), SP_LDS_BYTES, stream, A, lda, W, bias, R, ldr, Y, ldy,
                           M, N, K, n_tiles, total);
    return hipGetLastError();
}


}  // namespace

void set_f32_on_bf16(bool on) { g_f32_on_bf16.store(on ? 1 : 0, std::memory_order_relaxed); }
bool get_f32_on_bf16() { return f32_on_bf16(); }

hipError_t launch_gemm_split(const float* A, int64_t lda, const float* W, const float* bias, const float* R, int64_t ldr, float* Y,
                             int64_t ldy, int64_t M, int N, int K, GemmEpilogue epi, hipStream_t stream)
{
    if (M <= 0) return hipSuccess;
    if (N % BN != 0 || K % 64 != 0) return hipErrorInvalidValue;
    switch (epi) {
    case EPI_BIAS: return launch_split<EPI_BIAS>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    case EPI_BIAS_GELU: return launch_split<EPI_BIAS_GELU>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    case EPI_BIAS_GELU_NEW: return launch_split<EPI_BIAS_GELU_NEW>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    case EPI_BIAS_RELU: return launch_split<EPI_BIAS_RELU>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    case EPI_BIAS_TANH: return launch_split<EPI_BIAS_TANH>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    case EPI_BIAS_RESIDUAL: return launch_split<EPI_BIAS_RESIDUAL>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    case EPI_BIAS_MUL_SILU: return launch_split<EPI_BIAS_MUL_SILU>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_gemm_mid_split(const float* A, int64_t lda, const float* W, const float* bias, const float* R, int64_t ldr, float* Y,
                                 int64_t ldy, int M, int N, int K, int ksplit, float* partials, GemmEpilogue epi, hipStream_t stream)
{
    if (M <= 0) return hipSuccess;
    if (K % (32 * ksplit) != 0 || lda % 4 || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(W) & 15)) return hipErrorInvalidValue;
    const int m_tiles = (M + 63) / 64, n_tiles = (N + 63) / 64;
    const dim3 grid((unsigned)(m_tiles * n_tiles * ksplit));
#define KJ_MS(EPI_)                                                                                                                   \
    hipLaunchKernelGGL((gemm_nt_f32_mid_split<EPI_, false>), grid, dim3(256), 0, stream, A, lda, W, bias, R, ldr, Y, ldy, M, N, K, n_tiles, 1, \
                       nullptr);                                                                                                      \
    return hipGetLastError()
    if (partials) {
        hipLaunchKernelGGL((gemm_nt_f32_mid_split<EPI_BIAS, true>), grid, dim3(256), 0, stream, A, lda, W, bias, R, ldr, Y, ldy, M, N, K, n_tiles,
                           ksplit, partials);
        return hipGetLastError();
    }
    if (ksplit != 1) return hipErrorInvalidValue;
    switch (epi) {
    case EPI_BIAS: KJ_MS(EPI_BIAS);
    case EPI_BIAS_GELU: KJ_MS(EPI_BIAS_GELU);
    case EPI_BIAS_GELU_NEW: KJ_MS(EPI_BIAS_GELU_NEW);
    case EPI_BIAS_RELU: KJ_MS(EPI_BIAS_RELU);
    case EPI_BIAS_TANH: KJ_MS(EPI_BIAS_TANH);
    case EPI_BIAS_RESIDUAL: KJ_MS(EPI_BIAS_RESIDUAL);
    case EPI_BIAS_MUL_SILU: KJ_MS(EPI_BIAS_MUL_SILU);
    }
#undef KJ_MS
    return hipErrorInvalidValue;
}

hipError_t launch_prefill_tiles_bf16w(unsigned grid, const float* A, int64_t lda, const void* W_bf16, const float* bias, const float* R,
                                      int64_t ldr, float* Y, int64_t ldy, int M, int N, int K, int m_tiles, int ksplit, float* partials,
                                      hipStream_t stream)
{
    const uint16_t* W = static_cast<const uint16_t*>(W_bf16);
    if (R)
        hipLaunchKernelGGL((prefill_gemm_bf16w_kernel<true>), dim3(grid), dim3(256), 0, stream, A, lda, W, bias, R, ldr, Y, ldy, M, N, K, m_tiles,
                           ksplit, partials);
    else
        hipLaunchKernelGGL((prefill_gemm_bf16w_kernel<false>), dim3(grid), dim3(256), 0, stream, A, lda, W, bias, R, ldr, Y, ldy, M, N, K, m_tiles,
                           ksplit, partials);
    return hipGetLastError();
}

hipError_t launch_gemm_bf16_weights(const float* A, int64_t lda, const void* W_bf16, const float* bias, const float* R, int64_t ldr,
                                    float* Y, int64_t ldy, int64_t M, int N, int K, GemmEpilogue epi, hipStream_t stream)
{
    if (M <= 0) return hipSuccess;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (N % BN != 0 || K % 64 != 0 || lda % 4 || ldy % 4 || (R && ldr % 4) || !al16(A) || !al16(W_bf16) || !al16(Y) || !al16(bias) || !al16(R) ||
        (int64_t)BM * lda * 4 >= ((int64_t)1 << 31) || (int64_t)BN * K * 2 >= ((int64_t)1 << 31))
        return hipErrorInvalidValue;
    const uint16_t* W = static_cast<const uint16_t*>(W_bf16);
    switch (epi) {
    case EPI_BIAS: return launch_bf16w<EPI_BIAS>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    case EPI_BIAS_RESIDUAL: return launch_bf16w<EPI_BIAS_RESIDUAL>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    case EPI_BIAS_MUL_SILU: return launch_bf16w<EPI_BIAS_MUL_SILU>(A, lda, W, bias, R, ldr, Y, ldy, M, N, K, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace kjarni
