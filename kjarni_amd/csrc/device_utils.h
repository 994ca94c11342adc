// Device-side helpers shared by the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kjarni {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWave = 64;

// ---- cross-lane reductions without the LDS crossbar ---------------------------------------------------------------------------
// `__shfl_xor` compiles to ds_bpermute_b32: ~100 cycles of latency per butterfly step, six steps per wave-wide sum -- the
// longest chain inside a one-token decode stage (tools/lab/stage_lab.hip, profiles/r06h_stage_lab.log: a LayerNorm + GEMV
// stage 2.61 -> 2.19 us with the forms below).  The forms below give the SAME VALUES bit for bit (addition and max are
// commutative, and wherever a step reads a lane other than i ^ n that lane holds what lane i ^ n holds):
//   i ^ 32, i ^ 16   v_permlane32_swap / v_permlane16_swap of the register with a copy of itself (gfx950): the two results
//                    hold {v[i], v[i ^ n]} in some order -- combine them with a commutative op;
//   i ^ 2, i ^ 1     DPP quad permutations (exact for any data);
//   i ^ 8            DPP row_ror:8 (exact for any data: (i + 8) % 16 == i ^ 8 within the 16-lane row);
//   i ^ 4 etc.       a DPP rotation / mirror of the row, valid once the lanes it confuses are equal (see each helper).
// All of them need the lanes they read to be active: use them where the wave (or at least the 16-lane row) runs unmasked.
// They are VECTOR-ALU instructions where ds_bpermute_b32 is an LDS-pipe one: the kernels bound by the f32 matrix cores (whose
// MFMAs share the FP32 lanes with the VALU) keep their `__shfl_xor` -- the pipelined attention kernel lost 8 % with
// max_xor32 / sum_xor32 in its softmax (profiles/r06i); the latency-bound kernels (decode steps, row kernels, small calls) gain.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xF, 0xF, false));
}
constexpr int kDppQuadXor1 = 0xB1, kDppQuadXor2 = 0x4E, kDppRowMirror = 0x140, kDppRowHalfMirror = 0x141;
constexpr int kDppRowRor = 0x120;  // + n, n in 1..15

// {a[i], b[i]} = {v[i], v[i ^ 32]} (i < 32: a = own, b = partner; i >= 32: the other way round)
__device__ __forceinline__ void xor32_pair(float v, float& a, float& b)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const unsigned u = __float_as_uint(v);
    const u32x2 t = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    a = __uint_as_float(t[0]);
    b = __uint_as_float(t[1]);
}
// {a[i], b[i]} = {v[i], v[i ^ 16]} in some order
__device__ __forceinline__ void xor16_pair(float v, float& a, float& b)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const unsigned u = __float_as_uint(v);
    const u32x2 t = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    a = __uint_as_float(t[0]);
    b = __uint_as_float(t[1]);
}
__device__ __forceinline__ float sum_xor32(float v) { float a, b; xor32_pair(v, a, b); return a + b; }     // v[i] + v[i ^ 32]
__device__ __forceinline__ float sum_xor16(float v) { float a, b; xor16_pair(v, a, b); return a + b; }     // v[i] + v[i ^ 16]
__device__ __forceinline__ float max_xor32(float v) { float a, b; xor32_pair(v, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float max_xor16(float v) { float a, b; xor16_pair(v, a, b); return fmaxf(a, b); }

// Sum over the 16-lane row in the butterfly order 8, 4, 2, 1 (row_ror:n reads lane (i + n) % 16, which after the steps before it
// holds what lane i ^ n holds): == v += shfl_xor(v, 8); ... ; v += shfl_xor(v, 1), bit for bit.
__device__ __forceinline__ float row16_sum_desc(float v)
{
    v += dpp_mov<kDppRowRor + 8>(v);
    v += dpp_mov<kDppRowRor + 4>(v);
    v += dpp_mov<kDppRowRor + 2>(v);
    v += dpp_mov<kDppRowRor + 1>(v);
    return v;
}
__device__ __forceinline__ float row16_max_desc(float v)
{
    v = fmaxf(v, dpp_mov<kDppRowRor + 8>(v));
    v = fmaxf(v, dpp_mov<kDppRowRor + 4>(v));
    v = fmaxf(v, dpp_mov<kDppRowRor + 2>(v));
    v = fmaxf(v, dpp_mov<kDppRowRor + 1>(v));
    return v;
}
// Sum over aligned groups of 8 lanes in the butterfly order 1, 2, 4 (the quad permutations are exact; once a quad is uniform,
// row_half_mirror's lane 7 - i of the half row holds what lane i ^ 4 holds): == v += shfl_xor(v, 1); 2; 4, bit for bit.
__device__ __forceinline__ float group8_sum_asc(float v)
{
    v += dpp_mov<kDppQuadXor1>(v);
    v += dpp_mov<kDppQuadXor2>(v);
    v += dpp_mov<kDppRowHalfMirror>(v);
    return v;
}
__device__ __forceinline__ float group8_max_asc(float v)
{
    v = fmaxf(v, dpp_mov<kDppQuadXor1>(v));
    v = fmaxf(v, dpp_mov<kDppQuadXor2>(v));
    v = fmaxf(v, dpp_mov<kDppRowHalfMirror>(v));
    return v;
}

// Wave-wide sum / max in the butterfly order 32, 16, 8, 4, 2, 1: the values `for (off = 32; off; off >>= 1) v += shfl_xor(v, off)`
// gives, bit for bit, in every lane.
__device__ __forceinline__ float wave_sum(float v) { return row16_sum_desc(sum_xor16(sum_xor32(v))); }
__device__ __forceinline__ float wave_max(float v) { return row16_max_desc(max_xor16(max_xor32(v))); }

// ---- kernel arguments in ONE batch of scalar loads -----------------------------------------------------------------------------
// The compiler loads a kernel argument where it is first used; a latency-bound kernel with early exits and optional pointers
// (`if (n >= n_out) return;`, `bias ? bias[n] : 0`) then starts with a chain of scalar-load round trips -- argument, wait,
// branch, next argument, wait ... -- before its first vector load is issued.  kj_args_now(...) names the arguments in an empty asm
// statement with scalar-register constraints at the top of the kernel: all of them are loaded there, together, behind one wait
// (with -amdgpu-kernarg-preload-count=16 in the Makefile the first 16 dwords come in SGPRs with the wave).  In the decode steps'
// one-row GEMV the arguments are there 210-230 cycles after the wave's entry; worth ~1 % of a token (docs/history/r06.md 2b).
template <typename T>
__device__ __forceinline__ void kj_arg_now(const T& a)
{
    asm volatile("" ::"s"(a));
}
template <typename... T>
__device__ __forceinline__ void kj_args_now(const T&... a)
{
    (kj_arg_now(a), ...);
}

// Workgroups are dealt round-robin over the 8 XCDs (a private L2 each).  The unit workgroup `wg` of a launch of `total`
// workgroups should take so that every XCD owns ONE contiguous run of units (bijective for any total): the projections' tile
// order does this, and a row-wise kernel that follows it finds the rows its XCD's L2 already holds.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned wg, unsigned total)
{
    const unsigned q8 = total >> 3, r8 = total & 7u, xcd = wg & 7u, slot = wg >> 3;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
}

// activations.rs:56-59
__device__ __forceinline__ float gelu_erf(float x)
{
    return 0.5f * x * (1.0f + erff(x * 0.7071067811865475f));
}

// erf-GELU for the GEMM epilogues (activations.rs:56-59: 0.5 x (1 + erf(x / sqrt 2)) = x Phi(x)), with ONE transcendental:
//     GELU(x) = max(x, 0) - g(|x|),   g(a) = a Phi(-a) = a 2^-P(a),   P(a) = -log2 Phi(-a)
// P is smooth (1 at 0, ~ a^2 / (2 ln 2) far out) and a degree-6 polynomial, fitted on [0, 6] with the error weighted by g ln 2
// (what an error of P does to g; tools/fit_gelu.py), reproduces GELU in f32 to 2.8e-7 absolute over |x| <= 12 -- the
// rounding of the result itself near |x| = 4; beyond a = 6, g < 6e-9 and a is clamped.  Rounds 1-4 used Abramowitz & Stegun
// 7.1.26 (a reciprocal AND an exponential per element, 4.7e-7): FC1's epilogue runs once per 384 MFMA-flops-deep output
// element on the FP32 lanes the f32 matrix instructions also use, and the quarter-rate transcendentals were over half of it.
// max(x, 0) as 0.5 (x + |x|) (exact) so that a NaN stays a NaN.
constexpr float kGeluP0 = 0.999993085861206f, kGeluP1 = 1.1512017250061035f, kGeluP2 = 0.4587709307670593f, kGeluP3 = 0.05341215059161186f,
                kGeluP4 = -0.008080747909843922f, kGeluP5 = 0.0007692292565479875f, kGeluP6 = -3.3093903766712174e-05f;
__device__ __forceinline__ float gelu_erf_fast(float x)
{
    const float ax = fabsf(x);
    const float a = fminf(ax, 6.0f);
    float p = fmaf(kGeluP6, a, kGeluP5);
    p = fmaf(p, a, kGeluP4);
    p = fmaf(p, a, kGeluP3);
    p = fmaf(p, a, kGeluP2);
    p = fmaf(p, a, kGeluP1);
    p = fmaf(p, a, kGeluP0);
    const float g = a * __builtin_amdgcn_exp2f(-p);
    return fmaf(x + ax, 0.5f, -g);
}

// The same on a pair, with the polynomial on packed pairs (v_pk_fma_f32 / v_pk_mul_f32: two elements per VALU instruction;
// the same operations in the same order as above: the same bits) -- the epilogue has no MFMA to hide behind.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf_fast2(f32x2 x)
{
    const f32x2 ax = __builtin_elementwise_abs(x);
    const f32x2 a = {fminf(ax[0], 6.0f), fminf(ax[1], 6.0f)};
    auto k2 = [](float c) { return f32x2{c, c}; };
    f32x2 p = __builtin_elementwise_fma(k2(kGeluP6), a, k2(kGeluP5));
    p = __builtin_elementwise_fma(p, a, k2(kGeluP4));
    p = __builtin_elementwise_fma(p, a, k2(kGeluP3));
    p = __builtin_elementwise_fma(p, a, k2(kGeluP2));
    p = __builtin_elementwise_fma(p, a, k2(kGeluP1));
    p = __builtin_elementwise_fma(p, a, k2(kGeluP0));
    const f32x2 e = {__builtin_amdgcn_exp2f(-p[0]), __builtin_amdgcn_exp2f(-p[1])};
    const f32x2 g = a * e;
    return __builtin_elementwise_fma(x + ax, k2(0.5f), -g);
}

// activations.rs:62-66
__device__ __forceinline__ float gelu_tanh(float x)
{
    float x3 = x * x * x;
    float inner = 0.7978845608f * (x + 0.044715f * x3);
    return 0.5f * x * (1.0f + tanhf(inner));
}

// Row (reg, lane-half) of a 32x32 MFMA accumulator register.
__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

}  // namespace kjarni
