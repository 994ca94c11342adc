// Device-side helpers shared by the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kjarni {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
    return v;
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

// erf-GELU for the GEMM epilogue.  erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7, the
// same approximation the reference's own GPU shader uses: gpu_ops/blocks/ffn/fc1.wgsl:57-71):
// 2 transcendentals + ~12 VALU per element instead of ~34 for the libm-grade erff, which matters
// because FC1's epilogue runs once per 384 MFMA-flops-deep output element.
__device__ __forceinline__ float gelu_erf_fast(float x)
{
    const float z = fabsf(x) * 0.7071067811865475f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    const float erf_abs = fmaf(-p, e, 1.0f);
    const float erf_v = copysignf(erf_abs, x);
    return 0.5f * x * (1.0f + erf_v);
}

// The same on four elements, with the polynomial on packed pairs (v_pk_fma_f32 / v_pk_mul_f32:
// two elements per VALU instruction) -- the epilogue has no MFMA to hide behind.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf_fast2(f32x2 x)
{
    const f32x2 ax = __builtin_elementwise_abs(x);
    const f32x2 z = ax * 0.7071067811865475f;
    const f32x2 d = __builtin_elementwise_fma(z, f32x2{0.3275911f, 0.3275911f}, f32x2{1.0f, 1.0f});
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2 p = __builtin_elementwise_fma(t, f32x2{1.061405429f, 1.061405429f}, f32x2{-1.453152027f, -1.453152027f});
    p = __builtin_elementwise_fma(p, t, f32x2{1.421413741f, 1.421413741f});
    p = __builtin_elementwise_fma(p, t, f32x2{-0.284496736f, -0.284496736f});
    p = __builtin_elementwise_fma(p, t, f32x2{0.254829592f, 0.254829592f});
    p = p * t;
    const f32x2 a = z * z * -1.4426950408889634f;
    const f32x2 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
    const f32x2 erf_abs = __builtin_elementwise_fma(-p, e, f32x2{1.0f, 1.0f});  // erf(|x|/sqrt2) in [0,1]
    // 0.5*x*(1 + sign(x)*erf_abs) = 0.5*(x + |x|*erf_abs)
    return __builtin_elementwise_fma(ax, erf_abs, x) * 0.5f;
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
