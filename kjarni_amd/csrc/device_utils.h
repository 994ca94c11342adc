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
