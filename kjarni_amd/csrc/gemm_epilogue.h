// Epilogue arithmetic shared by the GEMM kernels (gemm.hip, gemm_split.hip).
#pragma once
#include "device_utils.h"
#include "kernels.h"

namespace kjarni {

constexpr int EPI_GELU_LIBM_ID = 100;  // (tuning build: gemm.hip's EPI_GELU_LIBM)

// silu_scalar, activations.rs:74-82
__device__ __forceinline__ float silu_ref(float x)
{
    if (x <= -20.0f) return 0.0f;
    if (x >= 20.0f) return x;
    return x / (1.0f + expf(-x));
}

template <int EPI>
__device__ __forceinline__ float epilogue(float v)
{
    if (EPI == EPI_BIAS_GELU) return gelu_erf_fast(v);
    if (EPI == EPI_GELU_LIBM_ID) return gelu_erf(v);
    if (EPI == EPI_BIAS_GELU_NEW) return gelu_tanh(v);
    if (EPI == EPI_BIAS_RELU) return fmaxf(v, 0.0f);
    if (EPI == EPI_BIAS_TANH) return tanhf(v);
    return v;
}


}  // namespace kjarni
