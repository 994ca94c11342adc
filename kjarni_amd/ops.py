"""Single operators of the forward pass on host arrays (kjarni_hip.h, "single operators"):
the reference's LinearLayer::matmul (+ fused epilogue), EncoderSelfAttention core and
LayerNorm::forward, each as one HIP kernel.  For parity tests and micro-benchmarks."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _ffi
from ._ffi import check_error, lib

EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GELU_NEW, EPI_BIAS_RELU, EPI_BIAS_TANH, EPI_BIAS_RESIDUAL, EPI_BIAS_MUL_SILU = range(7)
POOL_MEAN, POOL_CLS, POOL_MAX, POOL_LAST = 0, 1, 2, 3


def _f(a):
    return None if a is None else a.ctypes.data_as(_ffi._f32p)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, np.float32)


def linear(x, w, bias=None, residual=None, epilogue: int = EPI_BIAS, iters: int = 0, device: int = 0
           ) -> Tuple[np.ndarray, Optional[float]]:
    x, w, bias, residual = _c(x), _c(w), _c(bias), _c(residual)
    m, k = x.shape
    n = w.shape[0]
    y = np.empty((m, n), np.float32)
    ms = C.c_float(0)
    check_error(lib().kjarni_hip_op_linear(device, _f(x), _f(w), _f(bias), _f(residual), m, k, n, epilogue, _f(y),
                                           iters, C.byref(ms)))
    return y, (float(ms.value) if iters > 0 else None)


def to_bf16(w: np.ndarray) -> np.ndarray:
    """f32 -> bf16 bit patterns (uint16), round to nearest even (finite inputs)."""
    u = np.ascontiguousarray(w, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_to_f32(w16: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(w16, np.uint16).astype(np.uint32) << 16).view(np.float32)


def linear_bf16_weights(x, w_bf16, bias=None, residual=None, epilogue: int = EPI_BIAS, iters: int = 0, device: int = 0
                        ) -> Tuple[np.ndarray, Optional[float]]:
    """y = epilogue(x . w^T + bias (+ residual)) with w given as bf16 bit patterns [n, k] (kjarni_hip_op_linear_bf16_weights)."""
    x, bias, residual = _c(x), _c(bias), _c(residual)
    w16 = np.ascontiguousarray(w_bf16, np.uint16)
    m, k = x.shape
    n = w16.shape[0]
    y = np.empty((m, n), np.float32)
    ms = C.c_float(0)
    check_error(lib().kjarni_hip_op_linear_bf16_weights(device, _f(x), w16.ctypes.data_as(C.c_void_p), _f(bias), _f(residual), m, k, n,
                                                        epilogue, _f(y), iters, C.byref(ms)))
    return y, (float(ms.value) if iters > 0 else None)


def attention(qkv, mask, heads: int, mask_value: float = -1e9, iters: int = 0, device: int = 0
              ) -> Tuple[np.ndarray, Optional[float]]:
    qkv = _c(qkv)
    b, s, h3 = qkv.shape
    hidden = h3 // 3
    mask = None if mask is None else np.ascontiguousarray(mask, np.uint32)
    ctx = np.empty((b, s, hidden), np.float32)
    ms = C.c_float(0)
    check_error(lib().kjarni_hip_op_attention(device, _f(qkv), None if mask is None else mask.ctypes.data_as(_ffi._u32p),
                                              b, s, heads, hidden // heads, float(mask_value), _f(ctx), iters,
                                              C.byref(ms)))
    return ctx, (float(ms.value) if iters > 0 else None)


def pool(hidden_states, mask=None, pooling: int = 0, normalize: bool = False, device: int = 0) -> np.ndarray:
    """hidden_states [B, S, H] -> [B, H] (kjarni_hip_op_pool): pooling 0 mean, 1 cls, 2 max, 3 last token; optional L2."""
    h = _c(hidden_states)
    b, s, d = h.shape
    mask = None if mask is None else np.ascontiguousarray(mask, np.uint32)
    out = np.empty((b, d), np.float32)
    check_error(lib().kjarni_hip_op_pool(device, _f(h), None if mask is None else mask.ctypes.data_as(_ffi._u32p), b, s, d,
                                         int(pooling), 1 if normalize else 0, _f(out)))
    return out


def attention_biased(qkv, mask, heads: int, position_bias=None, scale_qk: bool = True, mask_value: float = -1e9,
                     device: int = 0) -> np.ndarray:
    """EncoderSelfAttention with the reference's full argument list (kjarni_hip_op_attention_biased): position_bias
    [1, heads, S', S'] or [heads, S', S'] with S' >= seq, added after the scale and before the padding mask."""
    qkv = _c(qkv)
    b, s, h3 = qkv.shape
    hidden = h3 // 3
    mask = None if mask is None else np.ascontiguousarray(mask, np.uint32)
    bias, bias_seq = None, 0
    if position_bias is not None:
        bias = np.ascontiguousarray(position_bias, np.float32)
        bias = bias.reshape(bias.shape[-3:])
        assert bias.shape[0] == heads and bias.shape[1] == bias.shape[2]
        bias_seq = int(bias.shape[1])
    ctx = np.empty((b, s, hidden), np.float32)
    check_error(lib().kjarni_hip_op_attention_biased(device, _f(qkv), None if mask is None else mask.ctypes.data_as(_ffi._u32p),
                                                     None if bias is None else _f(bias), bias_seq, b, s, heads, hidden // heads,
                                                     1 if scale_qk else 0, float(mask_value), _f(ctx)))
    return ctx


def layer_norm(x, gamma, beta, eps: float, iters: int = 0, device: int = 0) -> Tuple[np.ndarray, Optional[float]]:
    x, gamma, beta = _c(x), _c(gamma), _c(beta)
    hidden = x.shape[-1]
    rows = x.size // hidden
    y = np.empty_like(x)
    ms = C.c_float(0)
    check_error(lib().kjarni_hip_op_layer_norm(device, _f(x), _f(gamma), _f(beta), float(eps), rows, hidden, _f(y),
                                               iters, C.byref(ms)))
    return y, (float(ms.value) if iters > 0 else None)


def linear_layer_norm(x, w, bias, residual, gamma, beta, eps: float, iters: int = 0, device: int = 0
                      ) -> Tuple[np.ndarray, Optional[float]]:
    """LayerNorm(x . w^T + bias + residual) * gamma + beta (kjarni_hip_op_linear_layer_norm)."""
    x, w, bias, residual, gamma, beta = _c(x), _c(w), _c(bias), _c(residual), _c(gamma), _c(beta)
    m, k = x.shape
    n = w.shape[0]
    y = np.empty((m, n), np.float32)
    ms = C.c_float(0)
    check_error(lib().kjarni_hip_op_linear_layer_norm(device, _f(x), _f(w), _f(bias), _f(residual), _f(gamma), _f(beta),
                                                      float(eps), m, k, n, _f(y), iters, C.byref(ms)))
    return y, (float(ms.value) if iters > 0 else None)


def _tuning(name: str):
    """Kernel A/B switches exist only in the tuning build: `make -C kjarni_amd/csrc tuning`, then run the tool with
    KJARNI_FFI_LIB=kjarni_amd/lib/libkjarni_ffi_tuning.so.  The shipped library does not export them."""
    try:
        fn = getattr(lib(), name)
    except AttributeError:
        raise RuntimeError(f"{name} needs the tuning build (KJARNI_FFI_LIB=.../libkjarni_ffi_tuning.so)") from None
    fn.restype, fn.argtypes = None, [C.c_int32]
    return fn


def set_f32_on_bf16(on: bool) -> bool:
    """Process-wide opt-in (kjarni_hip_set_f32_on_bf16): the large-batch projections compute their f32 products on the bf16
    matrix cores from three exact bf16 pieces per operand.  Returns the previous setting."""
    return bool(lib().kjarni_hip_set_f32_on_bf16(1 if on else 0))


def clock_probe(out_dev_ptr: int, spin_us: int = 20, stream: int = 0) -> None:
    """Enqueues the one-wave clock probe (kjarni_hip_clock_probe): out_dev_ptr -> two uint64 on the device, [shader cycles,
    10 ns ticks]; GHz = cycles / ticks / 10."""
    check_error(lib().kjarni_hip_clock_probe(out_dev_ptr, int(spin_us), stream))


def clock_trace(out_dev_ptr: int, samples: int, window_us: int, stream: int) -> None:
    """Enqueues the repeated clock reading (kjarni_hip_clock_trace) on `stream` -- a stream of its own beside the work being
    measured: out_dev_ptr -> samples x [shader cycles, 10 ns ticks] uint64 on the device."""
    check_error(lib().kjarni_hip_clock_trace(out_dev_ptr, int(samples), int(window_us), stream))


def measurement_stream() -> int:
    """The library's own non-blocking stream for clock traces (kjarni_hip_measurement_stream); 0 if it cannot be made."""
    return int(lib().kjarni_hip_measurement_stream() or 0)


def measurement_stream_release() -> None:
    """Waits for the measurement stream and destroys it (kjarni_hip_measurement_stream_release)."""
    lib().kjarni_hip_measurement_stream_release()


def get_f32_on_bf16() -> bool:
    return bool(lib().kjarni_hip_get_f32_on_bf16())


def has_tuning() -> bool:
    return hasattr(lib(), "kjarni_hip_set_gemm_variant")


def set_gemm_variant(v: int):
    _tuning("kjarni_hip_set_gemm_variant")(int(v))


def set_attention_variant(v: int):
    _tuning("kjarni_hip_set_attention_variant")(int(v))


def set_cosine_variant(v: int):
    _tuning("kjarni_hip_set_cosine_variant")(int(v))


def topk(scores, k: int, device: int = 0):
    """Top-k of a score matrix [nq, n] on the GPU (kjarni_hip_cosine_topk): (idx int64 [nq,k], score f32 [nq,k]),
    score descending, equal scores by ascending index; entries past n are (-1, -inf)."""
    import ctypes as C
    from ._ffi import check_error
    scores = np.ascontiguousarray(scores, np.float32)
    if scores.ndim == 1:
        scores = scores[None, :]
    nq, n = scores.shape
    L = lib()
    ws_bytes = L.kjarni_hip_cosine_topk_workspace_bytes(nq, n, k)
    bufs = []

    def dmalloc(nbytes):
        p = C.c_void_p()
        check_error(L.kjarni_hip_malloc(device, max(nbytes, 16), C.byref(p)))
        bufs.append(p)
        return p
    try:
        d_sc, d_ws, d_idx, d_out = dmalloc(scores.nbytes), dmalloc(ws_bytes), dmalloc(nq * k * 8), dmalloc(nq * k * 4)
        check_error(L.kjarni_hip_memcpy_h2d(device, d_sc, scores.ctypes.data_as(C.c_void_p), scores.nbytes))
        check_error(L.kjarni_hip_cosine_topk(device, d_sc, nq, n, k, d_ws, d_idx, d_out, None))
        check_error(L.kjarni_hip_synchronize(device))
        idx = np.empty((nq, k), np.int64)
        out = np.empty((nq, k), np.float32)
        check_error(L.kjarni_hip_memcpy_d2h(device, idx.ctypes.data_as(C.c_void_p), d_idx, idx.nbytes))
        check_error(L.kjarni_hip_memcpy_d2h(device, out.ctypes.data_as(C.c_void_p), d_out, out.nbytes))
        return idx, out
    finally:
        for p in bufs:
            L.kjarni_hip_free(device, p)
