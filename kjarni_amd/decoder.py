"""Token-level access to the decoder-only (Llama / Qwen2) path on the GPU (kjarni_hip_decoder_*)."""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _ffi
from ._ffi import check_error, lib

WEIGHTS = {"auto": 0, "f32": 1, "bf16": 2}


class HipDecoder:
    def __init__(self, model_dir: str, device: int = 0, weights: str = "auto", max_context: int = 0):
        self._h = C.c_void_p()
        check_error(lib().kjarni_hip_decoder_load(model_dir.encode("utf-8"), device, WEIGHTS[weights], max_context, C.byref(self._h)))
        a, b, c, d, e = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        wb = C.c_uint64()
        check_error(lib().kjarni_hip_decoder_dims(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e), C.byref(wb)))
        self.hidden, self.layers, self.vocab, self.context, self.bf16, self.weight_bytes = a.value, b.value, c.value, d.value, bool(e.value), wb.value

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().kjarni_hip_decoder_free(self._h)
            self._h = C.c_void_p()

    def reset(self):
        check_error(lib().kjarni_hip_decoder_reset(self._h))

    def set_device_sampling(self, on: bool):
        lib().kjarni_hip_decoder_set_device_sampling(self._h, 1 if on else 0)

    def tile_gemm_calls(self) -> int:
        """Prompt projections that took the 128 x 128-tile GEMM route since load."""
        return int(lib().kjarni_hip_decoder_tile_gemm_calls(self._h))

    def forward(self, ids: Sequence[int], fetch: bool = True):
        a = np.ascontiguousarray(ids, np.uint32)
        hidden = np.empty(((a.size - 1) % 8 + 1, self.hidden), np.float32) if fetch else None   # rows of the last 8-row block
        logits = np.empty(self.vocab, np.float32) if fetch else None
        f = lambda x: x.ctypes.data_as(C.POINTER(C.c_float)) if x is not None else None  # noqa: E731
        check_error(lib().kjarni_hip_decoder_forward(self._h, a.ctypes.data_as(C.POINTER(C.c_uint32)), a.size, f(hidden), f(logits)))
        return hidden, logits

    def generate(self, prompt: Sequence[int], max_new_tokens: int, repetition_penalty: float = 1.0, no_repeat_ngram: int = 0,
                 on_token: Optional[Callable[[int], Optional[bool]]] = None) -> List[int]:
        p = np.ascontiguousarray(prompt, np.uint32)
        out = np.empty(max(max_new_tokens, 1), np.uint32)
        n = C.c_size_t(0)

        def cb(t, _u):
            r = on_token(int(t.token_id))
            return True if r is None else bool(r)
        fn = _ffi.KjarniTokenCallbackFn(cb) if on_token else _ffi.KjarniTokenCallbackFn()
        check_error(lib().kjarni_hip_decoder_generate(self._h, p.ctypes.data_as(C.POINTER(C.c_uint32)), p.size, max_new_tokens,
                                                      repetition_penalty, no_repeat_ngram, fn, None,
                                                      out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size, C.byref(n)))
        return out[:min(n.value, out.size)].tolist()
