"""Host-side BERT WordPiece tokenizer of libkjarni_ffi.so (kjarni_hip.h, tokenizer section).

Same configuration the reference applies at load time
(crates/kjarni-transformers/src/pipeline/encoder/loader.rs:98-115): truncation to
max_length, BatchLongest right padding with id 0."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from ._ffi import KjarniTokenBatch, check_error, lib


class Tokenizer:
    def __init__(self, tokenizer_json: str, max_length: int = 512):
        self._h = C.c_void_p()
        check_error(lib().kjarni_tokenizer_load(str(tokenizer_json).encode(), int(max_length), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().kjarni_tokenizer_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode_batch(self, texts: Sequence[str], pairs: Optional[Sequence[str]] = None
                     ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Returns (ids, attention_mask, type_ids), each uint32 [batch, longest]."""
        n = len(texts)
        a = (C.c_char_p * max(n, 1))(*[t.encode("utf-8") for t in texts])
        b = None
        if pairs is not None:
            assert len(pairs) == n
            b = (C.c_char_p * max(n, 1))(*[t.encode("utf-8") for t in pairs])
        out = KjarniTokenBatch()
        check_error(lib().kjarni_tokenizer_encode_batch(self._h, a, b, n, C.byref(out)))
        try:
            shape = (out.batch, out.seq)
            if out.batch == 0 or out.seq == 0:
                z = np.zeros(shape, np.uint32)
                return z, z.copy(), z.copy()
            ids = np.ctypeslib.as_array(out.ids, shape=shape).copy()
            mask = np.ctypeslib.as_array(out.attention_mask, shape=shape).copy()
            types = np.ctypeslib.as_array(out.type_ids, shape=shape).copy()
            return ids, mask, types
        finally:
            lib().kjarni_token_batch_free(C.byref(out))
