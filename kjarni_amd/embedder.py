"""High-level Embedder (mirror of crates/kjarni-ffi/bindings/python/kjarni/embedder.py)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from ._ffi import (KjarniDevice, KjarniEmbedderConfig, KjarniFloat2DArray, KjarniFloatArray, check_error, lib)


class Embedder:
    """Text embedding model.  `device` is accepted for signature compatibility:
    inference always runs on the AMD GPU."""

    def __init__(self, model: Optional[str] = None, device: str = "cpu", cache_dir: Optional[str] = None,
                 normalize: bool = True, quiet: bool = False, model_path: Optional[str] = None):
        config = lib().kjarni_embedder_config_default()
        config.device = KjarniDevice.GPU if device == "gpu" else KjarniDevice.CPU
        config.normalize = 1 if normalize else 0
        config.quiet = 1 if quiet else 0
        self._keep = [s.encode("utf-8") if s else None for s in (model, cache_dir, model_path)]
        config.model_name, config.cache_dir, config.model_path = self._keep
        self._handle = C.c_void_p()
        check_error(lib().kjarni_embedder_new(C.byref(config), C.byref(self._handle)))

    def __del__(self):
        if getattr(self, "_handle", None) and self._handle.value:
            lib().kjarni_embedder_free(self._handle)
            self._handle = C.c_void_p()

    def encode(self, text: str) -> List[float]:
        result = KjarniFloatArray()
        check_error(lib().kjarni_embedder_encode(self._handle, text.encode("utf-8"), C.byref(result)))
        out = result.to_numpy().tolist()
        result.free()
        return out

    def encode_batch(self, texts: Sequence[str]) -> np.ndarray:
        if not texts:
            return np.zeros((0, 0), np.float32)
        c_texts = (C.c_char_p * len(texts))(*[t.encode("utf-8") for t in texts])
        result = KjarniFloat2DArray()
        check_error(lib().kjarni_embedder_encode_batch(self._handle, c_texts, len(texts), C.byref(result)))
        out = result.to_numpy()
        result.free()
        return out

    def similarity(self, text1: str, text2: str) -> float:
        result = C.c_float()
        check_error(lib().kjarni_embedder_similarity(self._handle, text1.encode("utf-8"), text2.encode("utf-8"),
                                                     C.byref(result)))
        return float(result.value)

    @property
    def dim(self) -> int:
        return int(lib().kjarni_embedder_dim(self._handle))
