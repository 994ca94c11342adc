"""High-level Classifier (mirror of crates/kjarni-ffi/bindings/python/kjarni/classifier.py)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

from ._ffi import KjarniClassResults, KjarniDevice, KjarniStringArray, check_error, lib


class Classifier:
    def __init__(self, model: Optional[str] = None, device: str = "cpu", cache_dir: Optional[str] = None,
                 labels: Optional[Sequence[str]] = None, multi_label: bool = False, quiet: bool = False,
                 model_path: Optional[str] = None):
        config = lib().kjarni_classifier_config_default()
        config.device = KjarniDevice.GPU if device == "gpu" else KjarniDevice.CPU
        config.multi_label = 1 if multi_label else 0
        config.quiet = 1 if quiet else 0
        self._keep = [s.encode("utf-8") if s else None for s in (model, cache_dir, model_path)]
        config.model_name, config.cache_dir, config.model_path = self._keep
        if labels:
            self._labels = (C.c_char_p * len(labels))(*[l.encode("utf-8") for l in labels])
            config.labels = self._labels
            config.num_labels = len(labels)
        self._handle = C.c_void_p()
        check_error(lib().kjarni_classifier_new(C.byref(config), C.byref(self._handle)))

    def __del__(self):
        if getattr(self, "_handle", None) and self._handle.value:
            lib().kjarni_classifier_free(self._handle)
            self._handle = C.c_void_p()

    def classify(self, text: str) -> List[Tuple[str, float]]:
        """All (label, score) pairs, highest score first."""
        res = KjarniClassResults()
        check_error(lib().kjarni_classifier_classify(self._handle, text.encode("utf-8"), C.byref(res)))
        out = res.to_list()
        res.free()
        return out

    def labels(self) -> List[str]:
        arr = KjarniStringArray()
        check_error(lib().kjarni_classifier_labels(self._handle, C.byref(arr)))
        out = arr.to_list()
        arr.free()
        return out

    @property
    def num_labels(self) -> int:
        return int(lib().kjarni_classifier_num_labels(self._handle))
