"""High-level Reranker (mirror of crates/kjarni-ffi/bindings/python/kjarni/reranker.py)."""
from __future__ import annotations

import ctypes as C
from typing import List, NamedTuple, Optional

from ._ffi import KjarniDevice, KjarniRerankResults, check_error, lib


class RerankResult(NamedTuple):
    index: int
    score: float
    document: str


class Reranker:
    def __init__(self, model: Optional[str] = None, device: str = "cpu", cache_dir: Optional[str] = None,
                 quiet: bool = False, model_path: Optional[str] = None):
        config = lib().kjarni_reranker_config_default()
        config.device = KjarniDevice.GPU if device == "gpu" else KjarniDevice.CPU
        config.quiet = 1 if quiet else 0
        self._keep = [s.encode("utf-8") if s else None for s in (model, cache_dir, model_path)]
        config.model_name, config.cache_dir, config.model_path = self._keep
        self._handle = C.c_void_p()
        check_error(lib().kjarni_reranker_new(C.byref(config), C.byref(self._handle)))

    def __del__(self):
        if getattr(self, "_handle", None) and self._handle.value:
            lib().kjarni_reranker_free(self._handle)
            self._handle = C.c_void_p()

    def score(self, query: str, document: str) -> float:
        result = C.c_float()
        check_error(lib().kjarni_reranker_score(self._handle, query.encode("utf-8"), document.encode("utf-8"),
                                                C.byref(result)))
        return float(result.value)

    def _run(self, query: str, documents: List[str], top_k: Optional[int]) -> List[RerankResult]:
        if not documents:
            return []
        c_docs = (C.c_char_p * len(documents))(*[d.encode("utf-8") for d in documents])
        res = KjarniRerankResults()
        if top_k is None:
            rc = lib().kjarni_reranker_rerank(self._handle, query.encode("utf-8"), c_docs, len(documents),
                                              C.byref(res))
        else:
            rc = lib().kjarni_reranker_rerank_top_k(self._handle, query.encode("utf-8"), c_docs, len(documents),
                                                    int(top_k), C.byref(res))
        check_error(rc)
        out = [RerankResult(i, s, documents[i]) for i, s in res.to_list()]
        res.free()
        return out

    def rerank(self, query: str, documents: List[str]) -> List[RerankResult]:
        return self._run(query, documents, None)

    def rerank_top_k(self, query: str, documents: List[str], k: int) -> List[RerankResult]:
        return self._run(query, documents, k)
