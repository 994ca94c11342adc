"""High-level Searcher (mirror of crates/kjarni-ffi/bindings/python/kjarni/searcher.py)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

from ._ffi import (KjarniDevice, KjarniSearchResults, KjarniStringArray, check_error, lib)

MODE_KEYWORD, MODE_SEMANTIC, MODE_HYBRID = 0, 1, 2
_MODES = {"keyword": MODE_KEYWORD, "semantic": MODE_SEMANTIC, "hybrid": MODE_HYBRID}


class Searcher:
    def __init__(self, model: Optional[str] = None, rerank_model: Optional[str] = None, device: str = "cpu",
                 cache_dir: Optional[str] = None, default_mode: str = "hybrid", default_top_k: int = 10,
                 quiet: bool = False):
        config = lib().kjarni_searcher_config_default()
        config.device = KjarniDevice.GPU if device == "gpu" else KjarniDevice.CPU
        self._keep = [s.encode("utf-8") if s else None for s in (model, rerank_model, cache_dir)]
        config.model_name, config.rerank_model, config.cache_dir = self._keep
        config.default_mode = _MODES[default_mode]
        config.default_top_k = default_top_k
        config.quiet = 1 if quiet else 0
        self._handle = C.c_void_p()
        check_error(lib().kjarni_searcher_new(C.byref(config), C.byref(self._handle)))

    def __del__(self):
        if getattr(self, "_handle", None) and self._handle.value:
            lib().kjarni_searcher_free(self._handle)
            self._handle = C.c_void_p()

    def search(self, index_path: str, query: str, mode: Optional[str] = None, top_k: Optional[int] = None,
               rerank: Optional[bool] = None, threshold: Optional[float] = None,
               source_pattern: Optional[str] = None, filter_key: Optional[str] = None,
               filter_value: Optional[str] = None) -> List[Dict]:
        opts = lib().kjarni_search_options_default()
        if mode is not None:
            opts.mode = _MODES[mode]
        if top_k:
            opts.top_k = top_k
        if rerank is not None:
            opts.use_reranker = 1 if rerank else 0
        if threshold:
            opts.threshold = threshold
        keep = [s.encode("utf-8") if s else None for s in (source_pattern, filter_key, filter_value)]
        opts.source_pattern, opts.filter_key, opts.filter_value = keep
        res = KjarniSearchResults()
        check_error(lib().kjarni_searcher_search_with_options(self._handle, index_path.encode("utf-8"),
                                                              query.encode("utf-8"), C.byref(opts), C.byref(res)))
        out = res.to_list()
        res.free()
        return out

    @property
    def has_reranker(self) -> bool:
        return bool(lib().kjarni_searcher_has_reranker(self._handle))

    @property
    def default_top_k(self) -> int:
        return int(lib().kjarni_searcher_default_top_k(self._handle))

    @property
    def default_mode(self) -> int:
        return int(lib().kjarni_searcher_default_mode(self._handle))

    def _name(self, fn) -> str:
        need = fn(self._handle, None, 0)
        if need == 0:
            return ""
        buf = C.create_string_buffer(need)
        fn(self._handle, buf, need)
        return buf.value.decode("utf-8")

    @property
    def model_name(self) -> str:
        return self._name(lib().kjarni_searcher_model_name)

    @property
    def reranker_model(self) -> str:
        return self._name(lib().kjarni_searcher_reranker_model)


def search_keywords(index_path: str, query: str, top_k: int = 10) -> List[Dict]:
    """BM25 keyword search over an on-disk index (no model, no GPU)."""
    res = KjarniSearchResults()
    check_error(lib().kjarni_search_keywords(index_path.encode("utf-8"), query.encode("utf-8"), top_k, C.byref(res)))
    out = res.to_list()
    res.free()
    return out


def bm25_tokenize(text: str) -> List[str]:
    arr = KjarniStringArray()
    check_error(lib().kjarni_bm25_tokenize(text.encode("utf-8"), C.byref(arr)))
    out = arr.to_list()
    arr.free()
    return out


def glob_match(pattern: str, path: str) -> bool:
    return bool(lib().kjarni_glob_match(pattern.encode("utf-8"), path.encode("utf-8")))


def rrf_fuse(keyword_ids, semantic_ids, limit: int):
    kw = (C.c_size_t * max(len(keyword_ids), 1))(*keyword_ids)
    sem = (C.c_size_t * max(len(semantic_ids), 1))(*semantic_ids)
    cap = len(keyword_ids) + len(semantic_ids) + 1
    ids = (C.c_size_t * cap)()
    sc = (C.c_float * cap)()
    n = C.c_size_t(0)
    check_error(lib().kjarni_rrf_fuse(kw, len(keyword_ids), sem, len(semantic_ids), limit, ids, sc, C.byref(n)))
    return [(int(ids[i]), float(sc[i])) for i in range(n.value)]


def set_keyword_parallel_min_docs(docs: int) -> None:
    """Index size from which BM25 walks the segments on several host threads (kjarni_hip_set_keyword_parallel_min_docs)."""
    lib().kjarni_hip_set_keyword_parallel_min_docs(docs)


def search_breakdown() -> Dict[str, float]:
    """Where this thread's last search call spent its time, in microseconds (kjarni_hip_search_breakdown): re-opening the
    index, the scan over the device image, of it the device round trip, the whole call."""
    v = (C.c_double * 4)()
    lib().kjarni_hip_search_breakdown(v, 4)
    return {"open_us": v[0], "scan_us": v[1], "device_us": v[2], "total_us": v[3], "host_rest_us": v[3] - v[0] - v[1]}


def index_search(index_path: str, text_query: Optional[str] = None, query_emb=None, mode: Optional[str] = None,
                 top_k: Optional[int] = None, threshold: Optional[float] = None, source_pattern: Optional[str] = None,
                 filter_key: Optional[str] = None, filter_value: Optional[str] = None) -> List[Dict]:
    """Index retrieval with a caller-supplied query embedding (kjarni_hip_index_search)."""
    import numpy as np
    opts = lib().kjarni_search_options_default()
    if mode is not None:
        opts.mode = _MODES[mode]
    if top_k:
        opts.top_k = top_k
    if threshold:
        opts.threshold = threshold
    keep = [s.encode("utf-8") if s else None for s in (source_pattern, filter_key, filter_value)]
    opts.source_pattern, opts.filter_key, opts.filter_value = keep
    q, dim = None, 0
    if query_emb is not None:
        qa = np.ascontiguousarray(query_emb, dtype=np.float32)
        q, dim = qa.ctypes.data_as(C.POINTER(C.c_float)), qa.shape[0]
    res = KjarniSearchResults()
    check_error(lib().kjarni_hip_index_search(index_path.encode("utf-8"),
                                              text_query.encode("utf-8") if text_query is not None else None,
                                              q, dim, C.byref(opts), C.byref(res)))
    out = res.to_list()
    res.free()
    return out
