"""High-level Indexer (mirror of crates/kjarni-ffi/bindings/python/kjarni/indexer.py)."""
from __future__ import annotations

import ctypes as C
import json
from typing import Callable, Dict, List, NamedTuple, Optional, Sequence

from ._ffi import (KjarniDevice, KjarniIndexInfo, KjarniIndexStats, KjarniProgressCallbackFn, KjarniStringArray,
                   check_error, lib)

STAGES = ["scanning", "loading", "embedding", "writing", "committing", "searching", "reranking"]


class Progress(NamedTuple):
    stage: str
    current: int
    total: int
    message: Optional[str]


class IndexStats(NamedTuple):
    documents_indexed: int
    chunks_created: int
    dimension: int
    size_bytes: int
    files_processed: int
    files_skipped: int
    elapsed_ms: int


class IndexInfo(NamedTuple):
    path: str
    document_count: int
    segment_count: int
    dimension: int
    size_bytes: int
    embedding_model: Optional[str]


class CancelToken:
    def __init__(self):
        self._handle = C.c_void_p(lib().kjarni_cancel_token_new())

    def __del__(self):
        if getattr(self, "_handle", None) and self._handle.value:
            lib().kjarni_cancel_token_free(self._handle)
            self._handle = C.c_void_p()

    def cancel(self):
        lib().kjarni_cancel_token_cancel(self._handle)

    def is_cancelled(self) -> bool:
        return bool(lib().kjarni_cancel_token_is_cancelled(self._handle))

    def reset(self):
        lib().kjarni_cancel_token_reset(self._handle)


def _c_strings(items: Sequence[str]):
    arr = (C.c_char_p * max(len(items), 1))(*[s.encode("utf-8") for s in items])
    return arr


def _wrap(on_progress):
    if on_progress is None:
        return KjarniProgressCallbackFn()  # NULL function pointer

    def cb(p, _user):
        on_progress(Progress(STAGES[p.stage] if 0 <= p.stage < len(STAGES) else str(p.stage), int(p.current),
                             int(p.total), p.message.decode("utf-8") if p.message else None))
    return KjarniProgressCallbackFn(cb)


def make_config(model=None, device="cpu", cache_dir=None, chunk_size=512, chunk_overlap=50, batch_size=32,
                extensions: Optional[Sequence[str]] = None, exclude_patterns: Optional[Sequence[str]] = None,
                recursive=True, include_hidden=False, max_file_size=10 * 1024 * 1024, quiet=False):
    config = lib().kjarni_indexer_config_default()
    config.device = KjarniDevice.GPU if device == "gpu" else KjarniDevice.CPU
    keep = [s.encode("utf-8") if s else None for s in
            (model, cache_dir, ",".join(extensions) if extensions else None,
             ",".join(exclude_patterns) if exclude_patterns else None)]
    config.model_name, config.cache_dir, config.extensions, config.exclude_patterns = keep
    config.chunk_size, config.chunk_overlap, config.batch_size = chunk_size, chunk_overlap, batch_size
    config.recursive, config.include_hidden = int(recursive), int(include_hidden)
    config.max_file_size, config.quiet = max_file_size, int(quiet)
    return config, keep


class Indexer:
    def __init__(self, model: Optional[str] = None, **kw):
        config, self._keep = make_config(model, **kw)
        self._handle = C.c_void_p()
        check_error(lib().kjarni_indexer_new(C.byref(config), C.byref(self._handle)))

    def __del__(self):
        if getattr(self, "_handle", None) and self._handle.value:
            lib().kjarni_indexer_free(self._handle)
            self._handle = C.c_void_p()

    def create(self, index_path: str, inputs: Sequence[str], force: bool = False,
               on_progress: Optional[Callable[[Progress], None]] = None,
               cancel_token: Optional[CancelToken] = None) -> IndexStats:
        st = KjarniIndexStats()
        arr = _c_strings(inputs)
        if on_progress is None and cancel_token is None:
            rc = lib().kjarni_indexer_create(self._handle, index_path.encode("utf-8"), arr, len(inputs), int(force),
                                             C.byref(st))
        else:
            cb = _wrap(on_progress)
            rc = lib().kjarni_indexer_create_with_callback(
                self._handle, index_path.encode("utf-8"), arr, len(inputs), int(force), cb, None,
                cancel_token._handle if cancel_token else None, C.byref(st))
        check_error(rc)
        return IndexStats(*[int(getattr(st, f)) for f in IndexStats._fields])

    def add(self, index_path: str, inputs: Sequence[str], on_progress=None, cancel_token=None) -> int:
        n = C.c_size_t(0)
        arr = _c_strings(inputs)
        if on_progress is None and cancel_token is None:
            rc = lib().kjarni_indexer_add(self._handle, index_path.encode("utf-8"), arr, len(inputs), C.byref(n))
        else:
            cb = _wrap(on_progress)
            rc = lib().kjarni_indexer_add_with_callback(self._handle, index_path.encode("utf-8"), arr, len(inputs), cb,
                                                        None, cancel_token._handle if cancel_token else None,
                                                        C.byref(n))
        check_error(rc)
        return int(n.value)

    @staticmethod
    def info(index_path: str) -> IndexInfo:
        return index_info(index_path)

    @staticmethod
    def delete(index_path: str):
        index_delete(index_path)

    @property
    def model_name(self) -> str:
        need = lib().kjarni_indexer_model_name(self._handle, None, 0)
        buf = C.create_string_buffer(max(need, 1))
        lib().kjarni_indexer_model_name(self._handle, buf, need)
        return buf.value.decode("utf-8")

    @property
    def dimension(self) -> int:
        return int(lib().kjarni_indexer_dimension(self._handle))

    @property
    def chunk_size(self) -> int:
        return int(lib().kjarni_indexer_chunk_size(self._handle))


def index_info(index_path: str) -> IndexInfo:
    info = KjarniIndexInfo()
    check_error(lib().kjarni_index_info(index_path.encode("utf-8"), C.byref(info)))
    out = IndexInfo(C.string_at(info.path).decode("utf-8"), int(info.document_count), int(info.segment_count),
                    int(info.dimension), int(info.size_bytes),
                    C.string_at(info.embedding_model).decode("utf-8") if info.embedding_model else None)
    lib().kjarni_index_info_free(info)
    return out


def index_delete(index_path: str):
    check_error(lib().kjarni_index_delete(index_path.encode("utf-8")))


# ---- host-side pieces (kjarni_hip.h) -------------------------------------------------------------
def text_split(text: str, chunk_size: int = 1000, chunk_overlap: int = 200, separator: Optional[str] = None) -> List[str]:
    arr = KjarniStringArray()
    check_error(lib().kjarni_text_split(text.encode("utf-8"), chunk_size, chunk_overlap,
                                        separator.encode("utf-8") if separator is not None else None, C.byref(arr)))
    out = arr.to_list()
    arr.free()
    return out


def collect_files(inputs: Sequence[str], **kw) -> List[str]:
    config, _keep = make_config(None, **kw)
    arr = KjarniStringArray()
    check_error(lib().kjarni_collect_files(C.byref(config), _c_strings(inputs), len(inputs), C.byref(arr)))
    out = arr.to_list()
    arr.free()
    return out


def index_write(index_path: str, dimension: int, texts: Sequence[str], embeddings, metadata: Optional[Sequence[Dict]] = None,
                max_docs_per_segment: int = 0, embedding_model: Optional[str] = None, append: bool = False):
    import numpy as np
    emb = np.ascontiguousarray(embeddings, dtype=np.float32).reshape(len(texts), dimension) if len(texts) else \
        np.zeros((0, dimension), np.float32)
    t = _c_strings(texts)
    m = (C.c_char_p * max(len(texts), 1))(*[json.dumps(md).encode("utf-8") if md is not None else None
                                            for md in (metadata or [None] * len(texts))])
    check_error(lib().kjarni_index_write(index_path.encode("utf-8"), dimension, max_docs_per_segment,
                                         embedding_model.encode("utf-8") if embedding_model else None, t, m,
                                         emb.ctypes.data_as(C.POINTER(C.c_float)), len(texts), int(append)))
