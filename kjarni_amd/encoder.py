"""Token-level encoder on one MI355X (include/kjarni_hip.h).

Host-side mirror of the reference's token-level interface
(EncoderLanguageModel::get_hidden_states_batch_from_ids,
crates/kjarni-transformers/src/cpu/encoder/traits.rs:66-139;
SentenceEncoder::encode_batch_flat, kjarni-models/.../sentence_encoder/model.rs:201-218;
CrossEncoder::predict_pairs, .../cross_encoder/model.rs:170-240)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _ffi
from ._ffi import check_error, lib

POOL_MEAN, POOL_CLS, POOL_MAX, POOL_LAST_TOKEN = 0, 1, 2, 3
MASK_AUTO, MASK_NEG_1E9, MASK_NEG_INF = 0, 1, 2
COSINE_VECTOR_STORE, COSINE_SEGMENT = 0, 1

_POOL_NAMES = {"mean": POOL_MEAN, "cls": POOL_CLS, "max": POOL_MAX, "lasttoken": POOL_LAST_TOKEN,
               "last_token": POOL_LAST_TOKEN}


def device_count() -> int:
    return int(lib().kjarni_hip_device_count())


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


def _p(a: Optional[np.ndarray], typ):
    return None if a is None else a.ctypes.data_as(typ)


class HipEncoder:
    """A BERT-family encoder resident on one HIP device."""

    def __init__(self, model_dir: str, device: int = 0):
        self._h = C.c_void_p()
        check_error(lib().kjarni_hip_encoder_load(str(model_dir).encode(), int(device), C.byref(self._h)))
        L = lib()
        self.hidden_size = int(L.kjarni_hip_encoder_hidden_size(self._h))
        self.num_layers = int(L.kjarni_hip_encoder_num_layers(self._h))
        self.max_seq_len = int(L.kjarni_hip_encoder_max_seq_len(self._h))
        self.vocab_size = int(L.kjarni_hip_encoder_vocab_size(self._h))
        self.num_labels = int(L.kjarni_hip_encoder_num_labels(self._h))
        self.device = int(L.kjarni_hip_encoder_device(self._h))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().kjarni_hip_encoder_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_chunk_tokens(self, tokens: int):
        check_error(lib().kjarni_hip_encoder_set_chunk_tokens(self._h, int(tokens)))

    def set_packing(self, mode):
        """Ragged batches over the kept tokens only: 0 / False never, 1 / True (default) the host-array entry points, 2 also
        the device-pointer ones (embed_dev / logits_dev then synchronise their stream once per call: kjarni_hip.h)."""
        check_error(lib().kjarni_hip_encoder_set_packing(self._h, int(mode)))

    def set_combining(self, on: bool):
        """Small host-array calls that arrive while another is on the device ride along with the next one as one packed batch
        (opt-in, default off; kjarni_hip.h: kjarni_hip_encoder_set_combining)."""
        check_error(lib().kjarni_hip_encoder_set_combining(self._h, 1 if on else 0))

    def set_two_lanes(self, on: bool):
        """Mid-size host-array calls as two halves on two streams (default on; kjarni_hip.h: kjarni_hip_encoder_set_two_lanes)."""
        check_error(lib().kjarni_hip_encoder_set_two_lanes(self._h, 1 if on else 0))

    KINDS = ("embed_layernorm", "gemm_qkv", "attention", "gemm_out_proj", "layernorm", "gemm_fc1", "gemm_fc2",
             "pool", "head", "rope")

    def profile_begin(self, kinds=None):
        """Bracket kernel launches with HIP events on their stream until profile_end().
        kinds: iterable of names from KINDS to time only those kernels (default: all)."""
        if kinds is None:
            check_error(lib().kjarni_hip_encoder_profile_begin(self._h))
        else:
            mask = 0
            for k in kinds:
                mask |= 1 << self.KINDS.index(k)
            check_error(lib().kjarni_hip_encoder_profile_begin_kinds(self._h, mask))

    def profile_end(self):
        """Returns [{kind, symbol, launches, total_ms, flops, bytes}] (algorithmic flops/bytes)."""
        arr = (_ffi.KjarniHipKernelStat * 16)()
        n = C.c_size_t(0)
        check_error(lib().kjarni_hip_encoder_profile_end(self._h, arr, 16, C.byref(n)))
        return [dict(kind=arr[i].kind.decode(), symbol=arr[i].symbol.decode(), launches=int(arr[i].launches),
                     total_ms=float(arr[i].total_ms), flops=float(arr[i].flops), bytes=float(arr[i].bytes))
                for i in range(n.value)]

    # ---- host-array entry points (copy in, run, copy out, synchronise) ----
    def hidden_states(self, ids, mask, type_ids=None, fill: int = MASK_AUTO) -> np.ndarray:
        ids, mask = _u32(ids), _u32(mask)
        type_ids = None if type_ids is None else _u32(type_ids)
        B, S = ids.shape
        out = np.empty((B, S, self.hidden_size), np.float32)
        check_error(lib().kjarni_hip_encoder_hidden_states_host(
            self._h, _p(ids, _ffi._u32p), _p(mask, _ffi._u32p), _p(type_ids, _ffi._u32p), B, S, fill,
            _p(out, _ffi._f32p)))
        return out

    def embed(self, ids, mask, type_ids=None, pooling="mean", normalize: bool = True,
              fill: int = MASK_AUTO) -> np.ndarray:
        ids, mask = _u32(ids), _u32(mask)
        type_ids = None if type_ids is None else _u32(type_ids)
        B, S = ids.shape
        pool = _POOL_NAMES[pooling.lower()] if isinstance(pooling, str) else int(pooling)
        out = np.empty((B, self.hidden_size), np.float32)
        check_error(lib().kjarni_hip_encoder_embed_host(
            self._h, _p(ids, _ffi._u32p), _p(mask, _ffi._u32p), _p(type_ids, _ffi._u32p), B, S, pool,
            int(bool(normalize)), fill, _p(out, _ffi._f32p)))
        return out

    def logits(self, ids, mask, type_ids=None, fill: int = MASK_AUTO) -> np.ndarray:
        ids, mask = _u32(ids), _u32(mask)
        type_ids = None if type_ids is None else _u32(type_ids)
        B, S = ids.shape
        out = np.empty((B, self.num_labels), np.float32)
        check_error(lib().kjarni_hip_encoder_logits_host(
            self._h, _p(ids, _ffi._u32p), _p(mask, _ffi._u32p), _p(type_ids, _ffi._u32p), B, S, fill,
            _p(out, _ffi._f32p)))
        return out

    # ---- device-pointer entry points (ints are raw device addresses) ----
    def embed_dev(self, ids_ptr: int, mask_ptr: int, batch: int, seq: int, out_ptr: int,
                  type_ptr: int = 0, pooling: int = POOL_MEAN, normalize: bool = True,
                  fill: int = MASK_AUTO, stream: int = 0):
        check_error(lib().kjarni_hip_encoder_embed(self._h, ids_ptr, mask_ptr, type_ptr or None, batch, seq,
                                                   pooling, int(bool(normalize)), fill, out_ptr,
                                                   stream or None))

    def logits_dev(self, ids_ptr: int, mask_ptr: int, type_ptr: int, batch: int, seq: int, out_ptr: int,
                   fill: int = MASK_AUTO, stream: int = 0):
        check_error(lib().kjarni_hip_encoder_logits(self._h, ids_ptr, mask_ptr, type_ptr or None, batch, seq,
                                                    fill, out_ptr, stream or None))

    def hidden_states_dev(self, ids_ptr: int, mask_ptr: int, type_ptr: int, batch: int, seq: int,
                          out_ptr: int, fill: int = MASK_AUTO, stream: int = 0):
        check_error(lib().kjarni_hip_encoder_hidden_states(self._h, ids_ptr, mask_ptr, type_ptr or None, batch,
                                                           seq, fill, out_ptr, stream or None))


class HipEncoderGroup:
    """One replica of a BERT-family encoder per listed HIP device, driven from this process (kjarni_hip_group_*):
    batches are cut into balanced contiguous row blocks, one host thread + stream per device.  devices=None uses
    KJARNI_HIP_DEVICES / every visible device; a device may be listed twice."""

    def __init__(self, model_dir: str, devices=None):
        self._h = C.c_void_p()
        devs = list(devices or [])
        arr = (C.c_int32 * max(len(devs), 1))(*devs)
        check_error(lib().kjarni_hip_group_load(str(model_dir).encode(), arr if devs else None, len(devs),
                                                C.byref(self._h)))
        L = lib()
        self.size = int(L.kjarni_hip_group_size(self._h))
        self.devices = [int(L.kjarni_hip_group_device(self._h, i)) for i in range(self.size)]
        self.hidden_size = int(L.kjarni_hip_group_hidden_size(self._h))
        self.num_labels = int(L.kjarni_hip_group_num_labels(self._h))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().kjarni_hip_group_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def transport(self) -> str:
        return lib().kjarni_hip_group_transport(self._h).decode()

    def shard(self, rows: int, i: int):
        s, c = C.c_int64(0), C.c_int64(0)
        check_error(lib().kjarni_hip_group_shard(self._h, rows, i, C.byref(s), C.byref(c)))
        return int(s.value), int(c.value)

    def embed(self, ids, mask, type_ids=None, pooling="mean", normalize: bool = True, fill: int = MASK_AUTO) -> np.ndarray:
        ids, mask = _u32(ids), _u32(mask)
        type_ids = None if type_ids is None else _u32(type_ids)
        B, S = ids.shape
        pool = _POOL_NAMES[pooling.lower()] if isinstance(pooling, str) else int(pooling)
        out = np.empty((B, self.hidden_size), np.float32)
        check_error(lib().kjarni_hip_group_embed_host(
            self._h, _p(ids, _ffi._u32p), _p(mask, _ffi._u32p), _p(type_ids, _ffi._u32p), B, S, pool,
            int(bool(normalize)), fill, _p(out, _ffi._f32p)))
        return out

    def logits(self, ids, mask, type_ids=None, fill: int = MASK_AUTO) -> np.ndarray:
        ids, mask = _u32(ids), _u32(mask)
        type_ids = None if type_ids is None else _u32(type_ids)
        B, S = ids.shape
        out = np.empty((B, self.num_labels), np.float32)
        check_error(lib().kjarni_hip_group_logits_host(
            self._h, _p(ids, _ffi._u32p), _p(mask, _ffi._u32p), _p(type_ids, _ffi._u32p), B, S, fill,
            _p(out, _ffi._f32p)))
        return out

    def _ptr_array(self, ptrs):
        return (C.c_void_p * self.size)(*[C.c_void_p(int(p)) for p in ptrs])

    def embed_allgather(self, ids_ptrs, mask_ptrs, batch_total: int, seq: int, out_ptrs, type_ptrs=None,
                        pooling: int = POOL_MEAN, normalize: bool = True, fill: int = MASK_AUTO):
        """Raw device addresses, one per replica: its row block of ids / mask (/ types) and a full
        [batch_total, hidden] output buffer on its device.  Returns when every buffer holds every row."""
        check_error(lib().kjarni_hip_group_embed_allgather(
            self._h, self._ptr_array(ids_ptrs), self._ptr_array(mask_ptrs),
            self._ptr_array(type_ptrs) if type_ptrs else None, batch_total, seq, pooling, int(bool(normalize)), fill,
            self._ptr_array(out_ptrs)))

    def logits_allgather(self, ids_ptrs, mask_ptrs, type_ptrs, batch_total: int, seq: int, out_ptrs,
                         fill: int = MASK_AUTO):
        check_error(lib().kjarni_hip_group_logits_allgather(
            self._h, self._ptr_array(ids_ptrs), self._ptr_array(mask_ptrs),
            self._ptr_array(type_ptrs) if type_ptrs else None, batch_total, seq, fill, self._ptr_array(out_ptrs)))


def cosine_search(queries, corpus, k: int, mode: int = COSINE_VECTOR_STORE, device: int = 0):
    """Brute-force cosine top-k on the GPU (VectorStore::search / Segment::search_vectors).

    Returns (idx int64 [nq, k'], score f32 [nq, k']) with k' = min(k, n_docs)."""
    queries = np.ascontiguousarray(queries, np.float32)
    corpus = np.ascontiguousarray(corpus, np.float32)
    if queries.ndim == 1:
        queries = queries[None, :]
    nq, dim = queries.shape
    n = corpus.shape[0]
    if n == 0 or k <= 0 or corpus.shape[1] != dim:
        return np.zeros((nq, 0), np.int64), np.zeros((nq, 0), np.float32)
    idx = np.empty((nq, k), np.int64)
    sc = np.empty((nq, k), np.float32)
    hits = C.c_int64(0)
    check_error(lib().kjarni_hip_cosine_search_host(
        device, _p(queries, _ffi._f32p), nq, _p(corpus, _ffi._f32p), n, dim, mode, k,
        _p(idx, _ffi._i64p), _p(sc, _ffi._f32p), C.byref(hits)))
    return idx[:, :hits.value], sc[:, :hits.value]
