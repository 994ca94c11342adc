"""ctypes loader for oracle/kjarni_cpu_baseline.c -- the timed CPU leg of bench.py.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (see the C file's header): imported by bench.py's
cpu_baseline leg and tests/test_cpu_baseline.py, never by the product."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)


class KbLayer(C.Structure):
    _fields_ = [(n, _f32p) for n in ("wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2",
                                     "ln2_g", "ln2_b")]


class KbModel(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("hidden", "layers", "heads", "inter", "vocab", "max_pos", "type_vocab")] + \
               [("eps", C.c_float)] + [(n, _f32p) for n in ("word", "pos", "type", "emb_ln_g", "emb_ln_b")] + \
               [("L", C.POINTER(KbLayer))]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libkjarni_cpu_baseline.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "libkjarni_cpu_baseline.so"])
        L = C.CDLL(path)
        L.kb_buffers_new.restype = C.c_void_p
        L.kb_buffers_new.argtypes = [C.POINTER(KbModel), C.c_int32, C.c_int32]
        L.kb_buffers_free.argtypes = [C.c_void_p]
        L.kb_set_num_threads.argtypes = [C.c_int]
        L.kb_embed_batch.restype = C.c_int
        L.kb_embed_batch.argtypes = [C.POINTER(KbModel), C.c_void_p, _u32p, _u32p, C.c_int64, C.c_int, C.c_int, _f32p]
        _LIB = L
    return _LIB


def _f(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


class BaselineModel:
    """BERT-layout encoder (plain tensor names, sentence_encoder/configs.rs:218-366) with the
    reference's fused [3H,H] QKV weight (qkv_projection.rs:30-41)."""

    def __init__(self, tensors: Dict[str, np.ndarray], config: dict, max_batch: int = 256, max_seq: int = 128):
        t = {k: np.ascontiguousarray(v, np.float32) for k, v in tensors.items()}
        self.keep = [t]
        H, Lc = config["hidden_size"], config["num_hidden_layers"]
        self.hidden = H
        layers = (KbLayer * Lc)()
        for i in range(Lc):
            p = f"encoder.layer.{i}."
            wqkv = np.ascontiguousarray(np.concatenate(
                [t[p + f"attention.self.{n}.weight"] for n in ("query", "key", "value")], 0))
            bqkv = np.ascontiguousarray(np.concatenate(
                [t[p + f"attention.self.{n}.bias"] for n in ("query", "key", "value")], 0))
            self.keep += [wqkv, bqkv]
            layers[i].wqkv, layers[i].bqkv = _f(wqkv), _f(bqkv)
            for field, name in dict(wo="attention.output.dense.weight", bo="attention.output.dense.bias",
                                    ln1_g="attention.output.LayerNorm.weight", ln1_b="attention.output.LayerNorm.bias",
                                    w1="intermediate.dense.weight", b1="intermediate.dense.bias",
                                    w2="output.dense.weight", b2="output.dense.bias",
                                    ln2_g="output.LayerNorm.weight", ln2_b="output.LayerNorm.bias").items():
                setattr(layers[i], field, _f(t[p + name]))
        self.layers = layers
        m = KbModel()
        m.hidden, m.layers, m.heads, m.inter = H, Lc, config["num_attention_heads"], config["intermediate_size"]
        m.vocab = t["embeddings.word_embeddings.weight"].shape[0]
        m.max_pos = t["embeddings.position_embeddings.weight"].shape[0]
        m.type_vocab = t["embeddings.token_type_embeddings.weight"].shape[0]
        m.eps = float(config.get("layer_norm_eps", 1e-12))
        m.word, m.pos = _f(t["embeddings.word_embeddings.weight"]), _f(t["embeddings.position_embeddings.weight"])
        m.type = _f(t["embeddings.token_type_embeddings.weight"])
        m.emb_ln_g, m.emb_ln_b = _f(t["embeddings.LayerNorm.weight"]), _f(t["embeddings.LayerNorm.bias"])
        m.L = C.cast(layers, C.POINTER(KbLayer))
        self.m = m
        self.buf = lib().kb_buffers_new(C.byref(m), max_batch, max_seq)
        if not self.buf:
            raise MemoryError("kb_buffers_new")

    def __del__(self):
        if getattr(self, "buf", None):
            lib().kb_buffers_free(self.buf)
            self.buf = None

    def embed_batch(self, ids: np.ndarray, mask: np.ndarray, parallel_rowops: bool = False) -> np.ndarray:
        ids = np.ascontiguousarray(ids, np.uint32)
        mask = np.ascontiguousarray(mask, np.uint32)
        B, S = ids.shape
        out = np.empty((B, self.hidden), np.float32)
        rc = lib().kb_embed_batch(C.byref(self.m), self.buf, ids.ctypes.data_as(_u32p), mask.ctypes.data_as(_u32p),
                                  B, S, int(bool(parallel_rowops)), _f(out))
        if rc != 0:
            raise ValueError("batch does not fit the baseline's buffers")
        return out
