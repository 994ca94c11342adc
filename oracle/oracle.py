"""ctypes loader for the CPU oracle (oracle/kjarni_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (kjarni_amd/) never imports
this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkjarni_oracle.so")

ACT_GELU, ACT_GELU_NEW, ACT_RELU, ACT_TANH, ACT_NONE = 0, 1, 2, 3, 4
MASK_ALLOC = np.float32(-1e9)      # utils/masks.rs:4 (alloc path)
MASK_NOALLOC = np.float32(-np.inf)  # encoder_self_attention.rs:319 (no-alloc path)

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_i64p = C.POINTER(C.c_int64)


def build(force: bool = False) -> str:
    """Compile the oracle shared library with gcc (idempotent)."""
    src = os.path.join(_HERE, "kjarni_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class KoLayer(C.Structure):
    _fields_ = [(n, _f32p) for n in (
        "wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo", "ln1_g", "ln1_b",
        "w1", "b1", "w2", "b2", "ln2_g", "ln2_b", "wg")]


class KoModel(C.Structure):
    _fields_ = [
        ("hidden", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32), ("inter", C.c_int32),
        ("vocab", C.c_int32), ("max_pos", C.c_int32), ("type_vocab", C.c_int32),
        ("pos_offset", C.c_int32), ("act", C.c_int32), ("prenorm", C.c_int32),
        ("scale_embeddings", C.c_int32), ("scale_qk", C.c_int32), ("eps", C.c_float),
        ("blocked_gemm", C.c_int32),
        ("word", _f32p), ("pos", _f32p), ("type", _f32p), ("emb_ln_g", _f32p), ("emb_ln_b", _f32p),
        ("L", C.POINTER(KoLayer)),
        ("rope_cos", _f32p), ("rope_sin", _f32p), ("rope_len", C.c_int32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.ko_gelu.restype = C.c_float
        L.ko_gelu.argtypes = [C.c_float]
        L.ko_gelu_new.restype = C.c_float
        L.ko_gelu_new.argtypes = [C.c_float]
        L.ko_relu.restype = C.c_float
        L.ko_relu.argtypes = [C.c_float]
        L.ko_softmax_row.argtypes = [_f32p, C.c_int]
        L.ko_activation_array.argtypes = [_f32p, C.c_int64, C.c_int]
        L.ko_layer_norm.argtypes = [_f32p, _f32p, _f32p, C.c_float, C.c_int64, C.c_int, _f32p]
        for fn in (L.ko_linear, L.ko_linear_blocked):
            fn.argtypes = [_f32p, _f32p, _f32p, C.c_int64, C.c_int, C.c_int, _f32p]
        L.ko_embed.restype = C.c_int
        L.ko_embed.argtypes = [_u32p, _u32p, _f32p, _f32p, _f32p, C.c_int64, C.c_int, C.c_int,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
        L.ko_attention.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int64, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_float, _f32p]
        L.ko_encoder_layer.argtypes = [C.POINTER(KoModel), C.POINTER(KoLayer), _f32p, _f32p, _f32p,
                                       C.c_int64, C.c_int, C.c_float]
        L.ko_encoder_forward.restype = C.c_int
        L.ko_encoder_forward.argtypes = [C.POINTER(KoModel), _u32p, _u32p, _u32p, C.c_int64,
                                         C.c_int, C.c_float, _f32p]
        L.ko_embed_batch.restype = C.c_int
        L.ko_embed_batch.argtypes = [C.POINTER(KoModel), _u32p, _u32p, C.c_int64, C.c_int,
                                     C.c_float, _f32p]
        for fn in (L.ko_mean_pool, L.ko_max_pool, L.ko_last_token_pool):
            fn.argtypes = [_f32p, _f32p, C.c_int64, C.c_int, C.c_int, _f32p]
        L.ko_cls_pool.argtypes = [_f32p, C.c_int64, C.c_int, C.c_int, _f32p]
        L.ko_l2_normalize.argtypes = [_f32p, C.c_int64, C.c_int]
        L.ko_cls_head.argtypes = [_f32p, C.c_int64, C.c_int, C.c_int, _f32p, _f32p, C.c_int,
                                  _f32p, _f32p, C.c_int, _f32p]
        L.ko_cosine_ks.restype = C.c_float
        L.ko_cosine_ks.argtypes = [_f32p, _f32p, C.c_int]
        L.ko_cosine_kr.restype = C.c_float
        L.ko_cosine_kr.argtypes = [_f32p, _f32p, C.c_int, C.c_float]
        L.ko_cosine_k.restype = C.c_float
        L.ko_cosine_k.argtypes = [_f32p, _f32p, C.c_int]
        L.ko_cosine_scan.argtypes = [_f32p, _f32p, C.c_int64, C.c_int, C.c_int, _f32p]
        L.ko_search.restype = C.c_int64
        L.ko_search.argtypes = [_f32p, _f32p, C.c_int64, C.c_int, C.c_int, C.c_int64, _i64p, _f32p]
        L.ko_num_threads.restype = C.c_int
        L.ko_set_num_threads.argtypes = [C.c_int]
        # The checker runs tiny problems: an OpenMP team as wide as a 256-thread host spends its time
        # in barriers.  Callers that time the oracle (bench.py's cpu_baseline) set the width themselves.
        L.ko_set_num_threads(max(1, min(16, os.cpu_count() or 1)))
        _lib = L
    return _lib


def _f(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(_f32p)


def _u(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u32p)


def f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


# ----------------------------------------------------------------------------- primitives


def gelu(x: float) -> float:
    return float(lib().ko_gelu(float(x)))


def gelu_new(x: float) -> float:
    return float(lib().ko_gelu_new(float(x)))


ACT_GELU, ACT_GELU_NEW, ACT_RELU, ACT_TANH = 0, 1, 2, 3


def activation(x, act: int) -> np.ndarray:
    """gelu_scalar / gelu_new_scalar / relu / tanhf applied element by element (activations.rs:56-71)."""
    y = f32(x).copy()
    lib().ko_activation_array(_f(y), y.size, int(act))
    return y


def softmax_rows(x: np.ndarray) -> np.ndarray:
    x = f32(x).copy()
    rows = x.reshape(-1, x.shape[-1])
    for r in rows:
        lib().ko_softmax_row(_f(r), r.shape[0])
    return x


def layer_norm(x, gamma, beta, eps) -> np.ndarray:
    x = f32(x)
    out = np.empty_like(x)
    lib().ko_layer_norm(_f(x), _f(f32(gamma)), _f(f32(beta)), float(eps), x.size // x.shape[-1],
                        x.shape[-1], _f(out))
    return out


def linear(x, w, b=None, blocked: bool = False) -> np.ndarray:
    x, w = f32(x), f32(w)
    b = None if b is None else f32(b)
    m = x.size // x.shape[-1]
    n, k = w.shape
    assert x.shape[-1] == k
    out = np.empty(x.shape[:-1] + (n,), dtype=np.float32)
    fn = lib().ko_linear_blocked if blocked else lib().ko_linear
    fn(_f(x), _f(w), _f(b), m, k, n, _f(out))
    return out


def embed(ids, type_ids, word, pos, type_emb, pos_offset=0, scale=False) -> np.ndarray:
    ids = u32(ids)
    type_ids = None if type_ids is None else u32(type_ids)
    word = f32(word)
    pos = None if pos is None else f32(pos)
    type_emb = None if type_emb is None else f32(type_emb)
    B, S = ids.shape
    H = word.shape[1]
    out = np.empty((B, S, H), dtype=np.float32)
    rc = lib().ko_embed(_u(ids), _u(type_ids), _f(word), _f(pos), _f(type_emb), B, S, H,
                        word.shape[0], 0 if pos is None else pos.shape[0],
                        0 if type_emb is None else type_emb.shape[0], pos_offset, int(scale),
                        _f(out))
    if rc != 0:
        raise ValueError("token type id out of range")
    return out


def attention(q, k, v, mask, heads, position_bias=None, scale_qk=True,
              mask_value=MASK_ALLOC) -> np.ndarray:
    q, k, v = f32(q), f32(k), f32(v)
    B, S, H = q.shape
    mask = None if mask is None else f32(mask)
    pb = None if position_bias is None else f32(position_bias)
    ctx = np.empty_like(q)
    lib().ko_attention(_f(q), _f(k), _f(v), _f(mask), _f(pb), B, S, heads, H // heads,
                       int(scale_qk), float(mask_value), _f(ctx))
    return ctx


def mean_pool(hidden, mask) -> np.ndarray:
    hidden, mask = f32(hidden), f32(mask)
    B, S, H = hidden.shape
    out = np.empty((B, H), dtype=np.float32)
    lib().ko_mean_pool(_f(hidden), _f(mask), B, S, H, _f(out))
    return out


def max_pool(hidden, mask) -> np.ndarray:
    hidden, mask = f32(hidden), f32(mask)
    B, S, H = hidden.shape
    out = np.empty((B, H), dtype=np.float32)
    lib().ko_max_pool(_f(hidden), _f(mask), B, S, H, _f(out))
    return out


def last_token_pool(hidden, mask) -> np.ndarray:
    hidden, mask = f32(hidden), f32(mask)
    B, S, H = hidden.shape
    out = np.empty((B, H), dtype=np.float32)
    lib().ko_last_token_pool(_f(hidden), _f(mask), B, S, H, _f(out))
    return out


def cls_pool(hidden) -> np.ndarray:
    hidden = f32(hidden)
    B, S, H = hidden.shape
    out = np.empty((B, H), dtype=np.float32)
    lib().ko_cls_pool(_f(hidden), B, S, H, _f(out))
    return out


def l2_normalize(x) -> np.ndarray:
    x = f32(x).copy()
    lib().ko_l2_normalize(_f(x), x.shape[0], x.shape[1])
    return x


def cls_head(hidden, w_dense, b_dense, dense_act, w_cls, b_cls) -> np.ndarray:
    hidden = f32(hidden)
    B, S, H = hidden.shape
    w_cls = f32(w_cls)
    n = w_cls.shape[0]
    out = np.empty((B, n), dtype=np.float32)
    lib().ko_cls_head(_f(hidden), B, S, H, _f(None if w_dense is None else f32(w_dense)),
                      _f(None if b_dense is None else f32(b_dense)), int(dense_act), _f(w_cls),
                      _f(None if b_cls is None else f32(b_cls)), n, _f(out))
    return out


def cosine_ks(a, b) -> float:
    a, b = f32(a), f32(b)
    if a.shape != b.shape:
        return 0.0  # vector.rs:132-134
    return float(lib().ko_cosine_ks(_f(a), _f(b), a.shape[0]))


def cosine_k(a, b) -> float:
    a, b = f32(a), f32(b)
    return float(lib().ko_cosine_k(_f(a), _f(b), a.shape[0]))


def cosine_scan(query, corpus, mode: int = 0) -> np.ndarray:
    query, corpus = f32(query), f32(corpus)
    n, d = corpus.shape
    out = np.empty(n, dtype=np.float32)
    lib().ko_cosine_scan(_f(query), _f(corpus), n, d, mode, _f(out))
    return out


def search(query, corpus, limit: int, mode: int = 0):
    """Top-`limit` (idx, score), score descending, ties by ascending index."""
    query, corpus = f32(query), f32(corpus)
    if corpus.size == 0 or corpus.shape[1] != query.shape[0]:
        return np.zeros(0, np.int64), np.zeros(0, np.float32)
    n, d = corpus.shape
    k = max(0, min(int(limit), n))
    idx = np.empty(max(k, 1), dtype=np.int64)
    sc = np.empty(max(k, 1), dtype=np.float32)
    got = lib().ko_search(_f(query), _f(corpus), n, d, mode, k, idx.ctypes.data_as(_i64p), _f(sc))
    return idx[:got].copy(), sc[:got].copy()


# ----------------------------------------------------------------------------- model


class OracleModel:
    """ko_model built from a {hf_tensor_name: ndarray} dict + config dict.

    Weight-name layouts follow kjarni-models/src/models/sentence_encoder/
    configs.rs:218-366 (BERT: plain and "bert."-prefixed), :393-470 (MPNet: positions
    start at 2, tanh GELU, no token types, the relative attention bias is not read) and
    :638-687 (DistilBERT); sequence_classifier/configs.rs:149-280 (RoBERTa: BERT's layout
    under "roberta.", positions start at 2); sentence_encoder/configs.rs:140-275 (Nomic: `model_type` "nomic_bert",
    fused `attn.Wqkv`, no biases, no position table but RoPE with `rotary_emb_base`, SwiGLU with gate = `mlp.fc11` and
    up = `mlp.fc12`, `emb_ln`; BertConfig's serde aliases n_embd / n_layer / n_head / n_inner / layer_norm_epsilon)."""

    def __init__(self, tensors: Dict[str, np.ndarray], config: dict, blocked_gemm: bool = False):
        self.config = config
        self.t = {k: f32(v) for k, v in tensors.items()}
        t = self.t
        mt = config.get("model_type", "bert")
        self.keep = []
        H = config.get("hidden_size", config.get("dim", config.get("n_embd")))
        Lc = config.get("num_hidden_layers", config.get("n_layers", config.get("n_layer")))
        heads = config.get("num_attention_heads", config.get("n_heads", config.get("n_head")))
        inter = config.get("intermediate_size", config.get("hidden_dim", config.get("n_inner"))) or 4 * H
        act_s = config.get("hidden_act", config.get("activation", "gelu"))
        # configs.rs:194-200: "gelu" -> erf GELU, "gelu_new" -> tanh, "relu"
        act = {"gelu": ACT_GELU, "gelu_new": ACT_GELU_NEW, "relu": ACT_RELU}.get(act_s, ACT_GELU)
        layers = (KoLayer * Lc)()
        emb_ln = None
        rope = None
        if mt == "nomic_bert":
            pre = ""
            emb = "embeddings."
            emb_ln = ("emb_ln.weight", "emb_ln.bias")

            def names(i):
                l = f"encoder.layers.{i}."
                w = t[l + "attn.Wqkv.weight"]  # transformer_encoder.rs:75-83: rows [0,H) / [H,2H) / [2H,3H)
                parts = [f32(w[p * H:(p + 1) * H]) for p in range(3)]
                self.keep.extend(parts)
                return dict(wq=parts[0], wk=parts[1], wv=parts[2], bq=None, bk=None, bv=None,
                            wo=l + "attn.out_proj.weight", bo=None, ln1_g=l + "norm1.weight", ln1_b=l + "norm1.bias",
                            wg=l + "mlp.fc11.weight", w1=l + "mlp.fc12.weight", b1=None, w2=l + "mlp.fc2.weight", b2=None,
                            ln2_g=l + "norm2.weight", ln2_b=l + "norm2.bias")
            type_name = emb + "token_type_embeddings.weight"
            eps = config.get("layer_norm_eps", config.get("layer_norm_epsilon"))
            if config.get("rotary_emb_fraction") is not None or config.get("rotary_emb_base") is not None:
                rope = float(config.get("rotary_emb_base") or 10000.0)  # configs.rs:175-183
        elif mt == "distilbert":
            pre = "distilbert." if "distilbert.embeddings.word_embeddings.weight" in t else ""
            emb = pre + "embeddings."
            names = lambda i: dict(
                wq=f"{pre}transformer.layer.{i}.attention.q_lin.weight",
                bq=f"{pre}transformer.layer.{i}.attention.q_lin.bias",
                wk=f"{pre}transformer.layer.{i}.attention.k_lin.weight",
                bk=f"{pre}transformer.layer.{i}.attention.k_lin.bias",
                wv=f"{pre}transformer.layer.{i}.attention.v_lin.weight",
                bv=f"{pre}transformer.layer.{i}.attention.v_lin.bias",
                wo=f"{pre}transformer.layer.{i}.attention.out_lin.weight",
                bo=f"{pre}transformer.layer.{i}.attention.out_lin.bias",
                ln1_g=f"{pre}transformer.layer.{i}.sa_layer_norm.weight",
                ln1_b=f"{pre}transformer.layer.{i}.sa_layer_norm.bias",
                w1=f"{pre}transformer.layer.{i}.ffn.lin1.weight",
                b1=f"{pre}transformer.layer.{i}.ffn.lin1.bias",
                w2=f"{pre}transformer.layer.{i}.ffn.lin2.weight",
                b2=f"{pre}transformer.layer.{i}.ffn.lin2.bias",
                ln2_g=f"{pre}transformer.layer.{i}.output_layer_norm.weight",
                ln2_b=f"{pre}transformer.layer.{i}.output_layer_norm.bias")
            type_name = None
            eps = 1e-12
        elif mt == "mpnet":
            pre = ""
            emb = "embeddings."
            names = lambda i: dict(
                wq=f"encoder.layer.{i}.attention.attn.q.weight", bq=f"encoder.layer.{i}.attention.attn.q.bias",
                wk=f"encoder.layer.{i}.attention.attn.k.weight", bk=f"encoder.layer.{i}.attention.attn.k.bias",
                wv=f"encoder.layer.{i}.attention.attn.v.weight", bv=f"encoder.layer.{i}.attention.attn.v.bias",
                wo=f"encoder.layer.{i}.attention.attn.o.weight", bo=f"encoder.layer.{i}.attention.attn.o.bias",
                ln1_g=f"encoder.layer.{i}.attention.LayerNorm.weight", ln1_b=f"encoder.layer.{i}.attention.LayerNorm.bias",
                w1=f"encoder.layer.{i}.intermediate.dense.weight", b1=f"encoder.layer.{i}.intermediate.dense.bias",
                w2=f"encoder.layer.{i}.output.dense.weight", b2=f"encoder.layer.{i}.output.dense.bias",
                ln2_g=f"encoder.layer.{i}.output.LayerNorm.weight", ln2_b=f"encoder.layer.{i}.output.LayerNorm.bias")
            type_name = None
            eps = config.get("layer_norm_eps", 1e-5)
            act = ACT_GELU_NEW  # configs.rs:410
        else:
            if mt in ("roberta", "distilroberta"):
                pre = "roberta." if "roberta.embeddings.word_embeddings.weight" in t else ""
            else:
                pre = "bert." if "bert.embeddings.word_embeddings.weight" in t else ""
            emb = pre + "embeddings."
            names = lambda i: dict(
                wq=f"{pre}encoder.layer.{i}.attention.self.query.weight",
                bq=f"{pre}encoder.layer.{i}.attention.self.query.bias",
                wk=f"{pre}encoder.layer.{i}.attention.self.key.weight",
                bk=f"{pre}encoder.layer.{i}.attention.self.key.bias",
                wv=f"{pre}encoder.layer.{i}.attention.self.value.weight",
                bv=f"{pre}encoder.layer.{i}.attention.self.value.bias",
                wo=f"{pre}encoder.layer.{i}.attention.output.dense.weight",
                bo=f"{pre}encoder.layer.{i}.attention.output.dense.bias",
                ln1_g=f"{pre}encoder.layer.{i}.attention.output.LayerNorm.weight",
                ln1_b=f"{pre}encoder.layer.{i}.attention.output.LayerNorm.bias",
                w1=f"{pre}encoder.layer.{i}.intermediate.dense.weight",
                b1=f"{pre}encoder.layer.{i}.intermediate.dense.bias",
                w2=f"{pre}encoder.layer.{i}.output.dense.weight",
                b2=f"{pre}encoder.layer.{i}.output.dense.bias",
                ln2_g=f"{pre}encoder.layer.{i}.output.LayerNorm.weight",
                ln2_b=f"{pre}encoder.layer.{i}.output.LayerNorm.bias")
            type_name = emb + "token_type_embeddings.weight"
            eps = config.get("layer_norm_eps", 1e-5 if mt in ("roberta", "distilroberta") else 1e-12)
        for i in range(Lc):
            for field, name in names(i).items():
                setattr(layers[i], field, _f(t[name] if isinstance(name, str) else name))
        self.layers = layers
        m = KoModel()
        m.hidden, m.layers, m.heads, m.inter = H, Lc, heads, inter
        m.vocab = t[emb + "word_embeddings.weight"].shape[0]
        has_pos = emb + "position_embeddings.weight" in t
        # get_max_seq_len (configs.rs:145-149) when there is no position table
        m.max_pos = (t[emb + "position_embeddings.weight"].shape[0] if has_pos else
                     config.get("n_positions") or config.get("max_position_embeddings") or 512)
        m.type_vocab = t[type_name].shape[0] if type_name and type_name in t else 0
        m.pos_offset = 2 if mt in ("roberta", "distilroberta", "mpnet") else 0  # extra_pos_embeddings
        m.act, m.prenorm, m.scale_embeddings, m.scale_qk = act, 0, 0, 1
        m.eps = eps
        m.blocked_gemm = int(blocked_gemm)
        m.word = _f(t[emb + "word_embeddings.weight"])
        m.pos = _f(t[emb + "position_embeddings.weight"]) if has_pos else None
        m.type = _f(t[type_name]) if m.type_vocab else None
        m.emb_ln_g = _f(t[emb_ln[0] if emb_ln else emb + "LayerNorm.weight"])
        m.emb_ln_b = _f(t[emb_ln[1] if emb_ln else emb + "LayerNorm.bias"])
        m.L = C.cast(layers, C.POINTER(KoLayer))
        if rope is not None:  # RoPE::new(head_dim, max_seq_len, theta), transformer_encoder.rs:222-227
            from oracle.llm_oracle import rope_tables
            cos, sin = rope_tables(H // heads, int(m.max_pos), rope)
            self.rope_cos, self.rope_sin = f32(cos), f32(sin)
            m.rope_cos, m.rope_sin, m.rope_len = _f(self.rope_cos), _f(self.rope_sin), int(m.max_pos)
        self.m = m
        self.hidden = H
        self.prefix = pre

    def forward(self, ids, mask, type_ids=None, mask_value=MASK_ALLOC) -> np.ndarray:
        ids, mask = u32(ids), u32(mask)
        type_ids = None if type_ids is None else u32(type_ids)
        B, S = ids.shape
        out = np.empty((B, S, self.hidden), dtype=np.float32)
        rc = lib().ko_encoder_forward(C.byref(self.m), _u(ids), _u(mask), _u(type_ids), B, S,
                                      float(mask_value), _f(out))
        if rc != 0:
            raise ValueError("token type id out of range")
        return out

    def embed_batch(self, ids, mask, mask_value=None) -> np.ndarray:
        """encode_batch_flat semantics: mean-pool + L2, always."""
        ids, mask = u32(ids), u32(mask)
        B, S = ids.shape
        if mask_value is None:
            mask_value = strategy_mask_value(B * S)
        out = np.empty((B, self.hidden), dtype=np.float32)
        rc = lib().ko_embed_batch(C.byref(self.m), _u(ids), _u(mask), B, S, float(mask_value),
                                  _f(out))
        assert rc == 0
        return out

    def head_logits(self, hidden) -> np.ndarray:
        """Classification head auto-detected from tensor names
        (cpu/encoder/classifier.rs:103-202)."""
        t = self.t
        for stem in ("classification_head", "classifier"):  # dense + tanh + out_proj (BART / RoBERTa heads)
            if f"{stem}.dense.weight" in t:
                return cls_head(hidden, t[f"{stem}.dense.weight"], t.get(f"{stem}.dense.bias"), ACT_TANH,
                                t[f"{stem}.out_proj.weight"], t.get(f"{stem}.out_proj.bias"))
        if "pre_classifier.weight" in t:
            return cls_head(hidden, t["pre_classifier.weight"], t.get("pre_classifier.bias"),
                            ACT_RELU, t["classifier.weight"], t.get("classifier.bias"))
        if "bert.pooler.dense.weight" in t:
            return cls_head(hidden, t["bert.pooler.dense.weight"], t.get("bert.pooler.dense.bias"),
                            ACT_TANH, t["classifier.weight"], t.get("classifier.bias"))
        return cls_head(hidden, None, None, ACT_NONE, t["classifier.weight"],
                        t.get("classifier.bias"))

    def rerank_scores(self, ids, mask, type_ids) -> np.ndarray:
        """CrossEncoder::predict_pairs (cross_encoder/model.rs:170-240): always the
        alloc path (-1e9 mask), logits column 0."""
        h = self.forward(ids, mask, type_ids, MASK_ALLOC)
        return self.head_logits(h)[:, 0].copy()


def strategy_mask_value(tokens: int):
    """cpu/strategy.rs:43-44: scratch buffers (=> -inf mask) iff tokens <= 1 or >= 1000."""
    return MASK_NOALLOC if (tokens <= 1 or tokens >= 1000) else MASK_ALLOC
