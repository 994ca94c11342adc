"""Synthetic model directories (config.json + model.safetensors [+ tokenizer.json])
with seeded random weights, in the HF layouts the reference loads
(kjarni-models/src/models/sentence_encoder/configs.rs:218-366, :638-687;
sequence_classifier/configs.rs:99-143).  Test infrastructure only."""
from __future__ import annotations

import json
import os
from typing import Dict, Tuple

import numpy as np

MINILM = dict(hidden_size=384, num_hidden_layers=6, num_attention_heads=12, intermediate_size=1536,
              vocab_size=30522, max_position_embeddings=512, type_vocab_size=2)


def bert_tensors(cfg: dict, seed: int = 0, prefix: str = "", head: str = "") -> Dict[str, np.ndarray]:
    """head: "" | "cross" (bert.pooler + classifier[1]) | "plain2" (classifier[2] only)."""
    rng = np.random.default_rng(seed)
    H, L, I = cfg["hidden_size"], cfg["num_hidden_layers"], cfg["intermediate_size"]
    V, P, T = cfg["vocab_size"], cfg["max_position_embeddings"], cfg["type_vocab_size"]
    std = cfg.get("init_std", 0.02)

    def w(*shape, s=std):
        return (rng.standard_normal(shape) * s).astype(np.float32)

    def ln_g(n):
        return (1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32)

    t = {}
    e = prefix + "embeddings."
    t[e + "word_embeddings.weight"] = w(V, H)
    t[e + "position_embeddings.weight"] = w(P, H)
    t[e + "token_type_embeddings.weight"] = w(T, H)
    t[e + "LayerNorm.weight"] = ln_g(H)
    t[e + "LayerNorm.bias"] = w(H, s=0.05)
    for i in range(L):
        p = f"{prefix}encoder.layer.{i}."
        for nm in ("query", "key", "value"):
            t[p + f"attention.self.{nm}.weight"] = w(H, H, s=cfg.get("attn_std", std))
            t[p + f"attention.self.{nm}.bias"] = w(H, s=0.05)
        t[p + "attention.output.dense.weight"] = w(H, H)
        t[p + "attention.output.dense.bias"] = w(H, s=0.05)
        t[p + "attention.output.LayerNorm.weight"] = ln_g(H)
        t[p + "attention.output.LayerNorm.bias"] = w(H, s=0.05)
        t[p + "intermediate.dense.weight"] = w(I, H)
        t[p + "intermediate.dense.bias"] = w(I, s=0.05)
        t[p + "output.dense.weight"] = w(H, I)
        t[p + "output.dense.bias"] = w(H, s=0.05)
        t[p + "output.LayerNorm.weight"] = ln_g(H)
        t[p + "output.LayerNorm.bias"] = w(H, s=0.05)
    if head == "cross":
        t["bert.pooler.dense.weight"] = w(H, H, s=0.05)
        t["bert.pooler.dense.bias"] = w(H, s=0.05)
        t["classifier.weight"] = w(1, H, s=0.2)
        t["classifier.bias"] = w(1, s=0.1)
    elif head == "plain2":
        t["classifier.weight"] = w(2, H, s=0.2)
        t["classifier.bias"] = w(2, s=0.1)
    return t


def distilbert_tensors(cfg: dict, seed: int = 0) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    H, L, I = cfg["dim"], cfg["n_layers"], cfg["hidden_dim"]
    V, P = cfg["vocab_size"], cfg["max_position_embeddings"]

    def w(*shape, s=0.02):
        return (rng.standard_normal(shape) * s).astype(np.float32)

    def ln_g(n):
        return (1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32)

    t = {}
    e = "distilbert.embeddings."
    t[e + "word_embeddings.weight"] = w(V, H)
    t[e + "position_embeddings.weight"] = w(P, H)
    t[e + "LayerNorm.weight"] = ln_g(H)
    t[e + "LayerNorm.bias"] = w(H, s=0.05)
    for i in range(L):
        p = f"distilbert.transformer.layer.{i}."
        for nm in ("q_lin", "k_lin", "v_lin", "out_lin"):
            t[p + f"attention.{nm}.weight"] = w(H, H)
            t[p + f"attention.{nm}.bias"] = w(H, s=0.05)
        t[p + "sa_layer_norm.weight"] = ln_g(H)
        t[p + "sa_layer_norm.bias"] = w(H, s=0.05)
        t[p + "ffn.lin1.weight"] = w(I, H)
        t[p + "ffn.lin1.bias"] = w(I, s=0.05)
        t[p + "ffn.lin2.weight"] = w(H, I)
        t[p + "ffn.lin2.bias"] = w(H, s=0.05)
        t[p + "output_layer_norm.weight"] = ln_g(H)
        t[p + "output_layer_norm.bias"] = w(H, s=0.05)
    t["pre_classifier.weight"] = w(H, H, s=0.05)
    t["pre_classifier.bias"] = w(H, s=0.05)
    t["classifier.weight"] = w(2, H, s=0.2)
    t["classifier.bias"] = w(2, s=0.1)
    return t


def write_model_dir(path: str, config: dict, tensors: Dict[str, np.ndarray]) -> str:
    from safetensors.numpy import save_file
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(config, f, indent=1)
    save_file({k: np.ascontiguousarray(v) for k, v in tensors.items()},
              os.path.join(path, "model.safetensors"))
    return path


def minilm_embedder(path: str, seed: int = 0, **over) -> Tuple[dict, Dict[str, np.ndarray]]:
    cfg = dict(MINILM, model_type="bert", hidden_act="gelu", layer_norm_eps=1e-12,
               architectures=["BertModel"])
    cfg.update(over)
    t = bert_tensors(cfg, seed)
    write_model_dir(path, cfg, t)
    return cfg, t


def minilm_cross_encoder(path: str, seed: int = 1, **over) -> Tuple[dict, Dict[str, np.ndarray]]:
    cfg = dict(MINILM, model_type="bert", hidden_act="gelu", layer_norm_eps=1e-12,
               architectures=["BertForSequenceClassification"], id2label={"0": "LABEL_0"},
               label2id={"LABEL_0": 0})
    cfg.update(over)
    t = bert_tensors(cfg, seed, prefix="bert.", head="cross")
    write_model_dir(path, cfg, t)
    return cfg, t


def distilbert_sentiment(path: str, seed: int = 2, **over) -> Tuple[dict, Dict[str, np.ndarray]]:
    cfg = dict(model_type="distilbert", dim=768, n_layers=6, n_heads=12, hidden_dim=3072, vocab_size=30522,
               max_position_embeddings=512, activation="gelu",
               architectures=["DistilBertForSequenceClassification"],
               id2label={"0": "NEGATIVE", "1": "POSITIVE"}, label2id={"NEGATIVE": 0, "POSITIVE": 1})
    cfg.update(over)
    t = distilbert_tensors(cfg, seed)
    write_model_dir(path, cfg, t)
    return cfg, t


def synthetic_ids(n: int, seq: int, vocab: int = 30522, seed: int = 0, ragged: bool = False):
    """[CLS]=101 first, [SEP]=102 last real token, body uniform in
    1000..vocab-1; ragged=True draws lengths in 16..seq and right-pads with id 0 / mask 0."""
    rng = np.random.default_rng(seed)
    ids = rng.integers(1000, vocab, size=(n, seq), dtype=np.int64).astype(np.uint32)
    mask = np.ones((n, seq), np.uint32)
    if ragged:
        lens = rng.integers(min(16, seq), seq + 1, size=n)
    else:
        lens = np.full(n, seq)
    ids[:, 0] = 101
    for i, ln in enumerate(lens):
        ids[i, ln - 1] = 102
        ids[i, ln:] = 0
        mask[i, ln:] = 0
    return ids, mask


def synthetic_pairs(n: int, seq: int, vocab: int = 30522, seed: int = 1, qlen: int = 16):
    """[CLS] q [SEP] d [SEP] with type ids 0/1."""
    rng = np.random.default_rng(seed)
    ids = rng.integers(1000, vocab, size=(n, seq), dtype=np.int64).astype(np.uint32)
    ids[:, 0] = 101
    q_end = min(1 + qlen, seq - 2)
    ids[:, q_end] = 102
    ids[:, seq - 1] = 102
    types = np.zeros((n, seq), np.uint32)
    types[:, q_end + 1:] = 1
    mask = np.ones((n, seq), np.uint32)
    return ids, mask, types


GOLDEN_TOKENIZER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tokenizer_small.json")


def add_tokenizer(model_dir: str) -> str:
    """Drops the golden WordPiece tokenizer.json (tests/golden) into a model directory."""
    import shutil
    dst = os.path.join(model_dir, "tokenizer.json")
    shutil.copyfile(GOLDEN_TOKENIZER, dst)
    return dst
