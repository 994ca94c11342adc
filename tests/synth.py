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


TRAINED_OUTLIER_CHANNELS = 3     # per LayerNorm: gains x5
QK_GROWTH = 1.08
SUBLAYER_OUT = 0.5               # out-proj / FC2 outputs relative to the residual they are added to
TRAINED_MASSIVE_CHANNELS = 2     # shared by every LayerNorm: bias +-4 (the "massive activation" dimensions of trained BERTs)


def trained_bert_tensors(cfg: dict, seed: int = 0, prefix: str = "", head: str = "") -> Dict[str, np.ndarray]:
    """Seeded weights with the STATISTICS of a trained MiniLM-class checkpoint instead of N(0, 0.02) initialisation (under
    which every attention softmax is uniform to three digits and GELU never leaves its linear part): attention logits
    that spread over the keys of a query row with std 3-5 and reach |max| > 10 (peaked softmax), LayerNorm gains
    log-normal over 0.2 ... 2.5 with three x5 outlier channels per LayerNorm, two channels that carry a bias of +-4 through
    every layer, biases O(0.1 ... 1), FC1 pre-activations reaching +-6.  The matrices that read a LayerNorm's output are
    scaled by the second moment that output has by construction (sum of gain^2 + bias^2), so every layer sits in the same
    regime.  `regime_stats` measures what a batch actually sees; tests/golden/make_encoder_fixtures.py stores those
    numbers next to the float64 fixtures.  Stands in for the real checkpoints the reference pins
    (sentence_encoder/tests.rs:411-1184, cross_encoder/tests.rs:38-100), which are not available offline."""
    rng = np.random.default_rng([seed, 0x7261696E])
    H, L, I = cfg["hidden_size"], cfg["num_hidden_layers"], cfg["intermediate_size"]
    V, P, T = cfg["vocab_size"], cfg["max_position_embeddings"], cfg["type_vocab_size"]
    d = H // cfg["num_attention_heads"]
    massive = rng.choice(H, size=min(TRAINED_MASSIVE_CHANNELS, H), replace=False)
    massive_sign = rng.choice([-1.0, 1.0], size=len(massive))

    def w(*shape, s):
        return (rng.standard_normal(shape) * s).astype(np.float32)

    def layer_norm():
        g = np.clip(np.exp(rng.normal(np.log(0.8), 0.5, H)), 0.2, 2.5)
        g[rng.choice(H, size=min(TRAINED_OUTLIER_CHANNELS, H), replace=False)] *= 5.0
        g[massive] = 0.3                 # small gain, large bias: the value is re-created by every LayerNorm, it does not compound
        b = rng.standard_normal(H) * 0.1
        b[massive] += 4.0 * massive_sign
        return g.astype(np.float32), b.astype(np.float32), float(np.sqrt((g * g + b * b).sum()))

    t = {}
    e = prefix + "embeddings."
    t[e + "word_embeddings.weight"] = w(V, H, s=0.05)
    t[e + "position_embeddings.weight"] = w(P, H, s=0.02)
    t[e + "token_type_embeddings.weight"] = w(T, H, s=0.02)
    t[e + "LayerNorm.weight"], t[e + "LayerNorm.bias"], rms = layer_norm()
    for i in range(L):
        p = f"{prefix}encoder.layer.{i}."
        # q, k of std ~ 2 per component: logits = q.k / sqrt(d) then spread with std ~ 4 over the keys
        for nm, target in (("query", 2.0 * QK_GROWTH ** i), ("key", 2.0 * QK_GROWTH ** i), ("value", 1.0)):
            t[p + f"attention.self.{nm}.weight"] = w(H, H, s=target / rms)
            t[p + f"attention.self.{nm}.bias"] = w(H, s=0.1 if nm != "value" else 0.3)
            if nm != "value":   # content-addressed attention: what every token shares does not steer it
                t[p + f"attention.self.{nm}.weight"][:, massive] *= np.float32(0.05)
        t[p + "attention.output.dense.weight"] = w(H, H, s=0.5 * SUBLAYER_OUT / np.sqrt(H))
        t[p + "attention.output.dense.bias"] = w(H, s=0.2)
        t[p + "attention.output.LayerNorm.weight"], t[p + "attention.output.LayerNorm.bias"], rms = layer_norm()
        t[p + "intermediate.dense.weight"] = w(I, H, s=1.7 / rms)
        t[p + "intermediate.dense.bias"] = w(I, s=0.3)
        t[p + "output.dense.weight"] = w(H, I, s=SUBLAYER_OUT / np.sqrt(I))
        t[p + "output.dense.bias"] = w(H, s=0.2)
        t[p + "output.LayerNorm.weight"], t[p + "output.LayerNorm.bias"], rms = layer_norm()
    if head == "cross":
        t["bert.pooler.dense.weight"] = w(H, H, s=0.8 / rms)
        t["bert.pooler.dense.bias"] = w(H, s=0.1)
        t["classifier.weight"] = w(1, H, s=0.3)
        t["classifier.bias"] = w(1, s=0.3)
    elif head == "plain2":
        t["classifier.weight"] = w(2, H, s=0.3)
        t["classifier.bias"] = w(2, s=0.3)
    del d
    return t


def regime_stats(tensors: Dict[str, np.ndarray], cfg: dict, ids: np.ndarray, mask: np.ndarray, prefix: str = "",
                 types: np.ndarray = None) -> Dict[str, float]:
    """A float64 numpy forward of the BERT graph that reports WHICH numerical regime a weight set puts a batch in:
    attention-logit std over the keys of a query row (largest and smallest layer) / max, the mean of the largest softmax probability per query row, the range of FC1
    pre-activations, the largest hidden-state magnitude.  Real tokens only.  Test infrastructure (a third, independent
    statement of the graph besides oracle/ and Hugging Face)."""
    from scipy.special import erf
    H, L, nh = cfg["hidden_size"], cfg["num_hidden_layers"], cfg["num_attention_heads"]
    d, eps = H // nh, cfg.get("layer_norm_eps", 1e-12)
    g = lambda k: tensors[k].astype(np.float64)  # noqa: E731

    def ln(x, pre):
        mu = x.mean(-1, keepdims=True)
        var = ((x - mu) ** 2).mean(-1, keepdims=True)
        return (x - mu) / np.sqrt(var + eps) * g(pre + ".weight") + g(pre + ".bias")

    B, S = ids.shape
    e = prefix + "embeddings."
    ty = np.zeros_like(ids) if types is None else types
    x = g(e + "word_embeddings.weight")[ids] + g(e + "position_embeddings.weight")[:S][None] + g(e + "token_type_embeddings.weight")[ty]
    x = ln(x, e + "LayerNorm")
    real = mask.astype(bool)
    pair = real[:, None, :, None] & real[:, None, None, :]
    st = dict(logit_std=0.0, logit_absmax=0.0, softmax_top_mean=1.0, fc1_min=0.0, fc1_max=0.0, fc1_std=0.0, hidden_absmax=0.0)
    tops, per_layer = [], []
    for i in range(L):
        p = f"{prefix}encoder.layer.{i}."
        lin = lambda a, nm: a @ g(p + nm + ".weight").T + g(p + nm + ".bias")  # noqa: E731
        split = lambda a: a.reshape(B, S, nh, d).transpose(0, 2, 1, 3)  # noqa: E731
        q, k, v = (split(lin(x, "attention.self." + nm)) for nm in ("query", "key", "value"))
        logits = q @ k.transpose(0, 1, 3, 2) / np.sqrt(d)
        sel = logits[np.broadcast_to(pair, logits.shape)]
        cnt = real.sum(-1)[:, None, None, None]
        row_mean = np.where(real[:, None, None, :], logits, 0.0).sum(-1, keepdims=True) / cnt
        within = (logits - row_mean)[np.broadcast_to(pair, logits.shape)]   # spread over the keys of one query row
        per_layer.append((float(within.std()), float(np.abs(sel).max())))
        st["logit_std"] = max(st["logit_std"], float(within.std()))
        st["logit_absmax"] = max(st["logit_absmax"], float(np.abs(sel).max()))
        logits = np.where(real[:, None, None, :], logits, -1e9)
        pr = np.exp(logits - logits.max(-1, keepdims=True))
        pr /= pr.sum(-1, keepdims=True)
        tops.append(float(pr.max(-1)[np.broadcast_to(real[:, None, :], pr.shape[:3])].mean()))
        ctx = (pr @ v).transpose(0, 2, 1, 3).reshape(B, S, H)
        x = ln(lin(ctx, "attention.output.dense") + x, p + "attention.output.LayerNorm")
        a = lin(x, "intermediate.dense")
        ar = a[real]
        st["fc1_min"], st["fc1_max"] = min(st["fc1_min"], float(ar.min())), max(st["fc1_max"], float(ar.max()))
        st["fc1_std"] = max(st["fc1_std"], float(ar.std()))
        a = 0.5 * a * (1.0 + erf(a / np.sqrt(2.0)))
        x = ln(lin(a, "output.dense") + x, p + "output.LayerNorm")
        st["hidden_absmax"] = max(st["hidden_absmax"], float(np.abs(x[real]).max()))
    st["softmax_top_mean"] = float(np.mean(tops))
    st["logit_std_min_layer"] = min(a for a, _ in per_layer)
    st["last_hidden"] = x
    return st


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


def shard_model_dir(path: str, n_shards: int = 3) -> Dict[str, str]:
    """Rewrites <path>/model.safetensors as model-0000i-of-0000n.safetensors + model.safetensors.index.json, the layout of
    the registry's larger checkpoints (weights/safetensors_loader.rs:84-129).  Returns the weight_map."""
    from safetensors import safe_open
    from safetensors.torch import save_file
    single = os.path.join(path, "model.safetensors")
    with safe_open(single, framework="pt") as f:
        names = list(f.keys())
        tensors = {k: f.get_tensor(k) for k in names}
    weight_map, total = {}, 0
    for i in range(n_shards):
        part = names[i::n_shards]   # interleaved, so neighbouring tensors of a layer live in different files
        fname = f"model-{i + 1:05d}-of-{n_shards:05d}.safetensors"
        save_file({k: tensors[k].contiguous() for k in part}, os.path.join(path, fname))
        for k in part:
            weight_map[k] = fname
            total += tensors[k].numel() * tensors[k].element_size()
    with open(os.path.join(path, "model.safetensors.index.json"), "w") as f:
        json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=1)
    os.remove(single)
    return weight_map


def minilm_embedder(path: str, seed: int = 0, family: str = "init", **over) -> Tuple[dict, Dict[str, np.ndarray]]:
    """family: "init" = N(0, 0.02) initialisation; "trained" = trained-checkpoint statistics (trained_bert_tensors)."""
    cfg = dict(MINILM, model_type="bert", hidden_act="gelu", layer_norm_eps=1e-12,
               architectures=["BertModel"])
    cfg.update(over)
    t = trained_bert_tensors(cfg, seed) if family == "trained" else bert_tensors(cfg, seed)
    write_model_dir(path, cfg, t)
    return cfg, t


def minilm_cross_encoder(path: str, seed: int = 1, family: str = "init", **over) -> Tuple[dict, Dict[str, np.ndarray]]:
    cfg = dict(MINILM, model_type="bert", hidden_act="gelu", layer_norm_eps=1e-12,
               architectures=["BertForSequenceClassification"], id2label={"0": "LABEL_0"},
               label2id={"LABEL_0": 0})
    cfg.update(over)
    make = trained_bert_tensors if family == "trained" else bert_tensors
    t = make(cfg, seed, prefix="bert.", head="cross")
    write_model_dir(path, cfg, t)
    return cfg, t


def trained_distilbert_tensors(cfg: dict, seed: int = 0) -> Dict[str, np.ndarray]:
    """trained_bert_tensors in DistilBERT's layout (no token types; q_lin / k_lin / v_lin / out_lin, sa_layer_norm, ffn.lin1 /
    lin2, output_layer_norm; pre_classifier + classifier head)."""
    bert_cfg = dict(hidden_size=cfg["dim"], num_hidden_layers=cfg["n_layers"], num_attention_heads=cfg["n_heads"],
                    intermediate_size=cfg["hidden_dim"], vocab_size=cfg["vocab_size"],
                    max_position_embeddings=cfg["max_position_embeddings"], type_vocab_size=1)
    b = trained_bert_tensors(bert_cfg, seed)
    rng = np.random.default_rng([seed, 0x64697374])
    H = cfg["dim"]
    t = {}
    e = "distilbert.embeddings."
    t[e + "word_embeddings.weight"] = b["embeddings.word_embeddings.weight"]
    t[e + "position_embeddings.weight"] = b["embeddings.position_embeddings.weight"]
    t[e + "LayerNorm.weight"], t[e + "LayerNorm.bias"] = b["embeddings.LayerNorm.weight"], b["embeddings.LayerNorm.bias"]
    ren = {"attention.self.query": "attention.q_lin", "attention.self.key": "attention.k_lin", "attention.self.value": "attention.v_lin",
           "attention.output.dense": "attention.out_lin", "attention.output.LayerNorm": "sa_layer_norm",
           "intermediate.dense": "ffn.lin1", "output.dense": "ffn.lin2", "output.LayerNorm": "output_layer_norm"}
    for i in range(cfg["n_layers"]):
        for old, new in ren.items():
            for part in ("weight", "bias"):
                t[f"distilbert.transformer.layer.{i}.{new}.{part}"] = b[f"encoder.layer.{i}.{old}.{part}"]
    t["pre_classifier.weight"] = (rng.standard_normal((H, H)) * (0.8 / np.sqrt(H))).astype(np.float32)
    t["pre_classifier.bias"] = (rng.standard_normal(H) * 0.1).astype(np.float32)
    t["classifier.weight"] = (rng.standard_normal((2, H)) * 0.3).astype(np.float32)
    t["classifier.bias"] = (rng.standard_normal(2) * 0.3).astype(np.float32)
    return t


def distilbert_sentiment(path: str, seed: int = 2, family: str = "init", **over) -> Tuple[dict, Dict[str, np.ndarray]]:
    cfg = dict(model_type="distilbert", dim=768, n_layers=6, n_heads=12, hidden_dim=3072, vocab_size=30522,
               max_position_embeddings=512, activation="gelu",
               architectures=["DistilBertForSequenceClassification"],
               id2label={"0": "NEGATIVE", "1": "POSITIVE"}, label2id={"NEGATIVE": 0, "POSITIVE": 1})
    cfg.update(over)
    t = trained_distilbert_tensors(cfg, seed) if family == "trained" else distilbert_tensors(cfg, seed)
    write_model_dir(path, cfg, t)
    return cfg, t


def roberta_classifier(path: str, seed: int = 3, labels=("negative", "neutral", "positive"), family: str = "init",
                       **over) -> Tuple[dict, Dict[str, np.ndarray]]:
    """RobertaForSequenceClassification layout (sequence_classifier/configs.rs:149-280): "roberta."-prefixed BERT layers,
    514 positions (offset 2), one token type, classifier.dense + tanh + classifier.out_proj."""
    cfg = dict(model_type="roberta", hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256,
               hidden_act="gelu", max_position_embeddings=514, vocab_size=640, layer_norm_eps=1e-5, type_vocab_size=1,
               architectures=["RobertaForSequenceClassification"], id2label={str(i): l for i, l in enumerate(labels)},
               label2id={l: i for i, l in enumerate(labels)})
    cfg.update(over)
    t = (trained_bert_tensors if family == "trained" else bert_tensors)(cfg, seed, prefix="roberta.")
    rng = np.random.default_rng(seed + 100)
    H = cfg["hidden_size"]
    t["classifier.dense.weight"] = (rng.standard_normal((H, H)) * 0.05).astype(np.float32)
    t["classifier.dense.bias"] = (rng.standard_normal(H) * 0.05).astype(np.float32)
    t["classifier.out_proj.weight"] = (rng.standard_normal((len(labels), H)) * 0.2).astype(np.float32)
    t["classifier.out_proj.bias"] = (rng.standard_normal(len(labels)) * 0.1).astype(np.float32)
    write_model_dir(path, cfg, t)
    return cfg, t


def mpnet_embedder(path: str, seed: int = 4, family: str = "init", **over) -> Tuple[dict, Dict[str, np.ndarray]]:
    """MPNet layout (sentence_encoder/configs.rs:393-470), including the relative-attention-bias tensor of real
    checkpoints, which the reference never reads.  family "trained": trained_bert_tensors under MPNet's names."""
    cfg = dict(model_type="mpnet", hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256,
               hidden_act="gelu", max_position_embeddings=514, vocab_size=30527, layer_norm_eps=1e-5,
               architectures=["MPNetModel"], relative_attention_num_buckets=32)
    cfg.update(over)
    rng = np.random.default_rng(seed)
    if family == "trained":
        b = trained_bert_tensors(dict(cfg, type_vocab_size=1), seed)
        H = cfg["hidden_size"]
        t = {"embeddings.word_embeddings.weight": b["embeddings.word_embeddings.weight"],
             "embeddings.position_embeddings.weight": b["embeddings.position_embeddings.weight"],
             "embeddings.LayerNorm.weight": b["embeddings.LayerNorm.weight"], "embeddings.LayerNorm.bias": b["embeddings.LayerNorm.bias"],
             "encoder.relative_attention_bias.weight": (rng.standard_normal((32, cfg["num_attention_heads"])) * 0.5).astype(np.float32),
             "pooler.dense.weight": (rng.standard_normal((H, H)) * 0.02).astype(np.float32),
             "pooler.dense.bias": (rng.standard_normal(H) * 0.05).astype(np.float32)}
        ren = {"attention.self.query": "attention.attn.q", "attention.self.key": "attention.attn.k", "attention.self.value": "attention.attn.v",
               "attention.output.dense": "attention.attn.o", "attention.output.LayerNorm": "attention.LayerNorm",
               "intermediate.dense": "intermediate.dense", "output.dense": "output.dense", "output.LayerNorm": "output.LayerNorm"}
        for i in range(cfg["num_hidden_layers"]):
            for old, new in ren.items():
                for part in ("weight", "bias"):
                    t[f"encoder.layer.{i}.{new}.{part}"] = b[f"encoder.layer.{i}.{old}.{part}"]
        write_model_dir(path, cfg, t)
        return cfg, t
    H, L, I, V, P = cfg["hidden_size"], cfg["num_hidden_layers"], cfg["intermediate_size"], cfg["vocab_size"], cfg["max_position_embeddings"]
    w = lambda *shape, s=0.02: (rng.standard_normal(shape) * s).astype(np.float32)  # noqa: E731
    g = lambda n: (1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32)  # noqa: E731
    t = {"embeddings.word_embeddings.weight": w(V, H), "embeddings.position_embeddings.weight": w(P, H),
         "embeddings.LayerNorm.weight": g(H), "embeddings.LayerNorm.bias": w(H, s=0.05),
         "encoder.relative_attention_bias.weight": w(32, cfg["num_attention_heads"], s=0.5),
         "pooler.dense.weight": w(H, H), "pooler.dense.bias": w(H, s=0.05)}
    for i in range(L):
        p = f"encoder.layer.{i}."
        for nm in ("q", "k", "v", "o"):
            t[p + f"attention.attn.{nm}.weight"] = w(H, H)
            t[p + f"attention.attn.{nm}.bias"] = w(H, s=0.05)
        t[p + "attention.LayerNorm.weight"], t[p + "attention.LayerNorm.bias"] = g(H), w(H, s=0.05)
        t[p + "intermediate.dense.weight"], t[p + "intermediate.dense.bias"] = w(I, H), w(I, s=0.05)
        t[p + "output.dense.weight"], t[p + "output.dense.bias"] = w(H, I), w(H, s=0.05)
        t[p + "output.LayerNorm.weight"], t[p + "output.LayerNorm.bias"] = g(H), w(H, s=0.05)
    write_model_dir(path, cfg, t)
    return cfg, t


def nomic_embedder(path: str, seed: int = 6, **over) -> Tuple[dict, Dict[str, np.ndarray]]:
    """nomic-embed-text layout (sentence_encoder/configs.rs:218-275): config.json with the GPT-style key names, fused
    Wqkv, no biases, SwiGLU (fc11 = gate, fc12 = up), RoPE instead of a position table."""
    cfg = dict(model_type="nomic_bert", n_embd=128, n_layer=2, n_head=4, n_inner=256, activation_function="swiglu",
               n_positions=256, vocab_size=30528, layer_norm_epsilon=1e-12, type_vocab_size=2, rotary_emb_fraction=1.0,
               rotary_emb_base=1000, qkv_proj_bias=False, mlp_fc1_bias=False, mlp_fc2_bias=False,
               architectures=["NomicBertModel"])
    cfg.update(over)
    rng = np.random.default_rng(seed)
    H, L, I, V = cfg["n_embd"], cfg["n_layer"], cfg["n_inner"], cfg["vocab_size"]
    w = lambda *shape, s=0.02: (rng.standard_normal(shape) * s).astype(np.float32)  # noqa: E731
    g = lambda n: (1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32)  # noqa: E731
    t = {"embeddings.word_embeddings.weight": w(V, H), "embeddings.token_type_embeddings.weight": w(2, H),
         "emb_ln.weight": g(H), "emb_ln.bias": w(H, s=0.05)}
    for i in range(L):
        p = f"encoder.layers.{i}."
        t[p + "attn.Wqkv.weight"] = w(3 * H, H, s=0.08)
        t[p + "attn.out_proj.weight"] = w(H, H)
        t[p + "norm1.weight"], t[p + "norm1.bias"] = g(H), w(H, s=0.05)
        t[p + "mlp.fc11.weight"], t[p + "mlp.fc12.weight"], t[p + "mlp.fc2.weight"] = w(I, H, s=0.08), w(I, H, s=0.08), w(H, I)
        t[p + "norm2.weight"], t[p + "norm2.bias"] = g(H), w(H, s=0.05)
    write_model_dir(path, cfg, t)
    return cfg, t


def xlmr_embedder(path: str, seed: int = 8, **over) -> Tuple[dict, Dict[str, np.ndarray]]:
    """bge-m3's layout: config.json says "xlm-roberta", which the reference's factories read as plain BertConfig
    (sentence_encoder/model.rs:50-53): BERT tensor names without a prefix, positions from 0, one token type."""
    cfg = dict(model_type="xlm-roberta", hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
               hidden_act="gelu", layer_norm_eps=1e-5, vocab_size=1024, max_position_embeddings=130, type_vocab_size=1,
               architectures=["XLMRobertaModel"], pad_token_id=1, bos_token_id=0, eos_token_id=2)
    cfg.update(over)
    t = bert_tensors(cfg, seed)
    t["pooler.dense.weight"] = np.zeros((cfg["hidden_size"], cfg["hidden_size"]), np.float32)  # present in the checkpoint, never read
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


def synthetic_pairs_rows(start: int, count: int, seq: int, vocab: int = 30522, seed: int = 1, qlen: int = 16, block: int = 1024):
    """Rows start .. start + count of ONE fixed synthetic pair set (the layout of synthetic_pairs), generated in blocks of
    `block` rows seeded by (seed, block index): every rank of a sharded run builds only its own rows, and the rows are the same
    at every world size (strong scaling over the same 100 000 pairs)."""
    ids = np.empty((count, seq), np.uint32)
    b0, b1 = start // block, (start + count + block - 1) // block if count else start // block
    for b in range(b0, b1):
        rows = np.random.default_rng([seed, b]).integers(1000, vocab, size=(block, seq), dtype=np.int64).astype(np.uint32)
        lo, hi = max(start, b * block), min(start + count, (b + 1) * block)
        ids[lo - start:hi - start] = rows[lo - b * block:hi - b * block]
    ids[:, 0] = 101
    q_end = min(1 + qlen, seq - 2)
    ids[:, q_end] = 102
    ids[:, seq - 1] = 102
    types = np.zeros((count, seq), np.uint32)
    types[:, q_end + 1:] = 1
    mask = np.ones((count, seq), np.uint32)
    return ids, mask, types


GOLDEN_TOKENIZER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tokenizer_small.json")


def add_tokenizer(model_dir: str) -> str:
    """Drops the golden WordPiece tokenizer.json (tests/golden) into a model directory."""
    import shutil
    dst = os.path.join(model_dir, "tokenizer.json")
    shutil.copyfile(GOLDEN_TOKENIZER, dst)
    return dst


# ----------------------------------------------------------------------------- Whisper
WHISPER_BASE = dict(d_model=512, encoder_layers=6, decoder_layers=6, encoder_attention_heads=8, decoder_attention_heads=8,
                    encoder_ffn_dim=2048, decoder_ffn_dim=2048, vocab_size=51865, max_source_positions=1500,
                    max_target_positions=448, num_mel_bins=80)
WHISPER_TEST = dict(d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4, decoder_attention_heads=4,
                    encoder_ffn_dim=128, decoder_ffn_dim=128, vocab_size=51865, max_source_positions=1500,
                    max_target_positions=448, num_mel_bins=80)


def bytes_to_unicode() -> Dict[int, str]:
    """GPT-2 byte-level alphabet (`tokenizers` ByteLevel)."""
    bs = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return {b: chr(c) for b, c in zip(bs, cs)}


def whisper_tokenizer_json(path: str, vocab_size: int = 51865) -> str:
    """A byte-level BPE tokenizer.json with Whisper's id layout: ids < 50257 ordinary tokens (the 256 byte
    symbols first, then generated byte strings, some of them not valid UTF-8 on their own), 50257.. special
    tokens (<|endoftext|>, <|startoftranscript|>, languages, tasks, <|notimestamps|>, <|0.00|> ...)."""
    b2u = bytes_to_unicode()
    vocab = {}
    for b in range(256):
        vocab[b2u[b]] = b
    rng = np.random.default_rng(7)
    words = [b" the", b" quick", b" brown", b" fox", b"ing", b"ed", b" Reykjav", b"\xc3\xadk", b" \xe6\x9d\xb1", b"\xe4\xba\xac",
             b" caf", b"\xc3", b"\xa9", b"!", b" world", b" hello", b".", b",", b" \xf0\x9f", b"\x99\x82"]
    i = 256
    for w in words:
        tok = "".join(b2u[x] for x in w)
        if tok not in vocab:
            vocab[tok] = i
            i += 1
    while i < 50257:
        n = int(rng.integers(2, 6))
        raw = bytes(rng.integers(97, 123, n).tolist())
        if rng.random() < 0.5:
            raw = b" " + raw
        tok = "".join(b2u[x] for x in raw)
        if tok in vocab:
            continue
        vocab[tok] = i
        i += 1
    names = {50257: "<|endoftext|>", 50258: "<|startoftranscript|>", 50259: "<|en|>", 50260: "<|zh|>", 50261: "<|de|>",
             50262: "<|is|>", 50359: "<|transcribe|>", 50360: "<|translate|>", 50363: "<|notimestamps|>"}
    added = []
    for tid in range(50257, vocab_size):
        if tid >= 50364:
            content = f"<|{(tid - 50364) * 0.02:.2f}|>"
        else:
            content = names.get(tid, f"<|special_{tid}|>")
        added.append(dict(id=tid, content=content, single_word=False, lstrip=False, rstrip=False, normalized=False,
                          special=True))
    spec = {"version": "1.0", "truncation": None, "padding": None, "added_tokens": added, "normalizer": None,
            "pre_tokenizer": {"type": "ByteLevel", "add_prefix_space": False, "trim_offsets": True, "use_regex": True},
            "post_processor": None,
            "decoder": {"type": "ByteLevel", "add_prefix_space": True, "trim_offsets": True, "use_regex": True},
            "model": {"type": "BPE", "dropout": None, "unk_token": None, "continuing_subword_prefix": None,
                      "end_of_word_suffix": None, "fuse_unk": False, "byte_fallback": False, "ignore_merges": False,
                      "vocab": vocab, "merges": []}}
    with open(path, "w", encoding="utf-8") as f:
        json.dump(spec, f, ensure_ascii=False)
    return path


def whisper_tensors(cfg: dict, seed: int = 0, std: float = 0.05) -> Dict[str, np.ndarray]:
    """Random-init tensors with the HF names of crates/kjarni-models/src/models/whisper/config.rs:81-190
    (k_proj has no bias, as in the released checkpoints)."""
    rng = np.random.default_rng(seed)
    H, C = cfg["d_model"], cfg["num_mel_bins"]

    def w(*shape, s=std):
        return (rng.standard_normal(shape) * s).astype(np.float32)

    def ln(pre, t):
        t[f"{pre}.weight"] = (1.0 + 0.1 * rng.standard_normal(H)).astype(np.float32)
        t[f"{pre}.bias"] = w(H, s=0.02)

    def attn(pre, t):
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            t[f"{pre}.{n}.weight"] = w(H, H)
            if n != "k_proj":
                t[f"{pre}.{n}.bias"] = w(H, s=0.02)

    t: Dict[str, np.ndarray] = {}
    t["model.encoder.conv1.weight"], t["model.encoder.conv1.bias"] = w(H, C, 3, s=0.1), w(H, s=0.02)
    t["model.encoder.conv2.weight"], t["model.encoder.conv2.bias"] = w(H, H, 3), w(H, s=0.02)
    t["model.encoder.embed_positions.weight"] = w(cfg["max_source_positions"], H, s=0.1)
    for i in range(cfg["encoder_layers"]):
        pre = f"model.encoder.layers.{i}"
        attn(f"{pre}.self_attn", t)
        ln(f"{pre}.self_attn_layer_norm", t)
        t[f"{pre}.fc1.weight"], t[f"{pre}.fc1.bias"] = w(cfg["encoder_ffn_dim"], H), w(cfg["encoder_ffn_dim"], s=0.02)
        t[f"{pre}.fc2.weight"], t[f"{pre}.fc2.bias"] = w(H, cfg["encoder_ffn_dim"]), w(H, s=0.02)
        ln(f"{pre}.final_layer_norm", t)
    ln("model.encoder.layer_norm", t)
    t["model.decoder.embed_tokens.weight"] = w(cfg["vocab_size"], H, s=0.1)
    t["model.decoder.embed_positions.weight"] = w(cfg["max_target_positions"], H, s=0.1)
    for i in range(cfg["decoder_layers"]):
        pre = f"model.decoder.layers.{i}"
        attn(f"{pre}.self_attn", t)
        ln(f"{pre}.self_attn_layer_norm", t)
        attn(f"{pre}.encoder_attn", t)
        ln(f"{pre}.encoder_attn_layer_norm", t)
        t[f"{pre}.fc1.weight"], t[f"{pre}.fc1.bias"] = w(cfg["decoder_ffn_dim"], H), w(cfg["decoder_ffn_dim"], s=0.02)
        t[f"{pre}.fc2.weight"], t[f"{pre}.fc2.bias"] = w(H, cfg["decoder_ffn_dim"]), w(H, s=0.02)
        ln(f"{pre}.final_layer_norm", t)
    ln("model.decoder.layer_norm", t)
    return t


def whisper_model(path: str, seed: int = 0, base: bool = False, **over) -> Tuple[dict, Dict[str, np.ndarray]]:
    cfg = dict(WHISPER_BASE if base else WHISPER_TEST, model_type="whisper", activation_function="gelu",
               decoder_start_token_id=50258, eos_token_id=50257, bos_token_id=50257, pad_token_id=50256,
               scale_embedding=False, architectures=["WhisperForConditionalGeneration"])
    cfg.update(over)
    t = whisper_tensors(cfg, seed)
    write_model_dir(path, cfg, t)
    whisper_tokenizer_json(os.path.join(path, "tokenizer.json"), cfg["vocab_size"])
    return cfg, t


def synthetic_audio(seconds: float, seed: int = 0, rate: int = 16000) -> np.ndarray:
    """Tones + a chirp + broadband noise: no frequency bin is empty, so log-mel comparisons are well conditioned."""
    rng = np.random.default_rng(seed)
    n = int(seconds * rate)
    t = np.arange(n) / rate
    x = 0.3 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 1330 * t + 0.5) + \
        0.15 * np.sin(2 * np.pi * (300 + 900 * t / max(seconds, 1e-3)) * t) + 0.05 * rng.standard_normal(n)
    env = 0.6 + 0.4 * np.sin(2 * np.pi * 0.7 * t)
    return (x * env).astype(np.float32)


# ----------------------------------------------------------------------------- decoder-only LLMs
LLAMA_TEST = dict(model_type="llama", hidden_size=64, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                  intermediate_size=128, vocab_size=320, max_position_embeddings=256, rms_norm_eps=1e-5, rope_theta=500000.0,
                  tie_word_embeddings=True, bos_token_id=1, eos_token_id=[2, 3], hidden_act="silu",
                  rope_scaling=dict(rope_type="llama3", factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                    original_max_position_embeddings=64))
QWEN_TEST = dict(model_type="qwen2", hidden_size=64, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=1,
                 intermediate_size=96, vocab_size=300, max_position_embeddings=128, rms_norm_eps=1e-6, rope_theta=1000000.0,
                 tie_word_embeddings=False, bos_token_id=1, eos_token_id=2, hidden_act="silu")
# Llama-3.2-1B-Instruct geometry (registry.rs:517-530): the shape BASELINE.json configs[4] ("small LLM") is measured on
LLAMA_1B = dict(model_type="llama", hidden_size=2048, num_hidden_layers=16, num_attention_heads=32, num_key_value_heads=8,
                intermediate_size=8192, vocab_size=128256, max_position_embeddings=131072, rms_norm_eps=1e-5, rope_theta=500000.0,
                tie_word_embeddings=True, bos_token_id=128000, eos_token_id=[128001, 128008, 128009], hidden_act="silu",
                head_dim=64, torch_dtype="bfloat16",
                rope_scaling=dict(rope_type="llama3", factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                  original_max_position_embeddings=8192))


def llm_tensors(cfg: dict, seed: int = 0, std: float = 0.05, bf16: bool = False) -> Dict[str, np.ndarray]:
    """Random-init tensors with the HF names of llama/config.rs:283-330 (q/k/v biases for qwen2)."""
    rng = np.random.default_rng(seed)
    H, L, I = cfg["hidden_size"], cfg["num_hidden_layers"], cfg["intermediate_size"]
    d = cfg.get("head_dim") or H // cfg["num_attention_heads"]
    kv = cfg["num_key_value_heads"] * d

    def w(*shape, s=std):
        a = (rng.standard_normal(shape, dtype=np.float32) * np.float32(s))
        if bf16:  # values exactly representable in bf16, so f32 and bf16 storage hold the same numbers
            u = a.view(np.uint32)
            a = (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)
        return a

    t: Dict[str, np.ndarray] = {"model.embed_tokens.weight": w(cfg["vocab_size"], H, s=0.1),
                                "model.norm.weight": (1.0 + 0.1 * rng.standard_normal(H)).astype(np.float32)}
    if not cfg.get("tie_word_embeddings", True):
        t["lm_head.weight"] = w(cfg["vocab_size"], H, s=0.1)
    for i in range(L):
        p = f"model.layers.{i}"
        t[f"{p}.self_attn.q_proj.weight"], t[f"{p}.self_attn.k_proj.weight"] = w(H, H), w(kv, H)
        t[f"{p}.self_attn.v_proj.weight"], t[f"{p}.self_attn.o_proj.weight"] = w(kv, H), w(H, H)
        if cfg["model_type"] == "qwen2":
            t[f"{p}.self_attn.q_proj.bias"], t[f"{p}.self_attn.k_proj.bias"] = w(H, s=0.1).astype(np.float32), w(kv, s=0.1)
            t[f"{p}.self_attn.v_proj.bias"] = w(kv, s=0.1)
        t[f"{p}.mlp.gate_proj.weight"], t[f"{p}.mlp.up_proj.weight"], t[f"{p}.mlp.down_proj.weight"] = w(I, H), w(I, H), w(H, I)
        t[f"{p}.input_layernorm.weight"] = (1.0 + 0.1 * rng.standard_normal(H)).astype(np.float32)
        t[f"{p}.post_attention_layernorm.weight"] = (1.0 + 0.1 * rng.standard_normal(H)).astype(np.float32)
    return t


def llm_model(path: str, base: dict, seed: int = 0, bf16_values: bool = False, store_bf16: bool = False, std: float = 0.05,
              **over):
    """Writes config.json + model.safetensors.  store_bf16: the 2-D weights are stored as BF16 tensors."""
    cfg = dict(base)
    cfg.update(over)
    t = llm_tensors(cfg, seed, std=std, bf16=bf16_values or store_bf16)
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f, indent=1)
    if store_bf16:
        import torch
        from safetensors.torch import save_file
        save_file({k: (torch.from_numpy(v).to(torch.bfloat16) if v.ndim == 2 else torch.from_numpy(v)) for k, v in t.items()},
                  os.path.join(path, "model.safetensors"))
    else:
        from safetensors.numpy import save_file
        save_file({k: np.ascontiguousarray(v) for k, v in t.items()}, os.path.join(path, "model.safetensors"))
    return cfg, t


# Llama-3.1-8B-Instruct geometry (registry.rs:577-590)
LLAMA_8B = dict(model_type="llama", hidden_size=4096, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8,
                intermediate_size=14336, vocab_size=128256, max_position_embeddings=131072, rms_norm_eps=1e-5, rope_theta=500000.0,
                tie_word_embeddings=False, bos_token_id=128000, eos_token_id=[128001, 128008, 128009], hidden_act="silu",
                head_dim=128, torch_dtype="bfloat16",
                rope_scaling=dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                  original_max_position_embeddings=8192))


def llm_model_streamed(path: str, base: dict, seed: int = 0, std: float = 0.02, **over):
    """config.json + model.safetensors with BF16 matrices written tensor by tensor (multi-GB shapes never sit in memory whole)."""
    cfg = dict(base)
    cfg.update(over)
    H, L, I, V = cfg["hidden_size"], cfg["num_hidden_layers"], cfg["intermediate_size"], cfg["vocab_size"]
    d = cfg.get("head_dim") or H // cfg["num_attention_heads"]
    kv = cfg["num_key_value_heads"] * d
    specs = [("model.embed_tokens.weight", (V, H))]
    for i in range(L):
        p = f"model.layers.{i}"
        specs += [(p + ".self_attn.q_proj.weight", (H, H)), (p + ".self_attn.k_proj.weight", (kv, H)), (p + ".self_attn.v_proj.weight", (kv, H)),
                  (p + ".self_attn.o_proj.weight", (H, H)), (p + ".mlp.gate_proj.weight", (I, H)), (p + ".mlp.up_proj.weight", (I, H)),
                  (p + ".mlp.down_proj.weight", (H, I)), (p + ".input_layernorm.weight", (H,)), (p + ".post_attention_layernorm.weight", (H,))]
    specs.append(("model.norm.weight", (H,)))
    if not cfg.get("tie_word_embeddings", True):
        specs.append(("lm_head.weight", (V, H)))
    header, off = {}, 0
    for name, shape in specs:
        n = int(np.prod(shape))
        nbytes = n * (2 if len(shape) == 2 else 4)
        header[name] = {"dtype": "BF16" if len(shape) == 2 else "F32", "shape": list(shape), "data_offsets": [off, off + nbytes]}
        off += nbytes
    blob = json.dumps(header, separators=(",", ":")).encode()
    blob += b" " * ((8 - len(blob) % 8) % 8)
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f, indent=1)
    rng = np.random.default_rng(seed)
    with open(os.path.join(path, "model.safetensors"), "wb") as f:
        f.write(len(blob).to_bytes(8, "little"))
        f.write(blob)
        for name, shape in specs:
            n = int(np.prod(shape))
            if len(shape) == 1:
                f.write((1.0 + 0.1 * rng.standard_normal(n, dtype=np.float32)).astype("<f4").tobytes())
                continue
            for start in range(0, n, 1 << 26):
                m = min(1 << 26, n - start)
                x = rng.standard_normal(m, dtype=np.float32) * np.float32(std)
                f.write((x.view(np.uint32) >> 16).astype("<u2").tobytes())
    return cfg

