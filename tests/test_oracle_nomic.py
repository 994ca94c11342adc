"""The oracle's Nomic branch (RoPE + SwiGLU + fused Wqkv, sentence_encoder/configs.rs:140-275) against a float64 numpy
restatement written straight from the reference's formulas: rope/mod.rs:96-170 (caches, half-split rotation),
feedforward/swiglu.rs:33-57, encoder_self_attention.rs:61-140, encoder_layer.rs:216-232 (post-norm)."""
import numpy as np

from oracle import oracle as O
from tests import synth


def _ln(x, g, b, eps):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def _nomic_f64(t, cfg, ids, mask):
    t = {k: v.astype(np.float64) for k, v in t.items()}
    H, L, heads, eps = cfg["n_embd"], cfg["n_layer"], cfg["n_head"], cfg["layer_norm_epsilon"]
    d, half = H // heads, H // heads // 2
    B, S = ids.shape
    h = t["embeddings.word_embeddings.weight"][ids] + t["embeddings.token_type_embeddings.weight"][0]
    h = _ln(h, t["emb_ln.weight"], t["emb_ln.bias"], eps)
    inv = 1.0 / cfg["rotary_emb_base"] ** (2.0 * np.arange(half) / d)
    ang = np.arange(S)[:, None] * inv[None, :]
    cos, sin = np.cos(ang)[None, :, None, :], np.sin(ang)[None, :, None, :]

    def rot(x):  # [B,S,heads,d]
        x0, x1 = x[..., :half], x[..., half:]
        return np.concatenate([x0 * cos - x1 * sin, x0 * sin + x1 * cos], -1)

    for i in range(L):
        p = f"encoder.layers.{i}."
        qkv = h @ t[p + "attn.Wqkv.weight"].T
        q, k, v = [qkv[..., j * H:(j + 1) * H].reshape(B, S, heads, d) for j in range(3)]
        q, k = rot(q), rot(k)
        sc = np.einsum("bqhd,bkhd->bhqk", q, k) / np.sqrt(d)
        sc = np.where(mask[:, None, None, :] == 0, -1e9, sc)
        sc = np.exp(sc - sc.max(-1, keepdims=True))
        pr = sc / sc.sum(-1, keepdims=True)
        ctx = np.einsum("bhqk,bkhd->bqhd", pr, v).reshape(B, S, H)
        h = _ln(h + ctx @ t[p + "attn.out_proj.weight"].T, t[p + "norm1.weight"], t[p + "norm1.bias"], eps)
        gate, up = h @ t[p + "mlp.fc11.weight"].T, h @ t[p + "mlp.fc12.weight"].T
        ffn = (gate / (1.0 + np.exp(-gate)) * up) @ t[p + "mlp.fc2.weight"].T
        h = _ln(h + ffn, t[p + "norm2.weight"], t[p + "norm2.bias"], eps)
    return h


def test_nomic_oracle_matches_float64_restatement(tmp_path):
    cfg, t = synth.nomic_embedder(str(tmp_path / "nomic"))
    orc = O.OracleModel(t, cfg)
    ids, mask = synth.synthetic_ids(5, 40, vocab=cfg["vocab_size"], seed=3, ragged=True)
    want = _nomic_f64(t, cfg, ids, mask)
    got = orc.forward(ids, mask, None, O.MASK_ALLOC)
    valid = mask.astype(bool)
    assert np.abs(got[valid] - want[valid]).max() < 2e-5
    # RoPE really is position dependent: the same token at two positions embeds differently before pooling, and a model
    # read without the rotary keys has no position signal at all
    flat = dict(cfg)
    flat.pop("rotary_emb_fraction"), flat.pop("rotary_emb_base")
    plain = O.OracleModel(t, flat).forward(ids, mask, None, O.MASK_ALLOC)
    assert np.abs(plain[valid] - got[valid]).max() > 1e-3
    # embed_batch = mean pool + L2 (encode_batch_flat)
    e = orc.embed_batch(ids, mask)
    m = mask[..., None].astype(np.float64)
    pooled = (want * m).sum(1) / m.sum(1)
    pooled /= np.linalg.norm(pooled, axis=1, keepdims=True)
    assert np.abs(e - pooled).max() < 2e-5


def test_nomic_config_aliases_and_defaults(tmp_path):
    # BertConfig's serde aliases (configs.rs:15-27, 43-44): the BERT-style key names describe the same model
    cfg, t = synth.nomic_embedder(str(tmp_path / "a"))
    alias = dict(model_type="nomic_bert", hidden_size=cfg["n_embd"], num_hidden_layers=cfg["n_layer"],
                 num_attention_heads=cfg["n_head"], intermediate_size=cfg["n_inner"], layer_norm_eps=cfg["layer_norm_epsilon"],
                 max_position_embeddings=cfg["n_positions"], vocab_size=cfg["vocab_size"], rotary_emb_base=cfg["rotary_emb_base"])
    ids, mask = synth.synthetic_ids(2, 16, vocab=cfg["vocab_size"], seed=1)
    a = O.OracleModel(t, cfg).forward(ids, mask)
    b = O.OracleModel(t, alias).forward(ids, mask)
    assert np.array_equal(a, b)
    # rotary_emb_fraction alone switches RoPE on with theta 10000 (configs.rs:175-183)
    only_fraction = dict(alias)
    only_fraction.pop("rotary_emb_base")
    only_fraction["rotary_emb_fraction"] = 1.0
    c = O.OracleModel(t, only_fraction).forward(ids, mask)
    assert np.abs(c - a).max() > 1e-4
