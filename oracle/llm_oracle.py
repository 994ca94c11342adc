"""CPU restatement of the reference's decoder-only (Llama / Qwen2) path -- TEST INFRASTRUCTURE ONLY.

  RMSNorm                 crates/kjarni-transformers/src/cpu/normalization/rms_norm.rs:19-27
  RoPE (+ llama3 scaling) cpu/rope/mod.rs:20-176
  GQA attention + cache   cpu/decoder/decoder_attention.rs:44-196
  SwiGLU                  cpu/feedforward/swiglu.rs:32-57
  layer / model           cpu/decoder/rope_decoder_layer.rs:18-41, kjarni-models/src/models/llama/cpu_decoder.rs:142-219,
                          llama/config.rs:234-330 (tensor names), qwen/config.rs:225-275 (q/k/v biases)
  generation loop         decoder/generator.rs:228-381, common/sampling.rs:81-235 (greedy, repetition penalty,
                          no-repeat n-gram)
Weights stored as bf16 are widened exactly to f32 (what the reference's bf16 LinearLayer computes with);
activations, KV cache and accumulation are f32.

Parity status: PINNED by the reference's goldens (tests/test_llm_oracle.py): GQA attention with cache
(decoder_attention.rs:316-396), RoPE PyTorch parity (rope/tests.rs:24-311), RMSNorm PyTorch parity
(rms_norm.rs:209-246), SwiGLU / sampling unit cases.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import oracle as O

F32 = np.float32
MASK_VALUE = F32(-1e9)


def rms_norm(x: np.ndarray, weight: np.ndarray, eps: float) -> np.ndarray:
    """rms_norm.rs:19-27: x / sqrt(mean(x^2) + eps) * w."""
    x = x.astype(F32)
    ms = np.mean((x * x).astype(F32), axis=-1, keepdims=True, dtype=F32)
    return ((x / np.sqrt((ms + F32(eps)).astype(F32)).astype(F32)).astype(F32) * weight.astype(F32)).astype(F32)


def rope_inv_freq(head_dim: int, theta: float, scaling: Optional[dict] = None) -> np.ndarray:
    """rope/mod.rs:62-105."""
    half = head_dim // 2
    base = np.asarray([F32(1.0) / F32(np.power(F32(theta), F32(F32(2 * i) / F32(head_dim)))) for i in range(half)], F32)
    if not scaling or scaling.get("rope_type") != "llama3":
        return base
    factor, lo, hi = F32(scaling["factor"]), F32(scaling["low_freq_factor"]), F32(scaling["high_freq_factor"])
    orig = F32(scaling["original_max_position_embeddings"])
    low_wl, high_wl = orig / lo, orig / hi
    out = np.zeros(half, F32)
    for i in range(half):
        wl = F32(2.0) * F32(np.pi) / base[i]
        if wl < high_wl:
            out[i] = base[i]
        elif wl > low_wl:
            out[i] = base[i] / factor
        else:
            smooth = (orig / wl - lo) / (hi - lo)
            out[i] = base[i] / ((F32(1.0) - smooth) * factor + smooth)
    return out


def rope_tables(head_dim: int, max_len: int, theta: float, scaling: Optional[dict] = None):
    """rope/mod.rs:107-130: cos/sin of pos * inv_freq, duplicated over both halves."""
    inv = rope_inv_freq(head_dim, theta, scaling)
    ang = (np.arange(max_len, dtype=F32)[:, None] * inv[None, :]).astype(F32)
    c, s = np.cos(ang).astype(F32), np.sin(ang).astype(F32)
    return np.concatenate([c, c], 1), np.concatenate([s, s], 1)


def rope_rotate(x: np.ndarray, cos: np.ndarray, sin: np.ndarray, offset: int) -> np.ndarray:
    """rope/mod.rs:156-176 on [B, heads, S, d]: (x0, x1) = (x[i], x[i + d/2])."""
    B, Hh, S, d = x.shape
    half = d // 2
    c = cos[offset:offset + S, :half][None, None]
    s = sin[offset:offset + S, :half][None, None]
    x0, x1 = x[..., :half], x[..., half:]
    return np.concatenate([x0 * c - x1 * s, x0 * s + x1 * c], axis=-1).astype(F32)


def _softmax_rows(s: np.ndarray) -> np.ndarray:
    m = s.max(axis=-1, keepdims=True)
    e = np.exp((s - m).astype(F32)).astype(F32)
    z = e.sum(axis=-1, keepdims=True, dtype=F32)
    return (e * (F32(1.0) / z)).astype(F32)


def gqa_attention(hidden: np.ndarray, p: Dict[str, np.ndarray], heads: int, kv_heads: int, k_cache: np.ndarray,
                  v_cache: np.ndarray, offset: int, rope=None) -> np.ndarray:
    """DecoderAttention::forward (decoder_attention.rs:44-170).  k_cache / v_cache: [B, total, kv_heads*d]; the
    new rows are written at total - S .. (in place); mask of ones, causal by overwrite with -1e9."""
    B, S, H = hidden.shape
    d = H // heads
    total = k_cache.shape[1]
    start = total - S
    q = O.linear(hidden, p["q_w"], p.get("q_b"))
    k_new = O.linear(hidden, p["k_w"], p.get("k_b"))
    v_new = O.linear(hidden, p["v_w"], p.get("v_b"))
    qh = q.reshape(B, S, heads, d).transpose(0, 2, 1, 3)
    kh_new = k_new.reshape(B, S, kv_heads, d).transpose(0, 2, 1, 3)
    if rope is not None:
        qh = rope_rotate(qh, rope[0], rope[1], offset)
        kh_new = rope_rotate(kh_new, rope[0], rope[1], offset)
    k_cache[:, start:, :] = kh_new.transpose(0, 2, 1, 3).reshape(B, S, kv_heads * d)
    v_cache[:, start:, :] = v_new
    n_rep = heads // kv_heads
    kh = np.repeat(k_cache.reshape(B, total, kv_heads, d).transpose(0, 2, 1, 3), n_rep, axis=1)
    vh = np.repeat(v_cache.reshape(B, total, kv_heads, d).transpose(0, 2, 1, 3), n_rep, axis=1)
    scores = (np.matmul(qh, kh.transpose(0, 1, 3, 2)).astype(F32) * F32(1.0 / math.sqrt(d))).astype(F32)
    qpos = start + np.arange(S)[:, None]
    scores = np.where(np.arange(total)[None, :] > qpos, MASK_VALUE, scores).astype(F32)
    ctx = np.matmul(_softmax_rows(scores), vh).astype(F32)
    ctx = np.ascontiguousarray(ctx.transpose(0, 2, 1, 3)).reshape(B, S, H)
    return O.linear(ctx, p["o_w"], p.get("o_b"))


def silu(x: np.ndarray) -> np.ndarray:
    x = x.astype(F32)
    return (x / (F32(1.0) + np.exp(-x).astype(F32))).astype(F32)


def swiglu(x: np.ndarray, p: Dict[str, np.ndarray]) -> np.ndarray:
    """swiglu.rs:32-57: down(silu(gate(x)) * up(x))."""
    g = silu(O.linear(x, p["gate_w"], None))
    return O.linear((g * O.linear(x, p["up_w"], None)).astype(F32), p["down_w"], None)


def bf16_round(a: np.ndarray) -> np.ndarray:
    """f32 -> bf16 (round to nearest even) -> f32."""
    u = np.ascontiguousarray(a, F32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return (((u + r) >> 16) << 16).astype(np.uint32).view(F32)


class LlmOracle:
    """Llama / Qwen2 / Mistral (= the Llama decoder, mistral/model.rs:56-62) decoder over a {hf_tensor_name: ndarray} dict + HF config dict."""

    def __init__(self, tensors: Dict[str, np.ndarray], config: dict):
        self.t = {k: np.ascontiguousarray(v, F32) for k, v in tensors.items()}
        c = self.c = config
        self.H, self.heads = c["hidden_size"], c["num_attention_heads"]
        self.kv_heads = c.get("num_key_value_heads", self.heads)
        self.d = c.get("head_dim") or self.H // self.heads
        self.eps = c.get("rms_norm_eps", 1e-6 if c.get("model_type") == "qwen2" else 1e-5)
        self.L = c["num_hidden_layers"]
        self.rope = rope_tables(self.d, c["max_position_embeddings"], c.get("rope_theta", {"llama": 500000.0, "qwen2": 1000000.0}.get(c.get("model_type"), 10000.0)),
                                c.get("rope_scaling"))
        tie = c.get("tie_word_embeddings", c.get("model_type") == "llama")
        self.lm_head = self.t["model.embed_tokens.weight"] if tie else self.t["lm_head.weight"]
        eos = c.get("eos_token_id", [])
        self.stop_tokens = list(eos) if isinstance(eos, (list, tuple)) else [eos]

    def _attn(self, i: int):
        pre = f"model.layers.{i}.self_attn"
        d = dict(q_w=self.t[f"{pre}.q_proj.weight"], k_w=self.t[f"{pre}.k_proj.weight"], v_w=self.t[f"{pre}.v_proj.weight"],
                 o_w=self.t[f"{pre}.o_proj.weight"])
        for s, n in (("q_b", "q_proj"), ("k_b", "k_proj"), ("v_b", "v_proj"), ("o_b", "o_proj")):
            if f"{pre}.{n}.bias" in self.t:
                d[s] = self.t[f"{pre}.{n}.bias"]
        return d

    def new_cache(self):
        kv = self.kv_heads * self.d
        return [(np.zeros((1, 0, kv), F32), np.zeros((1, 0, kv), F32)) for _ in range(self.L)]

    def forward(self, ids: Sequence[int], cache) -> np.ndarray:
        """Embedding -> layers (rope_decoder_layer.rs:18-41) -> final RMSNorm (cpu_decoder.rs:196-219).
        Returns the normed hidden states [1, S, H]; `cache` grows by S positions."""
        ids = np.asarray(ids, np.int64)
        h = self.t["model.embed_tokens.weight"][ids][None].astype(F32)
        S = h.shape[1]
        offset = cache[0][0].shape[1]
        for i in range(self.L):
            pre = f"model.layers.{i}"
            k, v = cache[i]
            k = np.concatenate([k, np.zeros((1, S, k.shape[2]), F32)], 1)
            v = np.concatenate([v, np.zeros((1, S, v.shape[2]), F32)], 1)
            n1 = rms_norm(h, self.t[f"{pre}.input_layernorm.weight"], self.eps)
            h = (h + gqa_attention(n1, self._attn(i), self.heads, self.kv_heads, k, v, offset, self.rope)).astype(F32)
            cache[i] = (k, v)
            n2 = rms_norm(h, self.t[f"{pre}.post_attention_layernorm.weight"], self.eps)
            h = (h + swiglu(n2, dict(gate_w=self.t[f"{pre}.mlp.gate_proj.weight"], up_w=self.t[f"{pre}.mlp.up_proj.weight"],
                                     down_w=self.t[f"{pre}.mlp.down_proj.weight"]))).astype(F32)
        return rms_norm(h, self.t["model.norm.weight"], self.eps)

    def logits(self, hidden_row: np.ndarray) -> np.ndarray:
        return O.linear(hidden_row.reshape(1, -1), self.lm_head, None)[0]

    def generate(self, prompt: Sequence[int], max_new_tokens: int, repetition_penalty: float = 1.0,
                 no_repeat_ngram: int = 0, context_limit: Optional[int] = None, return_logits: bool = False):
        """run_generation_loop with DecodingStrategy::Greedy (decoder/generator.rs:228-381)."""
        cache = self.new_cache()
        logits = self.logits(self.forward(prompt, cache)[0, -1])
        all_tokens, out, trace = list(prompt), [], []
        limit = context_limit or self.c["max_position_embeddings"]
        for _ in range(max_new_tokens):
            if len(all_tokens) >= limit:
                break
            lg = logits.copy()
            if repetition_penalty != 1.0:
                apply_repetition_penalty(lg, all_tokens, repetition_penalty)
            if no_repeat_ngram > 0:
                apply_no_repeat_ngram(lg, all_tokens, no_repeat_ngram)
            trace.append(lg)
            nxt = greedy(lg)
            if nxt in self.stop_tokens:
                break
            all_tokens.append(nxt)
            out.append(nxt)
            logits = self.logits(self.forward([nxt], cache)[0, -1])
        return (out, trace) if return_logits else out


def greedy(logits: np.ndarray) -> int:
    """sampling.rs:83-88: Iterator::max_by keeps the LAST maximum."""
    best = logits.max()
    return int(np.nonzero(logits == best)[0][-1])


def apply_repetition_penalty(logits: np.ndarray, tokens: Sequence[int], penalty: float):
    """sampling.rs:207-219: once per OCCURRENCE."""
    p = F32(penalty)
    for t in tokens:
        if t < len(logits):
            logits[t] = logits[t] * p if logits[t] < 0 else logits[t] / p


def apply_no_repeat_ngram(logits: np.ndarray, tokens: Sequence[int], n: int):
    """sampling.rs:221-235."""
    if len(tokens) < n - 1:
        return
    prefix = list(tokens[len(tokens) - (n - 1):])
    for i in range(len(tokens) - n + 1):
        w = tokens[i:i + n]
        if list(w[:n - 1]) == prefix and w[n - 1] < len(logits):
            logits[w[n - 1]] = -np.inf
