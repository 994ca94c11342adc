"""Pins oracle/llm_oracle.py with the reference's goldens for the decoder-only path (CPU only):
GQA attention with cache (cpu/decoder/decoder_attention.rs:316-396), RoPE PyTorch parity (cpu/rope/tests.rs:24-311,
410-530), RMSNorm (cpu/normalization/rms_norm.rs:209-360), sampling helpers (common/sampling.rs)."""
import json
import os

import numpy as np

from oracle import llm_oracle as L

F32 = np.float32
G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "llm_goldens.json")))


def test_gqa_attention_with_cache_golden():
    g = G["gqa"]
    p = dict(q_w=F32(g["weight_q"]).reshape(16, 16), k_w=F32(g["weight_k"]).reshape(8, 16),
             v_w=F32(g["weight_v"]).reshape(8, 16), o_w=F32(g["weight_o"]).reshape(16, 16))
    k = np.zeros((1, 3, 8), F32)
    v = np.zeros((1, 3, 8), F32)
    k[:, :2] = F32(g["history_k"]).reshape(1, 2, 8)
    v[:, :2] = F32(g["history_v"]).reshape(1, 2, 8)
    out = L.gqa_attention(F32(g["hidden"]).reshape(1, 1, 16), p, 4, 2, k, v, 2, None)
    assert np.abs(out.reshape(-1) - F32(g["output"])).max() < 1e-4
    assert np.abs(k[0, 2] - F32(g["update_k"])).max() < 1e-4 and np.abs(v[0, 2] - F32(g["update_v"])).max() < 1e-4


def test_rope_pytorch_parity_golden():
    g = G["rope"]
    cos, sin = L.rope_tables(8, 14, 10000.0)
    q = L.rope_rotate(F32(g["q"]).reshape(1, 2, 4, 8), cos, sin, 10)
    k = L.rope_rotate(F32(g["k"]).reshape(1, 2, 4, 8), cos, sin, 10)
    assert np.abs(q.reshape(-1) - F32(g["expected_q"])).max() < 1e-5
    assert np.abs(k.reshape(-1) - F32(g["expected_k"])).max() < 1e-5


def test_rope_reference_unit_cases():
    cos, sin = L.rope_tables(4, 8, 10000.0)
    q = F32([1, 0, 1, 0]).reshape(1, 1, 1, 4)
    assert np.abs(L.rope_rotate(q, cos, sin, 0) - q).max() < 1e-3                 # rope/tests.rs:582-598: position 0 is identity
    x = np.random.default_rng(0).standard_normal((1, 2, 5, 8)).astype(F32)
    cos, sin = L.rope_tables(8, 32, 10000.0)
    r = L.rope_rotate(x, cos, sin, 3)
    assert np.allclose(np.linalg.norm(r, axis=-1), np.linalg.norm(x, axis=-1), atol=1e-5)   # :562-580 preserves norm
    assert np.abs(r - x).max() > 1e-3                                              # :312-328 actually rotates
    a = L.rope_rotate(x[:, :, 2:3], cos, sin, 5)
    b = L.rope_rotate(x, cos, sin, 3)[:, :, 2:3]
    assert np.abs(a - b).max() < 1e-6                                              # :508-530 offset = absolute position
    inv = L.rope_inv_freq(8, 10000.0)                                              # :357-370 frequencies
    assert abs(inv[0] - 1.0) < 1e-6 and np.all(np.diff(inv) < 0)
    sc = dict(rope_type="llama3", factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192)
    inv3 = L.rope_inv_freq(64, 500000.0, sc)
    base = L.rope_inv_freq(64, 500000.0)
    assert inv3[0] == base[0] and abs(inv3[-1] - base[-1] / 32.0) < 1e-12 and np.all(inv3 <= base)


def test_rmsnorm_goldens():
    g = G["rmsnorm"]
    out = L.rms_norm(F32(g["input"]).reshape(1, 1, 8), F32(g["gamma"]), 1e-5)
    assert np.abs(out.reshape(-1) - F32(g["expected"])).max() < 1e-5
    rms = np.sqrt((9 + 16 + 0) / 3.0)                                               # rms_norm.rs:263-299
    assert np.abs(L.rms_norm(F32([[3, 4, 0]]), np.ones(3, F32), 1e-6) - F32([3, 4, 0]) / rms).max() < 1e-4
    assert np.abs(L.rms_norm(F32([[3, 4, 0]]), F32([2, .5, 1.5]), 1e-6) - F32([3, 4, 0]) / rms * F32([2, .5, 1.5])).max() < 1e-4
    assert np.isfinite(L.rms_norm(F32([[1e-8, 2e-8, 1e-8]]), np.ones(3, F32), 1e-6)).all()


def test_swiglu_and_sampling_helpers():
    x = F32([[-2.0, -0.5, 0.0, 0.5, 2.0]])
    assert np.allclose(L.silu(x), x / (1 + np.exp(-x)), atol=1e-7)
    p = dict(gate_w=np.eye(3, dtype=F32), up_w=2 * np.eye(3, dtype=F32), down_w=np.eye(3, dtype=F32))
    v = F32([[[1.0, -1.0, 0.5]]])
    assert np.allclose(L.swiglu(v, p), L.silu(v) * 2 * v, atol=1e-6)
    lg = F32([1.0, 3.0, 3.0, -2.0])
    assert L.greedy(lg) == 2                                                       # last maximum (Iterator::max_by)
    L.apply_repetition_penalty(lg, [1, 3, 3], 2.0)                                 # sampling.rs:207-219, per occurrence
    assert lg.tolist() == [1.0, 1.5, 3.0, -8.0]
    lg = np.zeros(6, F32)
    L.apply_no_repeat_ngram(lg, [1, 2, 3, 1, 2], 3)                                # sampling.rs:221-235: (1,2)->3 is banned
    assert np.isneginf(lg[3]) and np.isfinite(np.delete(lg, 3)).all()
    assert np.array_equal(L.bf16_round(F32([1.0, 1.00390625, 3.1415927])), F32([1.0, 1.0, 3.140625]))
