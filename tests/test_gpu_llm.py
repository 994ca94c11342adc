"""Decoder-only generation on the GPU (kjarni_hip_decoder_*) vs oracle/llm_oracle.py (pinned by the reference's
GQA / RoPE / RMSNorm goldens in tests/test_llm_oracle.py): prefill, cached decode, grouped-query attention, llama3 RoPE
scaling, Qwen2 biases and untied heads, bf16-stored weights, the greedy loop with its stop rules and logits processors."""
import numpy as np
import pytest

from oracle import llm_oracle as L
from tests import synth

pytestmark = pytest.mark.gpu
F32 = np.float32
TOL = 1e-4


def _pair(tmp_path, base, **kw):
    import kjarni_amd
    d = str(tmp_path / base["model_type"])
    weights = kw.pop("weights", "auto")
    cfg, t = synth.llm_model(d, base, **kw)
    return L.LlmOracle(t, cfg), kjarni_amd.HipDecoder(d, weights=weights), cfg


@pytest.mark.parametrize("base", [synth.LLAMA_TEST, synth.QWEN_TEST], ids=["llama-gqa-rope-scaling", "qwen2-bias-mqa-untied"])
def test_prefill_and_cached_decode(tmp_path, base):
    orc, gpu, cfg = _pair(tmp_path, base, seed=3)
    rng = np.random.default_rng(0)
    cache = orc.new_cache()
    gpu.reset()
    for n in (5, 1, 1, 11, 1, 3, 1):                        # prefill blocks of <= 8 rows, single-token steps, a second prompt block
        ids = rng.integers(4, cfg["vocab_size"], n).tolist()
        ref_h = orc.forward(ids, cache)[0]
        h, logits = gpu.forward(ids)
        k = (n - 1) % 8 + 1                                   # the hook returns the rows of the last 8-row block
        assert np.abs(h[-k:] - ref_h[-k:]).max() < TOL, np.abs(h[-k:] - ref_h[-k:]).max()
        assert np.abs(logits - orc.logits(ref_h[-1])).max() < TOL
    # far positions: RoPE tables at the end of the context
    gpu.reset()
    cache = orc.new_cache()
    ids = rng.integers(4, cfg["vocab_size"], cfg["max_position_embeddings"] - 2).tolist()
    ref_h = orc.forward(ids, cache)[0]
    h, logits = gpu.forward(ids)
    assert np.abs(h[-1] - ref_h[-1]).max() < TOL and np.abs(logits - orc.logits(ref_h[-1])).max() < TOL


def _check(got, exp, trace):
    for i, (a, b) in enumerate(zip(got, exp)):
        if a != b:                                          # only where the oracle's own two best logits tie within noise
            assert abs(trace[i][a] - trace[i][b]) < 1e-4, (i, a, b)
            return
    assert len(got) == len(exp)


def test_greedy_generation_matches_oracle(tmp_path):
    orc, gpu, cfg = _pair(tmp_path, synth.LLAMA_TEST, seed=4)
    prompt = [1, 17, 44, 203, 9, 9, 250, 31, 77, 5, 120]
    exp, trace = orc.generate(prompt, 40, return_logits=True)
    got = gpu.generate(prompt, 40)
    assert len(exp) > 5
    _check(got, exp, trace)
    assert gpu.generate(prompt, 40) == got                  # deterministic, cache reset between calls
    assert gpu.generate(prompt, 3) == got[:3] and gpu.generate(prompt, 0) == []
    seen = []
    part = gpu.generate(prompt, 40, on_token=lambda t: (seen.append(t), len(seen) < 4)[1])
    assert part == got[:4] == seen
    # logits processors (common/sampling.rs:207-235) run per step
    exp, trace = orc.generate(prompt, 25, repetition_penalty=1.3, no_repeat_ngram=2, return_logits=True)
    _check(gpu.generate(prompt, 25, repetition_penalty=1.3, no_repeat_ngram=2), exp, trace)
    exp, trace = orc.generate(prompt, 25, repetition_penalty=1.7, return_logits=True)
    _check(gpu.generate(prompt, 25, repetition_penalty=1.7), exp, trace)
    # the processors run on the device by default (counts per token + n-gram windows, llm_kernels.hip); on a host copy of the
    # logits (the checker) the tokens are the same -- repeated tokens in the prompt (9, 9) are penalised once per occurrence
    for kw in (dict(repetition_penalty=1.3, no_repeat_ngram=2), dict(repetition_penalty=1.7), dict(no_repeat_ngram=1),
               dict(repetition_penalty=0.8, no_repeat_ngram=3)):
        on_device = gpu.generate(prompt, 30, **kw)
        gpu.set_device_sampling(False)
        on_host = gpu.generate(prompt, 30, **kw)
        gpu.set_device_sampling(True)
        assert on_device == on_host, kw


def test_stop_tokens_and_context_limit(tmp_path):
    import kjarni_amd
    # an lm head whose argmax is always token 2 (an eos id): generation stops at once, nothing is emitted
    d = str(tmp_path / "eos")
    cfg, t = synth.llm_model(d, synth.LLAMA_TEST, seed=5)
    orc = L.LlmOracle(t, cfg)
    gpu = kjarni_amd.HipDecoder(d)
    prompt = [1, 10, 20]
    exp = orc.generate(prompt, 300)
    got = gpu.generate(prompt, 300)
    assert got[:len(exp)] == exp[:len(got)]
    assert all(tok not in (2, 3) for tok in got)            # eos ids are never emitted (generator.rs:344-347)
    # context limit: prompt + generated never exceeds max_position_embeddings (generator.rs:319-322)
    long_prompt = list(range(4, 4 + 250))
    got = gpu.generate(long_prompt, 50)
    exp = orc.generate(long_prompt, 50)
    assert len(got) <= 256 - 250 and got == exp[:len(got)] and len(got) == len(exp)
    small = kjarni_amd.HipDecoder(d, max_context=64)
    assert small.context == 64
    with pytest.raises(Exception):
        small.generate(list(range(4, 80)), 5)               # prompt longer than the context
    with pytest.raises(Exception):
        gpu.generate([], 5)                                 # "cannot generate from empty prompt"


def test_bf16_weights(tmp_path):
    """Weights stored as BF16 stay bf16 in HBM; the arithmetic is the reference's (bf16 weight widened, f32 accumulate),
    so results equal the f32 path on the same (bf16-representable) values to 1e-4."""
    import kjarni_amd
    d16 = str(tmp_path / "bf16")
    cfg, t = synth.llm_model(d16, synth.LLAMA_TEST, seed=6, store_bf16=True)
    orc = L.LlmOracle(t, cfg)                               # t holds the bf16-rounded values as f32
    g16 = kjarni_amd.HipDecoder(d16)
    assert g16.bf16 and g16.weight_bytes < 0.6 * sum(v.nbytes for v in t.values())
    g32 = kjarni_amd.HipDecoder(d16, weights="f32")
    assert not g32.bf16
    rng = np.random.default_rng(1)
    ids = rng.integers(4, cfg["vocab_size"], 13).tolist()
    cache = orc.new_cache()
    ref = orc.forward(ids, cache)[0]
    for g in (g16, g32):
        g.reset()
        h, logits = g.forward(ids)
        assert np.abs(h - ref[-5:]).max() < TOL and np.abs(logits - orc.logits(ref[-1])).max() < TOL
    prompt = [1, 5, 9, 200]
    assert g16.generate(prompt, 30) == g32.generate(prompt, 30)
    # f32 file, bf16 requested: weights are rounded to nearest even on load
    d32 = str(tmp_path / "f32")
    cfg2, t2 = synth.llm_model(d32, synth.QWEN_TEST, seed=7)
    rounded = {k: (L.bf16_round(v) if v.ndim == 2 else v) for k, v in t2.items()}
    orc2 = L.LlmOracle(rounded, cfg2)
    g = kjarni_amd.HipDecoder(d32, weights="bf16")
    cache = orc2.new_cache()
    ids = rng.integers(4, cfg2["vocab_size"], 6).tolist()
    h, logits = g.forward(ids)
    assert np.abs(h - orc2.forward(ids, cache)[0]).max() < TOL


def _close(got, want):
    """1e-4 absolute on O(1) values, plus 1e-5 of the LARGEST logit: a 4096-term f32 dot product whose terms reach |logit| ~ 15
    carries that much rounding whatever its own value is (the error follows the sum of |terms|, not the result), and the
    oracle's single K chain and the prompt GEMMs' up-to-8 K slices do not round together.  Hidden states are held to 1e-4
    absolute separately."""
    got, want = np.asarray(got), np.asarray(want)
    return bool(np.all(np.abs(got - want) <= TOL + 1e-5 * np.abs(want).max()))


GEOMETRIES = {
    # head_dim 64 and 128 (the production sizes), GQA groups of 2 and 4, hidden sizes that take 1 and 2 K-chunks per lane
    "d64-gqa2": dict(synth.LLAMA_TEST, hidden_size=256, num_attention_heads=4, num_key_value_heads=2, intermediate_size=512, head_dim=64),
    "d128-gqa4": dict(synth.LLAMA_TEST, hidden_size=512, num_attention_heads=4, num_key_value_heads=1, intermediate_size=1024, head_dim=128,
                      num_hidden_layers=1),
    "d32-wide": dict(synth.QWEN_TEST, hidden_size=4096, num_attention_heads=128, num_key_value_heads=8, intermediate_size=8192 + 32 * 8, head_dim=32,
                     num_hidden_layers=1, vocab_size=320),
    # the registry's odd head counts: Qwen2.5-0.5B (14 query heads over 2 KV heads, hidden 896, inner 4864, biases, tied
    # embeddings) and Llama-3.2-3B (24 heads over 8, hidden 3072)
    "qwen0.5b-heads14-kv2": dict(synth.QWEN_TEST, hidden_size=896, num_attention_heads=14, num_key_value_heads=2, intermediate_size=4864,
                                 head_dim=64, num_hidden_layers=1, vocab_size=512, tie_word_embeddings=True),
    "llama3b-heads24-kv8": dict(synth.LLAMA_TEST, hidden_size=3072, num_attention_heads=24, num_key_value_heads=8, intermediate_size=8192,
                                head_dim=128, num_hidden_layers=1, vocab_size=320),
}


@pytest.mark.parametrize("name", sorted(GEOMETRIES))
def test_production_head_sizes_prefill_and_decode(tmp_path, name):
    """The matrix-core prefill (64 x 64 GEMM tiles, flash-style causal attention with 16 / 32 / 64 / 128-wide heads), the
    split-K decode GEMVs (1, 2 and 16-wave variants: k = 256 ... 8448) and the fused norm + QKV + RoPE step."""
    base = GEOMETRIES[name]
    orc, gpu, cfg = _pair(tmp_path, base, seed=9)
    rng = np.random.default_rng(2)
    ids = rng.integers(4, cfg["vocab_size"], 70).tolist()   # >= 24 rows: the GEMM route, two 32-query attention blocks + a tail
    cache = orc.new_cache()
    ref_h = orc.forward(ids, cache)[0]
    h, logits = gpu.forward(ids)
    k = (len(ids) - 1) % 8 + 1
    assert np.abs(h[-k:] - ref_h[-k:]).max() < TOL
    assert _close(logits, orc.logits(ref_h[-1]))
    for _ in range(3):                                        # single-token steps on top of the GEMM-filled cache
        t = [int(rng.integers(4, cfg["vocab_size"]))]
        ref_h = orc.forward(t, cache)[0]
        h, logits = gpu.forward(t)
        assert np.abs(h[-1] - ref_h[-1]).max() < TOL and _close(logits, orc.logits(ref_h[-1]))
    more = rng.integers(4, cfg["vocab_size"], 40).tolist()  # a second long block: attention over old + new keys
    ref_h = orc.forward(more, cache)[0]
    h, logits = gpu.forward(more)
    assert _close(logits, orc.logits(ref_h[-1]))
    gpu.reset()
    exp = orc.generate(ids[:30], 6)
    assert gpu.generate(ids[:30], 6) == exp


@pytest.mark.parametrize("store_bf16", [True, False], ids=["bf16-weights", "f32-weights"])
def test_full_width_decode_step(tmp_path, store_bf16):
    """The production widths of the decode step (hidden 2048, inner 8192: the Llama-3.2-1B row lengths, one layer, a
    vocabulary of 20 011 rows so that the head takes its looping form): single-token steps go through the weight-streaming GEMV (K = 2048 and 8192, SwiGLU pair, residual,
    RMSNorm folded in, the attention slabs merged by the output projection, the final norm by the vocabulary head), prompt
    blocks through the matrix-core route; both against the oracle."""
    base = dict(synth.LLAMA_TEST, hidden_size=2048, num_hidden_layers=1, num_attention_heads=32, num_key_value_heads=8,
                intermediate_size=8192, vocab_size=20011, max_position_embeddings=512, head_dim=64)  # > 16384 rows: the head loops
    base["rope_scaling"] = dict(base["rope_scaling"], original_max_position_embeddings=128)
    orc, gpu, cfg = _pair(tmp_path, base, seed=9, bf16_values=True, store_bf16=store_bf16)
    rng = np.random.default_rng(1)
    cache = orc.new_cache()
    gpu.reset()
    for n in (7, 1, 1, 1, 30, 1):
        ids = rng.integers(4, cfg["vocab_size"], n).tolist()
        ref_h = orc.forward(ids, cache)[0]
        h, logits = gpu.forward(ids)
        k = (n - 1) % 8 + 1                                     # the rows of the last 8-row block come back
        ref_l = orc.logits(ref_h[-1])
        scale = max(1.0, float(np.abs(ref_h).max()))
        assert np.abs(h[-k:] - ref_h[-k:]).max() < TOL * scale, np.abs(h[-k:] - ref_h[-k:]).max()
        assert np.abs(logits - ref_l).max() < TOL * max(1.0, float(np.abs(ref_l).max()))


def test_decode_over_several_key_ranges(tmp_path):
    """A 2 048-token context gives the decode attention four key ranges per head, merged inside the output projection:
    steps over 8 keys (three ranges empty), ~130 keys (register-held ranges) and ~700 keys (176 keys per range: the looping
    path), each against the oracle; the 690-token prompt in between goes through the prompt kernels."""
    base = dict(synth.LLAMA_TEST, hidden_size=2048, num_hidden_layers=1, num_attention_heads=32, num_key_value_heads=8,
                intermediate_size=8192, vocab_size=1003, max_position_embeddings=2048, head_dim=64)
    base["rope_scaling"] = dict(base["rope_scaling"], original_max_position_embeddings=512)
    orc, gpu, cfg = _pair(tmp_path, base, seed=11, bf16_values=True, store_bf16=True)
    rng = np.random.default_rng(5)
    cache = orc.new_cache()
    gpu.reset()
    for n in (7, 1, 1, 120, 1, 1, 560, 1, 1, 1):
        ids = rng.integers(4, cfg["vocab_size"], n).tolist()
        ref_h = orc.forward(ids, cache)[0]
        h, logits = gpu.forward(ids)
        k = (n - 1) % 8 + 1
        ref_l = orc.logits(ref_h[-1])
        scale = max(1.0, float(np.abs(ref_h).max()))
        assert np.abs(h[-k:] - ref_h[-k:]).max() < TOL * scale, (n, np.abs(h[-k:] - ref_h[-k:]).max())
        assert np.abs(logits - ref_l).max() < TOL * max(1.0, float(np.abs(ref_l).max())), (n, np.abs(logits - ref_l).max())


@pytest.mark.parametrize("store_bf16", [True, False], ids=["bf16-weights", "f32-weights"])
def test_long_prompt_blocks_take_the_tile_gemm(tmp_path, store_bf16):
    """A projection of a prompt block runs a 128 x 128-tile GEMM when its tiles number at least 208 (f32 weights: the encoder's
    f32 MFMA tiles; bf16 weights: the bf16-matrix-core tiles on three exact pieces of every activation, gemm_split.hip), else
    the 64 x 64 prompt kernel (f32 weights on the f32 matrix cores, bf16 weights on the bf16 ones).  Hidden 1 792 = 14 x 128 with a 1 792-wide FFN: a 2 048-row
    block has 16 x 14 = 224 tiles for q, o (in-place residual), gate, up * silu(gate) and down (in-place residual) -- five tiled
    projections per layer -- while k / v (256 wide: 32 tiles) and the 1 900-row prompt (15 x 14 = 210 ... also tiled) and the
    252-row tail (not tiled) take their own routes.  The route is asserted through the library's counter."""
    base = dict(synth.LLAMA_TEST, hidden_size=1792, num_hidden_layers=2, num_attention_heads=14, num_key_value_heads=2,
                intermediate_size=1792, vocab_size=777, max_position_embeddings=4096, head_dim=128)
    base["rope_scaling"] = dict(base["rope_scaling"], original_max_position_embeddings=1024)
    orc, gpu, cfg = _pair(tmp_path, base, seed=13, bf16_values=True, store_bf16=store_bf16, std=0.02)
    rng = np.random.default_rng(6)
    assert gpu.tile_gemm_calls() == 0
    for n_prompt, tiled_blocks in ((1500, 0), (2300, 1)):
        cache = orc.new_cache()
        gpu.reset()
        before = gpu.tile_gemm_calls()
        for n in (n_prompt, 1, 1):
            ids = rng.integers(4, cfg["vocab_size"], n).tolist()
            ref_h = orc.forward(ids, cache)[0]
            h, logits = gpu.forward(ids)
            k = (n - 1) % 8 + 1
            ref_l = orc.logits(ref_h[-1])
            scale = max(1.0, float(np.abs(ref_h).max()))
            assert np.abs(h[-k:] - ref_h[-k:]).max() < TOL * scale, (n, np.abs(h[-k:] - ref_h[-k:]).max())
            assert np.abs(logits - ref_l).max() < TOL * max(1.0, float(np.abs(ref_l).max())), (n, np.abs(logits - ref_l).max())
        # 1 500 rows: 12 x 14 = 168 tiles per projection, below the threshold; 2 300 rows = a 2 048-row block (224 tiles: q, o,
        # gate, up, down tiled in each of the 2 layers) + a 252-row block (below the 512-row floor)
        assert gpu.tile_gemm_calls() - before == tiled_blocks * 5 * cfg["num_hidden_layers"], (n_prompt, gpu.tile_gemm_calls() - before)


@pytest.mark.parametrize("head_dim,heads,kv_heads", [(64, 8, 2), (128, 4, 2), (128, 4, 4)], ids=["d64-gqa4", "d128-gqa2", "d128-mha"])
def test_prompt_attention_on_the_matrix_cores(tmp_path, head_dim, heads, kv_heads):
    """Prompt blocks of >= 256 rows with 64- / 128-wide heads take the MFMA causal attention kernel: a first block (no keys
    before it), a second one on top of the cache (base > 0, a block that ends inside a 128-key chunk), then single tokens."""
    base = dict(synth.LLAMA_TEST, hidden_size=heads * head_dim, num_hidden_layers=2, num_attention_heads=heads, num_key_value_heads=kv_heads,
                intermediate_size=512, vocab_size=600, max_position_embeddings=2048, head_dim=head_dim)
    base["rope_scaling"] = dict(base["rope_scaling"], original_max_position_embeddings=512)
    orc, gpu, cfg = _pair(tmp_path, base, seed=17)
    rng = np.random.default_rng(8)
    cache = orc.new_cache()
    gpu.reset()
    for n in (300, 1, 457, 1, 1):
        ids = rng.integers(4, cfg["vocab_size"], n).tolist()
        ref_h = orc.forward(ids, cache)[0]
        h, logits = gpu.forward(ids)
        k = (n - 1) % 8 + 1
        ref_l = orc.logits(ref_h[-1])
        scale = max(1.0, float(np.abs(ref_h).max()))
        assert np.abs(h[-k:] - ref_h[-k:]).max() < TOL * scale, (n, np.abs(h[-k:] - ref_h[-k:]).max())
        assert np.abs(logits - ref_l).max() < TOL * max(1.0, float(np.abs(ref_l).max())), (n, np.abs(logits - ref_l).max())
