"""Opt-in mode kjarni_hip_set_f32_on_bf16: the large-batch projections (more than 8 192 token rows) compute their f32 products
on the bf16 matrix cores -- every operand split exactly into three bf16 pieces, six of the nine cross products, f32
accumulation (gemm.hip, gemm_nt_f32_split).  Held to the same oracle and the same tolerances as the f32 MFMA kernels
(linear_layer.rs:160-282, feedforward/standard_new.rs:29-82, encoder_layer.rs:129-176), and to the default path to rounding."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def split_mode():
    from kjarni_amd import ops
    before = ops.set_f32_on_bf16(True)
    assert ops.get_f32_on_bf16()
    yield
    ops.set_f32_on_bf16(before)


@pytest.mark.parametrize("m,k,n", [(8193, 384, 1152), (8300, 1536, 384), (16384, 384, 1536), (9000, 768, 768), (8200, 3072, 768),
                                   (8193, 64, 128),
                                   # 6 144 .. 8 192 rows: in the mode the split kernel also replaces the 64 x 64-tile route
                                   (6144, 384, 1152), (7001, 1536, 384),
                                   # 257 .. 6 143 rows: the 64 x 64 tiles themselves on the bf16 matrix cores (with K slices for the
                                   # narrow outputs: 1 536 -> 384)
                                   (300, 384, 1152), (768, 384, 1152), (4096, 384, 1536), (1000, 1536, 384), (3000, 768, 768), (5000, 3072, 768)])
def test_projections_all_epilogues(split_mode, m, k, n):
    """Every epilogue of the split kernel against the oracle at 1e-5 (rows past the last full 128-row tile included), on sampled
    rows; and against float64: the error stays at the f32 kernels' level."""
    from kjarni_amd import ops
    rng = np.random.default_rng(m + n)
    x = rng.standard_normal((m, k)).astype(np.float32)
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    r = rng.standard_normal((m, n)).astype(np.float32)
    rows = np.unique(np.concatenate([np.arange(0, 40), np.arange(m - 200, m), rng.choice(m, 100, replace=False)]))
    lin = O.linear(x[rows], w, b, blocked=True)
    scale = max(1.0, float(np.abs(lin).max()))
    for epi, ref in ((ops.EPI_BIAS, lin), (ops.EPI_BIAS_GELU, O.activation(lin, O.ACT_GELU)),
                     (ops.EPI_BIAS_GELU_NEW, O.activation(lin, O.ACT_GELU_NEW)), (ops.EPI_BIAS_RELU, np.maximum(lin, 0.0)),
                     (ops.EPI_BIAS_TANH, np.tanh(lin))):
        got, _ = ops.linear(x, w, b, None, epi)
        assert float(np.abs(got[rows] - ref).max()) < 1e-5 * scale, epi
    got, _ = ops.linear(x, w, b, r, ops.EPI_BIAS_RESIDUAL)
    assert float(np.abs(got[rows] - (lin + r[rows])).max()) < 1e-5 * scale
    got, _ = ops.linear(x, w, None, None, ops.EPI_BIAS)
    f64 = x[rows].astype(np.float64) @ w.astype(np.float64).T
    assert float(np.abs(got[rows] - f64).max()) < 2e-6 * max(1.0, float(np.abs(f64).max())) * max(1.0, np.sqrt(k / 384.0))


def test_split_is_exact_on_bf16_representable_operands(split_mode):
    """Operands that are sums of at most three bf16 values with small integer products: every cross product and every partial
    sum is exact in f32, so the result must equal the integer arithmetic bit for bit -- the three pieces carry the whole
    significand and no product of kept pieces is lost."""
    from kjarni_amd import ops
    rng = np.random.default_rng(7)
    m, k, n = 8320, 128, 256
    x = rng.integers(-7, 8, (m, k)).astype(np.float32) * np.float32(2.0 ** -3) + rng.integers(-3, 4, (m, k)).astype(np.float32) * np.float32(2.0 ** -11)
    w = rng.integers(-5, 6, (n, k)).astype(np.float32) * np.float32(0.25)
    got, _ = ops.linear(x, w, None, None, ops.EPI_BIAS)
    ref = (x.astype(np.float64) @ w.astype(np.float64).T)
    assert np.array_equal(got, ref.astype(np.float32))


def test_residual_layernorm_route(split_mode):
    from kjarni_amd import ops
    rng = np.random.default_rng(11)
    for m, k, n in ((8400, 1536, 384), (6500, 1536, 384), (4100, 1536, 384), (900, 384, 384)):
        _residual_layernorm_case(ops, rng, m, k, n)


def _residual_layernorm_case(ops, rng, m, k, n):
    x = rng.standard_normal((m, k)).astype(np.float32)
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    r = (rng.standard_normal((m, n)) * 2 + 0.5).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(n)).astype(np.float32)
    rows = np.unique(np.concatenate([np.arange(0, 64), np.arange(m - 150, m)]))
    ref = O.layer_norm(O.linear(x[rows], w, b, blocked=True) + r[rows], g, beta, 1e-12)
    got, _ = ops.linear_layer_norm(x, w, b, r, g, beta, 1e-12)
    assert float(np.abs(got[rows] - ref).max()) < 1e-5 * max(1.0, float(np.abs(ref).max()))


def test_forward_equals_default_path_and_oracle(tmp_path):
    """A 2-layer MiniLM-shaped model, 100 x 128 tokens (12 800 rows: the large-batch route): embeddings in the opt-in mode against
    the default path (rounding) and the oracle (1e-4)."""
    import kjarni_amd
    from kjarni_amd import ops
    cfg, t = synth.minilm_embedder(str(tmp_path / "m"), seed=3, num_hidden_layers=2)
    enc = kjarni_amd.HipEncoder(str(tmp_path / "m"), 0)
    ids, mask = synth.synthetic_ids(100, 128, seed=9)
    base = enc.embed(ids, mask)
    before = ops.set_f32_on_bf16(True)
    try:
        got = enc.embed(ids, mask)
    finally:
        ops.set_f32_on_bf16(before)
    assert float(np.abs(got - base).max()) < 1e-5
    orc = O.OracleModel(t, cfg, blocked_gemm=True)
    sel = [0, 1, 50, 99]
    assert float(np.abs(got[sel] - orc.embed_batch(ids[sel], mask[sel])).max()) < 1e-4
    enc.close()


@pytest.mark.parametrize("m,k,n", [(2048, 2048, 2048), (513, 2048, 512), (700, 8192, 2048), (128, 64, 128), (1000, 4096, 1024)])
def test_bf16_weight_projections(m, k, n):
    """The decoder's prompt projections: f32 activations (three exact bf16 pieces) x bf16 weights (as they are) on the bf16 matrix
    cores, against float64 on the same bf16 weights at the f32 GEMM's error level, with every epilogue the decoder uses; and bit
    for bit where pieces, products and partial sums are exactly representable."""
    from kjarni_amd import ops
    rng = np.random.default_rng(m + k)
    x = rng.standard_normal((m, k)).astype(np.float32)
    w16 = ops.to_bf16((rng.standard_normal((n, k)) * 0.03).astype(np.float32))
    w = ops.bf16_to_f32(w16)
    b = rng.standard_normal(n).astype(np.float32)
    r = rng.standard_normal((m, n)).astype(np.float32)
    rows = np.unique(np.concatenate([np.arange(0, 40), np.arange(max(0, m - 140), m), rng.choice(m, 60, replace=False)]))
    lin = x[rows].astype(np.float64) @ w.astype(np.float64).T + b
    tol = 3e-6 * max(1.0, float(np.abs(lin).max())) * max(1.0, np.sqrt(k / 384.0))
    got, _ = ops.linear_bf16_weights(x, w16, b, None, ops.EPI_BIAS)
    assert float(np.abs(got[rows] - lin).max()) < tol
    got, _ = ops.linear_bf16_weights(x, w16, b, r, ops.EPI_BIAS_RESIDUAL)
    assert float(np.abs(got[rows] - (lin + r[rows])).max()) < tol
    got, _ = ops.linear_bf16_weights(x, w16, None, r, ops.EPI_BIAS_MUL_SILU)
    g = r[rows].astype(np.float64)
    silu = np.where(g <= -20.0, 0.0, np.where(g >= 20.0, g, g / (1.0 + np.exp(-g))))
    assert float(np.abs(got[rows] - silu * (lin - b)).max()) < tol * max(1.0, float(np.abs(silu).max()))
    xi = rng.integers(-7, 8, (m, k)).astype(np.float32) * np.float32(2.0 ** -3) + rng.integers(-3, 4, (m, k)).astype(np.float32) * np.float32(2.0 ** -11)
    wi = ops.to_bf16(rng.integers(-5, 6, (n, k)).astype(np.float32) * np.float32(0.25))
    got, _ = ops.linear_bf16_weights(xi, wi, None, None, ops.EPI_BIAS)
    assert np.array_equal(got, (xi.astype(np.float64) @ ops.bf16_to_f32(wi).astype(np.float64).T).astype(np.float32))
