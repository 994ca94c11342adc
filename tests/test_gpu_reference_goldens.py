"""The reference's OWN model-free golden vectors through the HIP operators (tests/test_oracle_goldens.py replays the same
vectors on the CPU oracle; here nothing but the C-ABI operator entry points -- kjarni_hip_op_* / kjarni_hip_cosine_search_host --
touches the numbers, the oracle is not involved).

Each test names the reference test it reproduces (paths relative to /root/reference/crates/kjarni-transformers/src unless
stated).  Tolerances are the reference's own (1e-4 for the layer / FFN goldens, 1e-5 for pooling, 1e-3 where its expected
values are written to four digits)."""
import numpy as np
import pytest

from tests.test_oracle_goldens import (GOLDEN_IN, GOLDEN_MASK, GOLDEN_POS_BIAS, GOLDEN_POSTNORM, GOLDEN_PRENORM,
                                       deterministic_layer)

pytestmark = pytest.mark.gpu


def hip_encoder_layer(t, hidden_in, mask, pos_bias, heads, prenorm, mask_value):
    """EncoderLayer::forward / forward_noalloc (cpu/encoder/encoder_layer.rs:100-176) composed from the HIP operators:
    fused [3H, H] QKV projection (qkv_projection.rs:30-41), attention with the additive position bias, output projection +
    residual, LayerNorm, FC1 + erf-GELU, FC2 + residual, LayerNorm -- post-norm or pre-norm order."""
    from kjarni_amd import ops
    B, S, H = hidden_in.shape
    x = hidden_in.reshape(B * S, H).astype(np.float32)
    wqkv = np.concatenate([t["wq"], t["wk"], t["wv"]], axis=0)
    bqkv = np.concatenate([t["bq"], t["bk"], t["bv"]], axis=0)
    eps = 1e-5

    def attn(inp):
        qkv, _ = ops.linear(inp, wqkv, bqkv, None, ops.EPI_BIAS)
        return ops.attention_biased(qkv.reshape(B, S, 3 * H), mask.astype(np.uint32), heads, position_bias=pos_bias,
                                    mask_value=float(mask_value)).reshape(B * S, H)

    def ffn_out(inp, residual):
        mid, _ = ops.linear(inp, t["w1"], t["b1"], None, ops.EPI_BIAS_GELU)
        out, _ = ops.linear(mid, t["w2"], t["b2"], residual, ops.EPI_BIAS_RESIDUAL)
        return out

    if prenorm:
        n1, _ = ops.layer_norm(x, t["ln1_g"], t["ln1_b"], eps)
        x, _ = ops.linear(attn(n1), t["wo"], t["bo"], x, ops.EPI_BIAS_RESIDUAL)
        n2, _ = ops.layer_norm(x, t["ln2_g"], t["ln2_b"], eps)
        x = ffn_out(n2, x)
    else:
        a, _ = ops.linear(attn(x), t["wo"], t["bo"], x, ops.EPI_BIAS_RESIDUAL)
        x, _ = ops.layer_norm(a, t["ln1_g"], t["ln1_b"], eps)
        x, _ = ops.layer_norm(ffn_out(x, x), t["ln2_g"], t["ln2_b"], eps)
    return x.reshape(B, S, H)


@pytest.mark.parametrize("mask_value", [-1e9, -np.inf])
def test_encoder_layer_golden_postnorm(mask_value):
    # cpu/encoder/encoder_layer.rs:395-448 test_golden_postnorm_noalloc, :733-780 test_golden_postnorm (hidden 4, 2 heads, bias)
    t = deterministic_layer(4, 8, 2)
    out = hip_encoder_layer(t, GOLDEN_IN, GOLDEN_MASK, GOLDEN_POS_BIAS, 2, False, mask_value)
    assert float(np.abs(out - GOLDEN_POSTNORM).max()) < 1e-4


@pytest.mark.parametrize("mask_value", [-1e9, -np.inf])
def test_encoder_layer_golden_prenorm(mask_value):
    # cpu/encoder/encoder_layer.rs:349-392 test_golden_prenorm_noalloc, :694-730 test_golden_prenorm
    t = deterministic_layer(4, 8, 2)
    out = hip_encoder_layer(t, GOLDEN_IN, GOLDEN_MASK, GOLDEN_POS_BIAS, 2, True, mask_value)
    assert float(np.abs(out - GOLDEN_PRENORM).max()) < 1e-4


def test_position_bias_is_added_after_the_scale_and_before_the_mask():
    """encoder_self_attention.rs:103-116, 244-262: scores = (Q K^T) * scale + bias, THEN the padding mask overwrites.  Against a
    float64 evaluation, on a shape the reference's goldens do not reach (3 heads of 5, 7 tokens, a masked tail, a bias slab
    larger than the sequence -- it is sliced [.., ..seq, ..seq])."""
    from kjarni_amd import ops
    rng = np.random.default_rng(0)
    B, S, heads, d, SB = 2, 7, 3, 5, 9
    H = heads * d
    qkv = rng.standard_normal((B, S, 3 * H)).astype(np.float32)
    bias = rng.standard_normal((1, heads, SB, SB)).astype(np.float32)
    mask = np.ones((B, S), np.uint32)
    mask[1, 5:] = 0
    for scale_qk in (True, False):
        got = ops.attention_biased(qkv, mask, heads, position_bias=bias, scale_qk=scale_qk, mask_value=-1e9)
        q, k, v = (qkv[..., i * H:(i + 1) * H].astype(np.float64).reshape(B, S, heads, d).transpose(0, 2, 1, 3) for i in range(3))
        s = q @ k.transpose(0, 1, 3, 2) * (1.0 / np.sqrt(d) if scale_qk else 1.0) + bias[:, :, :S, :S].astype(np.float64)
        s = np.where(mask[:, None, None, :] != 0, s, -1e9)
        p = np.exp(s - s.max(-1, keepdims=True))
        p /= p.sum(-1, keepdims=True)
        ref = (p @ v).transpose(0, 2, 1, 3).reshape(B, S, H)
        assert float(np.abs(got - ref).max()) < 1e-5
    # without a bias the entry equals the model path's operator
    plain, _ = ops.attention(qkv, mask, heads, mask_value=-1e9)
    assert float(np.abs(ops.attention_biased(qkv, mask, heads, mask_value=-1e9) - plain).max()) < 1e-6


def test_ffn_golden_gelu():
    # cpu/feedforward/standard_new.rs:155-191 test_ffn_golden_values_gelu (PyTorch-derived)
    from kjarni_amd import ops
    x = np.array([0.5, -0.2, 0.1, -0.5, 0.0, 0.8], np.float32).reshape(2, 3)
    w1 = np.array([0.4414, 0.4792, -0.1353, 0.5304, -0.1265, 0.1165, -0.2811, 0.3391, 0.509,
                   -0.4236, 0.5018, 0.1081], np.float32).reshape(4, 3)
    w2 = np.array([0.3694, 0.0677, 0.2411, -0.0706, 0.3854, 0.0739, -0.2334, 0.1274, -0.2304,
                   -0.0586, -0.2031, 0.3317], np.float32).reshape(3, 4)
    mid, _ = ops.linear(x, w1, None, None, ops.EPI_BIAS_GELU)
    out, _ = ops.linear(mid, w2, None, None, ops.EPI_BIAS)
    exp = np.array([0.0266, 0.0386, -0.0491, 0.0304, -0.1196, 0.0148], np.float32).reshape(2, 3)
    assert float(np.abs(out - exp).max()) <= 1e-4


def test_ffn_relu():
    # standard_new.rs:~120-150: identity FC1, x = [1, -1], relu, FC2 = 2 I -> [2, 0]
    from kjarni_amd import ops
    x = np.array([[1.0, -1.0]], np.float32)
    mid, _ = ops.linear(x, np.eye(2, dtype=np.float32), None, None, ops.EPI_BIAS_RELU)
    out, _ = ops.linear(mid, 2 * np.eye(2, dtype=np.float32), None, None, ops.EPI_BIAS)
    assert float(np.abs(out - np.array([[2.0, 0.0]])).max()) <= 1e-6


def test_layer_norm_reference_cases():
    # cpu/normalization/layer_norm.rs:228-307
    from kjarni_amd import ops
    one3, zero3 = np.ones(3, np.float32), np.zeros(3, np.float32)
    y, _ = ops.layer_norm(np.array([[1.0, 2.0, 3.0]], np.float32), one3, zero3, 1e-6)
    assert abs(float(y.mean())) < 1e-5
    assert abs(y[0, 0] + 1.2247) < 1e-3 and abs(y[0, 1]) < 1e-5 and abs(y[0, 2] - 1.2247) < 1e-3
    g, b = np.array([2.0, 0.5, 1.5], np.float32), np.array([1.0, -1.0, 0.5], np.float32)
    y, _ = ops.layer_norm(np.array([[1.0, 2.0, 3.0]], np.float32), g, b, 1e-6)
    std = np.sqrt(2.0 / 3.0 + 1e-6)
    exp = np.array([(1 - 2) / std * 2 + 1, (2 - 2) / std * 0.5 - 1, (3 - 2) / std * 1.5 + 0.5])
    assert float(np.abs(y[0] - exp).max()) < 1e-4
    x = np.array([1, 3, 2, 4, 5, 7, 6, 8], np.float32).reshape(4, 2)   # the reference's [2, 2, 2] batch, row by row
    y, _ = ops.layer_norm(x, np.ones(2, np.float32), np.zeros(2, np.float32), 1e-5)
    assert abs(y[0, 0] + 1.0) < 1e-3 and abs(y[0, 1] - 1.0) < 1e-3
    y, _ = ops.layer_norm(np.array([[1.0, 2.0, 3.0, 4.0]], np.float32), np.ones(4, np.float32), np.zeros(4, np.float32), 1e-5)
    assert float(np.abs(y[0] - np.array([-1.3416, -0.4472, 0.4472, 1.3416])).max()) < 1e-3


def test_pooling_goldens():
    # cpu/encoder/traits.rs:796-895 test_pooling_strategies_golden (mask from MockGoldenEncoder)
    from kjarni_amd import ops
    hs = np.array([
        -1.331580, -0.437194, 0.457193, 1.351581, -1.331581, -0.437193, 0.457194, 1.351580,
        -1.331581, -0.437194, 0.457194, 1.351581, -1.331581, -0.437194, 0.457194, 1.351581,
        -1.331581, -0.437193, 0.457194, 1.351580, -1.331580, -0.437194, 0.457193, 1.351581,
        -1.331581, -0.437193, 0.457193, 1.351581, -1.331581, -0.437193, 0.457194, 1.351580,
        -1.331581, -0.437193, 0.457193, 1.351581, -1.331581, -0.437193, 0.457193, 1.351581,
    ], np.float32).reshape(2, 5, 4)
    mask = np.ones((2, 5), np.uint32)
    mean = ops.pool(hs, mask, ops.POOL_MEAN)
    exp_mean = np.array([-1.331581, -0.437193, 0.457194, 1.351580, -1.331581, -0.437193, 0.457193, 1.351581], np.float32).reshape(2, 4)
    assert float(np.abs(mean - exp_mean).max()) < 1e-5
    exp_cls = np.array([-1.331580, -0.437194, 0.457193, 1.351581] * 2, np.float32).reshape(2, 4)
    assert float(np.abs(ops.pool(hs, mask, ops.POOL_CLS) - exp_cls).max()) < 1e-5
    exp_max = np.array([-1.331580, -0.437193, 0.457194, 1.351581] * 2, np.float32).reshape(2, 4)
    assert float(np.abs(ops.pool(hs, mask, ops.POOL_MAX) - exp_max).max()) < 1e-5
    exp_norm = np.array([-0.665787, -0.218596, 0.228596, 0.675787, -0.665787, -0.218595, 0.228595, 0.675787], np.float32).reshape(2, 4)
    assert float(np.abs(ops.pool(hs, mask, ops.POOL_MEAN, normalize=True) - exp_norm).max()) < 1e-5


def test_pooling_unit_cases():
    # pooling/mod.rs:70-153
    from kjarni_amd import ops
    hidden = np.array([[[1, 2], [3, 4]], [[5, 6], [7, 8]]], np.float32)
    mask = np.array([[1, 1], [1, 0]], np.uint32)
    assert float(np.abs(ops.pool(hidden, mask, ops.POOL_MEAN) - np.array([[2, 3], [5, 6]])).max()) < 1e-6
    assert (ops.pool(hidden, mask, ops.POOL_CLS) == np.array([[1, 2], [5, 6]])).all()
    assert float(np.abs(ops.pool(hidden, mask, ops.POOL_MAX) - np.array([[3, 4], [5, 6]])).max()) < 1e-6
    h3 = np.array([[[1, 2], [3, 4], [5, 6]], [[7, 8], [9, 10], [11, 12]]], np.float32)
    m3 = np.array([[1, 1, 0], [1, 1, 1]], np.uint32)
    assert (ops.pool(h3, m3, ops.POOL_LAST) == np.array([[3, 4], [11, 12]])).all()
    # empty sequence (all masked): the mean divides by max(count, 1e-9) -> token sums of nothing = 0 ... the reference's
    # mean_pool clamps the count to 1e-9, so an all-masked row is 0 / 1e-9 = 0 (pooling/mod.rs:22-28); max gives -1e9
    assert (ops.pool(np.array([[[1, 2]]], np.float32), np.array([[0]], np.uint32), ops.POOL_MAX) == -1e9).all()
    # l2 normalisation (cpu/encoder/traits.rs:~783-794 test_l2_normalize_inplace): rows [3, 4], [1, 1], zeros
    one_tok = np.array([[[3, 4]], [[1, 1]], [[0, 0]]], np.float32)
    d = ops.pool(one_tok, None, ops.POOL_CLS, normalize=True)
    assert abs(d[0, 0] - 0.6) < 1e-6 and abs(d[0, 1] - 0.8) < 1e-6 and abs(d[1, 0] - 1 / np.sqrt(2)) < 1e-6
    assert (d[2] == 0).all()


def test_cosine_and_search_reference_cases():
    # kjarni-search/src/vector.rs:169-433 through kjarni_hip_cosine_search_host (VectorStore mode)
    import kjarni_amd
    f = lambda a: np.asarray(a, np.float32)  # noqa: E731

    def cos(a, b):
        idx, sc = kjarni_amd.cosine_search(f(a), f([b]), 1)
        return float(sc[0, 0])
    assert abs(cos([1, 2, 3], [1, 2, 3]) - 1.0) < 1e-6
    assert abs(cos([1, 0], [0, 1])) < 1e-6
    assert abs(cos([1, 2, 3], [-1, -2, -3]) + 1.0) < 1e-6
    assert abs(cos([0, 0, 0], [1, 2, 3])) < 1e-6                                  # zero query: max(denominator, 1e-9)
    idx, sc = kjarni_amd.cosine_search(f([1, 0, 0]), f([[1, 0, 0], [0.9, 0.1, 0], [0, 1, 0]]), 10)
    assert list(idx[0]) == [0, 1, 2] and sc[0, 0] >= sc[0, 1] >= sc[0, 2]
    idx, _ = kjarni_amd.cosine_search(f([1, 0]), f([[1, 0], [0.9, 0.1], [0.8, 0.2], [0.7, 0.3], [0.6, 0.4]]), 3)
    assert idx.shape == (1, 3) and list(idx[0]) == [0, 1, 2]
    idx, _ = kjarni_amd.cosine_search(f([1, 0]), f([[1, 0], [0.9, 0.1]]), 10)    # k > n: every document
    assert idx.shape == (1, 2)
    idx, _ = kjarni_amd.cosine_search(f([1, 2]), np.zeros((0, 2), np.float32), 5)   # empty store
    assert idx.shape == (1, 0)
    idx, _ = kjarni_amd.cosine_search(f([1, 2]), f([[1, 2, 3]]), 5)                # dimension mismatch -> empty
    assert idx.shape == (1, 0)
    _, sc = kjarni_amd.cosine_search(f([1, 0]), f([[1, 0], [0.7, 0.7], [0, 1]]), 10)   # threshold case: 1.0, ~0.707, 0.0
    assert int((sc[0] >= 0.5).sum()) == 2 and abs(sc[0, 1] - 0.70710678) < 1e-6


def test_segment_scan_zero_guards():
    # kjarni-rag/src/segment.rs:307-371: zero query -> no hits; zero document -> score 0 (Segment mode)
    import kjarni_amd
    corpus = np.array([[0, 0], [1, 0], [0, 2]], np.float32)
    idx, sc = kjarni_amd.cosine_search(np.array([1, 0], np.float32), corpus, 3, mode=kjarni_amd.COSINE_SEGMENT)
    assert list(idx[0]) == [1, 0, 2] and sc[0, 0] == 1.0 and sc[0, 1] == 0.0 and sc[0, 2] == 0.0
    idx, sc = kjarni_amd.cosine_search(np.array([0, 0], np.float32), corpus, 3, mode=kjarni_amd.COSINE_SEGMENT)
    assert idx.shape[1] == 0 or (idx[0] < 0).all()   # |q| < 1e-9: no hits (segment.rs:315-317)
