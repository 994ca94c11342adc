"""Pins the CPU oracle against the reference's own model-free golden tests.

Each test names the reference test it reproduces (paths relative to
/root/reference/crates/kjarni-transformers/src unless stated)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O


def deterministic_layer(hidden, inter, heads):
    """cpu/encoder/encoder_layer.rs:244-307 create_deterministic_layer:
    weights = arange(1..)*0.001 in q,k,v,o,fc1,fc2 order, biases 0.01,
    LN gamma=1 beta=0.01 eps=1e-5, erf GELU."""
    count = [1]

    def w(rows, cols):
        n = rows * cols
        a = (np.arange(count[0], count[0] + n, dtype=np.float64).astype(np.float32)
             * np.float32(0.001)).reshape(rows, cols)
        count[0] += n
        return a

    b = lambda n: np.full(n, 0.01, np.float32)
    t = {}
    t["wq"], t["bq"] = w(hidden, hidden), b(hidden)
    t["wk"], t["bk"] = w(hidden, hidden), b(hidden)
    t["wv"], t["bv"] = w(hidden, hidden), b(hidden)
    t["wo"], t["bo"] = w(hidden, hidden), b(hidden)
    t["ln1_g"], t["ln1_b"] = np.ones(hidden, np.float32), b(hidden)
    t["w1"], t["b1"] = w(inter, hidden), b(inter)
    t["w2"], t["b2"] = w(hidden, inter), b(hidden)
    t["ln2_g"], t["ln2_b"] = np.ones(hidden, np.float32), b(hidden)
    return t


def run_layer(t, hidden_in, mask, pos_bias, heads, inter, prenorm, mask_value):
    B, S, H = hidden_in.shape
    layer = O.KoLayer()
    for k, v in t.items():
        setattr(layer, k, O._f(v))
    m = O.KoModel()
    m.hidden, m.layers, m.heads, m.inter = H, 1, heads, inter
    m.act, m.prenorm, m.scale_embeddings, m.scale_qk = O.ACT_GELU, int(prenorm), 0, 1
    m.eps = 1e-5
    h = O.f32(hidden_in).copy()
    O.lib().ko_encoder_layer(C.byref(m), C.byref(layer), O._f(h), O._f(O.f32(mask)),
                             O._f(O.f32(pos_bias)), B, S, float(mask_value))
    return h


GOLDEN_IN = (np.arange(24, dtype=np.float32) * np.float32(0.1)).reshape(2, 3, 4)
GOLDEN_MASK = np.array([[1, 1, 1], [1, 1, 0]], np.float32)
GOLDEN_POS_BIAS = (np.arange(18, dtype=np.float32) * np.float32(0.01)).reshape(1, 2, 3, 3)

GOLDEN_PRENORM = np.array([
    0.030466, 0.131298, 0.232129, 0.332961, 0.430466, 0.531298, 0.632130, 0.732961,
    0.830467, 0.931298, 1.032130, 1.132961, 1.230467, 1.331298, 1.432130, 1.532961,
    1.630466, 1.731298, 1.832129, 1.932961, 2.030467, 2.131298, 2.232130, 2.332961,
], np.float32).reshape(2, 3, 4)

GOLDEN_POSTNORM = np.array([
    -1.331634, -0.437211, 0.457211, 1.351634, -1.331634, -0.437212, 0.457212, 1.351634,
    -1.331634, -0.437211, 0.457212, 1.351634, -1.331634, -0.437211, 0.457211, 1.351634,
    -1.331634, -0.437211, 0.457211, 1.351634, -1.331634, -0.437211, 0.457211, 1.351634,
], np.float32).reshape(2, 3, 4)


@pytest.mark.parametrize("mask_value", [O.MASK_ALLOC, O.MASK_NOALLOC])
def test_golden_prenorm(mask_value):
    # encoder_layer.rs:349-392 test_golden_prenorm_noalloc and :694-730 test_golden_prenorm
    t = deterministic_layer(4, 8, 2)
    out = run_layer(t, GOLDEN_IN, GOLDEN_MASK, GOLDEN_POS_BIAS, 2, 8, True, mask_value)
    assert np.abs(out - GOLDEN_PRENORM).max() < 1e-4


@pytest.mark.parametrize("mask_value", [O.MASK_ALLOC, O.MASK_NOALLOC])
def test_golden_postnorm(mask_value):
    # encoder_layer.rs:395-448 test_golden_postnorm_noalloc and :733-780 test_golden_postnorm
    t = deterministic_layer(4, 8, 2)
    out = run_layer(t, GOLDEN_IN, GOLDEN_MASK, GOLDEN_POS_BIAS, 2, 8, False, mask_value)
    assert np.abs(out - GOLDEN_POSTNORM).max() < 1e-4


def test_ffn_golden_gelu():
    # cpu/feedforward/standard_new.rs:155-191 test_ffn_golden_values_gelu (PyTorch-derived)
    x = np.array([0.5, -0.2, 0.1, -0.5, 0.0, 0.8], np.float32).reshape(1, 2, 3)
    w1 = np.array([0.4414, 0.4792, -0.1353, 0.5304, -0.1265, 0.1165, -0.2811, 0.3391, 0.509,
                   -0.4236, 0.5018, 0.1081], np.float32).reshape(4, 3)
    w2 = np.array([0.3694, 0.0677, 0.2411, -0.0706, 0.3854, 0.0739, -0.2334, 0.1274, -0.2304,
                   -0.0586, -0.2031, 0.3317], np.float32).reshape(3, 4)
    mid = O.linear(x, w1)
    mid = np.vectorize(O.gelu, otypes=[np.float32])(mid)
    out = O.linear(mid, w2)
    exp = np.array([0.0266, 0.0386, -0.0491, 0.0304, -0.1196, 0.0148], np.float32).reshape(1, 2, 3)
    assert np.abs(out - exp).max() <= 1e-4


def test_ffn_relu():
    # standard_new.rs:~120-150: identity FC1, x=[1,-1], relu, FC2 = 2*I -> [2, 0]
    x = np.array([[[1.0, -1.0]]], np.float32)
    fc1 = np.eye(2, dtype=np.float32)
    fc2 = 2 * np.eye(2, dtype=np.float32)
    out = O.linear(np.maximum(O.linear(x, fc1), 0), fc2)
    assert np.abs(out - np.array([[[2.0, 0.0]]])).max() <= 1e-6


def test_activation_scalars():
    # activations.rs:312-329 test_scalars
    assert O.gelu(0.0) == 0.0
    assert abs(O.gelu(1.0) - 0.8413447) < 1e-5
    assert O.gelu_new(0.0) == 0.0
    assert abs(O.gelu_new(1.0) - 0.841192) < 1e-5
    assert O.lib().ko_relu(1.0) == 1.0 and O.lib().ko_relu(-1.0) == 0.0


def test_softmax_rows():
    # activations.rs:223-242 semantics: rows sum to 1, max-subtracted, uniform on equal input
    x = np.array([[1.0, 2.0, 3.0], [0.0, 0.0, 0.0]], np.float32)
    y = O.softmax_rows(x)
    assert np.allclose(y.sum(-1), 1.0, atol=1e-6)
    e = np.exp(x[0] - 3.0)
    assert np.allclose(y[0], e / e.sum(), atol=1e-7)
    assert np.allclose(y[1], 1.0 / 3.0)
    # -1e9-masked entries give exact zeros; a fully -inf row gives NaN (no-alloc path quirk)
    y = O.softmax_rows(np.array([[0.5, -1e9, 0.25]], np.float32))
    assert y[0, 1] == 0.0
    y = O.softmax_rows(np.array([[-np.inf, -np.inf]], np.float32))
    assert np.isnan(y).all()


def test_layer_norm_reference_cases():
    # cpu/normalization/layer_norm.rs:228-307
    one3, zero3 = np.ones(3, np.float32), np.zeros(3, np.float32)
    y = O.layer_norm(np.array([[[1.0, 2.0, 3.0]]], np.float32), one3, zero3, 1e-6)
    assert abs(y.mean()) < 1e-5
    assert abs(y[0, 0, 0] + 1.2247) < 1e-3 and abs(y[0, 0, 1]) < 1e-5 and abs(y[0, 0, 2] - 1.2247) < 1e-3
    # with scale and bias
    g, b = np.array([2.0, 0.5, 1.5], np.float32), np.array([1.0, -1.0, 0.5], np.float32)
    y = O.layer_norm(np.array([[[1.0, 2.0, 3.0]]], np.float32), g, b, 1e-6)
    std = np.sqrt(2.0 / 3.0 + 1e-6)
    exp = np.array([(1 - 2) / std * 2 + 1, (2 - 2) / std * 0.5 - 1, (3 - 2) / std * 1.5 + 0.5])
    assert np.abs(y[0, 0] - exp).max() < 1e-4
    # batch
    x = np.array([1, 3, 2, 4, 5, 7, 6, 8], np.float32).reshape(2, 2, 2)
    y = O.layer_norm(x, np.ones(2, np.float32), np.zeros(2, np.float32), 1e-5)
    assert abs(y[0, 0, 0] + 1.0) < 1e-3 and abs(y[0, 0, 1] - 1.0) < 1e-3
    # pytorch parity
    y = O.layer_norm(np.array([[[1.0, 2.0, 3.0, 4.0]]], np.float32), np.ones(4, np.float32),
                     np.zeros(4, np.float32), 1e-5)
    assert np.abs(y[0, 0] - np.array([-1.3416, -0.4472, 0.4472, 1.3416])).max() < 1e-3


@pytest.mark.parametrize("hidden", [64, 128, 384, 768])
def test_layer_norm_matches_numpy(hidden):
    # layer_norm.rs:362-470 SIMD == scalar at these widths; here oracle == float64 numpy at 1e-5
    rng = np.random.default_rng(hidden)
    x = rng.standard_normal((7, hidden)).astype(np.float32)
    g = rng.standard_normal(hidden).astype(np.float32)
    b = rng.standard_normal(hidden).astype(np.float32)
    y = O.layer_norm(x, g, b, 1e-12)
    xd = x.astype(np.float64)
    ref = (xd - xd.mean(-1, keepdims=True)) / np.sqrt(xd.var(-1, keepdims=True) + 1e-12) * g + b
    assert np.abs(y - ref).max() < 1e-5


def test_pooling_goldens():
    # cpu/encoder/traits.rs:796-895 test_pooling_strategies_golden (mask from MockGoldenEncoder)
    hs = np.array([
        -1.331580, -0.437194, 0.457193, 1.351581, -1.331581, -0.437193, 0.457194, 1.351580,
        -1.331581, -0.437194, 0.457194, 1.351581, -1.331581, -0.437194, 0.457194, 1.351581,
        -1.331581, -0.437193, 0.457194, 1.351580, -1.331580, -0.437194, 0.457193, 1.351581,
        -1.331581, -0.437193, 0.457193, 1.351581, -1.331581, -0.437193, 0.457194, 1.351580,
        -1.331581, -0.437193, 0.457193, 1.351581, -1.331581, -0.437193, 0.457193, 1.351581,
    ], np.float32).reshape(2, 5, 4)
    mask = np.ones((2, 5), np.float32)
    mean = O.mean_pool(hs, mask)
    exp_mean = np.array([-1.331581, -0.437193, 0.457194, 1.351580, -1.331581, -0.437193, 0.457193,
                         1.351581], np.float32).reshape(2, 4)
    assert np.abs(mean - exp_mean).max() < 1e-5
    exp_cls = np.array([-1.331580, -0.437194, 0.457193, 1.351581] * 2, np.float32).reshape(2, 4)
    assert np.abs(O.cls_pool(hs) - exp_cls).max() < 1e-5
    exp_max = np.array([-1.331580, -0.437193, 0.457194, 1.351581] * 2, np.float32).reshape(2, 4)
    assert np.abs(O.max_pool(hs, mask) - exp_max).max() < 1e-5
    exp_norm = np.array([-0.665787, -0.218596, 0.228596, 0.675787, -0.665787, -0.218595, 0.228595,
                         0.675787], np.float32).reshape(2, 4)
    assert np.abs(O.l2_normalize(mean) - exp_norm).max() < 1e-5


def test_pooling_unit_cases():
    # pooling/mod.rs:70-153
    hidden = np.array([[[1, 2], [3, 4]], [[5, 6], [7, 8]]], np.float32)
    mask = np.array([[1, 1], [1, 0]], np.float32)
    p = O.mean_pool(hidden, mask)
    assert np.abs(p - np.array([[2, 3], [5, 6]])).max() < 1e-6
    assert (O.cls_pool(hidden) == np.array([[1, 2], [5, 6]])).all()
    assert np.abs(O.max_pool(hidden, mask) - np.array([[3, 4], [5, 6]])).max() < 1e-6
    h3 = np.array([[[1, 2], [3, 4], [5, 6]], [[7, 8], [9, 10], [11, 12]]], np.float32)
    m3 = np.array([[1, 1, 0], [1, 1, 1]], np.float32)
    assert (O.last_token_pool(h3, m3) == np.array([[3, 4], [11, 12]])).all()
    # empty sequence (all masked): token 0's row
    assert (O.mean_pool(np.array([[[1, 2]]], np.float32), np.array([[0]], np.float32)) == [[1, 2]]).all()
    assert (O.max_pool(np.array([[[1, 2]]], np.float32), np.array([[0]], np.float32)) == -1e9).all()


def test_l2_normalize():
    # cpu/encoder/traits.rs:~783-794 test_l2_normalize_inplace
    d = O.l2_normalize(np.array([[3, 4], [1, 1]], np.float32))
    assert abs(d[0, 0] - 0.6) < 1e-6 and abs(d[0, 1] - 0.8) < 1e-6
    assert abs(d[1, 0] - 1 / np.sqrt(2)) < 1e-6
    z = O.l2_normalize(np.zeros((1, 4), np.float32))
    assert (z == 0).all()


def test_cosine_and_search_reference_cases():
    # kjarni-search/src/vector.rs:169-433
    assert abs(O.cosine_ks([1, 2, 3], [1, 2, 3]) - 1.0) < 1e-6
    assert abs(O.cosine_ks([1, 0], [0, 1])) < 1e-6
    assert abs(O.cosine_ks([1, 2, 3], [-1, -2, -3]) + 1.0) < 1e-6
    assert O.cosine_ks([1, 2], [1, 2, 3]) == 0.0
    assert abs(O.cosine_ks([0, 0, 0], [1, 2, 3])) < 1e-6
    idx, sc = O.search([1, 0, 0], [[1, 0, 0], [0.9, 0.1, 0], [0, 1, 0]], 10)
    assert list(idx) == [0, 1, 2] and sc[0] >= sc[1] >= sc[2]
    idx, _ = O.search([1, 0], [[1, 0], [0.9, 0.1], [0.8, 0.2], [0.7, 0.3], [0.6, 0.4]], 3)
    assert len(idx) == 3
    idx, _ = O.search([1, 0], [[1, 0], [0.9, 0.1]], 10)
    assert len(idx) == 2
    idx, _ = O.search([1, 2], np.zeros((0, 2), np.float32), 5)
    assert len(idx) == 0
    idx, _ = O.search([1, 2], [[1, 2, 3]], 5)  # dimension mismatch -> empty
    assert len(idx) == 0
    # threshold case: sims 1.0, ~0.707, 0.0
    _, sc = O.search([1, 0], [[1, 0], [0.7, 0.7], [0, 1]], 10)
    assert (sc >= 0.5).sum() == 2
    # kjarni/src/embedder/model.rs:247-257 zero guard
    assert O.cosine_k([0, 0], [1, 2]) == 0.0


def test_segment_scan_zero_guards():
    # kjarni-rag/src/segment.rs:307-371: zero query -> no hits; zero doc -> score 0
    corpus = np.array([[0, 0], [1, 0], [0, 2]], np.float32)
    idx, sc = O.search([0, 0], corpus, 3, mode=1)
    assert len(idx) == 0
    idx, sc = O.search([1, 0], corpus, 3, mode=1)
    assert list(idx) == [1, 0, 2] and sc[0] == 1.0 and sc[1] == 0.0 and sc[2] == 0.0


def test_linear_blocked_equals_plain():
    # cpu/ops/matmul.rs:571-686 blocking (64-row blocks, 4x3 tile, n%3 and m%4 tails)
    rng = np.random.default_rng(0)
    for (m, k, n) in [(1, 8, 3), (5, 17, 4), (64, 384, 384), (131, 100, 7), (70, 1536, 385)]:
        x = rng.standard_normal((m, k)).astype(np.float32)
        w = rng.standard_normal((n, k)).astype(np.float32) * np.float32(0.05)
        b = rng.standard_normal(n).astype(np.float32)
        ref = (x.astype(np.float64) @ w.astype(np.float64).T + b).astype(np.float32)
        assert np.abs(O.linear(x, w, b) - ref).max() < 1e-4
        assert np.abs(O.linear(x, w, b, blocked=True) - ref).max() < 1e-4


def test_embedding_semantics():
    # cpu/embeddings/tests.rs:60-222, 538-625: lookup + position broadcast + type ids + OOV zeros
    rng = np.random.default_rng(1)
    word = rng.standard_normal((10, 4)).astype(np.float32)
    pos = rng.standard_normal((6, 4)).astype(np.float32)
    typ = rng.standard_normal((2, 4)).astype(np.float32)
    ids = np.array([[1, 2, 3], [4, 99, 0]], np.uint32)
    out = O.embed(ids, None, word, pos, typ)
    assert np.allclose(out[0, 1], word[2] + pos[1] + typ[0])
    assert np.allclose(out[1, 1], pos[1] + typ[0])  # id >= vocab leaves zeros (mod.rs:232-236)
    tt = np.array([[0, 1, 1], [1, 0, 0]], np.uint32)
    out = O.embed(ids, tt, word, pos, typ)
    assert np.allclose(out[0, 2], word[3] + pos[2] + typ[1])
    out = O.embed(ids, None, word, pos, None, pos_offset=2)
    assert np.allclose(out[0, 0], word[1] + pos[2])
    with pytest.raises(ValueError):
        O.embed(ids, np.full((2, 3), 5, np.uint32), word, pos, typ)
