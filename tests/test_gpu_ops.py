"""Single-operator parity (tolerance 1e-5) through kjarni_hip_op_*."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _silu(x):
    x = x.astype(np.float32)
    y = x / (np.float32(1.0) + np.exp(-x, dtype=np.float32))
    return np.where(x <= -20, np.float32(0), np.where(x >= 20, x, y)).astype(np.float32)


@pytest.mark.parametrize("m,k,n", [(1, 384, 384), (127, 384, 1152), (128, 384, 1536), (300, 1536, 384),
                                   (1000, 768, 768), (77, 100, 7), (5, 17, 4), (257, 64, 128), (130, 32, 128),
                                   (4096, 384, 1536), (8192, 384, 384), (8193, 384, 384), (8300, 1536, 384), (2000, 3072, 768)])
def test_linear_all_epilogues(m, k, n):
    # LinearLayer::matmul shapes incl. decode (m=1), odd dims (cpu/ops/tests.rs:78-116) and M tails; the three routes:
    # up to 256 rows (K over the waves, row groups over the grid), 257 .. 8192 rows (64 x 64 tiles, K slices for narrow outputs),
    # more (128 x 128)
    from kjarni_amd import ops
    rng = np.random.default_rng(m * 7 + n)
    x = rng.standard_normal((m, k)).astype(np.float32)
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    r = rng.standard_normal((m, n)).astype(np.float32)
    base = O.linear(x, w, b)
    tol = 1e-5 * max(1.0, float(np.abs(base).max()))
    cases = [(ops.EPI_BIAS, None, base),
             (ops.EPI_BIAS_GELU, None, np.vectorize(O.gelu, otypes=[np.float32])(base)),
             (ops.EPI_BIAS_GELU_NEW, None, np.vectorize(O.gelu_new, otypes=[np.float32])(base)),
             (ops.EPI_BIAS_RELU, None, np.maximum(base, 0)),
             (ops.EPI_BIAS_TANH, None, np.tanh(base)),
             (ops.EPI_BIAS_RESIDUAL, r, base + r),
             # SwiGLU: silu(gate) * up, silu_scalar's cut-offs at +-20 included (activations.rs:74-82)
             (ops.EPI_BIAS_MUL_SILU, r * 12, _silu(r * 12) * base)]
    for epi, res, ref in cases:
        got, _ = ops.linear(x, w, b, res, epi)
        # (the SwiGLU product scales the projection, and its rounding, by silu(gate): tolerance relative to the result)
        t = tol if epi != ops.EPI_BIAS_MUL_SILU else 1e-5 * max(1.0, float(np.abs(ref).max()))
        assert np.abs(got - ref).max() < t, (epi, float(np.abs(got - ref).max()))
    got, _ = ops.linear(x, w, None, None, ops.EPI_BIAS)      # no bias
    assert np.abs(got - O.linear(x, w)).max() < tol


@pytest.mark.parametrize("m,k,n", [(1, 384, 384), (63, 384, 384), (64, 1536, 384), (65, 384, 384), (300, 1536, 384),
                                   (1000, 16, 384), (129, 256, 256), (200, 1024, 256), (77, 768, 768), (40, 100, 60),
                                   (1000, 3072, 768), (8192, 1536, 384), (8193, 1536, 384), (8400, 384, 384), (700, 4096, 1024),
                                   # up to 256 rows with a long K: the few-rows kernel's K slices (one to eight row groups,
                                   # the last rows of each), then the LayerNorm reduce
                                   (1, 1536, 384), (28, 1536, 384), (32, 1536, 384), (33, 1536, 384), (28, 3072, 768),
                                   (64, 4096, 1024), (17, 1024, 256), (130, 1536, 384), (256, 1536, 384), (200, 3072, 768)])
@pytest.mark.parametrize("eps", [1e-12, 1e-5])
def test_residual_projection_with_fused_layernorm(m, k, n, eps):
    """out-proj / FC2 + residual + LayerNorm of the post-norm layer (encoder_layer.rs:129-147, 155-176) as ONE kernel
    for 384- and 256-wide rows (whole rows per workgroup, M tails, a single K-step); other widths take GEMM + LayerNorm
    behind the same entry point.  Reference: Linear, add, LayerNorm of the oracle."""
    from kjarni_amd import ops
    rng = np.random.default_rng(m * 13 + k + n)
    x = rng.standard_normal((m, k)).astype(np.float32)
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    r = (rng.standard_normal((m, n)) * 2 + 0.5).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(n)).astype(np.float32)
    ref = O.layer_norm(O.linear(x, w, b) + r, g, beta, eps)
    got, _ = ops.linear_layer_norm(x, w, b, r, g, beta, eps)
    assert got.shape == ref.shape and float(np.abs(got - ref).max()) < 1e-5 * max(1.0, float(np.abs(ref).max()))
    if eps == 1e-5:  # Nomic's projections have no biases
        ref = O.layer_norm(O.linear(x, w) + r, g, beta, eps)
        got, _ = ops.linear_layer_norm(x, w, None, r, g, beta, eps)
        assert float(np.abs(got - ref).max()) < 1e-5 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("k,n", [(384, 384), (1536, 384), (768, 3072), (1024, 32),
                                 (384, 4096)])   # 128 column tiles: from 5 row groups on, two row groups per workgroup
def test_few_rows_results_do_not_depend_on_the_batch(k, n):
    """Up to 256 rows (128 for models wider than 512: the bound follows min(N, K)) the projections run the split-K kernel, one
    workgroup per 32 columns and group of 32 rows; a row's result must be bit-identical whatever other rows share the call
    (one to eight row groups, whole and partial), and oracle-equal."""
    from kjarni_amd import ops
    rng = np.random.default_rng(k + n)
    rows = 256 if min(k, n) <= 512 else 128
    x = rng.standard_normal((rows, k)).astype(np.float32)
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    r = rng.standard_normal((rows, n)).astype(np.float32)
    full, _ = ops.linear(x, w, b, r, ops.EPI_BIAS_RESIDUAL)
    ref = O.linear(x, w, b) + r
    assert float(np.abs(full - ref).max()) < 1e-5 * max(1.0, float(np.abs(ref).max()))
    for m in (1, 32, 33, 50, 64, 65, 127, 129, 255):
        if m >= rows:
            continue
        part, _ = ops.linear(x[:m], w, b, r[:m], ops.EPI_BIAS_RESIDUAL)
        assert np.array_equal(part, full[:m]), m
    gelu, _ = ops.linear(x[:7], w, b, None, ops.EPI_BIAS_GELU)
    assert float(np.abs(gelu - np.vectorize(O.gelu, otypes=[np.float32])(O.linear(x[:7], w, b))).max()) < 1e-5 * max(
        1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("k,n", [(384, 384), (1536, 384), (384, 1152), (3072, 768)])
def test_mid_size_results_do_not_depend_on_the_batch(k, n):
    """257 .. 8192 rows take 64 x 64 tiles with a K-slice count that depends on (N, K) only: a row's result is
    bit-identical whatever other rows share the call (also with the fused LayerNorm), and oracle-equal."""
    from kjarni_amd import ops
    rng = np.random.default_rng(k * 3 + n)
    M = 1300
    x = rng.standard_normal((M, k)).astype(np.float32)
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    r = rng.standard_normal((M, n)).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(n)).astype(np.float32)
    full, _ = ops.linear(x, w, b, r, ops.EPI_BIAS_RESIDUAL)
    ref = O.linear(x, w, b) + r
    assert float(np.abs(full - ref).max()) < 1e-5 * max(1.0, float(np.abs(ref).max()))
    full_ln, _ = ops.linear_layer_norm(x, w, b, r, g, beta, 1e-12)
    for m in (257, 300, 640, 1299):
        part, _ = ops.linear(x[:m], w, b, r[:m], ops.EPI_BIAS_RESIDUAL)
        assert np.array_equal(part, full[:m]), m
        if n <= 1024:
            part_ln, _ = ops.linear_layer_norm(x[:m], w, b, r[:m], g, beta, 1e-12)
            assert np.array_equal(part_ln, full_ln[:m]), m


def test_fused_layernorm_constant_rows():
    """A row whose 384 values are all equal has variance 0: the result is beta (eps inside the sqrt keeps it finite)."""
    from kjarni_amd import ops
    m, k, n = 70, 32, 384
    x = np.zeros((m, k), np.float32)
    w = np.ones((n, k), np.float32)
    b = np.full(n, 3.0, np.float32)
    r = np.full((m, n), -1.5, np.float32)
    g = np.linspace(0.5, 1.5, n).astype(np.float32)
    beta = np.linspace(-1, 1, n).astype(np.float32)
    got, _ = ops.linear_layer_norm(x, w, b, r, g, beta, 1e-12)
    assert np.isfinite(got).all() and float(np.abs(got - beta[None, :]).max()) < 1e-6


@pytest.mark.parametrize("B,S,heads,d", [(2, 128, 12, 32), (3, 37, 12, 32), (2, 200, 12, 32), (1, 512, 4, 32),
                                         (2, 128, 12, 64), (2, 77, 3, 64), (1, 300, 2, 64), (2, 9, 2, 8),
                                         (1, 1, 12, 32),
                                         # up to 128 (sentence, head) items and 128 tokens: the small-call kernel (a workgroup per
                                         # 32 queries, a 32-key tile per wave) -- its last size, partial query blocks and key tiles
                                         (10, 128, 12, 32), (10, 97, 12, 64), (5, 33, 12, 32), (4, 64, 16, 64), (7, 31, 12, 32)])
def test_attention_parity(B, S, heads, d):
    from kjarni_amd import ops
    rng = np.random.default_rng(B * 100 + S)
    H = heads * d
    qkv = rng.standard_normal((B, S, 3 * H)).astype(np.float32)
    mask = np.ones((B, S), np.uint32)
    for b in range(B):
        mask[b, rng.integers(max(1, S // 2), S + 1):] = 0
        if b % 2 and S >= 3:
            mask[b, S // 3] = 0   # a hole, not only right padding
    q, k, v = (np.ascontiguousarray(qkv[..., i * H:(i + 1) * H]) for i in range(3))
    for mv in (O.MASK_ALLOC, O.MASK_NOALLOC):
        ref = O.attention(q, k, v, mask.astype(np.float32), heads, mask_value=mv)
        got, _ = ops.attention(qkv, mask, heads, mask_value=float(mv))
        assert np.abs(got - ref).max() < 1e-5
    got, _ = ops.attention(qkv, None, heads)
    assert np.abs(got - O.attention(q, k, v, None, heads)).max() < 1e-5


def test_attention_online_softmax_rescale_branch():
    """Forces the multi-chunk rescale (cdna guide rule 26): one key in a LATER 128-key chunk
    dominates every row, so the running max jumps after the first chunk was accumulated."""
    from kjarni_amd import ops
    rng = np.random.default_rng(3)
    B, S, heads, d = 1, 384, 2, 32
    H = heads * d
    qkv = (rng.standard_normal((B, S, 3 * H)) * 0.3).astype(np.float32)
    qkv[0, :, 0:H] += 2.0                 # queries share a direction ...
    qkv[0, 300, H:2 * H] = 6.0            # ... that key 300 (chunk 2) aligns with strongly
    qkv[0, 140, H:2 * H] = 3.0            # and key 140 (chunk 1) moderately
    q, k, v = (np.ascontiguousarray(qkv[..., i * H:(i + 1) * H]) for i in range(3))
    ref = O.attention(q, k, v, None, heads)
    got, _ = ops.attention(qkv, None, heads)
    assert np.abs(got - ref).max() < 1e-5
    # masking the dominant key changes the answer (the branch mattered) and still matches
    mask = np.ones((B, S), np.uint32)
    mask[0, 300] = 0
    ref2 = O.attention(q, k, v, mask.astype(np.float32), heads)
    got2, _ = ops.attention(qkv, mask, heads)
    assert np.abs(ref2 - ref).max() > 1e-2 and np.abs(got2 - ref2).max() < 1e-5
    # a first chunk that is entirely masked with -inf must not poison later chunks
    mask = np.ones((B, S), np.uint32)
    mask[0, :128] = 0
    ref3 = O.attention(q, k, v, mask.astype(np.float32), heads, mask_value=O.MASK_NOALLOC)
    got3, _ = ops.attention(qkv, mask, heads, mask_value=float("-inf"))
    assert np.isfinite(got3).all() and np.abs(got3 - ref3).max() < 1e-5


@pytest.mark.parametrize("rows,hidden", [(1, 384), (1000, 384), (33, 768), (5, 64), (7, 1024), (9, 100), (3, 2048)])
def test_layer_norm_parity(rows, hidden):
    from kjarni_amd import ops
    rng = np.random.default_rng(hidden)
    x = (rng.standard_normal((rows, hidden)) * 3 + 1).astype(np.float32)
    g = rng.standard_normal(hidden).astype(np.float32)
    b = rng.standard_normal(hidden).astype(np.float32)
    for eps in (1e-12, 1e-5):
        got, _ = ops.layer_norm(x, g, b, eps)
        assert np.abs(got - O.layer_norm(x, g, b, eps)).max() < 1e-5


def test_clock_trace_reads_a_shader_clock():
    """kjarni_hip_clock_trace (bench.py's clock_ghz): windows of shader cycles against the 100 MHz counter on the library's own
    measurement stream -- every window a plausible clock (0.1 .. 3 GHz), window lengths as asked (within 20 %)."""
    import torch
    from kjarni_amd import ops
    out = torch.zeros((8, 2), dtype=torch.int64, device="cuda:0")
    stream = ops.measurement_stream()
    assert stream != 0
    ops.clock_trace(out.data_ptr(), 8, 500, stream)
    torch.cuda.synchronize()
    v = out.cpu().numpy().astype(np.float64)
    ghz = v[:, 0] / v[:, 1] / 10.0
    assert ((ghz > 0.1) & (ghz < 3.0)).all(), ghz
    assert ((v[:, 1] >= 50_000) & (v[:, 1] < 60_000)).all(), v[:, 1]  # 500 us = 50 000 ticks of 10 ns
    with pytest.raises(Exception):
        ops.clock_trace(out.data_ptr(), 0, 500, stream)
    ops.measurement_stream_release()
    assert ops.measurement_stream() != 0  # (a new one)
    ops.measurement_stream_release()


def test_reductions_without_the_lds_crossbar_equal_the_shfl_butterflies():
    """csrc/device_utils.h: wave_sum / wave_max, the 16-lane-row and 8-lane-group sums, i ^ 16 / i ^ 32 partners -- through
    v_permlane32_swap / v_permlane16_swap and DPP row rotations instead of ds_bpermute_b32 (what `__shfl_xor` compiles to).
    The kernels that use them claim the butterfly's VALUES bit for bit; the library's self-test runs each form against the
    butterfly it replaces on 2^20 waves of values over twelve binades, both signs and exact zeros."""
    import ctypes as C
    from kjarni_amd import _ffi
    L = _ffi.lib()
    for seed in (0, 12345):
        bad = C.c_uint32(7)
        _ffi.check_error(L.kjarni_hip_selftest_reductions(0, 1 << 20, seed, C.byref(bad)))
        assert bad.value == 0, f"{bad.value} lanes differ from the __shfl_xor butterflies (seed {seed})"
    assert L.kjarni_hip_selftest_reductions(0, 0, 0, C.byref(bad)) == 7          # INVALID_CONFIG
    assert L.kjarni_hip_selftest_reductions(0, 1, 0, None) == 1                   # NULL_POINTER
