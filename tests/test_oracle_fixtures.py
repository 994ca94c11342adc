"""The oracle at MODEL scale against a second opinion: tests/golden/encoder_fixtures.npz holds float64 evaluations of
Hugging Face `transformers.BertModel` / `BertForSequenceClassification` on the seeded random MiniLM weights of
tests/synth.py (written by tests/golden/make_encoder_fixtures.py in the build container; only the data travels).
The reference's own encoder goldens are one layer at hidden 4 (cpu/encoder/encoder_layer.rs:244-307); this pins the
model-level wiring -- tensor names, fused Q|K|V order, 12 x 32 heads, six post-norm layers, mean-pool + L2,
bert.pooler + classifier.  tests/test_gpu_encoder.py holds the HIP path to the same file."""
import hashlib
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import synth

FIXTURES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "encoder_fixtures.npz")
CASES = [(1, 8), (3, 8), (64, 8), (1, 128), (3, 128), (64, 128)]
ORACLE_TOL = 1e-5


def digest(tensors) -> str:
    h = hashlib.sha256()
    for k in sorted(tensors):
        h.update(k.encode())
        h.update(np.ascontiguousarray(tensors[k], dtype=np.float32).tobytes())
    return h.hexdigest()


@pytest.fixture(scope="module")
def fx():
    return np.load(FIXTURES)


FAMILIES = {"init": "", "trained": "trained_"}   # tests/synth.py: N(0, 0.02) initialisation | trained-checkpoint statistics


@pytest.fixture(scope="module", params=list(FAMILIES))
def family(request):
    return request.param


@pytest.fixture(scope="module")
def embedder(tmp_path_factory, fx, family):
    cfg, t = synth.minilm_embedder(str(tmp_path_factory.mktemp("fx_emb")), seed=0, family=family)
    assert digest(t) == str(fx[FAMILIES[family] + "embed_weights_sha256"]), "tests/synth.py no longer reproduces the fixture's weights"
    return O.OracleModel(t, cfg), O.OracleModel(t, cfg, blocked_gemm=True)


@pytest.fixture(scope="module")
def cross(tmp_path_factory, fx, family):
    cfg, t = synth.minilm_cross_encoder(str(tmp_path_factory.mktemp("fx_ce")), seed=1, family=family)
    assert digest(t) == str(fx[FAMILIES[family] + "cross_weights_sha256"]), "tests/synth.py no longer reproduces the fixture's weights"
    return O.OracleModel(t, cfg), O.OracleModel(t, cfg, blocked_gemm=True)


def test_trained_family_is_in_the_trained_regime(fx):
    """What the `trained_` fixtures were computed on (a float64 forward of 3 x 128 tokens, stored by the generating
    script): softmax rows are peaked, GELU sees its tails, LayerNorm outputs carry outliers -- and the `init` family
    does none of that, which is why it cannot stand in for a checkpoint."""
    r = lambda k: float(fx["trained_regime_" + k])  # noqa: E731
    assert 3.0 <= r("logit_std") <= 5.0 and r("logit_std_min_layer") >= 2.5 and r("logit_absmax") > 10.0
    assert r("softmax_top_mean") > 0.3
    assert r("fc1_min") < -6.0 and r("fc1_max") > 6.0
    assert r("hidden_absmax") > 15.0
    assert float(fx["regime_logit_std"]) < 0.2 and float(fx["regime_softmax_top_mean"]) < 0.05
    assert float(fx["trained_pairs_logit_absmax"]) > 1.0


@pytest.mark.parametrize("B,S", CASES)
def test_oracle_embeddings_equal_hf_bert(fx, embedder, family, B, S):
    plain, blocked = embedder
    tag = f"{FAMILIES[family]}embed_{B}x{S}"
    ids, mask, want = fx[tag + "_ids"], fx[tag + "_mask"], fx[tag + "_embeddings"]
    for orc in ((plain, blocked) if B * S <= 3 * 128 else (blocked,)):
        got = orc.embed_batch(ids, mask)
        assert got.shape == want.shape == (B, 384)
        assert float(np.abs(got - want).max()) < ORACLE_TOL
        for mv in (O.MASK_ALLOC, O.MASK_NOALLOC):   # both padding fills give the same vectors (masks.rs:4-36)
            assert float(np.abs(orc.embed_batch(ids, mask, mv) - want).max()) < ORACLE_TOL


@pytest.mark.parametrize("B,S", [(1, 8), (3, 8), (1, 128)])
def test_oracle_hidden_states_equal_hf_bert(fx, embedder, family, B, S):
    tag = f"{FAMILIES[family]}embed_{B}x{S}"
    if tag + "_hidden" not in fx:
        pytest.skip("hidden states of this size are stored for the trained family only")
    ids, mask, want = fx[tag + "_ids"], fx[tag + "_mask"], fx[tag + "_hidden"]
    real = mask.astype(bool)   # HF leaves padded QUERY rows attending to nothing special; only real tokens are defined alike
    # f32 against float64 on LayerNorm outputs of magnitude ~3 (init) / up to ~30 (trained: gains up to 12.5)
    tol = 2e-5 if family == "init" else 2e-6 * max(10.0, float(np.abs(want[real]).max()))
    for orc in embedder:
        got = orc.forward(ids, mask, None, O.MASK_ALLOC)
        assert float(np.abs(got - want)[real].max()) < tol


@pytest.mark.parametrize("B,S", CASES)
def test_oracle_rerank_logits_equal_hf_bert(fx, cross, family, B, S):
    plain, blocked = cross
    tag = f"{FAMILIES[family]}pairs_{B}x{S}"
    ids, mask, types, want = (fx[tag + s] for s in ("_ids", "_mask", "_types", "_logits"))
    for orc in ((plain, blocked) if B * S <= 3 * 128 else (blocked,)):
        got = orc.rerank_scores(ids, mask, types)
        assert got.shape == (B,)
        assert float(np.abs(got - want[:, 0]).max()) < ORACLE_TOL


def test_fixture_is_a_tight_reference(fx):
    # HF evaluated in f32 differs from its own f64 evaluation by a few 1e-6 on hidden states: the 1e-5 bar is meaningful
    assert float(fx["embed_hf_f32_vs_f64_hidden_max_abs"]) < 1e-5
    # ... and by a few 1e-5 in the trained regime, where hidden states reach 30: f32 itself, not the oracle
    assert float(fx["trained_embed_hf_f32_vs_f64_hidden_max_abs"]) < 1e-4
