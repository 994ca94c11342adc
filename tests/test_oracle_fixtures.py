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


@pytest.fixture(scope="module")
def embedder(tmp_path_factory, fx):
    cfg, t = synth.minilm_embedder(str(tmp_path_factory.mktemp("fx_emb")), seed=0)
    assert digest(t) == str(fx["embed_weights_sha256"]), "tests/synth.py no longer reproduces the fixture's weights"
    return O.OracleModel(t, cfg), O.OracleModel(t, cfg, blocked_gemm=True)


@pytest.fixture(scope="module")
def cross(tmp_path_factory, fx):
    cfg, t = synth.minilm_cross_encoder(str(tmp_path_factory.mktemp("fx_ce")), seed=1)
    assert digest(t) == str(fx["cross_weights_sha256"]), "tests/synth.py no longer reproduces the fixture's weights"
    return O.OracleModel(t, cfg), O.OracleModel(t, cfg, blocked_gemm=True)


@pytest.mark.parametrize("B,S", CASES)
def test_oracle_embeddings_equal_hf_bert(fx, embedder, B, S):
    plain, blocked = embedder
    tag = f"embed_{B}x{S}"
    ids, mask, want = fx[tag + "_ids"], fx[tag + "_mask"], fx[tag + "_embeddings"]
    for orc in ((plain, blocked) if B * S <= 3 * 128 else (blocked,)):
        got = orc.embed_batch(ids, mask)
        assert got.shape == want.shape == (B, 384)
        assert float(np.abs(got - want).max()) < ORACLE_TOL
        for mv in (O.MASK_ALLOC, O.MASK_NOALLOC):   # both padding fills give the same vectors (masks.rs:4-36)
            assert float(np.abs(orc.embed_batch(ids, mask, mv) - want).max()) < ORACLE_TOL


@pytest.mark.parametrize("B,S", [(1, 8), (3, 8)])
def test_oracle_hidden_states_equal_hf_bert(fx, embedder, B, S):
    tag = f"embed_{B}x{S}"
    ids, mask, want = fx[tag + "_ids"], fx[tag + "_mask"], fx[tag + "_hidden"]
    got = embedder[0].forward(ids, mask, None, O.MASK_ALLOC)
    real = mask.astype(bool)   # HF leaves padded QUERY rows attending to nothing special; only real tokens are defined alike
    assert float(np.abs(got - want)[real].max()) < 2e-5   # LayerNorm outputs of magnitude ~3


@pytest.mark.parametrize("B,S", CASES)
def test_oracle_rerank_logits_equal_hf_bert(fx, cross, B, S):
    plain, blocked = cross
    tag = f"pairs_{B}x{S}"
    ids, mask, types, want = (fx[tag + s] for s in ("_ids", "_mask", "_types", "_logits"))
    for orc in ((plain, blocked) if B * S <= 3 * 128 else (blocked,)):
        got = orc.rerank_scores(ids, mask, types)
        assert got.shape == (B,)
        assert float(np.abs(got - want[:, 0]).max()) < ORACLE_TOL


def test_fixture_is_a_tight_reference(fx):
    # HF evaluated in f32 differs from its own f64 evaluation by a few 1e-6 on hidden states: the 1e-5 bar is meaningful
    assert float(fx["embed_hf_f32_vs_f64_hidden_max_abs"]) < 1e-5
