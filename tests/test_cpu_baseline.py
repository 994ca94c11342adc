"""The timed CPU leg of bench.py (oracle/kjarni_cpu_baseline.c: the reference's no-alloc path with its blocking and
threading) must compute what the parity oracle computes -- a fast baseline that computes something else would make
the reported GPU/CPU ratio meaningless."""
import numpy as np
import pytest

from oracle import cpu_baseline as CB
from oracle import oracle as O
from tests import synth


@pytest.mark.parametrize("parallel_rowops", [False, True])
@pytest.mark.parametrize("batch,seq,ragged", [(3, 16, True), (9, 128, False), (8, 128, True)])
def test_baseline_equals_oracle(tmp_path, parallel_rowops, batch, seq, ragged):
    cfg, t = synth.minilm_embedder(str(tmp_path / "m"), seed=0, num_hidden_layers=2)
    ids, mask = synth.synthetic_ids(batch, seq, seed=batch, ragged=ragged)
    ref = O.OracleModel(t, cfg).embed_batch(ids, mask)
    got = CB.BaselineModel(t, cfg, max_batch=16, max_seq=128).embed_batch(ids, mask, parallel_rowops)
    assert got.shape == ref.shape
    assert float(np.abs(got - ref).max()) < 1e-5


def test_buffers_are_reused_and_bounded(tmp_path):
    cfg, t = synth.minilm_embedder(str(tmp_path / "m"), seed=1, num_hidden_layers=1)
    m = CB.BaselineModel(t, cfg, max_batch=4, max_seq=32)
    ids, mask = synth.synthetic_ids(4, 32, seed=0)
    a = m.embed_batch(ids, mask)
    b = m.embed_batch(ids[:2], mask[:2])       # smaller call in the same buffers
    c = m.embed_batch(ids, mask, True)
    assert np.array_equal(a[:2], b) and float(np.abs(a - c).max()) < 1e-6
    with pytest.raises(ValueError):
        m.embed_batch(*synth.synthetic_ids(5, 32, seed=0))
