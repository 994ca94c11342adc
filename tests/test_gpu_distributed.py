"""kjarni_amd.distributed through the real HIP encoder (world size 1: the GPU box has one device) against the
oracle -- the functions bench.py measures are the functions tested here and, at world size 2, in
tests/test_distributed_cpu.py."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()


def test_sharded_embed_matches_oracle(tmp_path):
    import kjarni_amd
    from kjarni_amd import distributed as D
    from oracle import oracle as O
    cfg, t = synth.minilm_embedder(str(tmp_path / "m"), seed=0, num_hidden_layers=2)
    enc = kjarni_amd.HipEncoder(str(tmp_path / "m"))
    ids, mask = synth.synthetic_ids(37, 128, seed=3, ragged=True)
    got = D.sharded_embed(enc, _dev(ids), _dev(mask)).cpu().numpy()
    got_local = D.sharded_embed(enc, _dev(ids), _dev(mask), n_total=37).cpu().numpy()
    ref = O.OracleModel(t, cfg).embed_batch(ids, mask)
    assert got.shape == ref.shape and float(np.abs(got - ref).max()) < 1e-4
    assert np.array_equal(got, got_local)


def test_sharded_rerank_matches_oracle_and_reference_order(tmp_path):
    import kjarni_amd
    from kjarni_amd import distributed as D
    from oracle import oracle as O
    cfg, t = synth.minilm_cross_encoder(str(tmp_path / "ce"), seed=1, num_hidden_layers=2)
    enc = kjarni_amd.HipEncoder(str(tmp_path / "ce"))
    ids, mask, types = synth.synthetic_pairs(29, 64, seed=5)
    sc = D.sharded_rerank_scores(enc, _dev(ids), _dev(mask), _dev(types))
    ref = O.OracleModel(t, cfg).rerank_scores(ids, mask, types)
    assert float(np.abs(sc.cpu().numpy() - ref).max()) < 1e-4
    # cross_encoder/model.rs:251-252: stable sort by score, descending; index = position in the input
    idx, srt = D.rerank_order_arrays(sc)
    want = sorted(range(29), key=lambda i: -float(sc[i]))  # Python's sort is stable
    assert idx.tolist() == want and bool((srt[:-1] >= srt[1:]).all())
    assert D.rerank_order(sc, 5) == [(i, float(sc[i])) for i in want[:5]]
