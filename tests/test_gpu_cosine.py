"""Cosine scan + top-k on the GPU vs the oracle: indices exact, scores 1e-4
(SURVEY.md section 8a R14)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _unit_rows(n, d, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


@pytest.mark.parametrize("n,dim,k,mode", [(1, 384, 10, 0), (7, 384, 3, 1), (1000, 384, 10, 0),
                                           (40000, 384, 10, 1), (40000, 384, 50, 0), (20000, 384, 1000, 0),
                                           (5000, 128, 5, 0), (3000, 100, 7, 1), (3000, 1024, 4, 0),
                                           (2000, 30, 9, 0)])
def test_search_matches_oracle(n, dim, k, mode):
    import kjarni_amd
    corpus = _unit_rows(n, dim, seed=n + dim)
    q = _unit_rows(1, dim, seed=99)[0] * np.float32(1.7)
    idx, sc = kjarni_amd.cosine_search(q, corpus, k, mode=mode)
    ridx, rsc = O.search(q, corpus, k, mode=mode)
    assert idx.shape == (1, min(k, n))
    # indices exact wherever the oracle's neighbouring scores are distinguishable at fp32 rounding
    assert np.abs(sc[0] - rsc).max() < 1e-4
    exact = idx[0] == ridx
    if not exact.all():
        # a swap is only acceptable between scores closer than the summation-order noise
        bad = np.nonzero(~exact)[0]
        full = O.cosine_scan(q, corpus, mode)
        assert np.abs(full[idx[0][bad]] - full[ridx[bad]]).max() < 2e-6
    assert (np.diff(sc[0]) <= 0).all()


def test_ties_resolve_to_lowest_index_and_k_over_1024():
    import kjarni_amd
    base = _unit_rows(50, 384, seed=1)
    corpus = np.concatenate([base] * 60, axis=0)       # every row appears 60 times: exact score ties
    q = base[7]
    idx, sc = kjarni_amd.cosine_search(q, corpus, 1500)
    full = O.cosine_scan(q, corpus, 0)
    # the returned list is ordered by (score desc, index asc)
    order = np.lexsort((idx[0], -sc[0]))
    assert (order == np.arange(1500)).all()
    # within each tie group indices ascend
    for s in np.unique(sc[0])[:20]:
        grp = idx[0][sc[0] == s]
        assert (np.diff(grp) > 0).all()
    assert set(idx[0][:60]) == set(range(7, 3000, 50))  # the 60 copies of the query row come first
    assert list(idx[0][:60]) == list(range(7, 3000, 50))
    assert np.abs(sc[0] - np.sort(full)[::-1][:1500]).max() < 1e-4


def test_multiple_queries_and_zero_vectors():
    import kjarni_amd
    corpus = _unit_rows(3000, 384, seed=5)
    corpus[10] = 0.0
    qs = _unit_rows(6, 384, seed=6)
    idx, sc = kjarni_amd.cosine_search(qs, corpus, 8, mode=1)
    for j in range(6):
        ridx, rsc = O.search(qs[j], corpus, 8, mode=1)
        assert list(idx[j]) == list(ridx) and np.abs(sc[j] - rsc).max() < 1e-4
    # zero document: score 0 in both modes (vector.rs:146 max(den,1e-9); segment.rs:366-368)
    for mode in (0, 1):
        full_idx, full_sc = kjarni_amd.cosine_search(qs[0], corpus, 3000, mode=mode)
        assert full_sc[0][list(full_idx[0]).index(10)] == 0.0


def test_full_size_properties():
    """Size-independent checks at a corpus the CPU oracle would not finish quickly:
    planted neighbours are found at the top and scores are sorted."""
    import kjarni_amd
    n, d = 1_000_000, 384
    rng = np.random.default_rng(0)
    corpus = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal(d, dtype=np.float32)
    planted = [123, 500_000, 999_999, 42]
    for r, i in enumerate(planted):
        corpus[i] = q * (1.0 + r) + rng.standard_normal(d, dtype=np.float32) * 0.02 * (r + 1) * (1.0 + r)
    idx, sc = kjarni_amd.cosine_search(q, corpus, 10)
    assert list(idx[0][:4]) == planted
    assert (np.diff(sc[0]) <= 0).all() and sc[0][0] > 0.999
    # spot-check the returned scores with the oracle's scalar formula
    for i, s in zip(idx[0], sc[0]):
        assert abs(O.cosine_ks(q, corpus[i]) - s) < 1e-4
