"""Cosine scan + top-k on the GPU vs the oracle: indices EXACT -- on inputs whose neighbouring oracle scores are
distinguishable, which the tests assert of their data --, exact ties by ascending index, one named near-tie case,
scores 1e-4."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.parity_report import report

pytestmark = pytest.mark.gpu


def _unit_rows(n, d, seed):
    rng = np.random.default_rng(seed)   # (an int or a list of ints)
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


GAP = 1e-5   # what "distinguishable" means below: 100 x the rounding of an f32 cosine


def _oracle_gaps(q, corpus, k, mode):
    """Smallest difference between neighbouring scores among the oracle's top-(k + 1)."""
    full = np.sort(O.cosine_scan(q, corpus, mode).astype(np.float64))[::-1][:k + 1]
    return float(np.min(-np.diff(full))) if len(full) > 1 else np.inf


def _planted_corpus(n, dim, k, q, seed):
    """Unit rows of which k + 1 are PLANTED at cosines that step down from 0.9 by >= 3e-4 (every other row's component along
    the query is shrunk, so nothing random comes near them): the oracle's top-(k + 1) scores are then more than GAP apart and
    the expected indices are a property of the data, not of a summation order.  Corpora too small to plant in (n < 4 (k + 1))
    are drawn from successive seeds until their top-(k + 1) gaps exceed GAP."""
    qh = q.astype(np.float64) / np.linalg.norm(q.astype(np.float64))
    m = min(k + 1, n)
    for attempt in range(200):
        rng = np.random.default_rng([seed, attempt])
        corpus = rng.standard_normal((n, dim))
        if n >= 4 * m and dim >= 8:
            along = corpus @ qh
            corpus -= np.outer(along * 0.7, qh)
            corpus /= np.linalg.norm(corpus, axis=1, keepdims=True)
            step = max(3e-4, min(0.02, 0.4 / m))
            for r, i in enumerate(rng.choice(n, m, replace=False)):
                c = 0.9 - r * step
                u = rng.standard_normal(dim)
                u -= (u @ qh) * qh
                corpus[i] = c * qh + np.sqrt(1.0 - c * c) * u / np.linalg.norm(u)
        else:
            corpus /= np.linalg.norm(corpus, axis=1, keepdims=True)
        corpus = corpus.astype(np.float32)
        if all(_oracle_gaps(q, corpus, k, mode) > GAP for mode in (0, 1)):
            return corpus
    raise AssertionError("no well-separated corpus found")


def _many_query_inputs(n, dim, nq, k, attempt):
    """Rows of norm 0.5 (one zero row), queries of norm 3 (query 2 zero).  Where the corpus has room, k + 1 rows are planted per
    query at cosines 0.9, 0.88, ... (as _planted_corpus); smaller corpora rely on the caller's seed loop."""
    rng = np.random.default_rng([n, dim, nq, attempt])
    corpus = _unit_rows(n, dim, seed=[n, attempt]).astype(np.float64)
    queries = _unit_rows(nq, dim, seed=[nq + 1, attempt]) * np.float32(3.0)
    if n >= nq * (k + 1) + 64 and dim >= 64:
        rows = rng.choice(n, nq * (k + 1), replace=False).reshape(nq, k + 1)
        for j in range(nq):
            qh = queries[j].astype(np.float64) / np.linalg.norm(queries[j].astype(np.float64))
            for r, i in enumerate(rows[j]):
                c = 0.9 - 0.02 * r
                u = rng.standard_normal(dim)
                u -= (u @ qh) * qh
                corpus[i] = c * qh + np.sqrt(1.0 - c * c) * u / np.linalg.norm(u)
    corpus = (corpus * 0.5).astype(np.float32)
    corpus[min(11, n - 1)] = 0.0
    queries[2] = 0.0
    return corpus, queries


@pytest.mark.parametrize("n,dim,k,mode", [(1, 384, 10, 0), (7, 384, 3, 1), (1000, 384, 10, 0),
                                           (40000, 384, 10, 1), (40000, 384, 50, 0), (20000, 384, 1000, 0),
                                           (5000, 128, 5, 0), (3000, 100, 7, 1), (3000, 1024, 4, 0),
                                           (2000, 30, 9, 0)])
def test_search_matches_oracle(n, dim, k, mode):
    """Indices EXACT (north_star: bit-exact for index work), on corpora whose top-(k + 1) oracle scores are more than 1e-5 apart
    (asserted, not assumed); scores within 1e-4.  Near-ties have a test of their own below; exact ties the one after."""
    import kjarni_amd
    q = _unit_rows(1, dim, seed=99)[0] * np.float32(1.7)
    corpus = _planted_corpus(n, dim, k, q, seed=n + dim)
    assert _oracle_gaps(q, corpus, k, mode) > GAP
    idx, sc = kjarni_amd.cosine_search(q, corpus, k, mode=mode)
    ridx, rsc = O.search(q, corpus, k, mode=mode)
    assert idx.shape == (1, min(k, n))
    assert np.abs(sc[0] - rsc).max() < 1e-4
    assert int((idx[0] != ridx).sum()) == 0, "index swaps against the oracle on a well-separated corpus"
    assert (np.diff(sc[0]) <= 0).all()


def test_near_ties_may_swap_only_within_rounding():
    """The ONE place where an index may differ from the oracle's: two documents whose cosines differ by less than f32 summation
    noise (here ~1e-8, built on purpose at ranks 2 and 3).  Whichever order the scan returns, the two must be exactly those
    two documents, every other rank must be exact, and the scores must agree to 2e-6."""
    import kjarni_amd
    n, dim, k = 30000, 384, 10
    q = _unit_rows(1, dim, seed=5)[0]
    corpus = _planted_corpus(n, dim, k, q, seed=77)
    ridx, _ = O.search(q, corpus, k, mode=0)
    a, b = int(ridx[2]), int(ridx[3])
    twin = corpus[a].astype(np.float64)
    bump = np.random.default_rng(3).standard_normal(dim)
    qd = q.astype(np.float64)
    bump -= (bump @ qd) * qd                                                # orthogonal to the query ...
    t_perp = twin - (twin @ qd) * qd
    bump -= (bump @ t_perp) / (t_perp @ t_perp) * t_perp                    # ... and to the twin: the cosine moves in second order
    corpus[b] = (twin + 2e-4 * bump / np.linalg.norm(bump)).astype(np.float32)
    full = O.cosine_scan(q, corpus, 0)
    assert abs(float(full[a]) - float(full[b])) < 2e-6 and not np.array_equal(corpus[a], corpus[b])
    ridx, rsc = O.search(q, corpus, k, mode=0)
    idx, sc = kjarni_amd.cosine_search(q, corpus, k, mode=0)
    assert {int(idx[0][2]), int(idx[0][3])} == {a, b} == {int(ridx[2]), int(ridx[3])}
    keep = [r for r in range(k) if r not in (2, 3)]
    assert (idx[0][keep] == ridx[keep]).all()
    assert np.abs(sc[0] - rsc).max() < 2e-6


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


@pytest.mark.parametrize("n,dim,nq,mode", [(5000, 384, 33, 0), (4100, 384, 64, 1), (1028, 768, 21, 0),
                                           (2051, 384, 40, 0), (3000, 100, 25, 1), (640, 384, 130, 0),
                                           # the fused matrix-core scan at its edges: 20 queries (its first size), exactly 64, 65
                                           # (two passes), document counts around the 256-document tile, the narrowest / widest rows
                                           (257, 384, 20, 0), (255, 16, 64, 1), (1000, 1024, 65, 0), (513, 32, 130, 1),
                                           (12, 384, 64, 0), (70_000, 384, 64, 0),
                                           # 2 .. 19 queries over >= 20 000 documents: the bf16 filter route (widths 128 .. 1 024)
                                           (30_000, 384, 3, 0), (30_000, 384, 5, 1), (45_000, 256, 19, 0), (21_000, 768, 7, 1)])
def test_many_queries_gemm_route_matches_streaming_and_oracle(n, dim, nq, mode):
    """From 20 queries on, blocks of 64 go through the fused matrix-core scan (corpus read once per block, document norms
    and cosines in the same launch); row widths it does not take (dim % 16 != 0) fall back to the streaming passes; from 20 000
    documents on, at the widths that have it, 2 and more queries take the bf16 filter pass + exact rescoring.  All routes must
    agree with the oracle and with one another (scores to 2e-6: the streaming passes sum in another order), zero rows and zero
    queries included."""
    import kjarni_amd
    k = 12
    # (inputs drawn from successive seeds until every query's top-(k + 1) oracle scores are more than GAP apart: the indices
    # below are then held EXACTLY)
    for attempt in range(200):
        corpus, queries = _many_query_inputs(n, dim, nq, k, attempt)
        if all(_oracle_gaps(queries[j], corpus, k, mode) > GAP for j in range(nq) if j != 2):
            break
    else:
        raise AssertionError("no well-separated inputs found")
    idx, sc = kjarni_amd.cosine_search(queries, corpus, k, mode=mode)
    # the same queries in groups of 16: below 20 queries a call takes the streaming passes (below 20 000 documents)
    parts = [kjarni_amd.cosine_search(queries[j0:j0 + 16], corpus, k, mode=mode) for j0 in range(0, nq, 16)]
    idx1, sc1 = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
    assert idx.shape == idx1.shape == (nq, k)
    fin = np.isfinite(sc1)
    assert (np.isfinite(sc) == fin).all() and np.abs(sc[fin] - sc1[fin]).max() < 2e-6
    swaps = 0
    for j in range(nq):
        if mode == 1 and j == 2:           # zero query: Segment mode returns no hits for it (idx -1)
            assert (idx[j] == -1).all() and (idx1[j] == -1).all()
            continue
        full = O.cosine_scan(queries[j], corpus, mode)
        kk = min(k, n)
        assert np.abs(full[idx[j][:kk]] - sc[j][:kk]).max() < 1e-4
        ridx, rsc = O.search(queries[j], corpus, k, mode=mode)
        assert np.abs(sc[j][:kk] - rsc).max() < 1e-4
        if j == 2:                          # zero query in VectorStore mode: every score is 0, order = index order
            assert list(idx[j][:kk]) == list(ridx)
            continue
        swaps += int((idx[j][:kk] != ridx).sum()) + int((idx1[j][:kk] != ridx).sum())
    assert swaps == 0, f"{swaps} index swaps against the oracle on well-separated inputs"


@pytest.mark.parametrize("n,nq,k", [(300_000, 1, 10), (300_000, 3, 100), (70_000, 2, 1500), (5000, 1, 5000),
                                    (2_500_000, 2, 16)])
def test_topk_adversarial_orders(n, nq, k):
    """The selection keeps only candidates above a running threshold; ascending scores make every
    candidate pass it (worst case), descending none, constant scores tie everywhere.  Expected order:
    score descending, equal scores by ascending index."""
    from kjarni_amd import ops
    rng = np.random.default_rng(n + k)
    rows = [np.arange(n, dtype=np.float32) / np.float32(n),                 # ascending: all pass
            -np.arange(n, dtype=np.float32),                                # descending
            np.zeros(n, np.float32),                                        # all tied
            rng.integers(0, 50, n).astype(np.float32),                      # heavy ties
            rng.standard_normal(n).astype(np.float32)]
    rows[4][rng.integers(0, n, 50)] = np.nan                               # NaN sorts lowest
    rows[4][rng.integers(0, n, 5)] = np.inf
    for r0 in range(0, len(rows), nq):
        sc = np.stack((rows + rows)[r0:r0 + nq])
        idx, out = ops.topk(sc, k)
        for j in range(nq):
            s = sc[j].copy()
            key = np.where(np.isnan(s), -np.inf, s)
            order = np.lexsort((np.arange(n), -key))[:k]                    # score desc, index asc
            kk = min(k, n)
            nan_free = ~np.isnan(s[order[:kk]])
            assert (idx[j][:kk][nan_free] == order[:kk][nan_free]).all()
            np.testing.assert_array_equal(out[j][:kk][nan_free], s[order[:kk]][nan_free])
            assert (idx[j][kk:] == -1).all()


@pytest.mark.parametrize("n,dim,k", [(1, 384, 1), (3, 384, 10), (4, 384, 4), (5, 384, 16), (999, 384, 17), (40_003, 384, 10),
                                     (40_003, 384, 64), (40_002, 384, 65), (40_001, 384, 128), (40_000, 384, 256), (300_001, 384, 100),
                                     (9_001, 128, 10), (9_001, 256, 33), (9_001, 512, 10), (9_001, 768, 129), (9_001, 1024, 10)])
def test_one_call_search_equals_scores_plus_topk(n, dim, k):
    """kjarni_hip_cosine_search (one query: a single fused pass, every wave keeping its best keys in registers) against
    kjarni_hip_cosine_scores + kjarni_hip_cosine_topk: the same indices and the same scores bit for bit -- on rows with exact
    duplicates (ties resolve to the lowest index), on a corpus whose scores ASCEND with the row index (every row beats the running
    threshold: the worst case for the insertions), with a zero row, a tail of rows past the last whole group, and in both modes."""
    import torch
    from kjarni_amd import _ffi
    L = _ffi.lib()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(n * 7 + dim + k)
    q = rng.standard_normal(dim).astype(np.float32)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    ramp = np.linspace(0.0, 3.0, n, dtype=np.float32)[:, None]
    corpora = [base.copy(), (base + ramp * q).astype(np.float32)]           # random; ascending cosine with the row index
    dup = base.copy()
    dup[n // 2:] = dup[: n - n // 2]                                          # the second half repeats the first: exact ties
    corpora.append(dup)
    for ci, corpus in enumerate(corpora):
        if n > 2:
            corpus[n // 3] = 0.0
        c_d = torch.from_numpy(corpus).to(dev)
        q_d = torch.from_numpy(q).to(dev)
        for mode in (0, 1):
            scores = torch.empty((1, n), dtype=torch.float32, device=dev)
            ws = torch.empty(L.kjarni_hip_cosine_topk_workspace_bytes(1, n, k), dtype=torch.uint8, device=dev)
            idx2 = torch.empty((1, k), dtype=torch.int64, device=dev)
            sc2 = torch.empty((1, k), dtype=torch.float32, device=dev)
            st = torch.cuda.current_stream().cuda_stream
            _ffi.check_error(L.kjarni_hip_cosine_scores(0, q_d.data_ptr(), 1, c_d.data_ptr(), n, dim, mode, scores.data_ptr(), st))
            _ffi.check_error(L.kjarni_hip_cosine_topk(0, scores.data_ptr(), 1, n, k, ws.data_ptr(), idx2.data_ptr(), sc2.data_ptr(), st))
            ws1 = torch.empty(L.kjarni_hip_cosine_search_workspace_bytes(1, n, dim, k), dtype=torch.uint8, device=dev)
            idx1 = torch.full((1, k), -7, dtype=torch.int64, device=dev)
            sc1 = torch.full((1, k), 7.0, dtype=torch.float32, device=dev)
            _ffi.check_error(L.kjarni_hip_cosine_search(0, q_d.data_ptr(), 1, c_d.data_ptr(), n, dim, mode, k, ws1.data_ptr(),
                                                        idx1.data_ptr(), sc1.data_ptr(), st))
            torch.cuda.synchronize()
            a, b = idx1.cpu().numpy()[0], idx2.cpu().numpy()[0]
            assert (a == b).all(), (ci, mode, a[:12], b[:12])
            sa, sb = sc1.cpu().numpy()[0], sc2.cpu().numpy()[0]
            assert (sa.view(np.uint32) == sb.view(np.uint32)).all(), (ci, mode)
            kk = min(k, n)
            assert (a[kk:] == -1).all() and (a[:kk] >= 0).all()
            # and against the oracle: order by (score descending, index ascending)
            ridx, rsc = O.search(q, corpus, k, mode=mode)
            assert np.abs(sa[:kk] - rsc).max() < 1e-4


def test_fused_many_query_search_against_the_oracle_at_its_own_size():
    """The route that selects INSIDE the matrix-core scan (>= 20 queries, >= 400 000 documents) held to the ORACLE directly, at a
    size where it engages: 450 000 x 384, 64 queries, k = 12, k + 1 planted neighbours per query at cosines 0.9, 0.88, ... (the
    expected indices are a property of the data) -- indices exact, scores 1e-4, both zero-norm conventions (vector.rs:146,
    segment.rs:355-371), zero row and zero query included.  (segment.rs:307-337, vector.rs:150-166)"""
    import kjarni_amd
    n, dim, nq, k = 450_000, 384, 64, 12
    corpus, queries = _many_query_inputs(n, dim, nq, k, 0)
    for mode, js in ((1, range(nq)), (0, range(0, nq, 4))):
        idx, sc = kjarni_amd.cosine_search(queries, corpus, k, mode=mode)
        assert idx.shape == (nq, k)
        worst = 0.0
        for j in js:
            if mode == 1 and j == 2:           # zero query: Segment mode returns no hits for it
                assert (idx[j] == -1).all()
                continue
            ridx, rsc = O.search(queries[j], corpus, k, mode=mode)
            if j != 2:
                assert _oracle_gaps(queries[j], corpus, k, mode) > GAP
            assert list(idx[j]) == list(ridx), (mode, j, idx[j], ridx)
            worst = max(worst, float(np.abs(sc[j] - rsc).max()))
        assert report(f"cosine/fused_many_query_450000x384_mode{mode}", worst, 1e-4) < 1e-4


@pytest.mark.parametrize("k", [12, 200])
@pytest.mark.parametrize("case", ["random", "ascending", "overflow", "nonfinite", "random_f32", "overflow_f32", "nonfinite_f32",
                                  "random_dim128", "random_dim256", "random_dim512", "random_dim768", "random_dim1024"])
def test_many_queries_one_call_selects_inside_the_scan(case, k):
    """From 400 000 documents on, kjarni_hip_cosine_search with >= 20 queries never writes a [queries, documents] score array: a
    strided sample of the corpus gives every query a lower bound of its k-th best score, one pass over the corpus keeps only what
    can reach it, a per-query selection finishes.  At widths 128 / 256 / 384 / 512 the pass is the bf16 FILTER (scores off by at most
    eta, bound relaxed by it) and the survivors' exact cosines come from a rescoring pass with the f32 scan's arithmetic (768 and
    1 024 the same with the tile in pieces); at other widths ("_f32" cases: 640) the f32 matrix-core scan itself selects.  Same indices and score bits as kjarni_hip_cosine_scores +
    kjarni_hip_cosine_topk: on random rows; on rows whose scores ascend with the index (the sample under-estimates every bound);
    and when the lists overflow (identical queries, every sampled row anti-correlated, every other row correlated), where the
    queued two-call form takes over.
    k = 12: the bound comes from the sampled tiles' per-wave maxima (k <= 128); k = 200: from the sample's own top-k.
    "nonfinite": one query holds a NaN, another an infinity -- their scores are NaN, which the cheap bound test must not drop:
    those queries come back as the two-call form returns them (k rows, NaN scores), the other 68 unchanged."""
    import torch
    from kjarni_amd import _ffi
    L = _ffi.lib()
    dev = torch.device("cuda", 0)
    n, nq = 450_123, 70
    dim = 640 if case.endswith("_f32") else 384   # (384: the bf16 filter pass + exact rescoring; 640: the f32 scan selects)
    case = case.removesuffix("_f32")
    if case.startswith("random_dim"):             # (the filter pass's other widths: 4, 8, 16 K-steps of 32; 24 and 32 in pieces)
        dim, case = int(case[len("random_dim"):]), "random"
    g = torch.Generator(device=dev).manual_seed(11)
    corpus = torch.randn((n, dim), generator=g, device=dev, dtype=torch.float32)
    q = torch.randn((nq, dim), generator=g, device=dev, dtype=torch.float32)
    if case == "ascending":
        corpus += torch.linspace(0.0, 2.0, n, device=dev)[:, None] * q[3][None, :]
    if case == "overflow":
        q[:] = q[0]
        if dim != 640:   # (the sample units of filter_plan, cosine.hip: tiles of 16 documents)
            tiles = (n + 15) // 16
            unit_tiles = min(16, max(1, (tiles + 4096 * 20 - 1) // (4096 * 20)))
            units = min(4096, max(1, tiles // unit_tiles // 8))
            ts = max(unit_tiles, tiles // units)
            row_tile = torch.arange(n, device=dev) // 16
            sampled = ((row_tile % ts) < unit_tiles) & ((row_tile // ts) < units)
        else:            # (the sample stride of launch_cosine_search's f32 sample pass: tiles of 256 documents)
            tiles = (n + 255) // 256
            sample_tiles = 512 if tiles >= 8192 else 256
            ts = max(1, (tiles + sample_tiles - 1) // sample_tiles)
            row_tile = torch.arange(n, device=dev) // 256
            sampled = (row_tile % ts) == 0
        corpus[sampled] = -q[0] + 0.01 * corpus[sampled]
        corpus[~sampled] = q[0] + 0.05 * corpus[~sampled]
    if case == "nonfinite":
        q[5, 7] = float("nan")
        q[9, 0] = float("inf")
    corpus[1000] = 0.0
    for mode in (0, 1):
        st = torch.cuda.current_stream().cuda_stream
        scores = torch.empty((nq, n), dtype=torch.float32, device=dev)
        ws = torch.empty(L.kjarni_hip_cosine_topk_workspace_bytes(nq, n, k), dtype=torch.uint8, device=dev)
        idx2 = torch.empty((nq, k), dtype=torch.int64, device=dev)
        sc2 = torch.empty((nq, k), dtype=torch.float32, device=dev)
        _ffi.check_error(L.kjarni_hip_cosine_scores(0, q.data_ptr(), nq, corpus.data_ptr(), n, dim, mode, scores.data_ptr(), st))
        _ffi.check_error(L.kjarni_hip_cosine_topk(0, scores.data_ptr(), nq, n, k, ws.data_ptr(), idx2.data_ptr(), sc2.data_ptr(), st))
        del scores, ws
        ws1 = torch.empty(L.kjarni_hip_cosine_search_workspace_bytes(nq, n, dim, k), dtype=torch.uint8, device=dev)
        idx1 = torch.full((nq, k), -7, dtype=torch.int64, device=dev)
        sc1 = torch.full((nq, k), 7.0, dtype=torch.float32, device=dev)
        _ffi.check_error(L.kjarni_hip_cosine_search(0, q.data_ptr(), nq, corpus.data_ptr(), n, dim, mode, k, ws1.data_ptr(),
                                                    idx1.data_ptr(), sc1.data_ptr(), st))
        torch.cuda.synchronize()
        if case == "nonfinite":
            odd = torch.zeros(nq, dtype=torch.bool, device=dev)
            odd[5] = odd[9] = True
            # the two-call form: k rows, NaN scores (the zero document alone scores 0 under the zero-norm rules) -- and the fused
            # call returns exactly those rows
            assert bool((idx2[odd] >= 0).all()) and bool(torch.isnan(sc2[odd][:, 1:]).all()), "the two-call form: k rows with NaN scores"
            assert bool(torch.equal(idx1[odd], idx2[odd])), (case, mode, idx1[odd][:, :4], idx2[odd][:, :4])
            same = (sc1[odd] == sc2[odd]) | (torch.isnan(sc1[odd]) & torch.isnan(sc2[odd]))
            assert bool(same.all()), (case, mode, sc1[odd][:, :4], sc2[odd][:, :4])
            assert bool(torch.equal(idx1[~odd], idx2[~odd])), (case, mode)
            assert bool(torch.equal(sc1[~odd].view(torch.int32), sc2[~odd].view(torch.int32))), (case, mode)
            assert bool((sc1[~odd][:, :-1] >= sc1[~odd][:, 1:]).all())
        else:
            assert bool(torch.equal(idx1, idx2)), (case, mode)
            assert bool(torch.equal(sc1.view(torch.int32), sc2.view(torch.int32))), (case, mode)
            assert bool((sc1[:, :-1] >= sc1[:, 1:]).all())
        del ws1


def _bf16_rne(x):
    """x rounded to bf16 (round to nearest even), as float32 -- what v_cvt_pk_bf16_f32 does to the filter pass's operands."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def test_filter_pass_keeps_documents_at_its_rounding_worst_case():
    """The bf16 filter pass of the many-query search (widths 128 .. 512, >= 400 000 documents) may be off by eta = 0.0081 in the
    cosine, and its bound is relaxed by exactly that.  Here the ten best documents of every query are the WORST case of the
    rounding: every component of the query and of those documents has magnitude 1 + 2^-8 - 2^-12 (x a power of two), a hair below
    the midpoint of two bf16 numbers, so all of them round DOWN by 2^-8 relative and the approximate cosine is 0.0073 below the
    exact one -- while 400 ordinary rows per query sit at cosine 0.940, a hair below the tenth-best (0.9427), and give the sample
    a tight bound.  An un-relaxed filter drops the tenth-best (approximate 0.9354 < 0.940); the search must return the planted
    rows in order, scores within 1e-4 of the oracle.  (vector.rs:150-166, segment.rs:307-337)"""
    import kjarni_amd
    n, dim, nq, k = 420_000, 384, 24, 10
    rng = np.random.default_rng(2024)
    corpus = _unit_rows(n, dim, seed=77)
    m_dn = np.float32(1.0 + 2.0 ** -8 - 2.0 ** -12)
    assert float(_bf16_rne(np.array([m_dn]))[0]) == 1.0
    signs = rng.choice(np.array([-1.0, 1.0], dtype=np.float32), size=(nq, dim))
    queries = (signs * m_dn * np.float32(2.0)).astype(np.float32)
    rows = rng.choice(n, nq * (k + 400), replace=False).reshape(nq, k + 400)
    for j in range(nq):
        qh = queries[j].astype(np.float64) / np.linalg.norm(queries[j].astype(np.float64))
        for r in range(k):                                   # planted: 2 + r sign flips -> cosine 1 - 2 (2 + r) / 384
            d = signs[j] * m_dn * np.float32(0.5)
            flip = rng.choice(dim, 2 + r, replace=False)
            d[flip] = -d[flip]
            corpus[rows[j, r]] = d
        for i in rows[j, k:]:                                # ordinary rows at cosine 0.940 +- 0.0004
            c = 0.940 + rng.uniform(-4e-4, 4e-4)
            u = rng.standard_normal(dim)
            u -= (u @ qh) * qh
            corpus[i] = (c * qh + np.sqrt(1.0 - c * c) * u / np.linalg.norm(u)).astype(np.float32)
    # the worst case is really there: the filter's arithmetic, emulated, is 0.006 .. eta below the exact cosine on the planted rows
    qb, db = _bf16_rne(queries[0]).astype(np.float64), _bf16_rne(corpus[rows[0, :k]]).astype(np.float64)
    q64, d64 = queries[0].astype(np.float64), corpus[rows[0, :k]].astype(np.float64)
    den = np.linalg.norm(q64) * np.linalg.norm(d64, axis=1)
    err = (d64 @ q64) / den - (db @ qb) / den
    assert 0.006 < err.min() and err.max() < 0.0081, err
    for mode in (0, 1):
        idx, sc = kjarni_amd.cosine_search(queries, corpus, k, mode=mode)
        worst = 0.0
        for j in range(nq):
            ridx, rsc = O.search(queries[j], corpus, k, mode=mode)
            assert list(ridx) == list(rows[j, :k]), "the planted rows are the oracle's top-k, in order"
            assert list(idx[j]) == list(ridx), (mode, j, idx[j], ridx)
            worst = max(worst, float(np.abs(sc[j] - rsc).max()))
        assert report(f"cosine/filter_worst_case_rounding_420000x384_mode{mode}", worst, 1e-4) < 1e-4


def test_many_query_search_fuzz_against_scores_plus_topk():
    """tools/search_fuzz.py, 24 random cases (corpus size, width -- with and without the bf16 filter pass --, 2 .. 130 queries, k, both
    zero-norm conventions, exact ties, zero rows, a zero query, ascending scores): kjarni_hip_cosine_search and
    kjarni_hip_cosine_scores + kjarni_hip_cosine_topk agree in every index and every score bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "search_fuzz.py"), "24", "5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "24 / 24 cases agree bit for bit" in r.stdout
