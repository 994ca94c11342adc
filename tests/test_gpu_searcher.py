"""Searcher on the GPU: kjarni_searcher_* and the raw-vector index retrieval hook vs the oracle
(oracle/search_oracle.py for index / BM25 / fusion / filters, oracle/oracle.py for the encoder and
the cosine scan).  Index fixtures are written in the reference's on-disk layout by the oracle."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from oracle import search_oracle as SO
from tests import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4

CORPUS = [
    ("Reykjavík is the capital of Iceland.", {"source": "docs/iceland.md", "lang": "en"}),
    ("Rust is a systems programming language.", {"source": "docs/rust.md", "lang": "en"}),
    ("Python is popular for scripting and data science.", {"source": "notes/python.txt", "lang": "en"}),
    ("The capital of France is Paris.", {"source": "docs/france.md", "lang": "en"}),
    ("Ísland er eyja í Norður-Atlantshafi.", {"source": "docs/island.md", "lang": "is"}),
    ("semantic search with sentence vectors", {"source": "notes/search.txt", "lang": "en"}),
    ("Iceland has many volcanoes and glaciers; Iceland is cold.", {"source": "docs/geo/iceland2.md", "lang": "en"}),
    ("a", {}),
    ("GPU kernels use matrix cores for fast matrix multiplication.", {"source": "gpu.md", "lang": "en"}),
    ("The quick brown fox jumps over the lazy dog.", {"source": "notes/fox.txt", "lang": "en"}),
]


def _same_hits(got, exp, tol):
    assert [g["document_id"] for g in got] == [e["document_id"] for e in exp]
    assert [g["text"] for g in got] == [e["text"] for e in exp]
    assert [g["metadata"] for g in got] == [e["metadata"] for e in exp]
    np.testing.assert_allclose([g["score"] for g in got], [e["score"] for e in exp], atol=tol, rtol=0)


@pytest.fixture(scope="module")
def env(tmp_path_factory):
    import kjarni_amd
    cache = str(tmp_path_factory.mktemp("cache"))
    d = os.path.join(cache, "sentence-transformers_all-MiniLM-L6-v2")
    cfg, t = synth.minilm_embedder(d, seed=3)
    synth.add_tokenizer(d)
    dr = os.path.join(cache, "cross-encoder_ms-marco-MiniLM-L-6-v2")
    cfg_r, t_r = synth.minilm_cross_encoder(dr, seed=4)
    synth.add_tokenizer(dr)
    orc, orc_r = O.OracleModel(t, cfg), O.OracleModel(t_r, cfg_r)
    tok = kjarni_amd.Tokenizer(os.path.join(d, "tokenizer.json"), 512)

    def embed(texts):
        ids, mask, _ = tok.encode_batch(texts)
        return orc.embed_batch(ids, mask)

    def rerank(query, texts):
        ids, mask, types = tok.encode_batch([query] * len(texts), texts)
        return orc_r.rerank_scores(ids, mask, types)

    # the index holds ORACLE embeddings, one text at a time, as the reference's indexer would store them
    docs = [(t_, embed([t_])[0], md) for t_, md in CORPUS]
    root = str(tmp_path_factory.mktemp("idx") / "corpus")
    SO.write_index(root, 384, docs, max_docs_per_segment=4)
    return dict(cache=cache, root=root, docs=docs, embed=embed, rerank=rerank, oracle=SO.IndexOracle(docs, 4))


# ---------------------------------------------------------------- raw-vector retrieval (no encoder involved)
def test_reference_lifecycle_semantic(tmp_path):
    """kjarni-rag/src/tests.rs:12-80."""
    from kjarni_amd.searcher import index_search
    root = str(tmp_path / "my_index")
    docs = [("Apple is a fruit", [1.0, 0.0, 0.0, 0.0], {"category": "fruit"}),
            ("Car is a vehicle", [0.0, 1.0, 0.0, 0.0], {}),
            ("Banana is yellow", [0.9, 0.1, 0.0, 0.0], {})]
    SO.write_index(root, 4, docs, max_docs_per_segment=2)
    r = index_search(root, None, [1.0, 0.0, 0.0, 0.0], mode="semantic", top_k=10)
    assert len(r) == 3 and r[0]["text"] == "Apple is a fruit" and r[0]["score"] > 0.99
    assert r[0]["metadata"] == {"category": "fruit"}
    assert r[1]["text"] == "Banana is yellow" and r[2]["text"] == "Car is a vehicle"
    _same_hits(r, SO.IndexOracle(docs, 2).search_semantic([1.0, 0.0, 0.0, 0.0], 10), 1e-6)
    # zero query -> no semantic results (segment.rs:315-317); wrong dimension -> INVALID_CONFIG
    assert index_search(root, None, [0.0, 0.0, 0.0, 0.0], mode="semantic", top_k=10) == []
    with pytest.raises(Exception):
        index_search(root, None, [1.0, 0.0, 0.0], mode="semantic", top_k=10)


def test_index_semantic_and_hybrid_multi_segment(tmp_path):
    from kjarni_amd.searcher import index_search
    rng = np.random.default_rng(9)
    words = ["alpha", "beta", "gamma", "delta", "rust", "python", "vector", "index", "search", "kernel"]
    docs = []
    for i in range(3000):
        text = " ".join(rng.choice(words, int(rng.integers(1, 10))))
        md = {"source": f"dir{i % 3}/file{i}.{'md' if i % 2 else 'txt'}", "bucket": str(i % 4)}
        docs.append((text, rng.standard_normal(64).astype(np.float32), md))
    docs[17] = (docs[17][0], np.zeros(64, np.float32), docs[17][2])      # zero-norm row is skipped
    root = str(tmp_path / "big")
    SO.write_index(root, 64, docs, max_docs_per_segment=700)             # 5 segments, ragged last
    orc = SO.IndexOracle(docs, 700)
    for qi in range(3):
        q = rng.standard_normal(64).astype(np.float32)
        for k in (1, 10, 200):
            _same_hits(index_search(root, None, q, mode="semantic", top_k=k), orc.search_semantic(q, k), 1e-5)
        got = index_search(root, "rust vector", q, mode="hybrid", top_k=20)
        _same_hits(got, orc.search_hybrid("rust vector", q, 20), 1e-7)
        # filters: 3x over-fetch, filter, cut
        f = SO.MetadataFilter().must("bucket", "1").source("*.md")
        exp = [r for r in orc.search_semantic(q, 60) if f.matches(r["metadata"])][:20]
        _same_hits(index_search(root, None, q, mode="semantic", top_k=20, filter_key="bucket", filter_value="1",
                                source_pattern="*.md"), exp, 1e-5)
        # threshold
        s = orc.search_semantic(q, 50)
        thr = s[25]["score"] + 1e-4
        got = index_search(root, None, q, mode="semantic", top_k=50, threshold=thr)
        assert [g["document_id"] for g in got] == [e["document_id"] for e in s if e["score"] >= thr]
    # the same segments again: served from the device-resident copies
    q = rng.standard_normal(64).astype(np.float32)
    _same_hits(index_search(root, None, q, mode="semantic", top_k=5), orc.search_semantic(q, 5), 1e-5)


def test_index_rewritten_in_place_is_reloaded(tmp_path):
    from kjarni_amd.searcher import index_search
    root = str(tmp_path / "idx")
    a = [("one", [1.0, 0.0], {}), ("two", [0.0, 1.0], {})]
    SO.write_index(root, 2, a)
    assert index_search(root, None, [1.0, 0.1], mode="semantic", top_k=1)[0]["text"] == "one"
    b = [("uno", [0.0, 1.0], {}), ("dos", [1.0, 0.0], {})]
    SO.write_index(root, 2, b)
    os.utime(os.path.join(root, "segments", "seg_000000", "vectors.bin"), ns=(1, 1))  # force a different mtime
    assert index_search(root, None, [1.0, 0.1], mode="semantic", top_k=1)[0]["text"] == "dos"


# ---------------------------------------------------------------- the full Searcher
def test_searcher_modes(env):
    import kjarni_amd
    s = kjarni_amd.Searcher(cache_dir=env["cache"])                      # default model: minilm-l6-v2
    assert (s.model_name, s.has_reranker, s.default_top_k, s.default_mode, s.reranker_model) == \
        ("minilm-l6-v2", False, 10, 2, "")
    orc = env["oracle"]
    for query in ["capital of Iceland", "programming language", "Ísland"]:
        q = env["embed"]([query])[0]
        _same_hits(s.search(env["root"], query, mode="semantic", top_k=5), orc.search_semantic(q, 5), TOL)
        _same_hits(s.search(env["root"], query, mode="keyword", top_k=5), orc.search_keywords(query, 5), 0)
        _same_hits(s.search(env["root"], query, mode="hybrid", top_k=5), orc.search_hybrid(query, q, 5), 1e-7)
        _same_hits(s.search(env["root"], query), orc.search_hybrid(query, q, 10), 1e-7)   # defaults
    # the stored embedding of a document is its own nearest neighbour with score ~1
    r = s.search(env["root"], CORPUS[5][0], mode="semantic", top_k=1)
    assert r[0]["document_id"] == 5 and abs(r[0]["score"] - 1.0) < TOL


def test_searcher_filters_threshold(env):
    import kjarni_amd
    s = kjarni_amd.Searcher(model="sentence-transformers/all-MiniLM-L6-v2", cache_dir=env["cache"],
                            default_mode="semantic", default_top_k=3)
    assert s.default_mode == 1 and s.default_top_k == 3
    orc = env["oracle"]
    query = "capital of Iceland"
    q = env["embed"]([query])[0]
    f = SO.MetadataFilter().source("docs/*.md")
    exp = [r for r in orc.search_semantic(q, 9) if f.matches(r["metadata"])][:3]
    _same_hits(s.search(env["root"], query, source_pattern="docs/*.md"), exp, TOL)
    f = SO.MetadataFilter().must("lang", "is")
    exp = [r for r in orc.search_semantic(q, 9) if f.matches(r["metadata"])][:3]
    got = s.search(env["root"], query, filter_key="lang", filter_value="is")
    _same_hits(got, exp, TOL)
    assert [g["document_id"] for g in got] == [4]
    full = orc.search_semantic(q, 10)
    thr = (full[2]["score"] + full[3]["score"]) / 2
    got = s.search(env["root"], query, top_k=10, threshold=thr)
    assert [g["document_id"] for g in got] == [e["document_id"] for e in full[:3]]


def test_searcher_rerank(env):
    import kjarni_amd
    s = kjarni_amd.Searcher(cache_dir=env["cache"], rerank_model="minilm-l6-v2-cross-encoder")
    assert s.has_reranker and s.reranker_model == "minilm-l6-v2-cross-encoder"
    orc = env["oracle"]
    query = "capital of Iceland"
    q = env["embed"]([query])[0]
    # candidates: top_k*5 hybrid hits; rescored by the cross-encoder; sorted desc; cut to top_k
    cand = orc.search_hybrid(query, q, 3 * 5)
    scores = env["rerank"](query, [c["text"] for c in cand])
    order = sorted(range(len(cand)), key=lambda i: -scores[i])[:3]
    exp = [dict(cand[i], score=float(scores[i])) for i in order]
    _same_hits(s.search(env["root"], query, top_k=3), exp, TOL)
    # use_reranker = 0 switches it off for one call
    _same_hits(s.search(env["root"], query, top_k=3, rerank=False), orc.search_hybrid(query, q, 3), 1e-7)


def test_searcher_errors(env, tmp_path):
    import ctypes as C
    import kjarni_amd
    from kjarni_amd import _ffi
    with pytest.raises(Exception):
        kjarni_amd.Searcher(model="no-such-model", cache_dir=env["cache"])
    with pytest.raises(Exception):                                        # model present, not in this cache
        kjarni_amd.Searcher(cache_dir=str(tmp_path))
    s = kjarni_amd.Searcher(cache_dir=env["cache"])
    res = _ffi.KjarniSearchResults()
    L = _ffi.lib()
    assert L.kjarni_searcher_search(s._handle, str(tmp_path / "none").encode(), b"q", C.byref(res)) == \
        _ffi.KjarniError.INFERENCE_FAILED
    wrong = str(tmp_path / "dim4")
    SO.write_index(wrong, 4, [("x y", [1.0, 0, 0, 0], {})])
    assert L.kjarni_searcher_search(s._handle, wrong.encode(), b"q", C.byref(res)) == _ffi.KjarniError.INVALID_CONFIG
    assert b"dimension" in L.kjarni_last_error_message().lower()
    empty = str(tmp_path / "empty")
    SO.write_index(empty, 384, [])
    assert s.search(empty, "anything") == []
