"""Host-side search logic (BM25, rank fusion, glob / metadata filter, index reading) through the
C ABI, against the reference's own unit-test expectations and oracle/search_oracle.py.

Reference tests restated here: crates/kjarni-search/src/bm25.rs:199-560, hybrid.rs:36-62,
crates/kjarni-rag/src/index_reader.rs:654-885, segment.rs:377-436, tests.rs:12-80."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

from kjarni_amd import _ffi
from kjarni_amd.searcher import (bm25_tokenize, glob_match, index_search, rrf_fuse, search_keywords,
                                 set_keyword_parallel_min_docs)
from oracle import search_oracle as SO


# ---------------------------------------------------------------- struct layout (searcher.rs:25-165)
def test_struct_sizes():
    assert C.sizeof(_ffi.KjarniSearchResult) == 32
    assert C.sizeof(_ffi.KjarniSearchResults) == 16
    assert C.sizeof(_ffi.KjarniSearchOptions) == 48
    assert C.sizeof(_ffi.KjarniSearcherConfig) == 56


def test_defaults():
    o = _ffi.lib().kjarni_search_options_default()
    assert (o.mode, o.top_k, o.use_reranker, o.threshold) == (-1, 0, -1, 0.0)
    assert o.source_pattern is None and o.filter_key is None and o.filter_value is None
    c = _ffi.lib().kjarni_searcher_config_default()
    assert (c.device, c.default_mode, c.default_top_k, c.quiet) == (0, 2, 10, 0)
    assert c.cache_dir is None and c.model_name is None and c.rerank_model is None


# ---------------------------------------------------------------- tokenizer (bm25.rs:216-241)
@pytest.mark.parametrize("text,expect", [
    ("Hello World", ["hello", "world"]),
    ("I am a test", ["am", "test"]),
    ("hello, world! how are you?", ["hello", "world", "how", "are", "you"]),
    ("", []),
    ("   ", []),
])
def test_tokenize_reference_cases(text, expect):
    assert bm25_tokenize(text) == expect
    assert SO.tokenize(text) == expect


def test_tokenize_unicode_matches_oracle():
    texts = ["Ólafur Jóhannsson skrifaði Kjarni", "naïve café — déjà vu", "東京 タワー 2024年", "ǅ ẞ İstanbul ΣΊΣΥΦΟΣ",
             "a_b c-d e.f x1 ٣٤٥ ①②", "tab\there\nnew line", "ﬁn ﬂ ½ ² ª", "🙂 emoji ok"]
    for t in texts:
        assert bm25_tokenize(t) == SO.tokenize(t), t


# ---------------------------------------------------------------- rank fusion (hybrid.rs:36-62)
def test_rrf_reference_cases():
    r = rrf_fuse([0, 1], [1, 2], 10)
    assert r[0][0] == 1
    assert rrf_fuse([], [], 10) == []
    assert len(rrf_fuse([0, 1, 2], [3, 4, 5], 2)) == 2


def test_rrf_matches_oracle():
    rng = np.random.default_rng(5)
    for _ in range(20):
        kw = rng.permutation(50)[:rng.integers(0, 30)].tolist()
        sem = rng.permutation(50)[:rng.integers(0, 30)].tolist()
        lim = int(rng.integers(1, 40))
        got = rrf_fuse(kw, sem, lim)
        exp = SO.hybrid_search([(i, 0.0) for i in kw], [(i, 0.0) for i in sem], lim)
        assert [g[0] for g in got] == [e[0] for e in exp]
        np.testing.assert_array_equal(np.float32([g[1] for g in got]), np.float32([e[1] for e in exp]))


# ---------------------------------------------------------------- glob (index_reader.rs:654-885)
@pytest.mark.parametrize("pattern,path,expect", [
    ("*.txt", "document.txt", True), ("*.txt", "document.md", False), ("docs/*.md", "docs/readme.md", True),
    ("README.md", "README.md", True), ("README.md", "README.txt", False), ("file[1].txt", "file1.txt", True),
    ("file\\[1\\].txt", "file[1].txt", True), ("test_*", "test_file.txt", True), ("test_*", "other_file.txt", False),
    ("file?.txt", "file1.txt", True), ("file?.txt", "file12.txt", False), ("**/*.md", "deep/nested/path/doc.md", True),
    ("*.md", "path/to/document.md", False), ("*.{md,txt}", "a.txt", True), ("*.{md,txt}", "a.rs", False),
    ("[!a-c]x", "dx", True), ("[!a-c]x", "bx", False), ("**", "a/b/c", True), ("a/**/z", "a/z", True),
])
def test_glob(pattern, path, expect):
    assert glob_match(pattern, path) is expect
    assert SO.glob_match(pattern, path) is expect


def test_glob_fuzz_against_oracle():
    rng = np.random.default_rng(11)
    atoms = ["a", "b", "/", "*", "**", "?", "[ab]", "[!a]", "{a,b}", "{a/b,*}", ".md", "\\*", "**/", "/**", "é", "!"]
    paths = ["", "a", "b", "a/b", "a/b/c.md", "ab.md", "b/a.md", "*", "a*", "é.md", "a/é", "a/b/a/b", "!a"]
    n = 0
    for _ in range(1500):
        pat = "".join(rng.choice(atoms, int(rng.integers(1, 6))))
        for path in paths:
            assert glob_match(pat, path) == SO.glob_match(pat, path), (pat, path)
            n += 1
    assert n > 10000


# ---------------------------------------------------------------- index fixtures
DOCS = [
    ("rust programming language", [1.0, 0.0, 0.0, 0.0], {"source": "docs/rust.md", "lang": "en"}),
    ("python scripting", [0.0, 1.0, 0.0, 0.0], {"source": "docs/python.md", "lang": "en"}),
    ("rust is fast", [0.5, 0.5, 0.0, 0.0], {"source": "notes/rust.txt", "lang": "en"}),
    ("Apple is a fruit", [0.9, 0.0, 0.1, 0.0], {"category": "fruit", "source": "fruit.txt"}),
    ("Car is a vehicle", [0.0, 0.0, 1.0, 0.0], {}),
    ("Banana is yellow", [0.0, 0.1, 0.9, 0.1], {"source": "docs/banana.md", "lang": "is"}),
    ("the rust belt is a region; rust never sleeps, rust rust", [0.0, 0.0, 0.0, 1.0], {"source": "x/y/rust2.md"}),
]


def _docs(n, dim, seed):
    rng = np.random.default_rng(seed)
    words = ["alpha", "beta", "gamma", "delta", "rust", "python", "vector", "index", "search", "kernel", "wave",
             "matrix", "ísland", "fjörður", "tokyo", "東京"]
    out = []
    for i in range(n):
        k = int(rng.integers(1, 12))
        text = " ".join(rng.choice(words, k))
        md = {"source": f"dir{i % 3}/file{i}.{'md' if i % 2 else 'txt'}", "bucket": str(i % 4)}
        out.append((text, rng.standard_normal(dim).astype(np.float32), md))
    return out


@pytest.fixture(scope="module")
def small_index(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("idx") / "small")
    SO.write_index(root, 4, DOCS, max_docs_per_segment=3)
    return root


@pytest.fixture(scope="module")
def big_index(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("idx") / "big")
    docs = _docs(500, 16, 3)
    SO.write_index(root, 16, docs, max_docs_per_segment=128)
    return root, docs


def _same(got, exp, tol=0.0):
    assert [g["document_id"] for g in got] == [e["document_id"] for e in exp]
    assert [g["text"] for g in got] == [e["text"] for e in exp]
    assert [g["metadata"] for g in got] == [e["metadata"] for e in exp]
    if tol == 0.0:
        np.testing.assert_array_equal(np.float32([g["score"] for g in got]), np.float32([e["score"] for e in exp]))
    else:
        np.testing.assert_allclose([g["score"] for g in got], [e["score"] for e in exp], atol=tol)


# ---------------------------------------------------------------- keyword search (segment.rs:413-436, bm25.rs)
def test_keywords_reference_segment_case(small_index):
    ids = [r["document_id"] for r in search_keywords(small_index, "rust", 10)]
    assert 0 in ids and 2 in ids and 6 in ids and 1 not in ids


def test_keywords_match_oracle_bit_exact(small_index, big_index):
    orc = SO.IndexOracle(DOCS, 3)
    for q in ["rust", "rust fast", "is", "banana yellow fruit", "nothing here", "", "a"]:
        for k in (1, 3, 10):
            _same(search_keywords(small_index, q, k), orc.search_keywords(q, k))
    root, docs = big_index
    orc = SO.IndexOracle(docs, 128)
    for q in ["rust python", "ísland fjörður", "東京", "alpha beta gamma delta kernel", "wave"]:
        for k in (5, 50, 600):
            _same(search_keywords(root, q, k), orc.search_keywords(q, k))


def test_keywords_repeated_terms_and_dense_matches_bit_exact(tmp_path):
    # The posting-list accumulation must add in the order score() does (bm25.rs:70-94): repeated query terms count
    # again, and a query that matches most of a segment is still ranked exactly.
    rng = np.random.default_rng(17)
    words = ["glacier", "fjord", "basalt", "wool", "geyser", "harbour", "lights", "river"]
    docs = [(" ".join(rng.choice(words, int(rng.integers(1, 14)))) + f" d{i}", rng.standard_normal(4).astype(np.float32),
             {"source": f"f{i % 7}.txt"}) for i in range(1500)]
    root = str(tmp_path / "dense")
    SO.write_index(root, 4, docs, max_docs_per_segment=400)
    orc = SO.IndexOracle(docs, 400)
    for q in ["glacier fjord basalt", "fjord fjord glacier fjord", "wool wool", "river d17 lights", "d3 d3 d4"]:
        for k in (1, 10, 2000):
            _same(search_keywords(root, q, k), orc.search_keywords(q, k))


def test_keywords_on_several_host_threads_are_the_same_hits(tmp_path):
    # Large indexes walk their segments on a few host threads (index.cpp, IndexReader::search_keywords): same hits, same scores,
    # same order as one thread and as the oracle.
    rng = np.random.default_rng(23)
    words = ["glacier", "fjord", "basalt", "wool", "geyser", "harbour", "lights", "river"]
    docs = [(" ".join(rng.choice(words, int(rng.integers(1, 14)))) + f" d{i}", rng.standard_normal(4).astype(np.float32),
             {"source": f"f{i % 7}.txt"}) for i in range(1300)]
    root = str(tmp_path / "threads")
    SO.write_index(root, 4, docs, max_docs_per_segment=100)   # 13 segments
    orc = SO.IndexOracle(docs, 100)
    try:
        for q in ["glacier fjord basalt", "wool wool", "river d17 lights", "nothing-matches"]:
            for k in (1, 10, 2000):
                set_keyword_parallel_min_docs(2 ** 62)
                one = search_keywords(root, q, k)
                set_keyword_parallel_min_docs(0)
                many = search_keywords(root, q, k)
                _same(many, one)
                _same(many, orc.search_keywords(q, k))
    finally:
        set_keyword_parallel_min_docs(50000)


def test_rewritten_index_is_not_served_from_the_segment_cache(tmp_path):
    # parsed segments are kept across queries, keyed by directory and revalidated by each file's stat()
    import shutil
    root = str(tmp_path / "again")
    for docs in (DOCS, [(t.replace("rust", "zig"), v, m) for t, v, m in DOCS], DOCS[:4]):
        SO.write_index(root, 4, docs, max_docs_per_segment=3)
        orc = SO.IndexOracle(docs, 3)
        for q in ("rust", "zig", "is"):
            _same(search_keywords(root, q, 10), orc.search_keywords(q, 10))
            _same(search_keywords(root, q, 10), orc.search_keywords(q, 10))   # second query: cached segments
        shutil.rmtree(root)


def test_index_search_keyword_mode_and_filters(small_index, big_index):
    # keyword mode of the retrieval hook needs no GPU
    orc = SO.IndexOracle(DOCS, 3)
    _same(index_search(small_index, "rust", mode="keyword", top_k=5), orc.search_keywords("rust", 5))
    got = index_search(small_index, "rust", mode="keyword", top_k=5, source_pattern="*.md")
    assert [r["document_id"] for r in got] == [r["document_id"] for r in orc.search_keywords("rust", 15)
                                               if r["metadata"].get("source", "").endswith(".md")][:5]
    got = index_search(small_index, "is", mode="keyword", top_k=5, filter_key="lang", filter_value="is")
    assert [r["text"] for r in got] == ["Banana is yellow"]
    got = index_search(small_index, "rust", mode="keyword", top_k=5, source_pattern="docs/*.md")
    assert [r["document_id"] for r in got] == [0]
    # threshold keeps score >= t
    allr = orc.search_keywords("rust", 5)
    t = allr[1]["score"]
    got = index_search(small_index, "rust", mode="keyword", top_k=5, threshold=t)
    assert [r["document_id"] for r in got] == [r["document_id"] for r in allr if np.float32(r["score"]) >= np.float32(t)]
    # filtered over-fetch on a bigger index: 3x candidates, then take top_k (index_reader.rs:107-158)
    root, docs = big_index
    orc = SO.IndexOracle(docs, 128)
    f = SO.MetadataFilter().must("bucket", "2").source("*.txt")
    cand = orc.search_keywords("rust python", 30)
    exp = [r for r in cand if f.matches(r["metadata"])][:10]
    _same(index_search(root, "rust python", mode="keyword", top_k=10, filter_key="bucket", filter_value="2",
                       source_pattern="*.txt"), exp)


def test_open_errors(tmp_path):
    res = _ffi.KjarniSearchResults()
    rc = _ffi.lib().kjarni_search_keywords(str(tmp_path / "missing").encode(), b"x", 5, C.byref(res))
    assert rc == _ffi.KjarniError.INFERENCE_FAILED
    assert b"Search failed" in _ffi.lib().kjarni_last_error_message()
    assert _ffi.lib().kjarni_search_keywords(None, b"x", 5, C.byref(res)) == _ffi.KjarniError.NULL_POINTER
    assert _ffi.lib().kjarni_search_keywords(b"x", b"\xff\xfe", 5, C.byref(res)) == _ffi.KjarniError.INVALID_UTF8
    # a segment that fails to load is skipped with a warning, not an error (index_reader.rs:182-187):
    # truncate the first segment's bm25.bin and only the second segment answers
    root = str(tmp_path / "broken")
    SO.write_index(root, 4, DOCS, max_docs_per_segment=3)
    p = os.path.join(root, "segments", "seg_000000", "bm25.bin")
    blob = open(p, "rb").read()
    open(p, "wb").write(blob[:len(blob) // 2])
    got = search_keywords(root, "rust", 10)
    exp = SO.IndexOracle(DOCS[3:], 3).search_keywords("rust", 10)
    _same(got, exp)
    # config.json missing is an error
    os.remove(os.path.join(root, "config.json"))
    rc = _ffi.lib().kjarni_search_keywords(root.encode(), b"rust", 5, C.byref(res))
    assert rc == _ffi.KjarniError.INFERENCE_FAILED


def test_empty_index(tmp_path):
    root = str(tmp_path / "empty")
    SO.write_index(root, 4, [])
    assert search_keywords(root, "rust", 5) == []


def test_searcher_null_handles():
    L = _ffi.lib()
    assert L.kjarni_searcher_has_reranker(None) is False
    assert L.kjarni_searcher_default_top_k(None) == 10
    assert L.kjarni_searcher_default_mode(None) == 2
    assert L.kjarni_searcher_model_name(None, None, 0) == 0
    res = _ffi.KjarniSearchResults()
    assert L.kjarni_searcher_search(None, b"a", b"b", C.byref(res)) == _ffi.KjarniError.NULL_POINTER
    L.kjarni_searcher_free(None)
    L.kjarni_search_results_free(None)
    L.kjarni_search_results_free(C.byref(_ffi.KjarniSearchResults()))


# ---------------------------------------------------------------- reference BM25 unit tests (bm25.rs:243-560)
def _manual(total_docs, doc_lengths, avg, df, inv):
    b = SO.Bm25Index()
    b.total_docs, b.doc_lengths, b.avg_doc_length = total_docs, list(doc_lengths), np.float32(avg)
    b.doc_frequencies, b.inverted_index = dict(df), {k: list(v) for k, v in inv.items()}
    return b


def _lib_search(tmp_path, bm, query, limit, name):
    """Serve a hand-built Bm25Index through the library: one segment whose bm25.bin is `bm`."""
    root = str(tmp_path / name)
    n = max(bm.total_docs, 1)
    SO.write_index(root, 2, [(f"doc{i}", [1.0, 0.0], {}) for i in range(n)])
    with open(os.path.join(root, "segments", "seg_000000", "bm25.bin"), "wb") as f:
        f.write(bm.to_bincode())
    return [(r["document_id"], r["score"]) for r in search_keywords(root, query, limit)]


def _both(tmp_path, bm, query, limit, name):
    exp = bm.search(query, limit)
    got = _lib_search(tmp_path, bm, query, limit, name)
    assert [g[0] for g in got] == [e[0] for e in exp]
    np.testing.assert_array_equal(np.float32([g[1] for g in got]), np.float32([e[1] for e in exp]))
    return exp


def test_bm25_reference_search_cases(tmp_path):
    assert SO.Bm25Index().search("test query", 10) == []                                   # empty index
    assert _manual(1, [10], 10.0, {}, {}).search("", 10) == []                             # empty query
    bm = _manual(3, [5, 3, 7], 5.0,
                 dict(rust=2, programming=2, language=1, python=1, fast=1, safe=1),
                 dict(rust=[(0, 1), (2, 1)], programming=[(0, 1), (1, 1)], language=[(0, 1)], python=[(1, 1)],
                      fast=[(2, 1)], safe=[(2, 1)]))
    ids = [d for d, _ in _both(tmp_path, bm, "rust", 10, "a")]
    assert 0 in ids and 2 in ids and 1 not in ids
    r = _both(tmp_path, _manual(2, [10, 10], 10.0, dict(test=2), dict(test=[(0, 1), (1, 3)])), "test", 10, "b")
    assert [d for d, _ in r] == [1, 0] and r[0][1] > r[1][1]                               # score ordering
    bm = _manual(10, [10] * 10, 10.0, dict(rare=1, common=9), dict(rare=[(0, 1)], common=[(i, 1) for i in range(9)]))
    assert bm.score(["rare"], 0) > bm.score(["common"], 0)                                 # idf effect
    assert _lib_search(tmp_path, bm, "rare", 1, "c")[0][1] > _lib_search(tmp_path, bm, "common", 10, "d")[0][1]
    bm = _manual(2, [5, 50], 27.5, dict(test=2), dict(test=[(0, 1), (1, 1)]))
    assert bm.score(["test"], 0) > bm.score(["test"], 1)                                   # length normalisation
    assert [d for d, _ in _both(tmp_path, bm, "test", 10, "e")] == [0, 1]
    bm = _manual(10, [10] * 10, 10.0, dict(test=10), dict(test=[(i, 1) for i in range(10)]))
    assert len(_both(tmp_path, bm, "test", 3, "f")) == 3                                   # limit
    bm = _manual(3, [10, 10, 10], 10.0, dict(rust=2, fast=2), dict(rust=[(0, 1), (2, 1)], fast=[(1, 1), (2, 1)]))
    assert _both(tmp_path, bm, "rust fast", 10, "g")[0][0] == 2                            # multi-term


def test_bm25_reference_add_document_cases(tmp_path):
    b = SO.Bm25Index()
    b.inverted_index["hello"] = [(0, 5), (2, 3)]
    assert [b.term_frequency("hello", i) for i in range(3)] == [5, 0, 3] and b.term_frequency("nonexistent", 0) == 0
    b = SO.Bm25Index()
    b.add_document(0, "rust programming language")
    assert (b.total_docs, b.doc_lengths[0], float(b.avg_doc_length)) == (1, 3, 3.0)
    assert b.doc_frequencies == dict(rust=1, programming=1, language=1) and b.term_frequency("rust", 0) == 1
    b = SO.Bm25Index()
    for i, t in enumerate(["rust is fast", "python is slow", "rust and python"]):
        b.add_document(i, t)
    assert b.total_docs == 3 and abs(float(b.avg_doc_length) - 3.0) < 0.01
    assert [b.doc_frequencies[k] for k in ("rust", "python", "is", "fast")] == [2, 2, 2, 1]
    b = SO.Bm25Index()
    b.add_document(0, "test test test hello")
    assert (b.term_frequency("test", 0), b.term_frequency("hello", 0), b.doc_frequencies["test"]) == (3, 1, 1)
    b = SO.Bm25Index()
    for i, t in enumerate(["the quick brown fox", "the lazy dog", "quick quick fox jumps"]):
        b.add_document(i, t)
    r = _both(tmp_path, b, "quick fox", 10, "h")
    assert r[0][0] == 2 and r[1][0] == 0 and all(d != 1 for d, _ in r)
    b = SO.Bm25Index()
    b.add_document(0, "")
    assert (b.total_docs, b.doc_lengths[0], b.inverted_index) == (1, 0, {})
    b = SO.Bm25Index()
    b.add_document(0, "first doc")
    b.add_document(5, "fifth doc")
    assert (b.total_docs, len(b.doc_lengths), b.doc_lengths[0], b.doc_lengths[5], b.doc_lengths[3]) == (6, 6, 2, 2, 0)
    _both(tmp_path, b, "doc fifth", 10, "i")


# ---------------------------------------------------------------- reference index tests (tests.rs:12-80, segment.rs:377-411)
def test_reference_lifecycle_documents_and_metadata(tmp_path):
    root = str(tmp_path / "my_index")
    docs = [("Apple is a fruit", [1.0, 0.0, 0.0, 0.0], {"category": "fruit"}),
            ("Car is a vehicle", [0.0, 1.0, 0.0, 0.0], {}),
            ("Banana is yellow", [0.9, 0.1, 0.0, 0.0], {})]
    SO.write_index(root, 4, docs, max_docs_per_segment=2)
    assert sorted(os.listdir(os.path.join(root, "segments"))) == ["seg_000000", "seg_000001"]
    r = search_keywords(root, "banana", 10)
    assert [(x["document_id"], x["text"], x["metadata"]) for x in r] == [(2, "Banana is yellow", {})]
    r = search_keywords(root, "fruit", 10)
    assert r[0]["text"] == "Apple is a fruit" and r[0]["metadata"] == {"category": "fruit"}


def test_by_value_free_twins():
    """The stale cbindgen header / C# / Python bindings free by value (Native.cs:376-394); the twins take
    exactly that calling convention."""
    L = _ffi.lib()
    arr = _ffi.KjarniStringArray()
    assert L.kjarni_bm25_tokenize(b"hello big world", C.byref(arr)) == 0 and arr.len == 3
    L.kjarni_string_array_free_by_value(arr)
    res = _ffi.KjarniSearchResults()
    L.kjarni_search_results_free_by_value(res)                    # empty = {NULL, 0}
    L.kjarni_float_array_free_by_value(_ffi.KjarniFloatArray())
    L.kjarni_float_2d_array_free_by_value(_ffi.KjarniFloat2DArray())
    L.kjarni_class_results_free_by_value(_ffi.KjarniClassResults())
    L.kjarni_rerank_results_free_by_value(_ffi.KjarniRerankResults())
