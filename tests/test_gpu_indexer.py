"""Indexer on the GPU (kjarni_indexer_*): files -> chunks -> embeddings -> the reference's on-disk
index, checked against the oracle's loader/splitter, the oracle encoder (1e-4) and an independent
reader of the format; progress events and cancellation points follow
crates/kjarni/src/indexer/model.rs:318-484 one for one.  Reference tests: model.rs:1480-1773."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O
from oracle import search_oracle as SO
from tests import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4

PARAS = ["Reykjavík is the capital of Iceland.", "Rust is a systems programming language focused on safety.",
         "Python is popular for scripting and data science.", "The capital of France is Paris.",
         "Ísland er eyja í Norður-Atlantshafi.", "GPU kernels use matrix cores for fast matrix multiplication.",
         "The quick brown fox jumps over the lazy dog.", "Volcanoes and glaciers shape the landscape."]


@pytest.fixture(scope="module")
def env(tmp_path_factory):
    import kjarni_amd
    cache = str(tmp_path_factory.mktemp("cache"))
    d = os.path.join(cache, "sentence-transformers_all-MiniLM-L6-v2")
    cfg, t = synth.minilm_embedder(d, seed=3)
    synth.add_tokenizer(d)
    orc = O.OracleModel(t, cfg)
    tok = kjarni_amd.Tokenizer(os.path.join(d, "tokenizer.json"), 512)

    def embed(texts):
        out = []
        for t_ in texts:                       # one text at a time: no padding involved at all
            ids, mask, _ = tok.encode_batch([t_])
            out.append(orc.embed_batch(ids, mask)[0])
        return np.stack(out)

    docs = tmp_path_factory.mktemp("docs")
    rng = np.random.default_rng(1)
    for i in range(7):
        paras = [PARAS[j] for j in rng.integers(0, len(PARAS), int(rng.integers(1, 9)))]
        (docs / f"doc{i}.txt").write_text("\n\n".join(paras))
    (docs / "sub").mkdir()
    (docs / "sub" / "notes.md").write_text("# Notes\n\n" + " ".join(PARAS) * 3)        # one oversized section
    (docs / "skip.pdf").write_text("not indexed")
    (docs / "bad.txt").write_bytes(b"\xff\xfe invalid utf-8")                          # counted as skipped
    (docs / "empty.txt").write_text("")                                                # processed, zero chunks
    return dict(cache=cache, embed=embed, docs=str(docs))


def _expected_chunks(docs_dir, chunk_size, overlap, **kw):
    sp = SO.TextSplitter(chunk_size, overlap, "\n\n")
    files = SO.collect_files([docs_dir], **kw)
    chunks, processed, skipped = [], 0, 0
    for f in files:
        try:
            c = SO.load_file_chunks(f, sp)
        except UnicodeDecodeError:
            skipped += 1
            continue
        processed += 1
        chunks += c
    return files, chunks, processed, skipped


def _expected_events(files, per_file_chunks, batch_size, commit_msg):
    """model.rs:318-484 as a trace."""
    ev = [("scanning", 0, 0, "Discovering files...")]
    total_docs, pending = 0, 0
    for i, f in enumerate(files):
        ev.append(("loading", i, len(files), f))
        for _ in range(per_file_chunks.get(f, 0)):
            pending += 1
            if pending >= batch_size:
                ev.append(("embedding", total_docs, 0, None))
                total_docs += pending
                pending = 0
    if pending:
        ev.append(("embedding", total_docs, 0, None))
        total_docs += pending
    ev.append(("committing", total_docs, total_docs, commit_msg))
    return ev


def test_create_matches_oracle_pipeline(env, tmp_path):
    import kjarni_amd
    ix = kjarni_amd.Indexer(cache_dir=env["cache"], chunk_size=120, chunk_overlap=20, batch_size=4, quiet=True)
    assert (ix.model_name, ix.dimension, ix.chunk_size) == ("minilm-l6-v2", 384, 120)
    root = str(tmp_path / "index")
    events = []
    stats = ix.create(root, [env["docs"]], on_progress=lambda p: events.append(tuple(p)))
    files, chunks, processed, skipped = _expected_chunks(env["docs"], 120, 20)
    assert len(chunks) > 20 and skipped == 1
    assert (stats.documents_indexed, stats.chunks_created, stats.dimension, stats.files_processed,
            stats.files_skipped) == (len(chunks), len(chunks), 384, processed, skipped)
    assert stats.size_bytes == kjarni_amd.index_info(root).size_bytes > 0
    per_file = {}
    for _, md in chunks:
        per_file[md["source"]] = per_file.get(md["source"], 0) + 1
    assert events == _expected_events(files, per_file, 4, "Finalizing index...")

    r = SO.read_index(root)
    assert r["config"]["embedding_model"] == "minilm-l6-v2" and r["config"]["dimension"] == 384
    assert r["index"] == {"total_docs": len(chunks), "segment_count": 1, "dimension": 384}
    seg = r["segments"][0]
    assert seg["texts"] == [c[0] for c in chunks]
    assert seg["metadata"] == [c[1] for c in chunks]
    ref = env["embed"]([c[0] for c in chunks])
    assert np.abs(seg["vectors"] - ref).max() < TOL
    np.testing.assert_allclose(np.linalg.norm(seg["vectors"], axis=1), 1.0, atol=1e-5)
    bm = SO.Bm25Index()
    for i, (t_, _) in enumerate(chunks):
        bm.add_document(i, t_)
    assert seg["bm25"].inverted_index == bm.inverted_index and seg["bm25"].doc_lengths == bm.doc_lengths

    # coalescing is invisible: one chunk per device pass gives the same index
    os.environ["KJARNI_HIP_INDEX_DEVICE_BATCH"] = "1"
    try:
        root2 = str(tmp_path / "index2")
        ix.create(root2, [env["docs"]])
    finally:
        del os.environ["KJARNI_HIP_INDEX_DEVICE_BATCH"]
    seg2 = SO.read_index(root2)["segments"][0]
    assert seg2["texts"] == seg["texts"] and np.abs(seg2["vectors"] - seg["vectors"]).max() < 1e-5

    # and the Searcher finds a chunk by its own text
    s = kjarni_amd.Searcher(cache_dir=env["cache"])
    hit = s.search(root, chunks[5][0], mode="semantic", top_k=1)[0]
    assert hit["text"] == chunks[5][0] and abs(hit["score"] - 1.0) < TOL


def test_create_exists_force_add_and_errors(env, tmp_path):
    import kjarni_amd
    from kjarni_amd import _ffi
    from kjarni_amd.indexer import index_write
    ix = kjarni_amd.Indexer(cache_dir=env["cache"], quiet=True)           # defaults: 512 / 50 / 32
    root = str(tmp_path / "index")
    d1 = tmp_path / "docs"
    d1.mkdir()
    (d1 / "doc1.txt").write_text("Hello world. This is a test document.")
    (d1 / "doc2.txt").write_text("Another document with different content.")
    st = ix.create(root, [str(d1)])
    assert st.documents_indexed == 2 and st.files_processed == 2 and os.path.isdir(root)
    info = kjarni_amd.index_info(root)
    assert (info.document_count, info.dimension, info.embedding_model) == (2, 384, "minilm-l6-v2")

    L = _ffi.lib()
    stats = _ffi.KjarniIndexStats()
    arr = (C.c_char_p * 1)(str(d1).encode())
    assert L.kjarni_indexer_create(ix._handle, root.encode(), arr, 1, 0, C.byref(stats)) == _ffi.KjarniError.INVALID_CONFIG
    assert b"already exists" in L.kjarni_last_error_message()
    assert ix.create(root, [str(d1)], force=True).documents_indexed == 2
    assert L.kjarni_indexer_create(ix._handle, str(tmp_path / "x").encode(), arr, 0, 0, C.byref(stats)) == \
        _ffi.KjarniError.INVALID_CONFIG                                     # NoInputs
    bad = (C.c_char_p * 1)(str(tmp_path / "missing").encode())
    assert L.kjarni_indexer_create(ix._handle, str(tmp_path / "y").encode(), bad, 1, 0, C.byref(stats)) == \
        _ffi.KjarniError.MODEL_NOT_FOUND
    assert b"Path not found" in L.kjarni_last_error_message()

    # add (model.rs:1588-1627)
    d2 = tmp_path / "more"
    d2.mkdir()
    (d2 / "doc3.txt").write_text("Additional document")
    events = []
    assert ix.add(root, [str(d2)], on_progress=lambda p: events.append(p.stage)) == 1
    assert events == ["scanning", "loading", "embedding", "committing"]
    r = SO.read_index(root)
    assert r["names"] == ["seg_000000", "seg_000001"] and r["index"]["total_docs"] == 3
    assert r["segments"][1]["texts"] == ["Additional document"]
    assert np.abs(r["segments"][1]["vectors"] - env["embed"](["Additional document"])).max() < TOL
    n = C.c_size_t(9)
    assert L.kjarni_indexer_add(ix._handle, root.encode(), arr, 0, C.byref(n)) == 0 and n.value == 0
    assert L.kjarni_indexer_add(ix._handle, str(tmp_path / "none").encode(), arr, 1, C.byref(n)) == \
        _ffi.KjarniError.MODEL_NOT_FOUND
    small = str(tmp_path / "dim4")
    index_write(small, 4, ["x y"], [[1, 0, 0, 0]])
    assert L.kjarni_indexer_add(ix._handle, small.encode(), arr, 1, C.byref(n)) == _ffi.KjarniError.INVALID_CONFIG
    assert b"Dimension mismatch: index has 4, model produces 384" in L.kjarni_last_error_message()


def test_segments_roll_over_and_cancel(env, tmp_path):
    import kjarni_amd
    from kjarni_amd import _ffi
    ix = kjarni_amd.Indexer(cache_dir=env["cache"], chunk_size=60, chunk_overlap=0, batch_size=8, quiet=True)
    # cancel from inside the progress callback at the second embedding step
    token = kjarni_amd.CancelToken()
    seen = []

    def on_progress(p):
        seen.append(p.stage)
        if seen.count("embedding") == 2:
            token.cancel()
    root = str(tmp_path / "cancelled")
    with pytest.raises(Exception):
        ix.create(root, [env["docs"]], on_progress=on_progress, cancel_token=token)
    assert "committing" not in seen and not os.path.exists(os.path.join(root, "index.json"))
    L = _ffi.lib()
    st = _ffi.KjarniIndexStats()
    arr = (C.c_char_p * 1)(env["docs"].encode())
    cb = _ffi.KjarniProgressCallbackFn()
    rc = L.kjarni_indexer_create_with_callback(ix._handle, root.encode(), arr, 1, 1, cb, None, token._handle, C.byref(st))
    assert rc == _ffi.KjarniError.CANCELLED and b"cancelled" in L.kjarni_last_error_message()
    token.reset()
    st2 = ix.create(root, [env["docs"]], force=True, cancel_token=token)
    assert st2.documents_indexed > 0 and os.path.exists(os.path.join(root, "index.json"))


def test_invalid_chunking_is_rejected_at_new(env):
    import kjarni_amd
    with pytest.raises(Exception):
        kjarni_amd.Indexer(cache_dir=env["cache"], chunk_size=50, chunk_overlap=50)
    with pytest.raises(Exception):
        kjarni_amd.Indexer(model="no-such-model", cache_dir=env["cache"])
