"""String-level kjarni-ffi surface on the GPU vs the oracle (texts tokenised by
the library's own tokenizer, which tests/test_tokenizer.py pins separately)."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4

TEXTS = ["Hello world!", "Reykjavík is the capital of Iceland.", "semantic search with sentence vectors",
         "a", "The quick brown fox jumps over the lazy dog 1234567890 times."]


@pytest.fixture(scope="module")
def embed_env(tmp_path_factory):
    import kjarni_amd
    cache = str(tmp_path_factory.mktemp("cache"))
    d = os.path.join(cache, "sentence-transformers_all-MiniLM-L6-v2")   # <cache>/<org>_<repo>/
    cfg, t = synth.minilm_embedder(d, seed=3)
    synth.add_tokenizer(d)
    return cache, d, O.OracleModel(t, cfg), kjarni_amd.Tokenizer(os.path.join(d, "tokenizer.json"), 512)


def test_embedder_registry_name_and_encode_batch(embed_env):
    import kjarni_amd
    cache, d, orc, tok = embed_env
    emb = kjarni_amd.Embedder("minilm-l6-v2", cache_dir=cache)          # registry name + cache dir
    assert emb.dim == 384
    got = emb.encode_batch(TEXTS)
    ids, mask, _ = tok.encode_batch(TEXTS)
    ref = orc.embed_batch(ids, mask)                                     # mean + L2, AUTO mask
    assert got.shape == (len(TEXTS), 384)
    assert np.abs(got - ref).max() < TOL
    # HF alias, case-insensitive (registry.rs:753-766)
    emb2 = kjarni_amd.Embedder("sentence-transformers/all-MiniLM-L6-v2", cache_dir=cache)
    # (a different call size: up to 256 tokens in all, 257 .. 8192 and more take projection kernels that sum in different orders)
    assert np.abs(emb2.encode_batch(TEXTS[:2]) - got[:2]).max() < 1e-6
    assert np.array_equal(emb2.encode_batch(TEXTS), got)


def test_embedder_encode_normalize_flag_and_similarity(embed_env):
    import kjarni_amd
    cache, d, orc, tok = embed_env
    ids, mask, _ = tok.encode_batch([TEXTS[1]])
    h = orc.forward(ids, mask, None, O.strategy_mask_value(ids.size))
    pooled = O.mean_pool(h, mask.astype(np.float32))
    e_norm = kjarni_amd.Embedder(model_path=d, normalize=True)
    e_raw = kjarni_amd.Embedder(model_path=d, normalize=False)
    assert np.abs(np.array(e_norm.encode(TEXTS[1])) - O.l2_normalize(pooled)[0]).max() < TOL
    assert np.abs(np.array(e_raw.encode(TEXTS[1])) - pooled[0]).max() < TOL      # embed honours normalize
    # encode_batch ignores normalize (always L2: sentence_encoder/model.rs:211)
    assert np.abs(e_raw.encode_batch([TEXTS[1]])[0] - O.l2_normalize(pooled)[0]).max() < TOL
    # similarity = cosine of the two (mean, config-normalize) embeddings of ONE batch of two
    ids2, mask2, _ = tok.encode_batch([TEXTS[0], TEXTS[2]])
    h2 = orc.forward(ids2, mask2, None, O.strategy_mask_value(ids2.size))
    p2 = O.mean_pool(h2, mask2.astype(np.float32))
    assert abs(e_raw.similarity(TEXTS[0], TEXTS[2]) - O.cosine_k(p2[0], p2[1])) < TOL
    assert abs(e_norm.similarity(TEXTS[0], TEXTS[0]) - 1.0) < 1e-5


def test_reranker_scores_order_and_top_k(tmp_path):
    import kjarni_amd
    d = str(tmp_path / "ce")
    cfg, t = synth.minilm_cross_encoder(d, seed=4)
    synth.add_tokenizer(d)
    orc = O.OracleModel(t, cfg)
    tok = kjarni_amd.Tokenizer(os.path.join(d, "tokenizer.json"), 512)
    rr = kjarni_amd.Reranker(model_path=d)
    query = "what is the capital of iceland"
    docs = TEXTS + ["Iceland's capital city is Reykjavík."]
    ids, mask, types = tok.encode_batch([query] * len(docs), docs)
    ref = orc.rerank_scores(ids, mask, types)
    res = rr.rerank(query, docs)
    assert sorted(r.index for r in res) == list(range(len(docs)))
    for r in res:
        assert abs(r.score - ref[r.index]) < TOL
    assert [r.index for r in res] == sorted(range(len(docs)), key=lambda i: -ref[i])   # desc, stable
    top = rr.rerank_top_k(query, docs, 2)
    assert [r.index for r in top] == [r.index for r in res[:2]]
    assert rr.rerank_top_k(query, docs, 100) == res
    assert rr.rerank(query, []) == []
    # single pair: tokenised alone (no padding), same score
    ids1, mask1, types1 = tok.encode_batch([query], [docs[1]])
    assert abs(rr.score(query, docs[1]) - orc.rerank_scores(ids1, mask1, types1)[0]) < TOL


def test_classifier_distilbert_softmax_labels_and_multilabel(tmp_path):
    import kjarni_amd
    d = str(tmp_path / "sst2")
    cfg, t = synth.distilbert_sentiment(d, seed=5, n_layers=2)   # 2 layers keep the CPU oracle quick
    synth.add_tokenizer(d)
    orc = O.OracleModel(t, cfg)
    tok = kjarni_amd.Tokenizer(os.path.join(d, "tokenizer.json"), 512)
    clf = kjarni_amd.Classifier(model_path=d)
    assert clf.num_labels == 2 and clf.labels() == ["NEGATIVE", "POSITIVE"]
    text = "this movie was surprisingly good"
    ids, mask, _ = tok.encode_batch([text])
    logits = orc.head_logits(orc.forward(ids, mask, None, O.MASK_ALLOC))[0]
    probs = O.softmax_rows(logits[None, :])[0]
    got = clf.classify(text)
    assert [l for l, _ in got] == [["NEGATIVE", "POSITIVE"][i] for i in np.argsort(-probs, kind="stable")]
    for label, score in got:
        assert abs(score - probs[["NEGATIVE", "POSITIVE"].index(label)]) < TOL
    assert abs(sum(s for _, s in got) - 1.0) < 1e-5
    # custom labels + multi-label (sigmoid)
    clf2 = kjarni_amd.Classifier(model_path=d, labels=["bad", "good"], multi_label=True)
    got2 = dict(clf2.classify(text))
    sig = 1.0 / (1.0 + np.exp(-logits))
    assert abs(got2["bad"] - sig[0]) < TOL and abs(got2["good"] - sig[1]) < TOL
    with pytest.raises(kjarni_amd.KjarniException) as ei:
        kjarni_amd.Classifier(model_path=d, labels=["only-one"])
    assert ei.value.code == kjarni_amd.KjarniError.LOAD_FAILED


def test_embedder_model_is_not_a_reranker(embed_env):
    import kjarni_amd
    cache, d, _, _ = embed_env
    with pytest.raises(kjarni_amd.KjarniException) as ei:
        kjarni_amd.Reranker(model_path=d)          # no classification head
    assert ei.value.code == kjarni_amd.KjarniError.LOAD_FAILED


def test_long_text_is_truncated_to_max_seq_len(embed_env):
    import kjarni_amd
    cache, d, orc, tok = embed_env
    emb = kjarni_amd.Embedder(model_path=d)
    long_text = "semantic search " * 600
    ids, mask, _ = tok.encode_batch([long_text])
    assert ids.shape[1] == 512                      # loader.rs:108-111
    got = emb.encode_batch([long_text, "short"])
    ids2, mask2, _ = tok.encode_batch([long_text, "short"])
    assert np.abs(got - orc.embed_batch(ids2, mask2)).max() < TOL
