"""RoBERTa (classification), MPNet and Nomic (embedding) checkpoints through the C ABI against the oracle: the other encoder
families the reference's SequenceClassifier / SentenceEncoder load (sequence_classifier/mod.rs:52-64,
sentence_encoder/model.rs:41-54; Nomic is BertConfig with model_type "nomic_bert", sentence_encoder/configs.rs:140-275).  Both place positions at offset 2; RoBERTa frames with <s> </s> over byte-level BPE,
MPNet with <s> </s> over WordPiece and uses the tanh GELU."""
import os
import shutil

import numpy as np
import pytest

from oracle import oracle as O
from tests import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TEXTS = ["I love this, it's wonderful!", "meh.", "The service was slow and the food was cold, never again", "þæö 日本語 😀", "a"]


def _tokenizer(d, name):
    shutil.copy(os.path.join(GOLDEN, f"{name}_tokenizer.json"), os.path.join(d, "tokenizer.json"))
    return os.path.join(d, "tokenizer.json")


@pytest.mark.parametrize("family", ["init", "trained"])
def test_roberta_classifier_matches_oracle(tmp_path, family):
    import kjarni_amd
    d = str(tmp_path / "roberta")
    cfg, t = synth.roberta_classifier(d, family=family)
    tok = kjarni_amd.Tokenizer(_tokenizer(d, "roberta"), cfg["max_position_embeddings"])
    orc = O.OracleModel(t, cfg)
    clf = kjarni_amd.Classifier(model_path=d)
    labels = ["negative", "neutral", "positive"]
    assert clf.num_labels == 3 and clf.labels() == labels
    for text in TEXTS:
        ids, mask, _ = tok.encode_batch([text])
        assert ids[0, 0] == 0 and ids[0, int(mask[0].sum()) - 1] == 2  # <s> ... </s>
        logits = orc.head_logits(orc.forward(ids, mask, None, O.strategy_mask_value(ids.size)))[0]
        probs = O.softmax_rows(logits[None, :])[0]
        got = dict(clf.classify(text))
        for i, label in enumerate(labels):
            assert abs(got[label] - probs[i]) < TOL, (text, label)
    # token level: a padded batch (pad id 0 = <s>, masked) and the position offset
    enc = kjarni_amd.HipEncoder(d)
    ids, mask, _ = tok.encode_batch(TEXTS)
    want = orc.forward(ids, mask, None, O.strategy_mask_value(ids.size))
    got = enc.hidden_states(ids, mask)
    valid = mask.astype(bool)
    assert np.abs(got[valid] - want[valid]).max() < TOL
    # the same weights read as plain BERT (positions from 0) must differ: the offset is really applied
    cfg0 = dict(cfg, model_type="bert")
    t0 = {k.replace("roberta.", "bert."): v for k, v in t.items()}
    want0 = O.OracleModel(t0, cfg0).forward(ids, mask, None, O.strategy_mask_value(ids.size))
    assert np.abs(want0[valid] - want[valid]).max() > 1e-2


@pytest.mark.parametrize("family", ["init", "trained"])
def test_mpnet_embedder_matches_oracle(tmp_path, family):
    import kjarni_amd
    d = str(tmp_path / "mpnet")
    cfg, t = synth.mpnet_embedder(d, family=family)
    tok = kjarni_amd.Tokenizer(_tokenizer(d, "mpnet"), cfg["max_position_embeddings"])
    orc = O.OracleModel(t, cfg)
    emb = kjarni_amd.Embedder(model_path=d)
    assert emb.dim == cfg["hidden_size"]
    ids, mask, _ = tok.encode_batch(TEXTS)
    want = orc.embed_batch(ids, mask)
    got = emb.encode_batch(TEXTS)
    assert np.abs(got - want).max() < TOL
    assert np.abs(np.linalg.norm(got, axis=1) - 1.0).max() < 1e-5
    one = emb.encode(TEXTS[2])
    i1, m1, _ = tok.encode_batch([TEXTS[2]])
    assert np.abs(one - orc.embed_batch(i1, m1)[0]).max() < TOL


@pytest.mark.parametrize("over", [{}, dict(n_embd=256, n_head=4, n_inner=512), dict(n_embd=40, n_head=4, n_inner=72)],
                         ids=["d32", "d64", "d10-generic"])
def test_nomic_embedder_matches_oracle(tmp_path, over):
    """RoPE on Q / K, SwiGLU through the GEMM epilogue, fused Wqkv, no biases, no position table."""
    import kjarni_amd
    d = str(tmp_path / "nomic")
    cfg, t = synth.nomic_embedder(d, **over)
    synth.add_tokenizer(d)
    tok = kjarni_amd.Tokenizer(os.path.join(d, "tokenizer.json"), cfg["n_positions"])
    orc = O.OracleModel(t, cfg)
    emb = kjarni_amd.Embedder(model_path=d)
    assert emb.dim == cfg["n_embd"]
    texts = ["the quick brown fox", "a", "rust is fast and the kernel is faster " * 6, "hello world, hello gpu"]
    ids, mask, _ = tok.encode_batch(texts)
    got = emb.encode_batch(texts)
    assert np.abs(got - orc.embed_batch(ids, mask)).max() < TOL
    # token level, ragged batch, both mask conventions; 1 200 tokens take the no-alloc (-inf) strategy
    enc = kjarni_amd.HipEncoder(d)
    assert enc.max_seq_len == cfg["n_positions"]
    for n, seq in ((3, 17), (12, 100)):
        ids, mask = synth.synthetic_ids(n, seq, vocab=cfg["vocab_size"], seed=n, ragged=True)
        mv = O.strategy_mask_value(ids.size)
        want = orc.forward(ids, mask, None, mv)
        got = enc.hidden_states(ids, mask)
        valid = mask.astype(bool)
        assert np.abs(got[valid] - want[valid]).max() < TOL, (n, seq)
    # position matters only through the rotation: the same sentence shifted right by padding on the LEFT is not
    # something the tokenizer produces, but reversing the token order must change the pooled vector
    ids, mask = synth.synthetic_ids(1, 24, vocab=cfg["vocab_size"], seed=9)
    a = enc.embed(ids, mask)
    b = enc.embed(np.ascontiguousarray(ids[:, ::-1]), mask)
    assert np.abs(a - b).max() > 5e-5  # without the rotation mean pooling would be order-blind (differences ~1e-7)


def test_bge_m3_layout_matches_oracle(tmp_path):
    """config.json says "xlm-roberta": read as plain BertConfig (positions from 0, one token type), SentencePiece-Unigram
    tokenizer.json with '<s> $A </s>' framing; ids are pinned against the `tokenizers` goldens in test_unigram_tokenizer.py."""
    import json
    import kjarni_amd
    d = str(tmp_path / "bge")
    cfg, t = synth.xlmr_embedder(d)
    tok = kjarni_amd.Tokenizer(_tokenizer(d, "unigram"), cfg["max_position_embeddings"])
    orc = O.OracleModel(t, cfg)
    emb = kjarni_amd.Embedder(model_path=d)
    texts = ["hello world", "Ísland er fallegt land með fjörðum", "日本語のテキストを東京で 🙂", "café naïve ﬁnance № ① ｆｕｌｌ ½ x² Ⅷ",
             "the quick brown fox jumps over the lazy dog " * 30]
    ids, mask, _ = tok.encode_batch(texts)
    assert ids.shape[1] == cfg["max_position_embeddings"] and ids[4, -1] == 2          # truncated to max_seq_len, </s> kept
    with open(os.path.join(GOLDEN, "unigram_goldens.json")) as f:
        by_text = {c["text"]: c["ids"] for c in json.load(f)["cases"]}
    for i in (0, 1, 3):
        assert ids[i, :int(mask[i].sum())].tolist() == by_text[texts[i]]
    got = emb.encode_batch(texts)
    assert np.abs(got - orc.embed_batch(ids, mask)).max() < TOL
    one = emb.encode(texts[2])
    i1, m1, _ = tok.encode_batch([texts[2]])
    assert np.abs(one - orc.embed_batch(i1, m1)[0]).max() < TOL


def test_nomic_long_sequence(tmp_path):
    """2 048 tokens in one sentence: RoPE rows far from 0 and the tiled attention path."""
    import kjarni_amd
    d = str(tmp_path / "nomic")
    cfg, t = synth.nomic_embedder(d, n_positions=2048, n_layer=1)
    orc = O.OracleModel(t, cfg)
    enc = kjarni_amd.HipEncoder(d)
    ids, mask = synth.synthetic_ids(1, 2048, vocab=cfg["vocab_size"], seed=2)
    want = orc.forward(ids, mask, None, O.strategy_mask_value(ids.size))
    got = enc.hidden_states(ids, mask)
    assert np.abs(got - want).max() < TOL


def test_registry_names_reach_the_new_families(tmp_path):
    import kjarni_amd
    cache = tmp_path / "cache"
    d = str(cache / "olafuraron_twitter-roberta-base-sentiment-latest-safetensors")
    cfg, t = synth.roberta_classifier(d)
    _tokenizer(d, "roberta")
    clf = kjarni_amd.Classifier("roberta-sentiment", cache_dir=str(cache))
    assert clf.labels() == ["negative", "neutral", "positive"]
    d2 = str(cache / "sentence-transformers_all-mpnet-base-v2")
    synth.mpnet_embedder(d2)
    _tokenizer(d2, "mpnet")
    emb = kjarni_amd.Embedder("mpnet-base-v2", cache_dir=str(cache))
    assert emb.dim == 128
    d3 = str(cache / "nomic-ai_nomic-embed-text-v1.5")
    synth.nomic_embedder(d3)
    synth.add_tokenizer(d3)
    assert kjarni_amd.Embedder("nomic-embed-text", cache_dir=str(cache)).dim == 128
    d4 = str(cache / "BAAI_bge-m3")
    synth.xlmr_embedder(d4)
    _tokenizer(d4, "unigram")
    assert kjarni_amd.Embedder("bge-m3", cache_dir=str(cache)).dim == 128
    with pytest.raises(kjarni_amd.KjarniException) as ei:  # a seq2seq model is not an embedder
        kjarni_amd.Embedder("flan-t5-base", cache_dir=str(cache))
    assert ei.value.code == kjarni_amd.KjarniError.LOAD_FAILED and "not compatible" in str(ei.value)


def test_distilbert_classifier_logits_in_the_trained_regime(tmp_path):
    """BASELINE.json configs[0]'s architecture (distilbert-sentiment: 6 x 768, pre_classifier + ReLU + classifier) on weights
    with the statistics of a trained checkpoint (tests/synth.py: trained_distilbert_tensors -- peaked softmax, LayerNorm gain
    outliers, GELU tails): token-level logits and string-level probabilities against the oracle at 1e-4, logits of several
    units.  (The N(0, 0.02) family is tests/test_gpu_ffi.py::test_classifier_distilbert_softmax_labels_and_multilabel.)"""
    import kjarni_amd
    from tests.parity_report import report
    d = str(tmp_path / "sst2")
    cfg, t = synth.distilbert_sentiment(d, seed=5, family="trained", n_layers=3)   # 3 of 6 layers keep the CPU oracle quick
    synth.add_tokenizer(d)
    orc = O.OracleModel(t, cfg, blocked_gemm=True)
    enc = kjarni_amd.HipEncoder(d)
    for B, S in ((1, 28), (8, 128), (40, 64)):
        ids, mask = synth.synthetic_ids(B, S, seed=B + S, ragged=True)
        want = orc.head_logits(orc.forward(ids, mask, None, O.strategy_mask_value(ids.size)))
        got = enc.logits(ids, mask)
        assert float(np.abs(want).max()) > 1.0
        assert report("families/trained/distilbert_logits", np.abs(got - want).max(), TOL) < TOL, (B, S)
    tok = kjarni_amd.Tokenizer(os.path.join(d, "tokenizer.json"), 512)
    clf = kjarni_amd.Classifier(model_path=d)
    for text in ("this movie was surprisingly good", "a dull, lifeless two hours"):
        ids, mask, _ = tok.encode_batch([text])
        probs = O.softmax_rows(orc.head_logits(orc.forward(ids, mask, None, O.MASK_ALLOC)))[0]
        got = dict(clf.classify(text))
        assert abs(got["NEGATIVE"] - probs[0]) < TOL and abs(got["POSITIVE"] - probs[1]) < TOL
