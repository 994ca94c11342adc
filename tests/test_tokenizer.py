"""Token ids are integer work: bit-exact against the HF `tokenizers` core the
reference links (Cargo.toml:34; configured as in pipeline/encoder/loader.rs:98-115)."""
import json
import os

import numpy as np
import pytest

from kjarni_amd.tokenizer import Tokenizer

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOK_JSON = os.path.join(GOLD, "tokenizer_small.json")


def _cases():
    with open(os.path.join(GOLD, "tokenizer_cases.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("case", _cases(), ids=lambda c: f"{c['kind']}-max{c['max_length']}")
def test_golden_cases(case):
    tok = Tokenizer(TOK_JSON, case["max_length"])
    if case["kind"] == "single":
        ids, mask, types = tok.encode_batch(case["texts"])
    else:
        ids, mask, types = tok.encode_batch([p[0] for p in case["texts"]], [p[1] for p in case["texts"]])
    assert ids.tolist() == case["ids"]
    assert mask.tolist() == case["mask"]
    assert types.tolist() == case["types"]


def _write_tokenizer(path, vocab, lowercase=True, strip_accents=None, chinese=True, clean=True):
    spec = {
        "version": "1.0", "truncation": None, "padding": None,
        "added_tokens": [dict(id=vocab[t], content=t, single_word=False, lstrip=False, rstrip=False,
                              normalized=False, special=True)
                         for t in ("[CLS]", "[SEP]", "[PAD]", "[UNK]") if t in vocab],
        "normalizer": {"type": "BertNormalizer", "clean_text": clean, "handle_chinese_chars": chinese,
                       "strip_accents": strip_accents, "lowercase": lowercase},
        "pre_tokenizer": {"type": "BertPreTokenizer"},
        "post_processor": {"type": "BertProcessing", "sep": ["[SEP]", vocab["[SEP]"]],
                           "cls": ["[CLS]", vocab["[CLS]"]]},
        "decoder": {"type": "WordPiece", "prefix": "##", "cleanup": True},
        "model": {"type": "WordPiece", "unk_token": "[UNK]", "continuing_subword_prefix": "##",
                  "max_input_chars_per_word": 100, "vocab": vocab},
    }
    with open(path, "w") as f:
        json.dump(spec, f)
    return path


REF_VOCAB = {"[CLS]": 0, "[SEP]": 1, "[PAD]": 2, "[UNK]": 3, "hello": 4, "world": 5, "##s": 6, "!": 7}


def test_reference_wordpiece_known_answers(tmp_path):
    # crates/kjarni-transformers/src/tokenizer/wordpiece.rs:137-254 (TEST_JSON vocab)
    tok = Tokenizer(_write_tokenizer(str(tmp_path / "t.json"), REF_VOCAB), 10)
    ids, mask, _ = tok.encode_batch(["hello"])           # test_tokenize_word_known
    assert ids.tolist() == [[0, 4, 1]]
    ids, _, _ = tok.encode_batch(["foobar"])             # test_tokenize_word_unknown -> [UNK]
    assert ids.tolist() == [[0, 3, 1]]
    ids, _, _ = tok.encode_batch(["worlds"])             # test_tokenize_word_with_subtokens
    assert ids.tolist() == [[0, 5, 6, 1]]
    ids, mask, _ = tok.encode_batch(["hello world!"])    # test_encode_basic
    assert ids.tolist() == [[0, 4, 5, 7, 1]] and mask.tolist() == [[1] * 5]
    tok5 = Tokenizer(_write_tokenizer(str(tmp_path / "t.json"), REF_VOCAB), 5)
    ids, mask, _ = tok5.encode_batch(["hello world! hello world!"])  # test_encode_truncation_and_padding
    assert ids.shape == (1, 5) and ids[0, -1] == 1 and mask[0, -1] == 1
    ids, mask, _ = tok5.encode_batch(["hello", "world"])  # test_encode_batch
    assert ids.tolist() == [[0, 4, 1], [0, 5, 1]]


def test_batch_longest_padding_and_empty():
    tok = Tokenizer(TOK_JSON, 512)
    ids, mask, types = tok.encode_batch(["hello world", "hello"])
    assert ids.shape[1] == mask.sum(1).max()
    assert (ids[mask == 0] == 0).all() and (types[mask == 0] == 0).all()
    ids, mask, types = tok.encode_batch([])
    assert ids.shape[0] == 0


def test_invalid_utf8_is_rejected():
    import ctypes as C
    from kjarni_amd import _ffi
    tok = Tokenizer(TOK_JSON, 16)
    arr = (C.c_char_p * 1)(b"\xff\xfe bad")
    out = _ffi.KjarniTokenBatch()
    rc = _ffi.lib().kjarni_tokenizer_encode_batch(tok._h, arr, None, 1, C.byref(out))
    assert rc == _ffi.KjarniError.INVALID_UTF8


def test_unsupported_pipeline_is_a_load_error(tmp_path):
    from kjarni_amd import KjarniException, KjarniError
    p = tmp_path / "bpe.json"
    p.write_text(json.dumps({"model": {"type": "BPE", "vocab": {}, "merges": []}}))
    with pytest.raises(KjarniException) as ei:
        Tokenizer(str(p))
    assert ei.value.code == KjarniError.LOAD_FAILED


# ---- live differential tests against the Rust core (python `tokenizers`) ----
tokenizers = pytest.importorskip("tokenizers")


def _hf(max_len):
    t = tokenizers.Tokenizer.from_file(TOK_JSON)
    t.enable_truncation(max_length=max_len)
    t.enable_padding()
    return t


def test_every_code_point_block():
    """One string per 256-code-point block covering the whole Unicode range: the
    probed property tables (tools/gen_unicode_tables.py) reproduce the crate."""
    ours = Tokenizer(TOK_JSON, 512)
    hf = _hf(512)
    texts = []
    for base in range(0, 0x110000, 64):
        cps = [cp for cp in range(base, base + 64) if not (0xD800 <= cp <= 0xDFFF)]
        if cps:
            texts.append("a" + "b ".join(chr(c) for c in cps if c != 0) + " z")
    for i in range(0, len(texts), 2048):
        chunk = texts[i:i + 2048]
        ids, mask, types = ours.encode_batch(chunk)
        enc = hf.encode_batch(chunk)
        ref = np.array([e.ids for e in enc], np.uint32)
        assert ids.shape == ref.shape
        bad = np.nonzero((ids != ref).any(1))[0]
        assert len(bad) == 0, f"first mismatch in block starting U+{(i + bad[0]) * 64:04X}"


def test_normalizer_flag_combinations(tmp_path):
    vocab = json.load(open(TOK_JSON))["model"]["vocab"]
    texts = ["Héllo Wörld ÀÉÎ", "日本語 and ＡＢＣ", "İ ß Σ ά", "tab\there\x01x​zero"]
    for lowercase in (True, False):
        for strip in (None, True, False):
            for chinese in (True, False):
                for clean in (True, False):
                    p = _write_tokenizer(str(tmp_path / "f.json"), vocab, lowercase, strip, chinese, clean)
                    hf = tokenizers.Tokenizer.from_file(p)
                    hf.enable_padding()
                    ours = Tokenizer(p, 512)
                    ref = np.array([e.ids for e in hf.encode_batch(texts)], np.uint32)
                    got, _, _ = ours.encode_batch(texts)
                    assert got.tolist() == ref.tolist(), (lowercase, strip, chinese, clean)


def test_hypothesis_random_text():
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    ours = Tokenizer(TOK_JSON, 24)
    hf = _hf(24)
    alphabet = st.characters(blacklist_categories=("Cs",))

    @settings(max_examples=400, deadline=None)
    @given(st.lists(st.text(alphabet=alphabet, max_size=60), min_size=1, max_size=6),
           st.lists(st.text(alphabet=alphabet, max_size=60), min_size=6, max_size=6))
    def run(a, b):
        a = [x.replace("\x00", "") for x in a]
        b = [x.replace("\x00", "") for x in b][:len(a)]
        got = ours.encode_batch(a)
        enc = hf.encode_batch(a)
        assert got[0].tolist() == [e.ids for e in enc]
        got = ours.encode_batch(a, b)
        enc = hf.encode_batch(list(zip(a, b)))
        assert got[0].tolist() == [e.ids for e in enc]
        assert got[1].tolist() == [e.attention_mask for e in enc]
        assert got[2].tolist() == [e.type_ids for e in enc]

    run()


def test_large_batch_is_split_over_threads_and_identical():
    """Batches of >= 512 rows are tokenised on several host threads (the `tokenizers` crate uses its
    rayon pool); the result must be what row-at-a-time encoding gives, pairs included."""
    from tokenizers import Tokenizer as HFTokenizer
    rng = np.random.default_rng(0)
    words = ["hello", "world", "unaffable", "Reykjavík", "naïve", "東京", "x", "1234", "!!", "the", "quick"]
    texts = [" ".join(rng.choice(words, int(rng.integers(0, 30)))) for _ in range(3000)]
    seconds = [" ".join(rng.choice(words, int(rng.integers(1, 20)))) for _ in range(3000)]
    tok = Tokenizer(TOK_JSON, 32)
    hf = HFTokenizer.from_file(TOK_JSON)
    hf.enable_truncation(max_length=32)
    hf.enable_padding(pad_id=0, pad_token="[PAD]")
    ids, mask, types = tok.encode_batch(texts)
    enc = hf.encode_batch(texts)
    assert ids.tolist() == [e.ids for e in enc] and mask.tolist() == [e.attention_mask for e in enc]
    ids, mask, types = tok.encode_batch(texts, seconds)
    enc = hf.encode_batch(list(zip(texts, seconds)))
    assert ids.tolist() == [e.ids for e in enc] and types.tolist() == [e.type_ids for e in enc]
    assert mask.tolist() == [e.attention_mask for e in enc]
