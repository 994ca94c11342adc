"""Byte-level BPE tokenizer (kjarni_amd/csrc/bpe.cpp) against the `tokenizers` package: committed goldens
(tests/golden/make_bpe_golden.py) and, when the package is importable, a live fuzz over mixed-script strings.
Host-only: no GPU needed."""
import json
import os
import random

import pytest

from kjarni_amd.chat import BpeTokenizer

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ["llama3", "qwen2", "gpt2"]


def _path(name):
    return os.path.join(GOLDEN, f"bpe_{name}_tokenizer.json")


@pytest.fixture(scope="module")
def goldens():
    with open(os.path.join(GOLDEN, "bpe_goldens.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", NAMES)
def test_encode_matches_tokenizers_goldens(goldens, name):
    tok = BpeTokenizer(_path(name))
    for case in goldens[name]["cases"]:
        assert tok.encode(case["text"]) == case["ids"], repr(case["text"])


@pytest.mark.parametrize("name", ["llama3", "qwen2"])
def test_split_regex_pieces(goldens, name):
    tok = BpeTokenizer(_path(name))
    for case in goldens[name]["cases"]:
        assert tok.pre_tokenize(case["text"]) == case["pieces"], repr(case["text"])


@pytest.mark.parametrize("name", NAMES)
def test_decode_whole_and_single_token(goldens, name):
    tok = BpeTokenizer(_path(name))
    for case in goldens[name]["cases"]:
        if "\x00" in case["decoded"]:
            continue  # NUL cannot cross a C string
        assert tok.decode(case["ids"]) == case["decoded"]
        assert tok.decode(case["ids"], skip_special=True) == case["decoded_skip"]
        # the generation loop decodes one token at a time: partial UTF-8 sequences become U+FFFD
        for i, text in zip(case["ids"][:24], case["single"]):
            assert tok.decode([i]) == text


@pytest.mark.parametrize("name", NAMES)
def test_truncation_keeps_the_head(goldens, name):
    tok = BpeTokenizer(_path(name))
    t = goldens[name]["truncated"]
    assert tok.encode(t["text"], t["max_length"]) == t["ids"]
    assert len(tok.encode(t["text"])) > t["max_length"]


def test_special_tokens_are_single_ids(goldens):
    tok = BpeTokenizer(_path("llama3"))
    ids = tok.encode("<|begin_of_text|>hi<|eot_id|>")
    assert ids[0] == 700 and ids[-1] == 704
    assert tok.decode(ids, skip_special=True) == "hi"
    # an unknown "<|...|>" is ordinary text
    assert len(tok.encode("<|nope|>")) > 1


def test_unsupported_pipelines_fail_loudly(tmp_path):
    with open(_path("llama3")) as f:
        j = json.load(f)
    j["pre_tokenizer"]["pretokenizers"][0]["pattern"]["Regex"] = r"\w+|\s+"
    p = tmp_path / "tokenizer.json"
    p.write_text(json.dumps(j))
    with pytest.raises(Exception, match="unsupported pre-tokenizer regex"):
        BpeTokenizer(str(p))
    j = json.load(open(_path("qwen2")))
    j["normalizer"] = {"type": "Lowercase"}
    p.write_text(json.dumps(j))
    with pytest.raises(Exception, match="unsupported normalizer"):
        BpeTokenizer(str(p))
    j = json.load(open(_path("gpt2")))
    j["model"]["byte_fallback"] = True
    p.write_text(json.dumps(j))
    with pytest.raises(Exception, match="byte_fallback"):
        BpeTokenizer(str(p))


POOLS = [
    "abcdefghij KLMNOP 0123456789 \n\t\r'.,!?-_()[]{}<>|/\\\"@#$%^&*+=~`",
    "éèêëāăąçčďđēėęěğßÞþðæøåÅ",
    "日本語漢字ひらがなカタカナ中文한국어",
    "абвгдеёжз АБВ",
    "αβγδσςω ΑΒΣ",
    "\U0001F600\U0001F389\U0001F44D\U0001F3FD\U0001F468‍\U0001F469‍\U0001F467\U0001F1EE\U0001F1F8✨",
    "ཱིུ̧̨̣̀́̂̃̈̊̈́ͅ",
    "            　",
    "٠١٢٣ ²³¹½¼ ⅠⅡ ①② ०१",
    "ﬁﬂﬀ Ω K Å ſ İ ı ǅ ǆ ᾳ ῼ",
    "العربية עברית हिन्दी ไทย",
    "각가각힣ㄱ",
]
SPECIALS = {"llama3": ["<|begin_of_text|>", "<|eot_id|>", "<|start_header_id|>", "<|end_header_id|>"],
            "qwen2": ["<|im_start|>", "<|im_end|>", "<|endoftext|>", "<tool_call>", "</tool_call>"], "gpt2": ["<|endoftext|>"]}


@pytest.mark.parametrize("name", NAMES)
def test_live_fuzz_against_tokenizers(name):
    tokenizers = pytest.importorskip("tokenizers")
    ref = tokenizers.Tokenizer.from_file(_path(name))
    mine = BpeTokenizer(_path(name))
    rng = random.Random(20240 + len(name))
    for _ in range(2500):
        s = ""
        for _ in range(rng.randint(1, 4)):
            pool = rng.choice(POOLS)
            s += "".join(rng.choice(pool) for _ in range(rng.randint(1, 12)))
            if rng.random() < 0.15:
                s += rng.choice(SPECIALS[name])
            if rng.random() < 0.1:
                s += "'" + rng.choice(["s", "S", "t", "re", "VE", "m", "ll", "Ll", "d", "ſ", "x"])
        assert mine.encode(s) == ref.encode(s, add_special_tokens=False).ids, repr(s)


# ---- SentencePiece-style BPE (Llama 2 / Mistral tokenizer.json): tests/golden/make_spbpe_golden.py ---------------

@pytest.fixture(scope="module")
def sp_goldens():
    with open(os.path.join(GOLDEN, "spbpe_goldens.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", ["legacy", "metaspace"])
def test_sentencepiece_style_bpe_matches_tokenizers(sp_goldens, name):
    tok = BpeTokenizer(os.path.join(GOLDEN, f"spbpe_{name}_tokenizer.json"))
    for case in sp_goldens[name]["cases"]:
        if "\x00" in case["text"]:
            continue  # NUL cannot cross a C string
        assert tok.encode(case["text"]) == case["ids"], repr(case["text"])
        assert tok.decode(case["ids"]) == case["decoded"], repr(case["text"])
        assert tok.decode(case["ids"], skip_special=True) == case["decoded_skip"]
        for i, text in zip(case["ids"][:32], case["single"]):
            assert tok.decode([i]) == text  # single-token decode: the leading space is stripped, lone bytes become U+FFFD
    t = sp_goldens[name]["truncated"]
    assert tok.encode(t["text"], t["max_length"]) == t["ids"]


@pytest.mark.parametrize("name", ["legacy", "metaspace"])
def test_sentencepiece_style_bpe_live_fuzz(name):
    tokenizers = pytest.importorskip("tokenizers")
    path = os.path.join(GOLDEN, f"spbpe_{name}_tokenizer.json")
    ref = tokenizers.Tokenizer.from_file(path)
    mine = BpeTokenizer(path)
    rng = random.Random(99)
    specials = ["[INST]", "[/INST]", "<s>", "</s>", "<unk>", "[TOOL_CALLS]"]
    for _ in range(1500):
        s = ""
        for _ in range(rng.randint(1, 4)):
            pool = rng.choice(POOLS[:5] + ["   ", "▁▁ ▁"])
            s += "".join(rng.choice(pool) for _ in range(rng.randint(1, 14)))
            if rng.random() < 0.2:
                s += rng.choice(specials)
        assert mine.encode(s) == ref.encode(s, add_special_tokens=False).ids, repr(s)
    # a whole prompt is one BPE word here: the merge loop must not be quadratic
    long_text = "The quick brown fox jumps over the lazy dog. " * 400
    assert mine.encode(long_text) == ref.encode(long_text, add_special_tokens=False).ids
