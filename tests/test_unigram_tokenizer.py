"""SentencePiece-Unigram tokenizer.json pipeline (kjarni_amd/csrc/unigram.cpp: Precompiled nmt_nfkc normaliser, WhitespaceSplit +
Metaspace, Unigram Viterbi, '<s> $A </s>' framing — bge-m3 / XLM-R's layout) against the `tokenizers` package: committed
goldens (tests/golden/make_unigram_golden.py) and, when the package is importable, a live fuzz.  Host-only."""
import json
import os
import random

import numpy as np
import pytest

import kjarni_amd

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
PATH = os.path.join(GOLDEN, "unigram_tokenizer.json")


@pytest.fixture(scope="module")
def goldens():
    with open(os.path.join(GOLDEN, "unigram_goldens.json")) as f:
        return json.load(f)


def _ids(tok, text):
    ids, mask, _ = tok.encode_batch([text])
    return ids[0, :int(mask[0].sum())].tolist()


def test_encode_matches_tokenizers_goldens(goldens):
    tok = kjarni_amd.Tokenizer(PATH, 8194)
    for case in goldens["cases"]:
        if "\x00" in case["text"]:
            continue
        assert _ids(tok, case["text"]) == case["ids"], repr(case["text"])


def test_pair_and_truncation(goldens):
    tok = kjarni_amd.Tokenizer(PATH, 8194)
    p = goldens["pair"]
    ids, mask, types = tok.encode_batch([p["a"]], [p["b"]])
    n = int(mask[0].sum())
    assert ids[0, :n].tolist() == p["ids"] and types[0, :n].tolist() == p["type_ids"]
    t = goldens["truncated"]
    short = kjarni_amd.Tokenizer(PATH, t["max_length"])
    assert _ids(short, t["text"]) == t["ids"]


def test_batch_padding_uses_id_zero():
    tok = kjarni_amd.Tokenizer(PATH, 64)
    ids, mask, _ = tok.encode_batch(["hello world", "the quick brown fox jumps over the lazy dog"])
    assert ids.shape == mask.shape and mask[0].sum() < mask[1].sum()
    assert (ids[0, int(mask[0].sum()):] == 0).all()  # PaddingParams::default(): pad id 0, not <pad> = 1 (loader.rs:112-115)


def test_unsupported_pieces_fail_loudly(tmp_path):
    j = json.load(open(PATH))
    j["normalizer"]["normalizers"].append({"type": "Lowercase"})
    p = tmp_path / "tokenizer.json"
    p.write_text(json.dumps(j))
    with pytest.raises(Exception, match="unsupported normalizer 'Lowercase'"):
        kjarni_amd.Tokenizer(str(p), 16)
    j = json.load(open(PATH))
    j["pre_tokenizer"] = {"type": "ByteLevel"}
    p.write_text(json.dumps(j))
    with pytest.raises(Exception, match="unsupported pre-tokenizer 'ByteLevel'"):
        kjarni_amd.Tokenizer(str(p), 16)


POOLS = [
    "abcdefghij KLMNOP 0123456789 \n\t\r'.,!?-_()[]{}<>|/\\\"@#$%^&*+=~`",
    "éèêëāăąçčďđēėęěğßÞþðæøåÅ",
    "日本語漢字ひらがなカタカナ中文한국어",
    "абвгдеёжз АБВ",
    "αβγδσςω ΑΒΣ",
    "\U0001F600\U0001F389\U0001F44D\U0001F3FD\U0001F468‍\U0001F469‍\U0001F467\U0001F1EE\U0001F1F8✨©®™‼️",
    "ཱིུ̧̨̣̀́̂̃̈̊̈́ͅ",
    "            　​‌‍⁠﻿­",
    "٠١٢٣ ²³¹½¼ ⅠⅡ ①② ०१",
    "ﬁﬂﬀ Ω K Å ſ İ ı ǅ ǆ ᾳ ῼ ｶﾞﾊﾟ ＡＢｃ １２ ㌔ ㍿ ℃ №",
    "العربية ؀؁۝ עברית हिन्दी क्ष ไทย กำ நி কো",
    "각가각힣ㄱ 각",
    "▁▁ ▁",
    "\x01\x02\x7f\x85  ",
]
SPECIALS = ["<s>", "</s>", "<pad>", "<unk>", "<mask>"]


def _fuzz(path, rounds, seed):
    tokenizers = pytest.importorskip("tokenizers")
    ref = tokenizers.Tokenizer.from_file(path)
    mine = kjarni_amd.Tokenizer(path, 1 << 20)
    rng = random.Random(seed)
    for _ in range(rounds):
        s = ""
        for _ in range(rng.randint(1, 5)):
            pool = rng.choice(POOLS)
            s += "".join(rng.choice(pool) for _ in range(rng.randint(1, 12)))
            if rng.random() < 0.15:
                s += rng.choice(SPECIALS)
            if rng.random() < 0.3:
                s += " " * rng.randint(1, 3)
        if "\x00" in s:
            continue
        assert _ids(mine, s) == ref.encode(s).ids, repr(s)


def test_normaliser_is_exact_under_a_character_complete_vocabulary(tmp_path):
    """With the trained vocabulary most exotic characters end as <unk> whatever the normaliser did.  Here every character
    the reference's normaliser can emit for the fuzz corpus is its own piece, so any difference in the Precompiled map walk
    (grapheme clusters, shortest-prefix lookup) changes the ids."""
    tokenizers = pytest.importorskip("tokenizers")
    base = tokenizers.Tokenizer.from_file(PATH)
    rng = random.Random(5)
    blocks = [(0x20, 0x7f), (0xa0, 0x24f), (0x300, 0x36f), (0x590, 0x6ff), (0x900, 0x97f), (0xe00, 0xe7f), (0x1100, 0x11ff),
              (0x2000, 0x206f), (0x20d0, 0x20ff), (0x2100, 0x214f), (0x2460, 0x24ff), (0x3000, 0x30ff), (0x3300, 0x33ff),
              (0xac00, 0xac80), (0xfb00, 0xfb4f), (0xfe00, 0xfe0f), (0xff00, 0xffef), (0x1f1e6, 0x1f1ff), (0x1f300, 0x1f64f),
              (0x1f3fb, 0x1f3ff), (0xe0020, 0xe007f), (0x1d400, 0x1d7ff), (0x1, 0x1f), (0x7f, 0x9f)]
    strs = ["".join(chr(rng.randint(*rng.choice(blocks))) for _ in range(rng.randint(1, 10)))
            for _ in range(6000 * int(os.environ.get("KJARNI_FUZZ_SCALE", "1")))]
    chars = set()
    for s in strs:
        chars.update(base.normalizer.normalize_str(s))
    j = json.load(open(PATH))
    have = {p for p, _ in j["model"]["vocab"]}
    j["model"]["vocab"] += [[c, -10.0 - i * 1e-4] for i, c in enumerate(sorted(chars - have - {" "}))]
    p = str(tmp_path / "tokenizer.json")
    with open(p, "w") as f:
        json.dump(j, f)
    ref = tokenizers.Tokenizer.from_file(p)
    mine = kjarni_amd.Tokenizer(p, 1 << 20)
    for s in strs:
        assert _ids(mine, s) == ref.encode(s).ids, ([hex(ord(c)) for c in s], ref.normalizer.normalize_str(s))


def test_live_fuzz_against_tokenizers():
    _fuzz(PATH, 3000 * int(os.environ.get("KJARNI_FUZZ_SCALE", "1")), 7)


@pytest.mark.parametrize("variant", ["legacy_add_prefix_space", "strip_and_meta_replace", "no_whitespace_split", "byte_fallback"])
def test_other_converter_layouts(tmp_path, variant):
    """The same model under the other layouts SentencePiece converters have emitted."""
    j = json.load(open(PATH))
    if variant == "legacy_add_prefix_space":   # pre-0.19 Metaspace fields, as in checkpoints uploaded before 2024
        j["pre_tokenizer"]["pretokenizers"][1] = {"type": "Metaspace", "replacement": "▁", "add_prefix_space": True}
    elif variant == "strip_and_meta_replace":  # SpmConverter since transformers 4.4x
        j["normalizer"]["normalizers"] = [j["normalizer"]["normalizers"][0], {"type": "Strip", "strip_left": False, "strip_right": True},
                                          {"type": "Replace", "pattern": {"Regex": " {2,}"}, "content": "▁"}]
    elif variant == "no_whitespace_split":     # T5 / mBART style: Metaspace alone
        j["pre_tokenizer"] = {"type": "Metaspace", "replacement": "▁", "prepend_scheme": "always", "split": True}
    elif variant == "byte_fallback":
        j["model"]["byte_fallback"] = True
        j["model"]["vocab"] += [[f"<0x{b:02X}>", -20.0] for b in range(256)]
    p = str(tmp_path / "tokenizer.json")
    with open(p, "w") as f:
        json.dump(j, f)
    _fuzz(p, 600, 11)
