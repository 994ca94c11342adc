#!/usr/bin/env python3
"""Generates tests/golden/tokenizer_small.json (a tokenizer.json) and
tests/golden/tokenizer_cases.json (inputs + expected ids/mask/type ids).

Expected outputs come from Python `tokenizers` (the Rust core the reference links,
Cargo.toml:34) configured exactly as the reference configures it at load time
(crates/kjarni-transformers/src/pipeline/encoder/loader.rs:98-115): truncation
max_length (default strategy LongestFirst, right) and BatchLongest padding.

Run from the repo root:  python tests/golden/make_tokenizer_golden.py
"""
import json
import os

from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors, trainers

HERE = os.path.dirname(os.path.abspath(__file__))

CORPUS = [
    "Hello world! This is a test of the WordPiece tokenizer, isn't it?",
    "Reykjavík is the capital of Iceland; café naïve façade coöperate.",
    "日本語のテキスト と 한국어 텍스트 中文字符",
    "unaffable unbelievably running runner runs walked walking talks",
    "The quick brown fox jumps over the lazy dog 1234567890 times.",
    "İstanbul ΣΊΣΥΦΟΣ straße ǅ Ångström Ñandú",
    "e-mail: someone@example.com, http://x.y/z?q=1&r=2 #hashtag $100 50%",
    "Machine learning models embed sentences into vectors for semantic search.",
    "The reranker scores query document pairs with a cross encoder.",
] * 4

TEXTS = [
    "Hello world!",
    "hello",
    "Reykjavík café naïve",
    "日本語 test 한국어 中文",
    "x" * 150,                      # > max_input_chars_per_word -> [UNK]
    "[CLS] literal [SEP] tokens [MASK] [PAD] [UNK]",
    "",
    "  \t\n \r ",
    "İstanbul ΣΊΣΥΦΟΣ ǅ ß",
    "a very long sentence " * 20,   # truncation
    "tabs\tand\nnewlines\r\nand nbsp line sep",
    "control chars\x01\x7f​‍zero width�﻿",
    "é ạ̈ ọ̈ ṩ ̈́ combining",
    "punctuation!?;:,.()[]{}<>«»“”‘’—–…·¿¡§¶",
    "emoji 😀 and 👍🏽 flags 🇮🇸 math ∑∫√ ≠ ≤",
    "mixedCASE WordPiece UNAFFABLE Unbelievably",
    "Ⅻ ﬁ ligature ｆｕｌｌ ｗｉｄｔｈ ＡＢＣ",
    "\U00020000\U0002a6df CJK ext B, 豈 compat, 㐀 ext A",
    "privateuse unassigned͸ here",
    "한국어 텍스트 분해 각 닭 읽다",
    "e-mail: someone@example.com, http://x.y/z?q=1",
]

PAIRS = [
    ("what is the capital of iceland", "Reykjavík is the capital of Iceland."),
    ("short", "a very long document " * 30),
    ("a very long query " * 30, "short"),
    ("a long query " * 20, "a long document " * 20),
    ("", ""),
    ("[SEP]", "日本語 한국어"),
    ("q", "d"),
]


def main():
    tok = Tokenizer(models.WordPiece(unk_token="[UNK]"))
    tok.normalizer = normalizers.BertNormalizer()
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    trainer = trainers.WordPieceTrainer(vocab_size=600,
                                        special_tokens=["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"])
    tok.train_from_iterator(CORPUS, trainer)
    cls, sep = tok.token_to_id("[CLS]"), tok.token_to_id("[SEP]")
    tok.post_processor = processors.TemplateProcessing(
        single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1",
        special_tokens=[("[CLS]", cls), ("[SEP]", sep)])
    tok.save(os.path.join(HERE, "tokenizer_small.json"))

    cases = []
    for max_len in (512, 32, 16, 7, 3):
        t = Tokenizer.from_file(os.path.join(HERE, "tokenizer_small.json"))
        t.enable_truncation(max_length=max_len)
        t.enable_padding()  # BatchLongest, pad id 0
        enc = t.encode_batch(TEXTS)
        cases.append(dict(max_length=max_len, kind="single", texts=TEXTS,
                          ids=[e.ids for e in enc], mask=[e.attention_mask for e in enc],
                          types=[e.type_ids for e in enc]))
        enc = t.encode_batch(PAIRS)
        cases.append(dict(max_length=max_len, kind="pair", texts=[list(p) for p in PAIRS],
                          ids=[e.ids for e in enc], mask=[e.attention_mask for e in enc],
                          types=[e.type_ids for e in enc]))
    with open(os.path.join(HERE, "tokenizer_cases.json"), "w") as f:
        json.dump(cases, f, ensure_ascii=True)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
