"""Builds tests/golden/unigram_tokenizer.json + unigram_goldens.json: an XLM-R style tokenizer (bge-m3's layout: SentencePiece
Unigram model, Precompiled nmt_nfkc normaliser + ' {2,}' -> ' ', WhitespaceSplit + Metaspace, '<s> $A </s>' framing) and the ids
the `tokenizers` package gives for a set of texts.  The SentencePiece model is trained here on a generated corpus; the
nmt_nfkc character map comes out of the trained model proto (it is compiled into the sentencepiece library).

Run from the repo root:  python tests/golden/make_unigram_golden.py"""
import json
import os
import random
import tempfile

import sentencepiece as spm
from sentencepiece import sentencepiece_model_pb2 as pb
from tokenizers import AddedToken, Regex, Tokenizer, decoders, normalizers, pre_tokenizers, processors
from tokenizers.models import Unigram

HERE = os.path.dirname(os.path.abspath(__file__))
WORDS = ["hello", "world", "the", "quick", "brown", "fox", "jumps", "over", "lazy", "dog", "Ísland", "fjörður", "þetta", "er", "próf",
         "日本語", "テキスト", "東京", "한국어", "텍스트", "привет", "мир", "Ελληνικά", "العربية", "café", "naïve", "ﬁnance", "№", "①",
         "ｆｕｌｌ", "½", "x²", "Ⅷ", "é", "å", "🙂", "👍🏽", "€100", "3.14", "don't", "co-op", "e-mail", "user@example.com",
         "http://x.y/z", "tokenization", "embedding", "retrieval", "multilingual", "हिन्दी", "ไทย", "עברית"]
TEXTS = [
    "hello world", "Hello, World!", "  leading and   multiple   spaces  ", "the quick brown fox jumps over the lazy dog",
    "Ísland er fallegt land með fjörðum", "日本語のテキストを東京で", "한국어 텍스트", "привет мир", "café naïve ﬁnance № ① ｆｕｌｌ ½ x² Ⅷ",
    "é å combining marks", "🙂 👍🏽 emoji 👨‍👩‍👧 🇮🇸", "tabs\tand\nnewlines\r\nhere", "<s> literal specials </s> <mask> <pad> <unk>",
    "unknown chars: ☃ ♞ 𝔘𝔫𝔦", "", " ", "a", "ＡＢＣ１２３", "ｶﾞ half-width kana ﾊﾟ", "zero​width nbsp　ideographic", "soft­hyphen",
    "x" * 300, "The retrieval of multilingual embedding tokenization " * 20, "가각힣ᄀ", "กำ thai sara am", "İstanbul ǅ ß ẞ",
    "control\x01chars\x7f", "العربية עברית हिन्दी ไทย", "▁already▁meta▁space", "a▁b ▁ c",
]


def build(tmp):
    random.seed(0)
    corpus = os.path.join(tmp, "corpus.txt")
    with open(corpus, "w") as f:
        for _ in range(6000):
            f.write(" ".join(random.choice(WORDS) for _ in range(random.randint(3, 12))) + "\n")
    prefix = os.path.join(tmp, "spm")
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=prefix, vocab_size=600, model_type="unigram",
                                   normalization_rule_name="nmt_nfkc", character_coverage=0.9995, bos_id=0, pad_id=1, eos_id=2,
                                   unk_id=3, bos_piece="<s>", pad_piece="<pad>", eos_piece="</s>", unk_piece="<unk>",
                                   hard_vocab_limit=False, minloglevel=2)
    m = pb.ModelProto()
    m.ParseFromString(open(prefix + ".model", "rb").read())
    # XLMRobertaConverter.vocab: the four specials, the model's pieces from index 3 on, then <mask>
    vocab = [("<s>", 0.0), ("<pad>", 0.0), ("</s>", 0.0), ("<unk>", 0.0)] + [(p.piece, p.score) for p in m.pieces[4:]] + [("<mask>", 0.0)]
    tok = Tokenizer(Unigram(vocab=vocab, unk_id=3, byte_fallback=False))
    tok.normalizer = normalizers.Sequence([normalizers.Precompiled(m.normalizer_spec.precompiled_charsmap),
                                           normalizers.Replace(Regex(" {2,}"), " ")])
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.WhitespaceSplit(),
                                                 pre_tokenizers.Metaspace(replacement="▁", prepend_scheme="always")])
    tok.decoder = decoders.Metaspace(replacement="▁", prepend_scheme="always")
    tok.post_processor = processors.TemplateProcessing(single="<s> $A </s>", pair="<s> $A </s> </s> $B </s>",
                                                       special_tokens=[("<s>", 0), ("</s>", 2)])
    tok.add_special_tokens([AddedToken("<s>", special=True), AddedToken("<pad>", special=True), AddedToken("</s>", special=True),
                            AddedToken("<unk>", special=True), AddedToken("<mask>", lstrip=True, special=True)])
    return tok


def main():
    with tempfile.TemporaryDirectory() as tmp:
        tok = build(tmp)
    path = os.path.join(HERE, "unigram_tokenizer.json")
    tok.save(path, pretty=False)
    cases = []
    for t in TEXTS:
        e = tok.encode(t)
        cases.append(dict(text=t, ids=e.ids, normalized=tok.normalizer.normalize_str(t),
                          pieces=[p for p, _ in tok.pre_tokenizer.pre_tokenize_str(tok.normalizer.normalize_str(t))]))
    pair = tok.encode("hello world", "the quick brown fox")
    tok.enable_truncation(max_length=16)
    trunc = tok.encode(TEXTS[22])
    with open(os.path.join(HERE, "unigram_goldens.json"), "w") as f:
        json.dump(dict(cases=cases, pair=dict(a="hello world", b="the quick brown fox", ids=pair.ids, type_ids=pair.type_ids),
                       truncated=dict(text=TEXTS[22], max_length=16, ids=trunc.ids)), f, ensure_ascii=False, indent=0)
    print("vocab", tok.get_vocab_size(), "cases", len(cases), "bytes", os.path.getsize(path))


if __name__ == "__main__":
    main()
