"""Generates the byte-level BPE fixtures: three small tokenizer.json files in the shapes the reference's decoder
models ship (Llama 3, Qwen 2, GPT-2) and bpe_goldens.json with what the `tokenizers` package -- the Python build of
the crate the reference links (tokenizers 0.22.x, Cargo.toml:34) -- produces for them.

    python tests/golden/make_bpe_golden.py

The tokenizers are trained here on a fixed corpus (no network): the vocabulary is small, the pipeline (added tokens,
normalizer, Split regex, ByteLevel, BPE with ignore_merges) is exactly the production one.
"""
import json
import os
import random

from tokenizers import AddedToken, Regex, Tokenizer, decoders, models, normalizers, pre_tokenizers, trainers

HERE = os.path.dirname(os.path.abspath(__file__))

LLAMA3 = r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+"
QWEN2 = r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+"

CORPUS = [
    "The quick brown fox jumps over the lazy dog. The dog doesn't mind; it's used to it.",
    "Hello, world! How are you today? I'm fine, thanks -- and you? We'll see what they've done.",
    "In 2024 the population was 8,123,456,789 people; by 2050 it may reach 9.7 billion.",
    "def tokenize(text):\n    return [t for t in text.split() if t]\n\n\nprint(tokenize('a b  c'))\n",
    "Kjarni er íslenskt orð sem þýðir kjarni eða miðja. Ég á heima á Íslandi og það er kalt.",
    "Größe, Straße, Äpfel und Öl: das sind deutsche Wörter mit Umlauten. Übung macht den Meister!",
    "日本語のテキストも含まれています。東京は日本の首都です。漢字とひらがなとカタカナ。",
    "Русский текст тоже есть: Москва — столица России. Привет, как дела?",
    "Emoji are fun 😀🎉👍 and so are symbols: ∑ ∫ √ ≈ ≠ ± × ÷ © ® ™ € £ ¥.",
    "Whitespace   matters \t tabs\tand\r\nwindows line endings\r\n\r\nand trailing spaces   \n",
    "URLs like https://example.com/path?query=1&other=2#frag and emails like someone@example.org appear.",
    "She said: \"It's 'quoted'\", then left. They'd've gone too, wouldn't they? I'LL SHOUT IT'S FINE.",
    "x = 3.14159; y = 2.71828; z = x * y + 42 - 7 / 3 % 2; arr[0] = {key: 'value'};",
    "Ελληνικά γράμματα: α β γ δ ε. Το Σίσυφος είναι όνομα. العربية أيضا هنا. עברית גם כן.",
    "é́ ñ ö ü å ø æ ç ß ÿ — combining: é ä ô ñ and precomposed: é ä ô ñ.",
    "한국어 텍스트: 서울은 대한민국의 수도입니다. 안녕하세요!",
]


def train(pre, special, normalizer=None, ignore_merges=False, vocab_size=700):
    tok = Tokenizer(models.BPE(ignore_merges=ignore_merges))
    if normalizer is not None:
        tok.normalizer = normalizer
    tok.pre_tokenizer = pre
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=vocab_size, special_tokens=[], initial_alphabet=pre_tokenizers.ByteLevel.alphabet(),
                                  show_progress=False)
    tok.train_from_iterator(CORPUS * 3, trainer)
    tok.add_special_tokens([AddedToken(t, special=True, normalized=False) for t in special])
    return tok


def texts():
    rng = random.Random(7)
    base = list(CORPUS) + [
        "", " ", "  ", "\n", " \n ", "a", "'", "''s", "I'M he'Ll it'ſ x'S 'K 'ſa",
        "12345 678 9 1000000 ٣٤٥٦ 1²½3",
        "a  b   \n\n  c \t\n d", "  hello!!!\r\n\r\nx", "x   y 　 z  w",
        "<|begin_of_text|><|start_header_id|>system<|end_header_id|>\n\nYou are helpful.<|eot_id|><|start_header_id|>user<|end_header_id|>\n\nHi!<|eot_id|><|start_header_id|>assistant<|end_header_id|>\n\n",
        "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n<|im_start|>user\nHello there<|im_end|>\n<|im_start|>assistant\n",
        "text with <|eot_id|> in the middle and <|unknown_special|> too <|endoftext|>",
        "<|begin_of_text|<|begin_of_text|>> nested-ish <|eot_id|><|eot_id|>",
        "Café vs Café; Å vs Å; 각 vs 각; ậ order; ̈́ क़ Ω ﬁ",
        "ＦＵＬＬＷＩＤＴＨ １２３ ｈｅｌｌｏ", "\U0001F468‍\U0001F469‍\U0001F467 family \U0001F1EE\U0001F1F8 flag",
        "tabs\t\tand\t spaces  \t mixed \r\n\r\n\n end",
        "don't can't won't I'd you're we've THEY'RE it'S", "'re're'll'LL'Ve",
        "0", "00", "000", "0000", "00000 0 00", "3.14 1,000 1e10 0x1F",
        "!!!", " !!!", "  !!!", "!!!\n", "!!!\n\n\nx", "...---...", "a.b,c;d:e",
    ]
    alphabet = "abc ABC 123 \n\t'.,!?-éßñ日本語ру😀́　"
    for _ in range(60):
        n = rng.randint(1, 40)
        base.append("".join(rng.choice(alphabet) for _ in range(n)))
    return base


def main():
    llama_special = ["<|begin_of_text|>", "<|end_of_text|>", "<|start_header_id|>", "<|end_header_id|>", "<|eot_id|>", "<|eom_id|>",
                     "<|python_tag|>", "<|finetune_right_pad_id|>"]
    qwen_special = ["<|endoftext|>", "<|im_start|>", "<|im_end|>", "<|object_ref_start|>", "<|vision_start|>"]
    split = lambda pattern: pre_tokenizers.Sequence([  # noqa: E731
        pre_tokenizers.Split(Regex(pattern), behavior="isolated", invert=False),
        pre_tokenizers.ByteLevel(add_prefix_space=False, trim_offsets=True, use_regex=False)])
    toks = {
        "llama3": train(split(LLAMA3), llama_special, ignore_merges=True),
        "qwen2": train(split(QWEN2), qwen_special, normalizer=normalizers.NFC()),
        "gpt2": train(pre_tokenizers.ByteLevel(add_prefix_space=False, trim_offsets=True, use_regex=True), ["<|endoftext|>"]),
    }
    # Qwen ships non-special added tokens too (<tool_call> ...): normalized=False, special=False.
    toks["qwen2"].add_tokens([AddedToken("<tool_call>", special=False, normalized=False), AddedToken("</tool_call>", special=False, normalized=False)])
    out = {}
    for name, tok in toks.items():
        path = os.path.join(HERE, f"bpe_{name}_tokenizer.json")
        tok.save(path, pretty=False)
        cases = []
        for t in texts():
            enc = tok.encode(t, add_special_tokens=False)
            pieces = [p for p, _ in tok.pre_tokenizer.pre_tokenize_str(tok.normalizer.normalize_str(t) if tok.normalizer else t)]
            cases.append({"text": t, "ids": enc.ids, "decoded": tok.decode(enc.ids, skip_special_tokens=False),
                          "decoded_skip": tok.decode(enc.ids, skip_special_tokens=True),
                          "single": [tok.decode([i], skip_special_tokens=False) for i in enc.ids[:24]]})
            if name != "gpt2":
                # ByteLevel pieces are in the mapped alphabet; keep the Split stage alone for the pre-tokenizer check
                sp = pre_tokenizers.Split(Regex(LLAMA3 if name == "llama3" else QWEN2), behavior="isolated", invert=False)
                cases[-1]["pieces"] = [p for p, _ in sp.pre_tokenize_str(tok.normalizer.normalize_str(t) if tok.normalizer else t)]
            del pieces
        long_text = " ".join(CORPUS)
        out[name] = {"cases": cases, "truncated": {"text": long_text, "max_length": 50,
                                                     "ids": tok.encode(long_text, add_special_tokens=False).ids[:50]},
                     "vocab_size": tok.get_vocab_size(with_added_tokens=True)}
    with open(os.path.join(HERE, "bpe_goldens.json"), "w") as f:
        json.dump(out, f, ensure_ascii=True, separators=(",", ":"))
    for name in toks:
        print(name, os.path.getsize(os.path.join(HERE, f"bpe_{name}_tokenizer.json")), "bytes,", len(out[name]["cases"]), "cases")


if __name__ == "__main__":
    main()
