"""Fixture for the SentencePiece-style BPE tokenizer.json shape of Llama 2 / Mistral checkpoints:

  normalizer  Sequence[Prepend("▁"), Replace(" " -> "▁")]   (or none + Metaspace pre-tokenizer, second fixture)
  model       BPE, byte_fallback, unk "<unk>", fuse_unk
  decoder     Sequence[Replace("▁" -> " "), ByteFallback, Fuse, Strip(" ", 1, 0)]

    python tests/golden/make_spbpe_golden.py   ->  spbpe_{legacy,metaspace}_tokenizer.json, spbpe_goldens.json
"""
import json
import os
import random

from tokenizers import AddedToken, Tokenizer, decoders, models, normalizers, pre_tokenizers, trainers

HERE = os.path.dirname(os.path.abspath(__file__))
from make_bpe_golden import CORPUS  # noqa: E402

SP = "▁"


def build(metaspace: bool) -> Tokenizer:
    tok = Tokenizer(models.BPE(unk_token="<unk>", fuse_unk=True, byte_fallback=True))
    tok.normalizer = normalizers.Sequence([normalizers.Prepend(SP), normalizers.Replace(" ", SP)])
    trainer = trainers.BpeTrainer(vocab_size=500, special_tokens=["<unk>", "<s>", "</s>"], show_progress=False)
    words = []
    for line in CORPUS[:6] + CORPUS[8:13]:  # mostly Latin text: other scripts fall back to bytes, as with the real vocabularies
        words += line.split(" ")
    tok.train_from_iterator(words * 3, trainer)
    j = json.loads(tok.to_str())
    vocab = j["model"]["vocab"]
    nxt = max(vocab.values()) + 1
    for b in range(256):
        vocab[f"<0x{b:02X}>"] = nxt
        nxt += 1
    if metaspace:
        j["normalizer"] = None
        j["pre_tokenizer"] = {"type": "Metaspace", "replacement": SP, "prepend_scheme": "first", "split": False}
    tok = Tokenizer.from_str(json.dumps(j))
    tok.decoder = decoders.Sequence([decoders.Replace(SP, " "), decoders.ByteFallback(), decoders.Fuse(), decoders.Strip(" ", 1, 0)])
    tok.add_special_tokens([AddedToken(t, special=True, normalized=False) for t in ("[INST]", "[/INST]", "[TOOL_CALLS]")])
    return tok


def texts():
    rng = random.Random(11)
    base = list(CORPUS) + ["", " ", "  ", "a", " a", "a ", "Hello world", "  two  spaces  ", "\n", "line\nbreak\ttab",
                           "<s>[INST] You are helpful.\n\nWhat is 2 + 2? [/INST]", "<s>[INST] Hello! [/INST] Hi there!</s>[INST] How are you? [/INST]",
                           "[INST][/INST]", "x[INST]y", "▁already", "emoji \U0001F600 and 日本", "<unk> literal", "</s>",
                           "café naïve þæö", "\x00 nul", "tab\tsep"]
    alphabet = "abc ABC 123 \n.,!?-é日\U0001F600" + SP
    for _ in range(40):
        base.append("".join(rng.choice(alphabet) for _ in range(rng.randint(1, 30))))
    return base


def main():
    out = {}
    for name, metaspace in (("legacy", False), ("metaspace", True)):
        tok = build(metaspace)
        tok.save(os.path.join(HERE, f"spbpe_{name}_tokenizer.json"), pretty=False)
        cases = []
        for t in texts():
            ids = tok.encode(t, add_special_tokens=False).ids
            cases.append({"text": t, "ids": ids, "decoded": tok.decode(ids, skip_special_tokens=False),
                          "decoded_skip": tok.decode(ids, skip_special_tokens=True),
                          "single": [tok.decode([i], skip_special_tokens=False) for i in ids[:32]]})
        long_text = " ".join(CORPUS[:4])
        out[name] = {"cases": cases, "truncated": {"text": long_text, "max_length": 40,
                                                     "ids": tok.encode(long_text, add_special_tokens=False).ids[:40]}}
    with open(os.path.join(HERE, "spbpe_goldens.json"), "w") as f:
        json.dump(out, f, ensure_ascii=True, separators=(",", ":"))
    print({k: len(v["cases"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
