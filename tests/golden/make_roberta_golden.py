"""Fixtures for the RoBERTa / MPNet tokenizer shapes (tests/test_tokenizer_families.py):

  roberta_tokenizer.json   byte-level BPE (GPT-2 regex) + RobertaProcessing, specials <s> <pad> </s> <unk> <mask>(lstrip)
  mpnet_tokenizer.json     the BERT WordPiece fixture (tokenizer_small.json) with <s> </s> framing (RobertaProcessing)
  roberta_goldens.json     batches encoded by the `tokenizers` package configured as the reference configures it
                           (pipeline/encoder/loader.rs:98-115: truncation max_length, BatchLongest padding, pad id 0)

    python tests/golden/make_roberta_golden.py
"""
import json
import os

from tokenizers import AddedToken, Tokenizer, decoders, models, pre_tokenizers, processors, trainers

HERE = os.path.dirname(os.path.abspath(__file__))
from make_bpe_golden import CORPUS  # noqa: E402


def roberta():
    tok = Tokenizer(models.BPE(unk_token=None))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, trim_offsets=True, use_regex=True)
    tok.decoder = decoders.ByteLevel()
    specials = ["<s>", "<pad>", "</s>", "<unk>"]
    trainer = trainers.BpeTrainer(vocab_size=600, special_tokens=specials, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(),
                                  show_progress=False)
    tok.train_from_iterator(CORPUS * 3, trainer)
    tok.add_special_tokens([AddedToken("<mask>", lstrip=True, special=True, normalized=False)])
    tok.post_processor = processors.RobertaProcessing(sep=("</s>", 2), cls=("<s>", 0), trim_offsets=True, add_prefix_space=False)
    return tok


def mpnet():
    with open(os.path.join(HERE, "tokenizer_small.json")) as f:
        j = json.load(f)
    vocab = j["model"]["vocab"]
    nxt = max(vocab.values()) + 1
    for t in ("<s>", "</s>"):
        vocab[t] = nxt
        j["added_tokens"].append({"id": nxt, "content": t, "single_word": False, "lstrip": False, "rstrip": False, "normalized": False,
                                  "special": True})
        nxt += 1
    j["post_processor"] = {"type": "RobertaProcessing", "sep": ["</s>", vocab["</s>"]], "cls": ["<s>", vocab["<s>"]], "trim_offsets": True,
                           "add_prefix_space": True}
    return Tokenizer.from_str(json.dumps(j))


TEXTS = ["Hello, world!", "The quick brown fox jumps over the lazy dog.", "", " leading and trailing  ", "It's a <mask> day, isn't it?",
         "<s>already framed</s>", "unicode: þæö 日本語 😀 naïve café", "numbers 12345 and symbols #$%", "a", "tabs\tand\nnewlines\r\n",
         "A much longer sentence that should be truncated when the maximum length is small enough to matter for this test case."]
PAIRS_A = ["what is the capital of iceland?", "short", "a long query " * 6, "", "q <mask> q"]
PAIRS_B = ["Reykjavik is the capital of Iceland.", "a long document " * 8, "short doc", "only b", "<pad> in the text"]


def encode(tok, max_length, texts, pairs=None):
    tok.enable_truncation(max_length=max_length)
    tok.enable_padding()  # PaddingParams::default(): BatchLongest, pad_id 0, pad_type_id 0
    encs = tok.encode_batch(list(zip(texts, pairs)) if pairs is not None else texts, add_special_tokens=True)
    return {"ids": [e.ids for e in encs], "mask": [e.attention_mask for e in encs], "types": [e.type_ids for e in encs]}


def main():
    out = {}
    for name, tok in (("roberta", roberta()), ("mpnet", mpnet())):
        tok.no_padding()
        tok.no_truncation()
        tok.save(os.path.join(HERE, f"{name}_tokenizer.json"), pretty=False)
        cases = []
        for max_length in (512, 16, 7, 3):
            cases.append({"max_length": max_length, "texts": TEXTS, "pairs": None, **encode(tok, max_length, TEXTS)})
            cases.append({"max_length": max_length, "texts": PAIRS_A, "pairs": PAIRS_B, **encode(tok, max_length, PAIRS_A, PAIRS_B)})
        out[name] = cases
    with open(os.path.join(HERE, "roberta_goldens.json"), "w") as f:
        json.dump(out, f, ensure_ascii=True, separators=(",", ":"))
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
