"""RoBERTa (byte-level BPE + <s> </s> framing) and MPNet (WordPiece + <s> </s> framing) tokenizer.json shapes through
the encoder tokenizer of the library, against `tokenizers` goldens (tests/golden/make_roberta_golden.py): ids, masks and
type ids of single and pair batches with the reference's truncation / BatchLongest padding.  Host-only."""
import json
import os

import numpy as np
import pytest

from kjarni_amd.tokenizer import Tokenizer

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["roberta", "mpnet"])
def test_batches_match_tokenizers(name):
    with open(os.path.join(GOLDEN, "roberta_goldens.json")) as f:
        cases = json.load(f)[name]
    for case in cases:
        tok = Tokenizer(os.path.join(GOLDEN, f"{name}_tokenizer.json"), case["max_length"])
        ids, mask, types = tok.encode_batch(case["texts"], case["pairs"])
        assert ids.tolist() == case["ids"], (name, case["max_length"], case["pairs"] is not None)
        assert mask.tolist() == case["mask"]
        assert types.tolist() == case["types"]


def test_roberta_pair_framing_and_mask_token():
    tok = Tokenizer(os.path.join(GOLDEN, "roberta_tokenizer.json"), 64)
    ids, mask, types = tok.encode_batch(["ab"], ["cd"])
    row = ids[0][mask[0] == 1].tolist()
    assert row[0] == 0 and row[-1] == 2 and row.count(2) == 3  # <s> A </s></s> B </s>
    assert not types.any()
    ids, _, _ = tok.encode_batch(["a <mask> b"])
    assert 600 in ids[0].tolist()  # the added <mask> token (lstrip: it swallows the space before it)
