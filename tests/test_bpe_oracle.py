"""The BPE restatement (oracle/bpe_oracle.py) pinned by the `tokenizers` goldens of tests/golden/make_bpe_golden.py."""
import json
import os

import pytest

from oracle.bpe_oracle import BpeOracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["llama3", "qwen2", "gpt2"])
def test_oracle_matches_tokenizers(name):
    with open(os.path.join(GOLDEN, "bpe_goldens.json")) as f:
        g = json.load(f)[name]
    tok = BpeOracle(os.path.join(GOLDEN, f"bpe_{name}_tokenizer.json"))
    for case in g["cases"]:
        assert tok.encode(case["text"]) == case["ids"], repr(case["text"])
        assert tok.decode(case["ids"]) == case["decoded"]
        assert tok.decode(case["ids"], skip_special_tokens=True) == case["decoded_skip"]
        for i, text in zip(case["ids"][:24], case["single"]):
            assert tok.decode([i]) == text
    assert tok.encode(g["truncated"]["text"], g["truncated"]["max_length"]) == g["truncated"]["ids"]
