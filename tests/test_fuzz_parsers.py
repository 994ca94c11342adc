"""Corrupted files through the host-side parsers of the C ABI: tokenizer.json (WordPiece and BPE loaders), WAV, and the
on-disk index (index.json, segment.json, docs.idx, bm25.bin, metadata.jsonl).  Every call must come back -- with data or
with an error code -- never crash; tools/asan_cpu_tests.sh runs the same cases under AddressSanitizer + UBSan."""
import json
import os
import random
import shutil
import struct

import numpy as np
import pytest

from kjarni_amd import _ffi
from kjarni_amd.chat import BpeTokenizer
from kjarni_amd.indexer import index_info, index_write
from kjarni_amd.searcher import search_keywords
from kjarni_amd.tokenizer import Tokenizer
from kjarni_amd.transcriber import load_wav

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
SCALE = int(os.environ.get("KJARNI_FUZZ_SCALE", "1"))  # tools/asan_cpu_tests.sh can run deeper


def _mutations(data: bytes, rng: random.Random, n: int):
    yield b""
    yield data[: len(data) // 2]
    yield data[:-1]
    for _ in range(n * SCALE):
        b = bytearray(data)
        kind = rng.randrange(5)
        if kind == 0:    # flip bytes
            for _ in range(rng.randint(1, 8)):
                b[rng.randrange(len(b))] = rng.randrange(256)
        elif kind == 1:  # truncate
            b = b[: rng.randrange(len(b))]
        elif kind == 2:  # delete a slice
            i = rng.randrange(len(b))
            del b[i: i + rng.randint(1, 64)]
        elif kind == 3:  # duplicate a slice
            i = rng.randrange(len(b))
            b[i:i] = b[i: i + rng.randint(1, 64)]
        else:            # plant extreme integers
            i = rng.randrange(max(1, len(b) - 8))
            b[i: i + 8] = struct.pack("<Q", rng.choice([0, 1, 2**31, 2**32 - 1, 2**63, 2**64 - 1]))
        yield bytes(b)


def _tolerant(fn):
    try:
        fn()
    except (_ffi.KjarniException, UnicodeDecodeError):
        pass


@pytest.mark.parametrize("name", ["bpe_llama3_tokenizer.json", "bpe_qwen2_tokenizer.json", "spbpe_legacy_tokenizer.json", "roberta_tokenizer.json",
                                  "tokenizer_small.json", "mpnet_tokenizer.json", "unigram_tokenizer.json"])
def test_corrupt_tokenizer_json(tmp_path, name):
    data = open(os.path.join(GOLDEN, name), "rb").read()
    rng = random.Random(len(name))
    p = str(tmp_path / "tokenizer.json")
    for blob in _mutations(data, rng, 120):
        with open(p, "wb") as f:
            f.write(blob)

        def bpe():
            t = BpeTokenizer(p)
            t.decode(t.encode("Hello [INST] wörld <|eot_id|> 12345\n"))

        def enc():
            Tokenizer(p, 32).encode_batch(["Hello wörld ﬁ e\u0301 \r\n <mask>", "x" * 300], ["pair", ""])

        _tolerant(bpe)
        _tolerant(enc)
    # structural edits that keep the JSON valid
    j = json.loads(data)
    for path, value in [(("model", "vocab"), {}), (("model", "merges"), [["a"]]), (("model", "merges"), ["nospace"]), (("added_tokens",), [{}]),
                        (("added_tokens",), [{"id": 2**40, "content": "x"}]), (("model", "vocab"), {"a": -5, "b": 2**40}),
                        (("post_processor",), {"type": "RobertaProcessing", "sep": [], "cls": []}), (("normalizer",), {"type": "Sequence"}),
                        (("pre_tokenizer",), {"type": "Sequence", "pretokenizers": [{}]}), (("model",), None), (("decoder",), {"type": "Sequence", "decoders": [1, 2]})]:
        k = json.loads(json.dumps(j))
        node = k
        for key in path[:-1]:
            node = node.get(key) if isinstance(node, dict) else None
            if node is None:
                break
        if isinstance(node, dict):
            node[path[-1]] = value
        with open(p, "w") as f:
            json.dump(k, f)
        _tolerant(lambda: BpeTokenizer(p).encode("abc def"))
        _tolerant(lambda: Tokenizer(p, 16).encode_batch(["abc def"]))


def test_corrupt_wav(tmp_path):
    import wave
    src = str(tmp_path / "a.wav")
    with wave.open(src, "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(22050)
        w.writeframes((np.sin(np.arange(4000) / 7.0) * 9000).astype("<i2").tobytes())
    data = open(src, "rb").read()
    rng = random.Random(5)
    p = str(tmp_path / "b.wav")
    for blob in _mutations(data, rng, 300):
        with open(p, "wb") as f:
            f.write(blob)
        _tolerant(lambda: load_wav(p))


def test_corrupt_index_files(tmp_path):
    rng = random.Random(9)
    src = str(tmp_path / "index")
    texts = [f"document number {i} about {'cats' if i % 2 else 'dogs'} and things" for i in range(40)]
    emb = np.random.default_rng(0).normal(size=(40, 8)).astype(np.float32)
    index_write(src, 8, texts, emb, [{"source": f"f{i}.txt"} for i in range(40)], max_docs_per_segment=16)
    files = []
    for root, _, names in os.walk(src):
        files += [os.path.join(root, n) for n in names]
    assert any(f.endswith("bm25.bin") for f in files)
    work = str(tmp_path / "work")
    for target in files:
        data = open(target, "rb").read()
        for blob in _mutations(data, rng, 25):
            if os.path.exists(work):
                shutil.rmtree(work)
            shutil.copytree(src, work)
            with open(os.path.join(work, os.path.relpath(target, src)), "wb") as f:
                f.write(blob)
            _tolerant(lambda: search_keywords(work, "cats and dogs", 5))
            _tolerant(lambda: index_info(work))
