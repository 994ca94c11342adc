#!/usr/bin/env python3
"""Extracts the real-weight golden VALUES the reference's own tests hold (numbers and input strings only) into
tests/golden/real_weight_goldens.json.  Run in the build container, where /root/reference exists:

    python tests/golden/make_real_weight_golden.py

Sources (data, not code):
  crates/kjarni-models/src/models/sentence_encoder/tests.rs:208-300, 411-1184   "ILoveEdgeGPT" CLS (raw) and mean+L2 vectors, tol 1e-3
  crates/kjarni-models/src/models/cross_encoder/tests.rs:38-100                 pair score 3.1776933670043945, rerank order [0, 2, 3, 1]
  crates/kjarni-ffi/bindings/csharp/Kjarni.Tests/EmbedderTests.cs:34-93         first five values, similarities
  crates/kjarni-ffi/bindings/csharp/Kjarni.Tests/RerankerTests.cs:21-52, 186    pair scores, top-1 of a rerank
  crates/kjarni-ffi/bindings/csharp/Kjarni.Tests/ClassifierTests.cs:28-60       sst-2 labels and scores
tests/test_real_weights.py asserts them when the three model directories are on disk (there is no network here, so
they usually are not; the test skips then)."""
import json
import os
import re

REF = "/root/reference/crates"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "real_weight_goldens.json")


def floats_after(text, marker):
    start = text.index(marker)
    body = text[text.index("[[", start) + 2:text.index("]]", start)]
    return [float(x) for x in re.findall(r"-?\d+\.\d+(?:[eE][-+]?\d+)?", body)]


def inline_data(text, method):
    """Argument tuples of the [InlineData(...)] attributes directly above `method`."""
    head = text[:text.index(method)]
    block = head[head.rindex("[Theory]"):]
    rows = []
    for m in re.finditer(r"\[InlineData\((.*?)\)\]", block, flags=re.S):
        args = re.findall(r'"((?:[^"\\]|\\.)*)"|(-?\d+\.\d+)f', m.group(1))
        rows.append([s if s or not f else float(f) for s, f in args])
    return rows


def main():
    se = open(f"{REF}/kjarni-models/src/models/sentence_encoder/tests.rs").read()
    cls = floats_after(se, "const CLS_DATA")
    mean = floats_after(se, "const DATA")
    assert len(cls) == 384 and len(mean) == 384
    ce = open(f"{REF}/kjarni-models/src/models/cross_encoder/tests.rs").read()
    pair = float(re.search(r"let torch_value = ([\d.]+);", ce).group(1))
    docs = re.findall(r'"([^"]+)",', ce[ce.index("let documents = vec!["):ce.index("let ranked = encoder.rerank")])
    order = [int(x) for x in re.search(r"vec!\[(\d+), (\d+), (\d+), (\d+)\];", ce).groups()]
    cs = f"{REF}/kjarni-ffi/bindings/csharp/Kjarni.Tests"
    emb = open(f"{cs}/EmbedderTests.cs").read()
    first5 = [float(x) for x in re.findall(r"Assert\.Equal\(\s*(-?\d+\.\d+)f, embedding\[\d\], 4\)", emb)]
    sims = inline_data(emb, "public void Similarity_ExactValues")
    rr = open(f"{cs}/RerankerTests.cs").read()
    scores = inline_data(rr, "public void Score_ExactValues")
    cl = open(f"{cs}/ClassifierTests.cs").read()
    pos = inline_data(cl, "public void Classify_PositiveText")
    neg = inline_data(cl, "public void Classify_NegativeText")
    out = {
        "sentence_encoder": {"model": "minilm-l6-v2", "text": "ILoveEdgeGPT", "tolerance": 1e-3,
                             "cls_raw": cls, "mean_l2": mean},
        "cross_encoder": {"model": "minilm-l6-v2-cross-encoder", "tolerance": 1e-3,
                          "pair": {"query": "i love edgeGPT", "document": "edgeGPT is a new model inference library",
                                   "score": pair},
                          "rerank": {"query": "machine learning algorithms", "documents": docs, "order": order}},
        "csharp_embedder": {"model": "minilm-l6-v2", "hello_world_first5": first5, "first5_decimals": 4,
                            "similarities": [{"a": a, "b": b, "value": v} for a, b, v in sims], "similarity_decimals": 3},
        "csharp_reranker": {"model": "minilm-l6-v2-cross-encoder", "decimals": 2,
                            "scores": [{"query": q, "document": d, "value": v} for q, d, v in scores]},
        "csharp_classifier": {"model": "distilbert-sentiment", "decimals": 3,
                              "cases": [{"text": t, "label": l, "score": s} for t, l, s in pos + neg] +
                                       [{"text": "This is a test sentence.", "label": "NEGATIVE", "score": 0.981}]},
    }
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print(f"wrote {OUT}: {len(cls)} + {len(mean)} vector values, {len(sims)} similarities, {len(scores)} pair scores, "
          f"{len(pos) + len(neg) + 1} classifications, order {order}, {len(docs)} documents")


if __name__ == "__main__":
    main()
