#!/usr/bin/env python3
"""Extracts the conv-front-end golden vectors the reference holds as test data
(crates/kjarni-transformers/src/audio/mel.rs:487-2076: conv weights, biases, the first rows of the
position table, the mel input and the expected output of test_conv_frontend_golden) into
tests/golden/whisper_conv_frontend.json.  Run in the build container only (/root/reference is not
available on the GPU box)."""
import json
import os
import re

SRC = "/root/reference/crates/kjarni-transformers/src/audio/mel.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "whisper_conv_frontend.json")
text = open(SRC).read()


def vec(fn_name):
    m = re.search(r"fn %s\(\) -> Vec<f32> \{(.*?)\n    \}" % fn_name, text, re.S)
    body = m.group(1)
    body = body[body.index("vec!["):]
    return [float(x) for x in re.findall(r"-?\d+\.\d+(?:e-?\d+)?", body[:body.index("]")])]


pos = vec("get_pos_embed_data")
assert len(pos) % 8 == 0
out = dict(
    source="crates/kjarni-transformers/src/audio/mel.rs:487-2119 (test_conv_frontend_golden, tolerance 1e-4)",
    conv1_weight=dict(shape=[8, 4, 3], data=vec("get_conv1_weight_data")),
    conv1_bias=dict(shape=[8], data=vec("get_conv1_bias_data")),
    conv2_weight=dict(shape=[8, 8, 3], data=vec("get_conv2_weight_data")),
    conv2_bias=dict(shape=[8], data=vec("get_conv2_bias_data")),
    # the test passes all 1500 rows; only the first 5 (= output frames) are read
    embed_positions=dict(shape=[16, 8], data=pos[:16 * 8], rows_in_reference=len(pos) // 8),
    mel_input=dict(shape=[1, 4, 10], data=vec("get_mel_input_data")),
    frontend_output=dict(shape=[1, 5, 8], data=vec("get_frontend_output_data")),
)
for k, v in out.items():
    if isinstance(v, dict):
        n = 1
        for d in v["shape"]:
            n *= d
        assert len(v["data"]) == n, (k, len(v["data"]), n)
json.dump(out, open(OUT, "w"))
print("wrote", OUT)
