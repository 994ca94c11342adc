#!/usr/bin/env python3
"""Extracts the golden vectors the reference's tests hold for the decoder-only path into
tests/golden/llm_goldens.json (run in the build container only):
  GQA attention with cache   crates/kjarni-transformers/src/cpu/decoder/decoder_attention.rs:202-396
  RoPE PyTorch parity        cpu/rope/tests.rs:24-311
  RMSNorm PyTorch parity     cpu/normalization/rms_norm.rs:209-246
"""
import json
import os
import re

ROOT = "/root/reference/crates/kjarni-transformers/src/"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "llm_goldens.json")
NUM = r"-?\d+\.\d+(?:e-?\d+)?"


def vec_after(text, name):
    m = re.search(r"let %s(?:: Vec<f32>)? = vec!\[(.*?)\];" % re.escape(name), text, re.S)
    assert m, name
    return [float(x) for x in re.findall(NUM, m.group(1))]


att = open(ROOT + "cpu/decoder/decoder_attention.rs").read()
rope = open(ROOT + "cpu/rope/tests.rs").read()
rms = open(ROOT + "cpu/normalization/rms_norm.rs").read()
rope_test = rope[rope.index("fn test_rope_pytorch_parity()"):rope.index("fn test_rope_actually_rotates")]
rms_test = rms[rms.index("fn test_rmsnorm_pytorch_parity()"):]
out = dict(
    gqa=dict(source="cpu/decoder/decoder_attention.rs:202-396 (hidden 16, 4 heads, 2 kv heads, tolerance 1e-4)",
             weight_q=vec_after(att, "weight_q_data"), weight_k=vec_after(att, "weight_k_data"),
             weight_v=vec_after(att, "weight_v_data"), weight_o=vec_after(att, "weight_o_data"),
             hidden=vec_after(att, "attn_input_hidden_data"), history_k=vec_after(att, "attn_history_k_data"),
             history_v=vec_after(att, "attn_history_v_data"), output=vec_after(att, "attn_output_data"),
             update_k=vec_after(att, "attn_update_k_data"), update_v=vec_after(att, "attn_update_v_data")),
    rope=dict(source="cpu/rope/tests.rs:24-311 (head_dim 8, seq 4, offset 10, theta 10000, shape [1,2,4,8], tolerance 1e-5)",
              q=vec_after(rope_test, "q_input_vec"), k=vec_after(rope_test, "k_input_vec"),
              expected_q=vec_after(rope_test, "expected_q_vec"), expected_k=vec_after(rope_test, "expected_k_vec")),
    rmsnorm=dict(source="cpu/normalization/rms_norm.rs:209-246 (eps 1e-5)",
                 input=vec_after(rms_test, "input_vec"), gamma=vec_after(rms_test, "gamma_vec"),
                 expected=vec_after(rms_test, "expected_output_vec")),
)
assert len(out["gqa"]["weight_q"]) == 256 and len(out["gqa"]["weight_k"]) == 128 and len(out["rope"]["q"]) == 64
json.dump(out, open(OUT, "w"))
print("wrote", OUT, {k: {kk: len(vv) for kk, vv in v.items() if isinstance(vv, list)} for k, v in out.items()})
