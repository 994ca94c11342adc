"""The five hot-path launches of one MiniLM layer at chunk size (262 144 token rows), HIP-event timed through the
kjarni_hip_op_* entry points: python tools/kernel_probe.py [rows] [iters]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import torch  # noqa: F401  (HIP runtime first)
from kjarni_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(0)
PEAK = 157.3


def report(name, flops, ms):
    tf = flops / (ms * 1e-3) / 1e12
    print(f"{name:22s} rows={M} {ms:8.4f} ms {tf:7.2f} TFLOP/s ({tf / PEAK * 100:5.1f}% of the f32 MFMA peak)", flush=True)


x384 = rng.standard_normal((M, 384), dtype=np.float32)
# clocks and power state settle over the first seconds of load: an untimed burn first
_w = (rng.standard_normal((1536, 384), dtype=np.float32) * 0.05).astype(np.float32)
ops.linear(x384, _w, None, None, ops.EPI_BIAS, iters=1500)
for name, K, N, epi in (("qkv", 384, 1152, ops.EPI_BIAS), ("fc1 + gelu", 384, 1536, ops.EPI_BIAS_GELU)):
    w = (rng.standard_normal((N, K), dtype=np.float32) * 0.05).astype(np.float32)
    b = rng.standard_normal(N, dtype=np.float32)
    _, ms = ops.linear(x384, w, b, None, epi, iters=iters)
    report(name, 2.0 * M * N * K, ms)
g = (1 + 0.1 * rng.standard_normal(384)).astype(np.float32)
beta = (0.1 * rng.standard_normal(384)).astype(np.float32)
for name, K in (("out-proj + LN", 384), ("fc2 + LN", 1536)):
    x = x384 if K == 384 else rng.standard_normal((M, K), dtype=np.float32)
    w = (rng.standard_normal((384, K), dtype=np.float32) * 0.05).astype(np.float32)
    b = rng.standard_normal(384, dtype=np.float32)
    _, ms = ops.linear_layer_norm(x, w, b, x384, g, beta, 1e-12, iters=iters)
    report(name, 2.0 * M * 384 * K, ms)
    del x
B = M // 128
qkv = rng.standard_normal((B, 128, 1152), dtype=np.float32)
mask = np.ones((B, 128), np.uint32)
_, ms = ops.attention(qkv, mask, 12, iters=iters)
report("attention", 4.0 * B * 128 * 128 * 384, ms)
