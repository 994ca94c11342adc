"""The 64 x 64-tile route on the QKV / FC1 shapes at a few call sizes, HIP-event timed back to back (kjarni_hip_op_linear); with
the tuning build GEMM_VARIANT selects knock-outs (tuning.h: 21 no global loads in the K-loop, 22 no MFMAs, 23 no barrier, 24 no LDS stores, 25 no LDS reads, 26 MFMAs only).
python tools/mid_gemm_probe.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import torch  # noqa: F401
from kjarni_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(0)
w0 = (rng.standard_normal((1536, 384), dtype=np.float32) * 0.05).astype(np.float32)
ops.linear(rng.standard_normal((4096, 384), dtype=np.float32), w0, None, None, ops.EPI_BIAS, iters=3000)  # clocks up
for variant in ([int(v) for v in os.environ.get('VARIANTS', '0,21,22,23,24,25,26').split(',')] if ops.has_tuning() else [0]):
    if ops.has_tuning():
        ops.set_gemm_variant(variant)
    for M in (1024, 2048, 4096, 8192):
        for name, K, N, epi in (("qkv", 384, 1152, ops.EPI_BIAS), ("fc1 + gelu", 384, 1536, ops.EPI_BIAS_GELU)):
            x = rng.standard_normal((M, K), dtype=np.float32)
            w = (rng.standard_normal((N, K), dtype=np.float32) * 0.05).astype(np.float32)
            b = rng.standard_normal(N, dtype=np.float32)
            _, ms = ops.linear(x, w, b, None, epi, iters=iters)
            tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
            print(f"variant {variant:2d} rows {M:5d} {name:12s} {ms * 1e3:8.2f} us  {tf:6.1f} TFLOP/s ({tf / 157.3 * 100:4.1f} % of the f32 MFMA peak)", flush=True)
if ops.has_tuning():
    ops.set_gemm_variant(0)
