"""The projections of one MiniLM layer at a few rows (one sentence: 28 tokens; 64), HIP-event timed back to back through
kjarni_hip_op_linear: python tools/few_rows_probe.py [iters].  With the tuning build (KJARNI_FFI_LIB=...tuning.so) GEMM_VARIANT
selects a variant (tuning.h)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import torch  # noqa: F401  (HIP runtime first)
from kjarni_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
if os.environ.get("GEMM_VARIANT"):
    ops.set_gemm_variant(int(os.environ["GEMM_VARIANT"]))
rng = np.random.default_rng(0)
ops.linear(rng.standard_normal((4096, 384), dtype=np.float32), rng.standard_normal((1536, 384), dtype=np.float32), None, None,
           ops.EPI_BIAS, iters=3000)  # clocks up
for M in (28, 64):
    for name, K, N, epi, res in (("qkv", 384, 1152, ops.EPI_BIAS, False), ("fc1 + gelu", 384, 1536, ops.EPI_BIAS_GELU, False),
                                 ("out-proj + residual", 384, 384, ops.EPI_BIAS_RESIDUAL, True),
                                 ("fc2 + residual", 1536, 384, ops.EPI_BIAS_RESIDUAL, True),
                                 ("768 -> 768", 768, 768, ops.EPI_BIAS, False), ("3072 -> 768", 3072, 768, ops.EPI_BIAS_RESIDUAL, True)):
        x = rng.standard_normal((M, K), dtype=np.float32)
        w = (rng.standard_normal((N, K), dtype=np.float32) * 0.05).astype(np.float32)
        b = rng.standard_normal(N, dtype=np.float32)
        r = rng.standard_normal((M, N), dtype=np.float32) if res else None
        y, ms = ops.linear(x, w, b, r, epi, iters=iters)
        ref = x.astype(np.float64) @ w.astype(np.float64).T + b + (r if res else 0.0)
        if epi == ops.EPI_BIAS_GELU:
            from scipy.special import erf
            ref = 0.5 * ref * (1.0 + erf(ref / np.sqrt(2.0)))
        print(f"rows {M:3d} {name:20s} {ms * 1e3:7.2f} us per launch   max |err| vs float64 {float(np.abs(y - ref).max()):.1e}", flush=True)
g = (1 + 0.1 * rng.standard_normal(384)).astype(np.float32)
beta = (0.1 * rng.standard_normal(384)).astype(np.float32)
for M in (1, 28, 64):
    for name, K in (("out-proj + LN", 384), ("fc2 + LN", 1536)):
        x = rng.standard_normal((M, K), dtype=np.float32)
        w = (rng.standard_normal((384, K), dtype=np.float32) * 0.05).astype(np.float32)
        b = rng.standard_normal(384, dtype=np.float32)
        r = rng.standard_normal((M, 384), dtype=np.float32)
        y, ms = ops.linear_layer_norm(x, w, b, r, g, beta, 1e-12, iters=iters)
        t = x.astype(np.float64) @ w.astype(np.float64).T + b + r
        ref = (t - t.mean(-1, keepdims=True)) / np.sqrt(t.var(-1, keepdims=True) + 1e-12) * g + beta
        print(f"rows {M:3d} {name:20s} {ms * 1e3:7.2f} us per call (all its launches)   max |err| vs float64 {float(np.abs(y - ref).max()):.1e}", flush=True)
