"""The per-call tile kernel (gemm_flex.hip, 16 x 16 x 4 MFMAs) against the large-batch tiles (gemm.hip, 32 x 32 x 2) on the
headline's shapes at chunk size (tuning build): python tools/flex_large_probe.py [rows=131072] [iters=10]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import torch  # noqa: F401
from kjarni_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(0)
for name, K, N, epi in (("qkv", 384, 1152, ops.EPI_BIAS), ("fc1+gelu", 384, 1536, ops.EPI_BIAS_GELU)):
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = (rng.standard_normal((N, K), dtype=np.float32) * 0.05).astype(np.float32)
    b = rng.standard_normal(N, dtype=np.float32)
    line = f"rows {M} {name:9s}"
    for label, v in (("large tiles", 0), ("flex 128x96", 5206), ("flex 128x128", 5208), ("flex 128x144", 5209), ("flex 128x192", 5212),
                     ("flex 64x192", 5112), ("large tiles", 0)):
        ops.set_gemm_variant(v)
        _, ms = ops.linear(x, w, b, None, epi, iters=iters)
        line += f" | {label} {ms * 1e3:.0f} us {2.0 * M * N * K / (ms * 1e-3) / 1e12:.1f} TF/s"
    print(line, flush=True)
ops.set_gemm_variant(0)
