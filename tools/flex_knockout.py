"""gemm_flex.hip knock-outs (tuning build: KJARNI_FFI_LIB=kjarni_amd/lib/libkjarni_ffi_tuning.so): the 128 x 144 (QKV) and 128 x 192
(FC1) tiles at 4 096 rows with parts of the kernel switched off (tuning.h: 10000 d + 2000 + 100 RA + CB).  HIP-event time per launch,
back to back; run under `rocprofv3 --kernel-trace --stats` for the kernels' own durations.
python tools/flex_knockout.py [iters] [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import torch  # noqa: F401
from kjarni_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rng = np.random.default_rng(0)
NAMES = {0: "whole kernel", 1: "no output stores", 2: "no global loads in the K-loop", 3: "MFMAs only in the K-loop", 4: "no barrier in the K-loop",
         5: "no staging in the K-loop", 6: "staging loads sc1 (bypass L1)", 7: "staging loads sc0", 8: "staging loads nt", 9: "output stores: streaming, or KJARNI_HIP_FLEX_STORE=10/11/12 (sc1 / sc0 / both)"}
ops.linear(rng.standard_normal((4096, 384), dtype=np.float32), (rng.standard_normal((1536, 384), dtype=np.float32) * 0.05), None, None,
           ops.EPI_BIAS, iters=3000)  # clocks up
for name, K, N, epi, cfg in (("qkv 128x144", 384, 1152, ops.EPI_BIAS, 209), ("fc1+gelu 128x192", 384, 1536, ops.EPI_BIAS_GELU, 212),
                             ("fc1 (no gelu) 128x192", 384, 1536, ops.EPI_BIAS, 212)):
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = (rng.standard_normal((N, K), dtype=np.float32) * 0.05).astype(np.float32)
    b = rng.standard_normal(N, dtype=np.float32)
    for d in ((0, 9, 0, 9, 0, 9) if os.environ.get("ONLY_NT") else (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 0)):
        ops.set_gemm_variant(10000 * d + 2000 + cfg)
        _, ms = ops.linear(x, w, b, None, epi, iters=iters)
        print(f"rows {M} {name:22s} {NAMES[d]:32s} {ms * 1e3:7.2f} us", flush=True)
ops.set_gemm_variant(0)
