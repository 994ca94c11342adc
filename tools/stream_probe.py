"""A/B of the continuous K-stream tile kernel (gemm_nt_f32_stream) against the per-tile-prologue kernel (tuning build, gemm
variant 53), on the headline's QKV and FC1 + GELU shapes; also that both give the same bits.
usage: KJARNI_FFI_LIB=.../libkjarni_ffi_tuning.so python tools/stream_probe.py [rows] [variants...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kjarni_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
variants = [int(v) for v in sys.argv[2:]] or [53, 0]
rng = np.random.default_rng(0)
for name, K, N, epi in [("qkv", 384, 1152, ops.EPI_BIAS), ("fc1", 384, 1536, ops.EPI_BIAS_GELU)]:
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = (rng.standard_normal((N, K), dtype=np.float32) * 0.05).astype(np.float32)
    b = rng.standard_normal(N, dtype=np.float32)
    rows = rng.choice(M, 64, replace=False)
    ref = x[rows].astype(np.float64) @ w.astype(np.float64).T + b
    if epi == ops.EPI_BIAS_GELU:
        from math import erf
        ref = 0.5 * ref * (1 + np.vectorize(erf)(ref / np.sqrt(2)))
    outs = {}
    for rounds in range(3):
        for v in variants:
            ops.set_gemm_variant(v)
            y, ms = ops.linear(x, w, b, None, epi, iters=20)
            tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
            err = np.abs(y[rows] - ref).max()
            print(f"{name} M={M} variant={v:3d} {ms:8.4f} ms {tf:7.2f} TFLOP/s ({tf/157.3*100:5.1f} %) max_err={err:.2e}", flush=True)
            if v in (0, 53):
                outs[v] = y
    if 0 in outs and 53 in outs:
        print(f"{name}: stream == per-tile kernel bit for bit: {np.array_equal(outs[0], outs[53])}", flush=True)
ops.set_gemm_variant(0)
