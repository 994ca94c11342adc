"""gemm_flex.hip (the mid-size projections' kernel) on the MiniLM shapes at a few call sizes: checked against float64, then
HIP-event timed back to back (kjarni_hip_op_linear / kjarni_hip_op_linear_layer_norm).  With the tuning build
(KJARNI_FFI_LIB=kjarni_amd/lib/libkjarni_ffi_tuning.so) every tile (RA, CB) is timed next to the launcher's choice and next to
the 64 x 64-tile route it replaced (gemm variant 8).
python tools/flex_probe.py [iters] [rows,rows,...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import torch  # noqa: F401
from kjarni_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rows = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1024, 2048, 3200, 4096, 8192]
rng = np.random.default_rng(0)
SHAPES = (("qkv", 384, 1152, ops.EPI_BIAS, False), ("fc1+gelu", 384, 1536, ops.EPI_BIAS_GELU, False),
          ("out+ln", 384, 384, None, True), ("fc2+ln", 1536, 384, None, True))
CONFIGS = [(ra, cb) for ra in (1, 2) for cb in (3, 4, 6, 8, 9, 12)]


def run(name, K, N, epi, ln, M, check):
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = (rng.standard_normal((N, K), dtype=np.float32) * 0.05).astype(np.float32)
    b = rng.standard_normal(N, dtype=np.float32)
    if ln:
        r = rng.standard_normal((M, N), dtype=np.float32)
        gm = rng.standard_normal(N, dtype=np.float32)
        bt = rng.standard_normal(N, dtype=np.float32)
        y, ms = ops.linear_layer_norm(x, w, b, r, gm, bt, 1e-12, iters=iters)
        if check:
            sel = rng.choice(M, size=min(M, 256), replace=False)
            v = x[sel].astype(np.float64) @ w.astype(np.float64).T + b + r[sel]
            mu = v.mean(-1, keepdims=True)
            ref = (v - mu) / np.sqrt(v.var(-1, keepdims=True) + 1e-12) * gm + bt
            err = float(np.abs(y[sel] - ref).max())
        else:
            err = float("nan")
    else:
        y, ms = ops.linear(x, w, b, None, epi, iters=iters)
        if check:
            sel = rng.choice(M, size=min(M, 256), replace=False)
            ref = x[sel].astype(np.float64) @ w.astype(np.float64).T + b
            if epi == ops.EPI_BIAS_GELU:
                from math import erf
                ref = 0.5 * ref * (1.0 + np.vectorize(erf)(ref * 0.7071067811865475))
            err = float(np.abs(y[sel] - ref).max())
        else:
            err = float("nan")
    return y, ms, err


ops.linear(rng.standard_normal((4096, 384), dtype=np.float32), (rng.standard_normal((1536, 384), dtype=np.float32) * 0.05), None, None,
           ops.EPI_BIAS, iters=3000)  # clocks up
for M in rows:
    for name, K, N, epi, ln in SHAPES:
        if ops.has_tuning():
            ops.set_gemm_variant(0)
        y0, ms, err = run(name, K, N, epi, ln, M, True)
        tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        line = f"rows {M:5d} {name:9s} chosen {ms * 1e3:7.2f} us {tf:6.1f} TF/s  err {err:.2e}"
        if ops.has_tuning():
            rng2 = np.random.default_rng(0)
            ops.set_gemm_variant(8)
            _, ms_old, err_old = run(name, K, N, epi, ln, M, True)
            line += f" | 64x64 tiles {ms_old * 1e3:7.2f} us err {err_old:.2e} |"
            best = None
            for ra, cb in CONFIGS:
                if 16 * cb > N:
                    continue
                ops.set_gemm_variant(2000 + 100 * ra + cb)
                _, ms_c, err_c = run(name, K, N, epi, ln, M, True)
                line += f" {64 * ra}x{16 * cb}:{ms_c * 1e3:.1f}"
                if err_c > 1e-4:
                    line += f"(ERR {err_c:.1e})"
                if best is None or ms_c < best[0]:
                    best = (ms_c, ra, cb)
            line += f" | best {64 * best[1]}x{16 * best[2]} {best[0] * 1e3:.2f} us | one workgroup per CU:"
            for ra, cb in CONFIGS:
                if 16 * cb > N:
                    continue
                ops.set_gemm_variant(3000 + 100 * ra + cb)
                _, ms_c, err_c = run(name, K, N, epi, ln, M, False)
                line += f" {64 * ra}x{16 * cb}:{ms_c * 1e3:.1f}"
            ops.set_gemm_variant(0)
        print(line, flush=True)
