"""The large-batch projections with their f32 products on the bf16 matrix cores (KJARNI_HIP_F32_ON_BF16=1: three bf16 pieces per
operand, six cross products) against the f32 MFMA kernels: time per launch and the error of both against float64 on sampled
rows.  python tools/split_probe.py [rows] [iters]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch  # noqa: F401
    from kjarni_amd import ops
    M, iters = int(sys.argv[2]), int(sys.argv[3])
    rng = np.random.default_rng(0)
    ops.linear(rng.standard_normal((8192, 384), dtype=np.float32), rng.standard_normal((1536, 384), dtype=np.float32), None, None,
               ops.EPI_BIAS, iters=300)
    for name, K, N, epi, res in (("qkv", 384, 1152, ops.EPI_BIAS, False), ("fc1 + gelu", 384, 1536, ops.EPI_BIAS_GELU, False),
                                 ("out-proj + residual", 384, 384, ops.EPI_BIAS_RESIDUAL, True),
                                 ("fc2 + residual", 1536, 384, ops.EPI_BIAS_RESIDUAL, True)):
        x = rng.standard_normal((M, K), dtype=np.float32)
        w = (rng.standard_normal((N, K), dtype=np.float32) * 0.05).astype(np.float32)
        b = rng.standard_normal(N, dtype=np.float32)
        r = rng.standard_normal((M, N), dtype=np.float32) if res else None
        y, ms = ops.linear(x, w, b, r, epi, iters=iters)
        rows = rng.choice(M, 64, replace=False)
        ref = x[rows].astype(np.float64) @ w.astype(np.float64).T + b + (r[rows] if res else 0.0)
        if epi == ops.EPI_BIAS_GELU:
            from scipy.special import erf
            ref = 0.5 * ref * (1.0 + erf(ref / np.sqrt(2.0)))
        tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        print(f"  {name:20s} {ms:8.4f} ms  {tf:7.1f} TFLOP/s of f32 products   max |err| vs float64 {float(np.abs(y[rows] - ref).max()):.2e}",
              flush=True)
    sys.exit(0)
rows = sys.argv[1] if len(sys.argv) > 1 else "262144"
iters = sys.argv[2] if len(sys.argv) > 2 else "10"
for mode in ("0", "1"):
    print("f32 MFMA" if mode == "0" else "f32 products on the bf16 matrix cores (3 pieces, 6 products)", flush=True)
    env = dict(os.environ, KJARNI_HIP_F32_ON_BF16=mode)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", rows, iters], env=env, check=False)
