"""Random-shape parity sweep of the projection routes (few rows / 64 x 64 tiles with K slices / 128 x 128 tiles, with and
without the fused LayerNorm) against a float64 numpy reference: python tools/gemm_fuzz.py [cases] [seed]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import torch  # noqa: F401  (HIP runtime first)
from kjarni_amd import ops


def gelu(x):
    from math import erf
    return 0.5 * x * (1.0 + np.vectorize(erf)(x / np.sqrt(2.0)))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = 0.0
    for i in range(cases):
        m = int(rng.choice([1, 7, 33, 64, 65, 100, 129, 500, 1000, 2049, 4096, 4100, 8192, 8193, 9000]))
        if rng.random() < 0.5:
            m = int(rng.integers(1, 9000))
        n = int(rng.choice([4, 32, 60, 64, 128, 256, 384, 388, 512, 768, 1024, 1152, 1536, 2048]))
        k = int(rng.choice([32, 64, 96, 128, 384, 512, 768, 1024, 1536, 2048, 3072]))
        x = rng.standard_normal((m, k)).astype(np.float32)
        w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
        b = rng.standard_normal(n).astype(np.float32) if rng.random() < 0.8 else None
        r = rng.standard_normal((m, n)).astype(np.float32)
        base = x.astype(np.float64) @ w.astype(np.float64).T + (b.astype(np.float64) if b is not None else 0.0)
        kind = int(rng.integers(0, 5))
        if kind == 4 and n < 64:  # LayerNorm over a handful of columns is ill-conditioned in f32 whatever computes it
            kind = 2
        if kind == 0:
            got, _ = ops.linear(x, w, b, None, ops.EPI_BIAS)
            ref = base
        elif kind == 1:
            got, _ = ops.linear(x, w, b, None, ops.EPI_BIAS_GELU)
            ref = gelu(base)
        elif kind == 2:
            got, _ = ops.linear(x, w, b, r, ops.EPI_BIAS_RESIDUAL)
            ref = base + r
        elif kind == 3:
            got, _ = ops.linear(x, w, b, r, ops.EPI_BIAS_MUL_SILU)
            ref = (r / (1.0 + np.exp(-r.astype(np.float64)))) * base
        else:
            g = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32)
            beta = (0.1 * rng.standard_normal(n)).astype(np.float32)
            got, _ = ops.linear_layer_norm(x, w, b, r, g, beta, 1e-12)
            v = base + r
            mu = v.mean(axis=1, keepdims=True)
            ref = (v - mu) / np.sqrt(((v - mu) ** 2).mean(axis=1, keepdims=True) + 1e-12) * g + beta
        err = float(np.abs(got - ref).max()) / max(1.0, float(np.abs(ref).max()))
        worst = max(worst, err)
        flag = "" if err < 2e-5 else "   <-- FAIL"
        if flag or i % 20 == 0:
            print(f"case {i}: kind {kind} m={m} n={n} k={k} bias={'y' if b is not None else 'n'} rel err {err:.2e}{flag}", flush=True)
        if flag:
            sys.exit(1)
    print(f"{cases} cases ok, worst relative error {worst:.2e}")


if __name__ == "__main__":
    main()
