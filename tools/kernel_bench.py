"""Kernel micro-benchmarks on the GPU box (kjarni_hip_op_* entry points, HIP-event timed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kjarni_amd import ops

M = int(os.environ.get("KB_M", 65536))
rng = np.random.default_rng(0)
SHAPES = [("qkv", 384, 1152, ops.EPI_BIAS), ("out", 384, 384, ops.EPI_BIAS_RESIDUAL),
          ("fc1", 384, 1536, ops.EPI_BIAS_GELU), ("fc2", 1536, 384, ops.EPI_BIAS_RESIDUAL)]
variants = [int(v) for v in os.environ.get("KB_VARIANTS", "0,1").split(",")]
for name, K, N, epi in SHAPES:
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = (rng.standard_normal((N, K), dtype=np.float32) * 0.05).astype(np.float32)
    b = rng.standard_normal(N, dtype=np.float32)
    r = rng.standard_normal((M, N), dtype=np.float32) if epi == ops.EPI_BIAS_RESIDUAL else None
    rows = rng.choice(M, 64, replace=False)
    ref = x[rows].astype(np.float64) @ w.astype(np.float64).T + b
    if r is not None:
        ref += r[rows]
    if epi == ops.EPI_BIAS_GELU:
        from math import erf
        ref = 0.5 * ref * (1 + np.vectorize(erf)(ref / np.sqrt(2)))
    for rounds in range(2):
        for v in variants:
            ops.set_gemm_variant(v)
            y, ms = ops.linear(x, w, b, r, epi, iters=int(os.environ.get("KB_ITERS", 20)))
            err = np.abs(y[rows] - ref).max()
            tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
            print(f"gemm {name:4s} M={M} N={N:5d} K={K:5d} variant={v} {ms:8.4f} ms {tf:7.2f} TFLOP/s "
                  f"({tf/157.3*100:5.1f}% peak) max_err={err:.2e}", flush=True)
ops.set_gemm_variant(0)

B, S, heads, d = M // 128, 128, 12, 32
qkv = rng.standard_normal((B, S, 3 * heads * d), dtype=np.float32)
mask = np.ones((B, S), np.uint32)
for ragged in (False, True):
    if ragged:
        for b in range(B):
            mask[b, rng.integers(16, S + 1):] = 0
    for v in (0, 1, 0, 1):
        ops.set_attention_variant(v)
        ctx, ms = ops.attention(qkv, mask, heads, iters=int(os.environ.get("KB_ITERS", 20)))
        fl = 4.0 * B * S * S * heads * d
        print(f"attention B={B} S={S} h={heads} d={d} ragged={ragged} variant={v}: {ms:.4f} ms "
              f"{fl/(ms*1e-3)/1e12:.2f} TFLOP/s {(4.0*B*S*heads*d*4)/(ms*1e-3)/1e9:.0f} GB/s algorithmic", flush=True)
ops.set_attention_variant(0)
x = rng.standard_normal((M, 384), dtype=np.float32)
g = np.ones(384, np.float32); bb = np.zeros(384, np.float32)
y, ms = ops.layer_norm(x, g, bb, 1e-12, iters=50)
print(f"layernorm rows={M}: {ms:.4f} ms {2*M*384*4/(ms*1e-3)/1e9:.0f} GB/s", flush=True)
