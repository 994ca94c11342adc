"""GEMM probes on the GPU box (HIP-event timed through kjarni_hip_op_*; >= 0.3 s per measurement so the clock settles).
  1. K sweep of the plain projection GEMM: where does the main loop settle once prologue / epilogue are amortised?
  2. residual projection + LayerNorm: the fused kernel against GEMM + LayerNorm (needs the tuning build for the A/B:
     KJARNI_FFI_LIB=kjarni_amd/lib/libkjarni_ffi_tuning.so)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import torch  # noqa: F401  (HIP runtime first, as in bench.py)
from kjarni_amd import ops

M = int(os.environ.get("KB_M", 131072))
rng = np.random.default_rng(0)
PEAK = 157.3


def iters_for(flops):
    return max(10, int(0.35 / (flops / 120e12)))


def prio():
    """s_setprio 2 around the K-loop (default) against none (tuning variant 5), plain and GELU epilogues."""
    if not ops.has_tuning():
        return
    for name, k, n, epi in (("qkv ", 384, 1152, ops.EPI_BIAS), ("fc1 ", 384, 1536, ops.EPI_BIAS_GELU), ("k1536", 1536, 384, ops.EPI_BIAS)):
        x = rng.standard_normal((M, k), dtype=np.float32)
        w = (rng.standard_normal((n, k), dtype=np.float32) * 0.05).astype(np.float32)
        b = rng.standard_normal(n, dtype=np.float32)
        fl = 2.0 * M * n * k
        for _ in range(2):
            for variant in (0, 5):
                ops.set_gemm_variant(variant)
                _, ms = ops.linear(x, w, b, None, epi, iters=iters_for(fl))
                tf = fl / (ms * 1e-3) / 1e12
                print(f"gemm {name} epi={epi} {'setprio' if variant == 0 else 'no prio'} {ms:8.4f} ms {tf:7.2f} TFLOP/s ({tf / PEAK * 100:5.1f}% peak)", flush=True)
    ops.set_gemm_variant(0)


def persist():
    """Persistent tile loops (default) against one workgroup per tile (tuning variant 10): the four shapes of the headline."""
    if not ops.has_tuning():
        return
    for name, k, n, epi, ln in (("qkv  ", 384, 1152, ops.EPI_BIAS, False), ("fc1  ", 384, 1536, ops.EPI_BIAS_GELU, False),
                                ("out+ln", 384, 384, None, True), ("fc2+ln", 1536, 384, None, True)):
        x = rng.standard_normal((M, k), dtype=np.float32)
        w = (rng.standard_normal((n, k), dtype=np.float32) * 0.05).astype(np.float32)
        b = rng.standard_normal(n, dtype=np.float32)
        r = rng.standard_normal((M, n), dtype=np.float32) if ln else None
        g = np.ones(n, np.float32)
        fl = 2.0 * M * n * k
        for _ in range(2):
            for variant in (10, 0):
                ops.set_gemm_variant(variant)
                if ln:
                    _, ms = ops.linear_layer_norm(x, w, b, r, g, b, 1e-12, iters=iters_for(fl))
                else:
                    _, ms = ops.linear(x, w, b, None, epi, iters=iters_for(fl))
                tf = fl / (ms * 1e-3) / 1e12
                print(f"gemm {name} {'persistent' if variant == 0 else 'tile per WG'} {ms:8.4f} ms {tf:7.2f} TFLOP/s ({tf / PEAK * 100:5.1f}% peak)", flush=True)
    ops.set_gemm_variant(0)


def lnparams():
    """Fused LayerNorm GEMM: bias / gamma / beta from LDS (default) against global loads in the epilogue (tuning variant 14)."""
    if not ops.has_tuning():
        return
    m, n = 262144, 384
    for name, k in (("out+ln", 384), ("fc2+ln", 1536)):
        x = rng.standard_normal((m, k), dtype=np.float32)
        w = (rng.standard_normal((n, k), dtype=np.float32) * 0.05).astype(np.float32)
        b = rng.standard_normal(n, dtype=np.float32)
        r = rng.standard_normal((m, n), dtype=np.float32)
        g = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32)
        beta = (0.1 * rng.standard_normal(n)).astype(np.float32)
        fl = 2.0 * m * n * k
        ref = None
        for _ in range(3):
            for variant in (0, 14):
                ops.set_gemm_variant(variant)
                y, ms = ops.linear_layer_norm(x, w, b, r, g, beta, 1e-12, iters=iters_for(fl))
                if ref is None:
                    ref = y
                tf = fl / (ms * 1e-3) / 1e12
                print(f"{name} {'params in LDS   ' if variant == 0 else 'params from global'} {ms:8.4f} ms {tf:7.2f} TFLOP/s "
                      f"({tf / PEAK * 100:5.1f}% peak) max diff vs first {float(np.abs(y - ref).max()):.1e}", flush=True)
    ops.set_gemm_variant(0)


def attn(d=32):
    B, S, heads = 1024, 128, 12
    qkv = rng.standard_normal((B, S, 3 * heads * d), dtype=np.float32)
    for ragged in (False, True):
        mask = np.ones((B, S), np.uint32)
        if ragged:
            for b in range(B):
                mask[b, rng.integers(16, S + 1):] = 0
        names = {0: "kernel", 11: "no softmax", 12: "no LDS staging", 13: "no output stores", 14: "no prefetch", 15: "memory only", 20: "2 WG per CU", 16: "nt loads"}
        for _ in range(2):
            for variant in ((0, 20, 16, 11, 13, 14, 15) if ops.has_tuning() and not ragged and d == 32 else (0,)):
                if ops.has_tuning():
                    ops.set_attention_variant(variant)
                ctx, ms = ops.attention(qkv, mask, heads, iters=1500)
                fl = 4.0 * B * S * S * heads * d
                print(f"attention B={B} S={S} h={heads} d={d} ragged={ragged} {names[variant]:18s}: {ms:.4f} ms "
                      f"{fl / (ms * 1e-3) / 1e12:.2f} TFLOP/s {(4.0 * B * S * heads * d * 4) / (ms * 1e-3) / 1e9:.0f} GB/s algorithmic",
                      flush=True)
        if ops.has_tuning():
            ops.set_attention_variant(0)


def sweep():
    for n, ks in ((384, (384, 1536, 6144)), (1536, (384, 1536))):
        for k in ks:
            m = M if k * M * 4 < 3.3e9 else M // 2
            x = rng.standard_normal((m, k), dtype=np.float32)
            w = (rng.standard_normal((n, k), dtype=np.float32) * 0.05).astype(np.float32)
            b = rng.standard_normal(n, dtype=np.float32)
            fl = 2.0 * m * n * k
            for _ in range(2):
                _, ms = ops.linear(x, w, b, None, ops.EPI_BIAS, iters=iters_for(fl))
                tf = fl / (ms * 1e-3) / 1e12
                print(f"gemm bias      M={m} N={n:5d} K={k:5d} {ms:8.4f} ms {tf:7.2f} TFLOP/s ({tf / PEAK * 100:5.1f}% peak)", flush=True)


def fused():
    for name, k in (("out_proj", 384), ("fc2", 1536)):
        n = 384
        x = rng.standard_normal((M, k), dtype=np.float32)
        w = (rng.standard_normal((n, k), dtype=np.float32) * 0.05).astype(np.float32)
        b = rng.standard_normal(n, dtype=np.float32)
        r = rng.standard_normal((M, n), dtype=np.float32)
        g = np.ones(n, np.float32)
        beta = np.zeros(n, np.float32)
        fl = 2.0 * M * n * k
        it = iters_for(fl)
        for rnd in range(2):
            for variant in ((0, 4) if ops.has_tuning() else (0,)):
                if ops.has_tuning():
                    ops.set_gemm_variant(variant)
                y, ms = ops.linear_layer_norm(x, w, b, r, g, beta, 1e-12, iters=it)
                tf = fl / (ms * 1e-3) / 1e12
                tag = "fused           " if variant == 0 else "gemm+layernorm  "
                print(f"{name:8s} {tag} M={M} K={k:5d} {ms:8.4f} ms {tf:7.2f} TFLOP/s ({tf / PEAK * 100:5.1f}% peak) "
                      f"row0 mean {float(y[0].mean()):+.2e} var {float(y[0].var()):.4f}", flush=True)
        if ops.has_tuning():
            ops.set_gemm_variant(0)


if __name__ == "__main__":
    what = sys.argv[1:] or ["fused", "sweep"]
    if "attn" in what:
        attn()
    if "attn64" in what:
        attn(64)
    if "lnparams" in what:
        lnparams()
    if "prio" in what:
        prio()
    if "persist" in what:
        persist()
    if "fused" in what:
        fused()
    if "sweep" in what:
        sweep()
