"""Where a wave of the K-stream tile kernel spends its cycles (tuning build, gemm variant 54).
usage: KJARNI_FFI_LIB=.../libkjarni_ffi_tuning.so python tools/stream_cycles.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kjarni_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
rng = np.random.default_rng(0)
for name, K, N, epi in [("qkv", 384, 1152, ops.EPI_BIAS), ("fc1", 384, 1536, ops.EPI_BIAS_GELU)]:
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = (rng.standard_normal((N, K), dtype=np.float32) * 0.05).astype(np.float32)
    b = rng.standard_normal(N, dtype=np.float32)
    ops.set_gemm_variant(54)
    y, ms = ops.linear(x, w, b, None, epi, iters=3)
    ops.set_gemm_variant(0)
    c = y.reshape(-1)[:512 * 4].reshape(512, 4).astype(np.float64)
    tiles = c[:, 2]
    loop, edge = c[:, 0] / tiles, c[:, 1] / np.maximum(tiles - 1, 1)
    print(f"{name}: {ms:.4f} ms; per tile of wave 0 -- K-steps {loop.mean():8.0f} cycles (min {loop.min():.0f}, max {loop.max():.0f}; "
          f"12 K-steps x 64 MFMAs x 64 = 49152 matrix cycles, two waves per SIMD), edge {edge.mean():7.0f} (min {edge.min():.0f}, max {edge.max():.0f}); "
          f"tiles per workgroup {tiles.min():.0f}..{tiles.max():.0f}", flush=True)
