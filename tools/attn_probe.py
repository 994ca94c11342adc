"""The d = 64 attention kernel alone at long sequence (for rocprofv3 --pmc passes): 32 sentences x 2048 tokens x 12 heads."""
import sys
import numpy as np
sys.path.insert(0, ".")
from kjarni_amd import ops

B, S, heads, d = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 2048, 12, 64
rng = np.random.default_rng(0)
qkv = (rng.standard_normal((B, S, 3 * heads * d)) * 0.5).astype(np.float32)
mask = np.ones((B, S), np.uint32)
out, ms = ops.attention(qkv, mask, heads, iters=5)
fl = 4.0 * B * S * S * heads * d
print(f"attention d={d} B={B} S={S}: {ms:.3f} ms, {fl / ms / 1e9:.1f} TFLOP/s")
