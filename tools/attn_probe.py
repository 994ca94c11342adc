"""One attention kernel alone (for rocprofv3 --pmc passes).  Default: d = 64 at long sequence, 32 sentences x 2048 tokens x 12
heads (attention_kernel<64>); `python tools/attn_probe.py 4096 128 12 32` is the headline's shape (attention_pipe_kernel<32>)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from kjarni_amd import ops

a = [int(x) for x in sys.argv[1:]]
B, S, heads, d = (a + [32, 2048, 12, 64][len(a):])[:4]
rng = np.random.default_rng(0)
qkv = (rng.standard_normal((B, S, 3 * heads * d)) * 0.5).astype(np.float32)
mask = np.ones((B, S), np.uint32)
out, ms = ops.attention(qkv, mask, heads, iters=50)
fl = 4.0 * B * S * S * heads * d
print(f"attention d={d} B={B} S={S}: {ms:.3f} ms, {fl / ms / 1e9:.1f} TFLOP/s")
