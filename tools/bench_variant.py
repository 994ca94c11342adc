"""bench.py with a tuning-build kernel variant selected first (same-box A/B of a dispatch decision on the whole step).
usage: KJARNI_FFI_LIB=.../libkjarni_ffi_tuning.so python tools/bench_variant.py <gemm variant> [bench.py flags]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()  # (before the library touches the device: torch.cuda.is_available() is what bench.py asks first)
from kjarni_amd import ops
ops.set_gemm_variant(int(sys.argv[1]))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
