#!/bin/bash
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os; print('affinity', len(os.sched_getaffinity(0)))"
( time python -m pytest tests -m gpu -q -x 2>&1 | tail -25 ) 2>&1 | tee gpurun_out/pytest_gpu.log
KJARNI_FFI_LIB=$PWD/kjarni_amd/lib/libkjarni_ffi_tuning.so python tools/gemm_probe.py 2>&1 | tee gpurun_out/gemm_probe.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/bench.err | tee gpurun_out/bench.json
tail -3 gpurun_out/bench.err
python - <<'PY' 2>&1 | tee gpurun_out/cpu_threads.log
import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
from oracle import cpu_baseline as CB
from tests import synth
with tempfile.TemporaryDirectory() as d:
    cfg, t = synth.minilm_embedder(d, seed=0)
m = CB.BaselineModel(t, cfg, 256, 128)
ids, mask = synth.synthetic_ids(256, 128, seed=0)
for th in (8, 16, 32, 64, 128, 256):
    CB.lib().kb_set_num_threads(th)
    for B in (32, 256):
        m.embed_batch(ids[:B], mask[:B], True)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 2.0:
            m.embed_batch(ids[:B], mask[:B], True); n += B
        print(th, B, round(n / (time.perf_counter() - t0), 1), flush=True)
PY
