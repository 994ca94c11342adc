#!/bin/bash
# mid-size GEMM route: op parity, full suite, call-size sweep
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r02i
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q -x 2>&1 | tail -12 > gpurun_out/r02i/ops.log; cat gpurun_out/r02i/ops.log
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r02i/tests.log; cat gpurun_out/r02i/tests.log
timeout 900 python tools/bench_more.py sweep 2>gpurun_out/r02i/sweep.err > gpurun_out/r02i/sweep.jsonl
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02i/sweep.jsonl").read().strip().split("\n")[-1])
for r in d["calls"]: print(r)
PY
