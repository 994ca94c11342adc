#!/bin/bash
# prefill with split-K: parity, then prompt times at 32 .. 2048 tokens and per-kernel stats at 128
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r02h
timeout 1500 python -m pytest tests/test_gpu_llm.py tests/test_gpu_chat.py -m gpu -q 2>&1 | tail -6 > gpurun_out/r02h/tests.log
cat gpurun_out/r02h/tests.log
for n in 32 128 512 2048; do timeout 300 python tools/prefill_probe.py $n 3 2>/dev/null | tail -1; done | tee gpurun_out/r02h/prefill.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02h/prof -o p -- python tools/prefill_probe.py 128 5 > gpurun_out/r02h/prof.log 2>&1
for f in $(find gpurun_out/r02h/prof -name '*kernel_stats.csv'); do cp $f gpurun_out/r02h/prefill128_kernel_stats.csv; done
rm -rf gpurun_out/r02h/prof
python - <<'PY'
import csv,re
for r in list(csv.DictReader(open("gpurun_out/r02h/prefill128_kernel_stats.csv")))[:12]:
    nm=re.sub(r"kjarni::\(anonymous namespace\)::","",r["Name"]); nm=re.sub(r"\(.*","",nm)
    print(f"  {nm[:70]:70s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.2f}us {r['Percentage']}%")
PY
