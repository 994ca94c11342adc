#!/bin/bash
# rocprofv3 per-kernel table of a 2 048-token prompt (tools/prefill_probe.py)
set -u
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof_pf
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pf -- python tools/prefill_probe.py 2048 5 > gpurun_out/prof_pf.log 2>&1
f=$(find gpurun_out/prof_pf -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    n = r['Name'].replace('kjarni::(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    print(f"  {n[:52]:52s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1000:9.2f} us  share {float(r['TotalDurationNs']) / tot * 100:5.1f} %")
PY
find gpurun_out/prof_pf -name "*kernel_trace.csv" -delete
cp "$f" gpurun_out/prefill2048_kernel_stats.csv
