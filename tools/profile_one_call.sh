#!/bin/bash
# rocprofv3 per-kernel table of one sentence of $1 tokens (default 28) through the encoder (tools/mid_probe.py 1 TOKENS)
set -u
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof_x
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_x -- python tools/mid_probe.py 1 ${1:-28} 500 > gpurun_out/prof_x.log 2>&1
grep "^batch" gpurun_out/prof_x.log
f=$(find gpurun_out/prof_x -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].replace('kjarni::(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    print(f"  {n[:48]:48s} calls {r['Calls']:>6s} avg {float(r['AverageNs']) / 1000:8.2f} us")
    tot += float(r['TotalDurationNs'])
print('total kernel us per call', tot / 501 / 1000)
PY
t=$(find gpurun_out/prof_x -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:]) for r in csv.DictReader(open(sys.argv[1]))))
# one call in the middle: 46 kernels
k = len(rows) // 2
seg = rows[k:k + 60]
for i in range(1, len(seg)):
    print(f"{seg[i][2]:42s} dur {(seg[i][1] - seg[i][0]) / 1000:6.2f} gap_before {(seg[i][0] - seg[i - 1][1]) / 1000:6.2f}")
PY
find gpurun_out/prof_x -name "*kernel_trace.csv" -delete
