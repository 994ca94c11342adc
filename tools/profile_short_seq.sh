#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for bs in "256 32" "128 32" "512 16" "64 64"; do
  set -- $bs
  b=$1; s=$2
  rm -rf gpurun_out/prof_q
  KJARNI_HIP_TWO_LANES=${LANES:-1} rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -- python tools/mid_probe.py $b $s 200 > gpurun_out/prof_q.log 2>&1
  grep "^batch" gpurun_out/prof_q.log
  f=$(find gpurun_out/prof_q -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].replace('kjarni::(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    print(f"  {n[:48]:48s} calls {r['Calls']:>6s} avg {float(r['AverageNs']) / 1000:8.2f} us  {r['Percentage']}%")
PY
  python tools/mid_probe.py $b $s 500 | grep ^batch
  KJARNI_HIP_TWO_LANES=0 python tools/mid_probe.py $b $s 500 | grep ^batch
done
