#!/bin/bash
# rocprofv3 kernel-trace stats of encoder calls of 32 x 128, 1 x 128 and 1 x 28 tokens (the sizes the reference's callers
# make: Indexer batch_size 32, one sentence to embed / classify) -> gpurun_out/call_size_b<B>_s<S>_kernel_stats.csv
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for bs in "32 128" "1 128" "1 28"; do
  set -- $bs
  b=$1; s=$2
  rm -rf gpurun_out/prof_b${b}_s$s
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b${b}_s$s -- python tools/mid_probe.py $b $s 200 > gpurun_out/prof_b${b}_s$s.log 2>&1
  find gpurun_out/prof_b${b}_s$s -name "*kernel_trace.csv" -delete
  grep "^batch" gpurun_out/prof_b${b}_s$s.log
  f=$(find gpurun_out/prof_b${b}_s$s -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/call_size_b${b}_s${s}_kernel_stats.csv
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].replace('kjarni::(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    print(f"  {n[:48]:48s} calls {r['Calls']:>6s} avg {float(r['AverageNs']) / 1000:8.2f} us")
PY
done
python tools/mid_probe.py 32 128 300; python tools/mid_probe.py 1 128 1000; python tools/mid_probe.py 1 28 1000; python tools/mid_probe.py 8 128 1000; python tools/mid_probe.py 64 128 300
