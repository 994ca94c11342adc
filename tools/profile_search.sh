#!/bin/bash
# rocprofv3 per-kernel table of the fused one-query search at a few corpus sizes (tools/search_probe.py)
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for n in 100000 200000 1000000; do
  rm -rf gpurun_out/prof_s
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_s -- python tools/search_probe.py $n ${K:-10} 200 > gpurun_out/prof_s.log 2>&1
  grep "^n " gpurun_out/prof_s.log
  f=$(find gpurun_out/prof_s -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].replace('kjarni::(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if 'cosine' in n or 'topk' in n:
        print(f"  {n[:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs']) / 1000:8.2f} us")
PY
done
rm -rf gpurun_out/prof_s
python tools/search_probe.py 100000; python tools/search_probe.py 200000; python tools/search_probe.py 1000000; python tools/search_probe.py 200000 60
