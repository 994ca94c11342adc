#!/bin/bash
# rocprofv3 per-kernel table of the many-query search (64 queries: selection inside the matrix-core scan) at 1e6 and 1e7 documents
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for n in ${SIZES:-1000000 10000000}; do
  rm -rf gpurun_out/prof_sm
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sm -- python tools/search_many_probe.py $n ${NQ:-64} ${K:-10} 20 > gpurun_out/prof_sm.log 2>&1
  grep "^n " gpurun_out/prof_sm.log
  f=$(find gpurun_out/prof_sm -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].replace('kjarni::(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    print(f"  {n[:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs']) / 1000:9.2f} us  total {float(r['TotalDurationNs']) / 1e6:9.3f} ms")
PY
done
rm -rf gpurun_out/prof_sm
for n in ${SIZES:-1000000 10000000}; do python tools/search_many_probe.py $n ${NQ:-64} ${K:-10} 20; done
