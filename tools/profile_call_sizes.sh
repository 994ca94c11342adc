#!/bin/bash
# rocprofv3 kernel-trace stats of encoder calls of 32 x 128 and 1 x 128 tokens (the sizes the reference's callers make:
# Indexer batch_size 32, one sentence to embed / classify) -> gpurun_out/prof_b32, gpurun_out/prof_b1
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for b in 32 1; do
  rm -rf gpurun_out/prof_b$b
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b$b -- python tools/mid_probe.py $b 128 200 > gpurun_out/prof_b$b.log 2>&1
  find gpurun_out/prof_b$b -name "*kernel_trace.csv" -delete
  cat gpurun_out/prof_b$b.log | tail -1
  f=$(find gpurun_out/prof_b$b -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/call_size_b${b}_kernel_stats.csv
  cut -d, -f1-4 "$f" | head -14
done
python tools/mid_probe.py 32 128 300; python tools/mid_probe.py 1 128 1000; python tools/mid_probe.py 8 128 1000; python tools/mid_probe.py 64 128 300
