#!/bin/bash
mkdir -p gpurun_out
for c in 32768 65536 131072 262144; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --chunk-tokens $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk_tokens', $c, d['value'], 'sent/s', d['e2e_frac_fp32_mfma_peak'])"
done | tee gpurun_out/chunk_sweep.log
