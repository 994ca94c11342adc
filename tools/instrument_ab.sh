#!/bin/bash
# Does bench.py's in-region instrumentation (per-step events, the one-wave clock trace beside the launch stream, the sysfs sampler)
# cost throughput?  The headline leg with and without --no-instrument, alternating processes, one box.
F="--steps 6 --warmup 3 --no-scan --no-cpu-baseline --no-extras --no-rerank-leg --no-models --no-parity-check"
for round in 1 2 3; do
  for extra in "" "--no-sensors" "--no-clock-trace" "--no-instrument" "--no-instrument --no-profile"; do
    python bench.py $F $extra 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$extra]'.ljust(34), d['value'], 'sentences/s', d['ms_per_step'], 'ms/step')"
  done
done
