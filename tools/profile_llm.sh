#!/bin/bash
# rocprofv3 kernel stats of the LLM decode measurement (tools/bench_more.py llm).
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof_l
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_l -- python tools/bench_more.py llm > gpurun_out/llm_prof.json 2>gpurun_out/prof_l.err
f=$(find gpurun_out/prof_l -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(r["Name"][:100].ljust(100), r["Calls"].rjust(7), f'{float(r["AverageNs"])/1e3:9.2f} us', r["Percentage"])
PY
