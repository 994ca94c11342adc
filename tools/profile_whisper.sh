#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_w -- python tools/bench_more.py whisper > gpurun_out/whisper_prof.json 2>gpurun_out/prof_w.err
find gpurun_out/prof_w -name "*kernel_trace.csv" -delete   # gpurun_out travels back (64 MiB limit): keep the summaries only
f=$(find gpurun_out/prof_w -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(7), f'{float(r["AverageNs"])/1e3:9.2f} us', r["Percentage"])
PY
