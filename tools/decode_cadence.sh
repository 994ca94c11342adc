#!/bin/bash
# What every launch of a decode step costs IN the chain: rocprofv3 kernel trace of tools/decode_probe.py, then per kernel name the
# mean of (its end - the previous kernel's end) over the steady decode steps -- the profiler's own intervals overlap, the ends do not.
# usage: bash tools/decode_cadence.sh whisper|llm
set -u
which=${1:-whisper}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof_c
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_c -- python tools/decode_probe.py $which ${2:-2} > gpurun_out/prof_c.log 2>&1
f=$(find gpurun_out/prof_c -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
gap = collections.defaultdict(list)
dur = collections.defaultdict(list)
prev_end = None
for s, e, n in rows:
    n = n.replace("void kjarni::(anonymous namespace)::", "").replace("kjarni::(anonymous namespace)::", "").split("(")[0]
    if prev_end is not None and 0 < e - prev_end < 200000:
        gap[n].append(e - prev_end)
        dur[n].append(e - s)
    prev_end = e
tot = sum(sum(v) for v in gap.values())
print(f"{'kernel':70s} {'calls':>8s} {'end-to-end us':>14s} {'interval us':>12s} {'share':>7s}")
for n, v in sorted(gap.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 50:
        continue
    print(f"{n[:70]:70s} {len(v):8d} {sum(v) / len(v) / 1e3:14.2f} {sum(dur[n]) / len(dur[n]) / 1e3:12.2f} {100.0 * sum(v) / tot:6.1f}%")
PY
rm -rf gpurun_out/prof_c
