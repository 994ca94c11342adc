#!/bin/bash
# measurement: are the streaming kernels faster when their weights sit in the memory-side cache?
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r02f
export KJARNI_BENCH_NO_CPU=1 KJARNI_FFI_LIB=kjarni_amd/lib/libkjarni_ffi_tuning.so KJARNI_HIP_LLM_TOUCH=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02f/prof -o t -- python tools/bench_more.py llm > gpurun_out/r02f/prof.log 2>&1
for f in $(find gpurun_out/r02f/prof -name '*kernel_stats.csv'); do cp $f gpurun_out/r02f/touch_kernel_stats.csv; done
rm -rf gpurun_out/r02f/prof
python - <<'PY'
import csv,re
for r in list(csv.DictReader(open("gpurun_out/r02f/touch_kernel_stats.csv")))[:10]:
    nm=re.sub(r"kjarni::\(anonymous namespace\)::","",r["Name"]); nm=re.sub(r"\(.*","",nm)
    print(f"  {nm[:84]:84s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.2f}us {r['Percentage']}%")
PY
