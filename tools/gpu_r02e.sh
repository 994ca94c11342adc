#!/bin/bash
# round 2, decode path: parity (LLM, chat, Whisper), tokens/s, replayed vs eager launches, per-kernel stats
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r02e
timeout 1500 python -m pytest tests/test_gpu_llm.py tests/test_gpu_chat.py tests/test_gpu_whisper.py -m gpu -q 2>&1 | tail -8 > gpurun_out/r02e/tests.log
cat gpurun_out/r02e/tests.log
T=kjarni_amd/lib/libkjarni_ffi_tuning.so
export KJARNI_BENCH_NO_CPU=1
for i in 1 2; do timeout 600 python tools/bench_more.py llm 2>gpurun_out/r02e/llm_new.err | cut -c1-140,560-700; done
for sp in 2 8; do KJARNI_FFI_LIB=$T KJARNI_HIP_LLM_SPLITS=$sp timeout 600 python tools/bench_more.py llm 2>gpurun_out/r02e/llm_sp.err | cut -c1-140,560-700; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02e/prof_new -o new -- python tools/bench_more.py llm > gpurun_out/r02e/prof_new.log 2>&1
for f in $(find gpurun_out/r02e/prof_new -name '*kernel_stats.csv'); do cp $f gpurun_out/r02e/new_kernel_stats.csv; done
rm -rf gpurun_out/r02e/prof_new
python - <<'PY'
import csv,re
for r in list(csv.DictReader(open("gpurun_out/r02e/new_kernel_stats.csv")))[:12]:
    nm=re.sub(r"kjarni::\(anonymous namespace\)::","",r["Name"]); nm=re.sub(r"\(.*","",nm)
    print(f"  {nm[:84]:84s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.2f}us {r['Percentage']}%")
PY
