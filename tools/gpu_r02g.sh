#!/bin/bash
# full GPU suite + decode / chat / whisper measurements after the decode-path changes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r02g
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r02g/tests.log
cat gpurun_out/r02g/tests.log
timeout 900 python tools/bench_more.py llm chat whisper > gpurun_out/r02g/bench_more.jsonl 2> gpurun_out/r02g/bench_more.err
cut -c1-420 gpurun_out/r02g/bench_more.jsonl
