#!/bin/bash
# Native threads on one encoder handle (tools/lab/threads_bench.cpp): calls/s of single-sentence embeds with and without call combining.
set -u
d=$(mktemp -d)
python - "$d" <<'PY'
import sys
sys.path.insert(0, ".")
from tests import synth
synth.minilm_embedder(sys.argv[1] + "/m", seed=0)
PY
[ -x tools/lab/threads_bench ] || g++ -O2 -std=c++17 -pthread -o tools/lab/threads_bench tools/lab/threads_bench.cpp -ldl
tools/lab/threads_bench kjarni_amd/lib/libkjarni_ffi.so "$d/m" ${1:-28} ${2:-400}
rm -rf "$d"
