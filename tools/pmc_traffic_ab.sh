#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the headline kernels with an alternative library (KJARNI_FFI_LIB=$1) against the shipped one:
# bash tools/pmc_traffic_ab.sh kjarni_amd/lib_nt/libkjarni_ffi.so
set -u
alt=$1
mkdir -p gpurun_out/pmct
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for which in base alt; do
  if [ $which = alt ]; then export KJARNI_FFI_LIB=$PWD/$alt; else unset KJARNI_FFI_LIB; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmct/${which}_$c -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-profile --no-rerank-leg --no-parity-check --sentences 16384 > gpurun_out/pmct/${which}_$c.log 2>&1
  done
  mkdir -p gpurun_out/pmct/$which && rm -rf gpurun_out/pmct/$which/* && mv gpurun_out/pmct/${which}_FETCH_SIZE gpurun_out/pmct/$which/f && mv gpurun_out/pmct/${which}_WRITE_SIZE gpurun_out/pmct/$which/w
  python tools/pmc_to_traffic.py gpurun_out/pmct/$which gpurun_out/pmct/${which}_traffic.json > /dev/null
  python3 - gpurun_out/pmct/${which}_traffic.json $which <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["kernels"]
for k, v in d.items():
    if k.startswith(("gemm", "attention")):
        print(f"{sys.argv[2]:5s} {k:36s} fetch x2 {2 * v['fetch_size_kib_raw'] * 1024 / 1e9:6.3f} GB  write {v['write_size_kib'] * 1024 / 1e9:6.3f} GB  total {v['hbm_bytes_per_launch'] / 1e9:6.3f} GB")
PY
done
find gpurun_out/pmct -name "*.csv" -delete
