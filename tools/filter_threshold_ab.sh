#!/bin/bash
# Where the bf16 filter route of the many-query search starts to pay (tuning build): 64 queries x n documents, both routes.
export KJARNI_FFI_LIB=$PWD/kjarni_amd/lib/libkjarni_ffi_tuning.so
for n in ${SIZES:-20000 50000 100000 200000 400000}; do
  echo -n "two-call form  : "; KJARNI_HIP_FILTER_MIN_DOCS=1000000000 python tools/search_many_probe.py $n 64 10 100 2>&1 | grep "^n "
  echo -n "filter route   : "; KJARNI_HIP_FILTER_MIN_DOCS=0 python tools/search_many_probe.py $n 64 10 100 2>&1 | grep "^n "
done
