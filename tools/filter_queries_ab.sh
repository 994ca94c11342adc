#!/bin/bash
# From how many queries on the bf16 filter route of the search beats the streaming passes (4 queries per pass) + selection
# (tuning build): nq queries x n documents, both routes.
export KJARNI_FFI_LIB=$PWD/kjarni_amd/lib/libkjarni_ffi_tuning.so
for n in ${SIZES:-100000 1000000}; do for q in ${QUERIES:-2 4 5 8 12 16 19}; do
  echo -n "streaming passes: "; KJARNI_HIP_FILTER_MIN_QUERIES=1000 python tools/search_many_probe.py $n $q 10 50 2>&1 | grep "^n "
  echo -n "filter route    : "; KJARNI_HIP_FILTER_MIN_QUERIES=2 python tools/search_many_probe.py $n $q 10 50 2>&1 | grep "^n "
done; done
