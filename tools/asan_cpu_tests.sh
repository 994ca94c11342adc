#!/bin/bash
# Host-side sanitizer run: builds the C++ host sources with AddressSanitizer + UBSan (device code untouched: GPU
# sanitizers are not available on this pool) into build/asan/libkjarni_ffi.so and runs the CPU test suite against it.
set -e
cd "$(dirname "$0")/../kjarni_amd/csrc"
mkdir -p ../../build/asan
make -s >/dev/null
for f in $(grep '^CPP_SRCS' Makefile | cut -d= -f2); do
  ( /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -fvisibility=hidden -I../../include -x hip --offload-arch=gfx950 \
      -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer \
      -c "$f" -o "../../build/asan/${f%.cpp}.o" ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fsanitize=undefined -shared-libsan -Wno-option-ignored \
  -o ../../build/asan/libkjarni_ffi.so ../../build/csrc/{gemm,gemm_flex,gemm_split,attention,rowops,cosine,whisper_kernels,llm_kernels}.o ../../build/asan/*.o \
  -Wl,-soname,libkjarni_ffi.so
cd ../..
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 KJARNI_FFI_LIB=$PWD/build/asan/libkjarni_ffi.so \
  python -m pytest tests -x -q -s -m "not gpu" -p no:cacheprovider "$@"
