#!/bin/bash
# AddressSanitizer + UBSan over the C++ host library on the CPU-only tests (parsers, state space, dense transitions,
# error paths).  Builds an instrumented liblinearham_host.so in place, runs the tests, restores the normal build.
# (GPU AddressSanitizer is not available on the pool; the HIP side is covered by the parity tests.)
set -e
cd "$(dirname "$0")/.."
hostdir=linearham_amd/csrc/host
lib=linearham_amd/lib/liblinearham_host.so
cp $lib /tmp/liblinearham_host.backup.so
trap 'cp /tmp/liblinearham_host.backup.so $lib' EXIT
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -fPIC -shared -pthread -I include -I $hostdir \
  $(ls $hostdir/*.cpp | grep -v _main.cpp) -o $lib -L linearham_amd/lib -llinearham_hip -Wl,-rpath,'$ORIGIN'
# libstdc++ next to libasan: the interpreter does not link it, and ASan's __cxa_throw interceptor needs the real one
LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libstdc++.so.6)" ASAN_OPTIONS=detect_leaks=0 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python -m pytest tests/test_host_goldens.py tests/test_structured_cpu.py -q -m "not gpu"
