#!/bin/bash
# CPU-only sanitizer run (GPU AddressSanitizer is not available on this pool): the host C code
# (sp_host.c, mc64.c, fiedler.c, awbm.c, awbm_dist.c, rcm.c, matio.c, idx32.c) built with -fsanitize=address,undefined and driven by the CPU
# test-suite, plus the oracle built the same way.  Leak checking is off: the interpreter itself never frees everything.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
make -s -C "$ROOT/spike-petsc_amd/csrc" asan
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fopenmp -fPIC -shared -o "$ROOT/oracle/liboracle_asan.so" "$ROOT/oracle/spike_oracle.c" -lm
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export SPIKE_HOST_LIB="$ROOT/spike-petsc_amd/libspike_petsc_host_asan.so" SPIKE_ORACLE_LIB="$ROOT/oracle/liboracle_asan.so"
cd "$ROOT" && python -m pytest tests/test_host_cpu.py tests/test_awbm_dist_cpu.py tests/test_mc64_oracle.py tests/test_golden.py tests/test_oracle.py -x -q -m "not gpu" "$@"
