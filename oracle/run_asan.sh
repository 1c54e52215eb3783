#!/bin/bash
# The CPU suite's host logic + oracle under AddressSanitizer / UBSan (sanitizers belong on the CPU build: the GPU pool
# refuses them).  Builds oracle/_build/libhamsm_oracle_asan.so from the same source and runs the CPU tests against it.
set -e
cd "$(dirname "$0")/.."
make -s -C oracle asan
export HAMSM_ORACLE_SO="$PWD/oracle/_build/libhamsm_oracle_asan.so"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1"
export UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"
if [ "$1" = "--gpu-logic" ]; then
  # the GPU suite's host logic (fixtures, wrappers, buffer sizes) with the oracle standing in for the HIP library; tests
  # that need HIP-only hooks fail with AttributeError -- what matters here is that the sanitizer stays silent
  shift
  export HAMSM_HIP_STANDIN="$HAMSM_ORACLE_SO"
  exec python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_fullsize.py "$@"
fi
exec python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@" \
  --deselect tests/test_bench_contract.py --deselect tests/test_distributed_gloo.py
