#!/bin/bash
# CPU-side sanitizer run of the C restatement (oracle/ssmq_oracle.c): AddressSanitizer + UndefinedBehaviorSanitizer build
# (oracle/Makefile: asan), tests/test_oracle_c.py on it.  CPU only - GPU sanitizers are not available on this pool.
# python itself is not instrumented, so the ASan runtime is preloaded and leak checking (python's own arenas) is off.
set -e
cd "$(dirname "$0")/.."
make -C oracle asan
ASAN_LIB=$(gcc -print-file-name=libasan.so)
UBSAN_LIB=$(gcc -print-file-name=libubsan.so)
LD_PRELOAD="$ASAN_LIB:$UBSAN_LIB" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    SSMQ_ORACLE_LIB="$PWD/oracle/libssmq_oracle_asan.so" OMP_NUM_THREADS=4 \
    python -m pytest tests/test_oracle_c.py -x -q -p no:cacheprovider "$@"
