#!/bin/bash
# The oracle (the parity checker) under AddressSanitizer + UndefinedBehaviorSanitizer: every CPU test that drives it, with liboracle_san.so in liboracle.so's place.
#   bash tools/oracle_sanitized_tests.sh [pytest args]        (CPU only; ~10 x slower than the plain suite)
set -e
cd "$(dirname "$0")/.."
make -C oracle liboracle_san.so
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
export PLP_ORACLE_SO=liboracle_san.so LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:allocator_may_return_null=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
exec python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"
