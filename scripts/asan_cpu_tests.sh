#!/bin/bash
# Runs the CPU-tier tests that exercise the C99 host layer and the C oracle with both libraries
# built under AddressSanitizer + UndefinedBehaviorSanitizer (gcc).  The sanitized builds replace
# the in-tree .so files for the duration of the run and are put back afterwards.
# Usage: bash scripts/asan_cpu_tests.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
SAN="-O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer"
cp "$R/oracle/libfsea_oracle.so" "$T/oracle.orig"
cp "$R/frequensea_amd/libfsea_nrf.so" "$T/nrf.orig"
restore() { cp "$T/oracle.orig" "$R/oracle/libfsea_oracle.so"; cp "$T/nrf.orig" "$R/frequensea_amd/libfsea_nrf.so"; rm -rf "$T"; }
trap restore EXIT
gcc -std=c99 $SAN -shared -o "$R/oracle/libfsea_oracle.so" "$R/oracle/fsea_oracle.c" -lm -lpthread -ldl
gcc -std=c99 $SAN -I"$R/include" -shared -o "$R/frequensea_amd/libfsea_nrf.so" "$R"/frequensea_amd/host/*.c \
    -L"$R/frequensea_amd" -lfsea_hip -Wl,-rpath,"$R/frequensea_amd" -lm -lpthread -lz
cd "$R"
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD=$(gcc -print-file-name=libasan.so) \
    python -m pytest tests/test_oracle.py tests/test_host_api.py tests/test_ntt_font.py tests/test_reference_tools.py -x -q -p no:cacheprovider
