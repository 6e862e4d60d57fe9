#!/bin/bash
# The kernel source itself (fsea_fft_core.h on the CPU shim of tests/emu) under AddressSanitizer + UBSan: LDS indexing,
# table and twiddle indexing, ragged units -- every size, mode, hop, tiled output and both frame distributions.
# The sanitized build is slow (round 2: 56 min with -g as one translation unit; round 3: four units in parallel, -g0: minutes);
# the in-tree libfsea_emu.so is put back afterwards.  Usage: bash scripts/asan_emu_kernels.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
python -c "from tests.emu_util import emu_lib; emu_lib()"          # make sure the normal build exists
T=$(mktemp -d)
cp tests/emu/libfsea_emu.so "$T/orig.so"
restore() { cp "$T/orig.so" "$R/tests/emu/libfsea_emu.so"; touch "$R/tests/emu/libfsea_emu.so"; rm -rf "$T"; }
trap restore EXIT
SAN="-std=c++20 -O1 -g0 -fPIC -pthread -DFSEA_TUNE -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-unknown-pragmas -Itests/emu -Ifrequensea_amd/csrc"
for u in emu_main emu_variants_a emu_variants_b emu_variants_c; do g++ $SAN -c tests/emu/$u.cpp -o "$T/$u.o" & done   # in parallel
wait
g++ -shared -pthread -fsanitize=address,undefined -o tests/emu/libfsea_emu.so "$T"/emu_main.o "$T"/emu_variants_a.o "$T"/emu_variants_b.o "$T"/emu_variants_c.o
touch tests/emu/libfsea_emu.so
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD=$(gcc -print-file-name=libasan.so) \
    python -m pytest tests/test_emu_kernels.py -x -q -p no:cacheprovider \
    -k "all_sizes or compile_time or runtime_dispatch or overlapped or tiled or static_unit or ragged or no_flip or single_wave or half_overlap or window or per_mode"
