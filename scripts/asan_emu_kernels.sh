#!/bin/bash
# The kernel source itself (fsea_fft_core.h on the CPU shim of tests/emu) under AddressSanitizer + UBSan: LDS indexing,
# table and twiddle indexing, ragged units -- every size, mode, hop, tiled output and both frame distributions.
# The sanitized build of the one big translation unit is slow (tens of minutes; round 2: 56 min with -g, 78 tests green);
# the in-tree libfsea_emu.so is put back afterwards.  Usage: bash scripts/asan_emu_kernels.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
python -c "from tests.emu_util import emu_lib; emu_lib()"          # make sure the normal build exists
T=$(mktemp -d)
cp tests/emu/libfsea_emu.so "$T/orig.so"
restore() { cp "$T/orig.so" "$R/tests/emu/libfsea_emu.so"; touch "$R/tests/emu/libfsea_emu.so"; rm -rf "$T"; }
trap restore EXIT
g++ -std=c++20 -O1 -g0 -fPIC -shared -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-unknown-pragmas \
    -Itests/emu -Ifrequensea_amd/csrc tests/emu/emu_main.cpp tests/emu/emu_variants_a.cpp tests/emu/emu_variants_b.cpp tests/emu/emu_variants_c.cpp -o tests/emu/libfsea_emu.so
touch tests/emu/libfsea_emu.so
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD=$(gcc -print-file-name=libasan.so) \
    python -m pytest tests/test_emu_kernels.py -x -q -p no:cacheprovider \
    -k "all_sizes or compile_time or runtime_dispatch or overlapped or tiled or static_unit or ragged or no_flip"
