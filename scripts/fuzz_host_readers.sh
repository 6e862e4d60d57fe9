#!/bin/bash
# Fuzzes the host layer's two file readers under ASan + UBSan: the TrueType reader / rasteriser
# (frequensea_amd/host/ntt_font.c) with damaged copies of the fonts on this machine, and read_gray_png
# (frequensea_amd/host/easypng.c) with damaged PNGs.  Usage: bash scripts/fuzz_host_readers.sh [cases, default 2000]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
SAN="-std=c99 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -I$R/include"
gcc $SAN -o "$T/font" "$R/tests/fuzz/ntt_font_fuzz.c" "$R/frequensea_amd/host/ntt_font.c" -lm
gcc $SAN -o "$T/png" "$R/tests/fuzz/easypng_fuzz.c" "$R/frequensea_amd/host/easypng.c" -lz -lm
N=${1:-2000}
for font in /usr/share/fonts/truetype/dejavu/DejaVuSans.ttf /usr/share/fonts/truetype/dejavu/DejaVuSansMono-Bold.ttf \
            /root/reference/fonts/RobotoCondensed-Regular.ttf /root/reference/fonts/RobotoCondensed-Bold.ttf; do
  [ -f "$font" ] || continue
  for seed in 1 100001; do
    "$T/font" "$font" $seed $N "$T/case.ttf" 2>&1 | grep -v "^ERROR ntt_font" | tail -3
  done
done
"$T/png" 1 $((10 * N)) "$T/case.png" 2>&1 | tail -3
