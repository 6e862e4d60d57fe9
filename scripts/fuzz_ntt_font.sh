#!/bin/bash
# Fuzzes the TrueType reader / rasteriser (frequensea_amd/host/ntt_font.c) under ASan + UBSan with damaged copies of
# the fonts on this machine.  Usage: bash scripts/fuzz_ntt_font.sh [cases per font, default 2000]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
gcc -std=c99 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -I"$R/include" \
    -o "$T/fuzz" "$R/tests/fuzz/ntt_font_fuzz.c" "$R/frequensea_amd/host/ntt_font.c" -lm
N=${1:-2000}
for font in /usr/share/fonts/truetype/dejavu/DejaVuSans.ttf /usr/share/fonts/truetype/dejavu/DejaVuSansMono-Bold.ttf \
            /root/reference/fonts/RobotoCondensed-Regular.ttf /root/reference/fonts/RobotoCondensed-Bold.ttf; do
  [ -f "$font" ] || continue
  for seed in 1 100001; do
    "$T/fuzz" "$font" $seed $N "$T/case.ttf" 2>&1 | grep -v "^ERROR ntt_font" | tail -3
  done
done
