#!/bin/bash
# the sweep tools at the reference's own constants, on a GPU box (through gpurun):
#   broad   471 captures x 256-pt x 4096 rows -> fsea-fft-batch --broad -> 471 PNGs -> fsea-fft-stitch --broad AND the
#           reference's own c/fft-stitch-broad.c (oracle/_ref/fft-stitch-broad): the two 120576 x 4096 images bit for bit
#   narrow  300 captures x 1024-pt x 16384 rows -> fsea-fft-batch -> fsea-fft-stitch --rows 11211 --footer 600
#           (c/fft-stitch.c's layout, 154112 x 11811, overlapping max) against the oracle's composite
# The test file does the work and writes each stage's wall time; this script runs it and leaves the records where
# profiles/ takes them from:  gpurun_out/r06_reference_geometry_{broad,narrow}.json
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_reference_geometry.py -q -m gpu -s --durations=5 2>&1 | tee gpurun_out/r06_reference_geometry.log
for f in gpurun_out/r06_reference_geometry_*.json; do echo "== $f"; cat "$f"; done
