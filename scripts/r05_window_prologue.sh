#!/bin/bash
# Round 5, VERDICT r04 item 4: what of the windowed headline launch's extra time (bench shape: 4096 frames per launch) is the
# prologue's table traffic -- the 32 KB of permuted weights and the DC table every workgroup fetches from L2 in front of its
# first frame -- and what is per-frame work (the 16 packed ops of the weighted first butterfly level, the DC add)?  Three
# measurement-only builds of the product library (wrong rows by design, built by this script into scripts/ab/, never shipped):
#   FSEA_WIN_ABL=1  the lane's weights are constants: no weight-table loads
#   FSEA_WIN_ABL=2  ... and the DC table is not fetched either (zeros go to LDS)
#   FSEA_WIN_ABL=3  ... and the per-frame DC add (two LDS reads, two packed adds per lane-frame) is left out
# run beside the product library in ONE process (scripts/ab_window.py: interleaved rounds after a common pre-warm).
# An in-prologue computation of cosine-sum weights can at best reach build 1's time.
# The -DFSEA_WIN_ABL hooks these builds need lived in fsea_fft_core.h from commit 1b31553 to 347805d and were taken out of the
# product header again once the measurement was recorded (profiles/r05_window_prologue.txt): check one of those commits out to
# repeat it.
# Usage: bash scripts/r05_window_prologue.sh build   (in the build container: cross-compiles the three libraries)
#        bash scripts/r05_window_prologue.sh run     (on the GPU box)
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
if [ "$1" = "build" ]; then
  for v in 1 2 3; do
    make -s -j8 -C frequensea_amd/csrc product OUT=$R/scripts/ab/libfsea_hip_winabl$v.so BUILD=$R/frequensea_amd/csrc/build_winabl$v EXTRA=-DFSEA_WIN_ABL=$v || exit 1
    rm -rf frequensea_amd/csrc/build_winabl$v
  done
  ls -la scripts/ab/libfsea_hip_winabl*.so
else
  mkdir -p gpurun_out
  for form in "200 15" "20 60" "200 15" "20 60"; do
    set -- $form
    AB_REGION=$1 AB_ROUNDS=$2 python scripts/ab_window.py scripts/ab/libfsea_hip_winabl1.so scripts/ab/libfsea_hip_winabl2.so scripts/ab/libfsea_hip_winabl3.so 2>&1 | grep -v amdgpu.ids
  done | tee gpurun_out/r05_window_prologue.txt
fi
