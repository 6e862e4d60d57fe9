#!/bin/bash
# The store-data hazard guard as a mechanism: a build of libfsea_hip.so WITHOUT the wait states behind
# the 16-byte buffer stores (-DFSEA_STORE_GUARD=0) must fail
#   (1) the disassembly test of tests/test_shipped_artifacts.py -- no GPU needed, and
#   (2) on a GPU: the soak run of the GPU tier (scripts/soak.py, 12 s) and the identical-launches test.
# Usage: bash scripts/store_guard_regression.sh [outfile]     (exit 0 = the unguarded build was caught everywhere it can be)
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/dev/stdout}
B=/tmp/fsea_unguarded
rm -rf $B; mkdir -p $B
make -s -j8 -C $R/frequensea_amd/csrc product BUILD=$B/build OUT=$B/libfsea_hip.so EXTRA=-DFSEA_STORE_GUARD=0 || exit 2
{
echo "# unguarded build: $B/libfsea_hip.so (-DFSEA_STORE_GUARD=0)"
cd $R
echo "== (1) disassembly test on the unguarded build (must FAIL)"
FSEA_ARTIFACT_LIB=$B/libfsea_hip.so python -m pytest tests/test_shipped_artifacts.py -q -k wait_states 2>&1 | tail -3
FSEA_ARTIFACT_LIB=$B/libfsea_hip.so python -m pytest tests/test_shipped_artifacts.py -q -k wait_states >/dev/null 2>&1 && { echo "NOT CAUGHT by the disassembly test"; exit 1; }
echo "caught by the disassembly test"
if python -c "import torch,sys; sys.exit(0 if torch.cuda.is_available() else 1)" 2>/dev/null; then
  echo "== (2) soak, 12 s, on the unguarded build (must FAIL)"
  caught=0
  for seed in 11 $RANDOM; do
    FSEA_HIP_LIB=$B/libfsea_hip.so timeout 300 python scripts/soak.py 12 $seed > $B/soak_$seed.log 2>&1 && echo "seed $seed: soak passed on the unguarded build" || { echo "seed $seed: soak FAILED on the unguarded build (as it must): $(grep -m1 -E 'AssertionError|assert' $B/soak_$seed.log | cut -c1-200)"; caught=1; }
  done
  echo "== (3) identical launches, unguarded build (must FAIL)"
  FSEA_HIP_LIB=$B/libfsea_hip.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k identical_launches 2>&1 | tail -3
  [ $caught = 1 ] || { echo "NOT CAUGHT by the soak"; exit 1; }
  echo "== the same two on the shipped (guarded) library (must PASS)"
  timeout 300 python scripts/soak.py 12 11 2>&1 | tail -1
fi
echo "store_guard_regression: OK"
} > $OUT 2>&1
