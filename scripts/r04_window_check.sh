#!/bin/bash
# Round 4, first GPU visit: the windowed kernels' parity tier, the widened recorded-capture tier, and what the fused window costs.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4a; mkdir -p $O
cd $R; export TMPDIR=/tmp
echo "== pytest window + captures"
timeout 1200 python -m pytest tests/test_gpu_window.py -x -q 2>&1 | tail -8 | tee $O/pytest_window.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "every_recorded_capture or whole_block" 2>&1 | tail -5 | tee $O/pytest_captures.log
echo "== window rates"
timeout 900 python -u scripts/window_rate.py 2>&1 | tee $O/window_rate.txt
