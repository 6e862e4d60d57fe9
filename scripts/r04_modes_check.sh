#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c; mkdir -p $O
cd $R; export TMPDIR=/tmp
echo "== per-mode configuration tests + window tests"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_window.py -x -q -k "per_mode or window or compile_time" 2>&1 | tail -5 | tee $O/pytest.log
echo "== every (size, mode): round 2 / round 3 / current in one process"
timeout 900 python -u scripts/ab_modes.py scripts/ab/libfsea_hip_r02.so scripts/ab/libfsea_hip_r03.so 2>&1 | grep -v amdgpu.ids | tee $O/mode_rates.txt
