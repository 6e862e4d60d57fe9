#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
timeout 1500 python -m pytest tests -m gpu -q -k "tools or scene or nrf_fft" 2>&1 | grep -v "^Written\|^Composing\|^Frequency" | tail -30
for regime in resident; do
  timeout 300 python bench.py --workload broad --regime $regime --steps 20 --warmup 3 2>/dev/null | cut -c1-400
done
timeout 300 python bench.py --workload stft16384stream --steps 5 --warmup 2 2>/dev/null | cut -c1-400
