#!/bin/bash
# After the product / tuning split: the whole GPU tier, the tuning library's variants against numpy, smoke.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4d; mkdir -p $O
cd $R; export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.log
echo "== tuning variants (rel OK = rows equal numpy's)"
TUNE_SETS=3 timeout 900 python scripts/tune.py 8192 4096 2>&1 | grep -v amdgpu.ids | tee $O/tune_variants.txt | awk '{print $1, $2, $3, $(NF-1), $NF}' | head -60
