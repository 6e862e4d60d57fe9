#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
/opt/rocm/bin/rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head -2
timeout 600 python scripts/energy_probe.py - r1 nd v2 abl_v2l abl_io abl_nolds abl_noflop abl_nostore abl_noload 2>&1 | tee $OUT/r02_energy_per_launch.txt
ENERGY_CONST_INPUT=1 ENERGY_SECONDS=2.5 timeout 600 python scripts/energy_probe.py - r1 abl_io 2>&1 | tee -a $OUT/r02_energy_per_launch.txt
ENERGY_N=16384 ENERGY_SECONDS=2.5 timeout 300 python scripts/energy_probe.py - r1 2>&1 | tee -a $OUT/r02_energy_per_launch.txt
ENERGY_N=1024 ENERGY_SECONDS=2.5 timeout 300 python scripts/energy_probe.py - r1 2>&1 | tee -a $OUT/r02_energy_per_launch.txt
