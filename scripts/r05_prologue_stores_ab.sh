#!/bin/bash
# Round 5: the prologue's balancing stores (fsea_fft_core.h: balance_vmcnt) against the library without them
# (scripts/ab/libfsea_hip_r05b.so), one process each, repeated: selected (size, mode) pairs, the config-4 sweep + headline,
# the windowed headline in both region lengths.
for rep in 1 2 3; do
  timeout 900 python -u scripts/ab_modes.py scripts/ab/libfsea_hip_r05b.so -- 512 1024 4096 8192 2>&1 | grep -v amdgpu.ids
done
for rep in 1 2; do
  timeout 600 python -u scripts/ab_sweep.py scripts/ab/libfsea_hip_r05b.so 2>&1 | grep -v amdgpu.ids
  AB_N=8192 AB_REGION=200 AB_ROUNDS=15 python scripts/ab_window.py scripts/ab/libfsea_hip_r05b.so 2>&1 | grep -v amdgpu.ids
  AB_N=8192 AB_REGION=20 AB_ROUNDS=60 python scripts/ab_window.py scripts/ab/libfsea_hip_r05b.so 2>&1 | grep -v amdgpu.ids
done
