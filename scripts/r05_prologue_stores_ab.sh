#!/bin/bash
# Round 5: a library before a change to the kernels' prologue (scripts/ab/libfsea_hip_r05c.so: ticket request behind unit 0's loads)
# against the product (ticket request in front), one process each, repeated: selected (size, mode) pairs in long launches
# (dynamic unit distribution), the config-4 sweep + headline.
for rep in 1 2 3; do
  timeout 900 python -u scripts/ab_modes.py scripts/ab/libfsea_hip_r05c.so -- 1024 4096 8192 2>&1 | grep -v amdgpu.ids
done
for rep in 1 2; do
  timeout 600 python -u scripts/ab_sweep.py scripts/ab/libfsea_hip_r05c.so 2>&1 | grep -v amdgpu.ids
done
