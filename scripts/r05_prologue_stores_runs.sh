for rep in 1 2; do
AB_N=16384 AB_HOP=8192 AB_REGION=100 AB_ROUNDS=15 python scripts/ab_window.py scripts/ab/libfsea_hip_runs.so 2>&1 | grep -v amdgpu.ids
AB_N=8192 AB_HOP=4096 AB_REGION=100 AB_ROUNDS=15 python scripts/ab_window.py scripts/ab/libfsea_hip_runs.so 2>&1 | grep -v amdgpu.ids
AB_N=16384 AB_REGION=100 AB_ROUNDS=15 python scripts/ab_window.py scripts/ab/libfsea_hip_runs.so 2>&1 | grep -v amdgpu.ids
AB_N=4096 AB_REGION=100 AB_ROUNDS=15 python scripts/ab_window.py scripts/ab/libfsea_hip_runs.so 2>&1 | grep -v amdgpu.ids
AB_N=2048 AB_REGION=100 AB_ROUNDS=15 python scripts/ab_window.py scripts/ab/libfsea_hip_runs.so 2>&1 | grep -v amdgpu.ids
done
