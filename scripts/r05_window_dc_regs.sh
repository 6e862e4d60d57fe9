for n in 8192 256 512 128; do
  AB_N=$n AB_REGION=200 AB_ROUNDS=15 python scripts/ab_window.py scripts/ab/libfsea_hip_r05a.so 2>&1 | grep -v amdgpu.ids
done
AB_N=8192 AB_REGION=20 AB_ROUNDS=60 python scripts/ab_window.py scripts/ab/libfsea_hip_r05a.so 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_window.py -x -q -p no:cacheprovider 2>&1 | tail -3
