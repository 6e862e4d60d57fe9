#!/bin/bash
# Round 5: the static unit distribution of short launches, interleaved (unit = workgroup + k * grid, the product) against
# contiguous (workgroup b takes units [b * per, (b + 1) * per): scripts/ab/libfsea_hip_contig.so, -DFSEA_STATIC_CONTIG=1),
# in the bench's launch shape (64 MiB of samples per launch, regions of 200 and of 20 launches).
mkdir -p gpurun_out
{
  python - <<'PY'
import ctypes, numpy as np, torch
vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
outs = []
for path in ("scripts/ab/libfsea_hip_contig.so", "frequensea_amd/libfsea_hip.so"):
    L = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    L.fsea_plan_create.argtypes = [ctypes.POINTER(vp), ci, ci, ci, ci]
    L.fsea_exec_u8_device.argtypes = [vp, vp, sz, ci, vp, vp]
    res = []
    for n, frames in ((8192, 4096), (8192, 4099), (8192, 37), (2048, 16384), (1024, 20001)):
        x = torch.from_numpy(np.random.default_rng(n + frames).integers(-70, 70, 2 * n * frames, dtype=np.int8)).cuda()
        y = torch.zeros(n * frames, dtype=torch.float32, device="cuda")
        p = vp()
        assert L.fsea_plan_create(ctypes.byref(p), n, n, 0, 0) == 0
        assert L.fsea_exec_u8_device(p, x.data_ptr(), frames, 1, y.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        res.append(y.cpu())
    outs.append(res)
print("contiguous build's rows identical to the product's:", all(torch.equal(a, b) for a, b in zip(*outs)))
PY
  for n in 8192 4096 2048; do
    AB_N=$n timeout 300 python -u scripts/ab_window.py scripts/ab/libfsea_hip_contig.so 2>&1 | grep -v amdgpu.ids
    AB_N=$n AB_REGION=20 AB_ROUNDS=60 timeout 300 python -u scripts/ab_window.py scripts/ab/libfsea_hip_contig.so 2>&1 | grep -v amdgpu.ids
  done
} > gpurun_out/r05_static_contiguous.txt 2>&1
tail -40 gpurun_out/r05_static_contiguous.txt
