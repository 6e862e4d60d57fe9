#!/usr/bin/env python3
"""Diagnostics: when do the workgroups of one launch start and finish? (FSEA_TRACE=1)"""
import ctypes, os, sys
import numpy as np
os.environ["FSEA_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frequensea_amd import fsea
fsea.use_tune_library()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
L = fsea.hip_lib()
L.fsea_plan_read_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
def dev_alloc(nbytes):
    p = ctypes.c_void_p(); fsea._check(L.fsea_device_alloc(0, nbytes, ctypes.byref(p))); return p
sets = 8
d_in = [dev_alloc(2 * n * frames) for _ in range(sets)]
d_out = [dev_alloc(4 * n * frames) for _ in range(sets)]
host = np.random.default_rng(1).integers(-70, 70, 2 * n * frames, dtype=np.int8).view(np.uint8)
for d in d_in:
    fsea._check(L.fsea_copy_to_device(0, d, host.ctypes.data, host.nbytes))
plan = fsea.Plan(n, variant=os.environ.get("FSEA_VARIANT"))
if os.environ.get("FSEA_UNITS"):          # static | tickets: pin the frame distribution (default: per launch)
    plan.set_unit_distribution({"static": fsea.UNITS_STATIC, "tickets": fsea.UNITS_TICKETS}[os.environ["FSEA_UNITS"]])
grid = plan.grid(frames)[0]
for k in range(200):                      # warm clocks, rotate buffers
    plan.exec_device(d_in[k % sets], frames, d_out[k % sets])
plan.synchronize()
ms = plan.time_device(d_in[0], frames, d_out[0], 1)
tr = np.zeros((grid, 32), dtype=np.uint64)
fsea._check(L.fsea_plan_read_trace(plan._p, tr.ctypes.data, grid))
t = (tr[:, :2].astype(np.int64) - int(tr[:, 0].min())) / 100.0      # wall_clock64 ticks at 100 MHz -> us
st, en = t[:, 0], t[:, 1]
clk = (tr[:, 3].astype(np.int64) - tr[:, 2].astype(np.int64)) / np.maximum(en - st, 1e-9) / 1e3   # GHz
hw = tr[:, 4].astype(np.int64)
xcc = tr[:, 5].astype(np.int64) & 0xf
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
print("N=%d frames=%d grid=%d  event time %.1f us" % (n, frames, grid, ms * 1e3))
print("start: min %.2f  p50 %.2f  p99 %.2f  max %.2f us" % (st.min(), np.median(st), np.percentile(st, 99), st.max()))
print("end  : min %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f us" % (en.min(), np.percentile(en, 10), np.median(en), np.percentile(en, 90), en.max()))
dur = en - st
print("workgroup duration: min %.2f  p50 %.2f  max %.2f us;  mean %.2f" % (dur.min(), np.median(dur), dur.max(), dur.mean()))
print("idle tail if perfectly balanced: %.2f us (max end - mean end)" % (en.max() - en.mean()))
print("shader clock seen by workgroups: min %.3f  mean %.3f  max %.3f GHz" % (clk.min(), clk.mean(), clk.max()))
for x in range(8):
    m = xcc == x
    if m.any():
        print("  XCC %d: %3d workgroups, mean duration %.2f us, clock %.3f GHz, distinct (se,sh,cu) %d"
              % (x, m.sum(), dur[m].mean(), clk[m].mean(), len(set(zip(se[m], sh[m], cu[m])))))
key = xcc * 1000 + se * 100 + sh * 10 * 2 + cu
import collections
cnt = collections.Counter(zip(xcc, se, sh, cu))
print("workgroups per physical CU: " + str(sorted(collections.Counter(cnt.values()).items())))
its = (tr[:, 8:32].astype(np.int64) - tr[:, 0:1].astype(np.int64)) / 100.0
# the trace buffer is not cleared between launches: keep only stamps inside this launch's window
valid = (tr[:, 8:32] >= tr[:, 0:1]) & (tr[:, 8:32] <= tr[:, 1:2])
prev = np.concatenate([np.zeros((grid, 1)), its[:, :-1]], axis=1)
per = np.where(valid, its - prev, np.nan)
print("mean time of iteration k (us): " + " ".join("%.2f" % x for x in np.nanmean(per, axis=0)[:min(24, int(valid.sum(axis=1).max()))]))
pro = (tr[:, 6].astype(np.int64) - tr[:, 0].astype(np.int64)) / 100.0
p0 = (tr[:, 7].astype(np.int64) - tr[:, 0].astype(np.int64)) / 100.0
print("prologue done after %.2f us (mean), first pass 0 + barrier after %.2f us, first iteration ends after %.2f us"
      % (pro.mean(), p0.mean(), np.nanmean(its[:, 0])))
# which workgroups share a CU, and how their durations and frame counts relate
pairs = collections.defaultdict(list)
for blk in range(grid):
    pairs[(int(xcc[blk]), int(se[blk]), int(sh[blk]), int(cu[blk]))].append(blk)
diffs = collections.Counter(tuple(sorted(v))[1] - tuple(sorted(v))[0] for v in pairs.values() if len(v) == 2)
print("block-index distance of the two workgroups of a CU: " + str(sorted(diffs.items(), key=lambda kv: -kv[1])[:6]))
n_iter = valid.sum(axis=1)
two = [sorted(v) for v in pairs.values() if len(v) == 2]
if two:
    lo = np.array([min(dur[a], dur[b]) for a, b in two])
    hi = np.array([max(dur[a], dur[b]) for a, b in two])
    first_faster = np.mean([dur[a] < dur[b] for a, b in two])
    cu_end = np.array([max(en[a], en[b]) for a, b in two])
    print("per CU: faster workgroup %.2f us, slower %.2f us (means); the lower block index is the faster one in %.0f %% of the CUs"
          % (lo.mean(), hi.mean(), 100 * first_faster))
    print("per CU end (last of its two workgroups): min %.2f  p50 %.2f  max %.2f us; iterations per workgroup: min %d max %d"
          % (cu_end.min(), np.median(cu_end), cu_end.max(), n_iter.min(), n_iter.max()))
# prologue detail (tuning build): tables arrived / tables in LDS (barrier) / register twiddles built / first unit's bytes converted
sub = (tr[:, 28:31].astype(np.int64) - tr[:, 0:1].astype(np.int64)) / 100.0
ok = (tr[:, 28] >= tr[:, 0]) & (tr[:, 28] <= tr[:, 1])
if ok.any():
    print("prologue: tables arrived %.2f us, in LDS (barrier) %.2f us, register twiddles built %.2f us, first bytes converted %.2f us (means)"
          % (sub[ok, 0].mean(), sub[ok, 1].mean(), pro[ok].mean(), sub[ok, 2].mean()))
