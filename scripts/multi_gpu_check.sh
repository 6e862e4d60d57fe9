#!/bin/bash
# First-contact kit for a multi-GPU node: this repository's multi-GPU code (the C sweep tool over RCCL, bench.py over
# torch.distributed/RCCL) has never run on more than one GPU -- the pool offers one-GPU boxes.  This script does all of it
# in one unattended go and leaves a log that tells the reader, by itself, what happened:
#   0. what the node is: GPUs, RCCL version, HIP device <-> rank map
#   1. the C tool: fsea-fft-sweep --devices 0-(N-1) over synthetic captures, must report "Gather backend: rccl", and its
#      stitched image must equal the max-composite of fsea-fft-batch's tiles made on one GPU; if RCCL cannot initialise
#      the tool falls back to the "copy" backend with a loud line (fsea_comm.hip) and this script says so
#   2. bench.py at 1, 2, 4 ... N ranks: the headline (frames sharded, no collective), the fft-batch-broad sweep with
#      resident and with ingested captures (its chunks alternate on two streams per rank), the halo-sharded 16384-point
#      stream, rectangular and with the fused Hann taper -- every line carries its regime and window in `config`
#   3. the driver's own line, `bench.py --gpus n --steps 20 --warmup 5`, at 1, 2, 4 ... N ranks: headline + the multi-GPU leg in
#      `extra` (census, checksums of what arrived, gather rate) -- what the round-end SCALE record will contain
# Every step runs under `timeout`; nothing can hang the node.
# Usage: bash scripts/multi_gpu_check.sh [N] [--dry-run] [--log FILE]
#   --dry-run   print the commands instead of running them (no GPU needed; tests/test_host_api.py checks the plumbing)
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
N=""; DRY=0; LOG=""
while [ $# -gt 0 ]; do
  case "$1" in
    --dry-run) DRY=1;;
    --log) shift; LOG="$1";;
    *) N="$1";;
  esac
  shift
done
if [ -z "$N" ]; then
  if [ $DRY = 1 ]; then N=8; else N=$(python -c "from frequensea_amd import fsea; print(fsea.device_count())" 2>/dev/null || echo 0); fi
fi
[ -n "$LOG" ] || LOG="$R/gpurun_out/multi_gpu_check_${N}gpu.log"
mkdir -p "$(dirname "$LOG")"
STEP_TIMEOUT=${STEP_TIMEOUT:-420}
run() {   # run "<label>" cmd...   -> logs the command, its tail and its verdict
  local label="$1"; shift
  echo "---- $label" | tee -a "$LOG"
  echo "\$ $*" | tee -a "$LOG"
  if [ $DRY = 1 ]; then return 0; fi
  local t0=$(date +%s)
  timeout "$STEP_TIMEOUT" "$@" > "$LOG.step" 2>&1
  local rc=$?
  tail -n 12 "$LOG.step" | cut -c1-600 | tee -a "$LOG"
  if [ $rc = 124 ]; then echo "VERDICT[$label]: TIMED OUT after ${STEP_TIMEOUT}s" | tee -a "$LOG"
  elif [ $rc = 0 ]; then echo "VERDICT[$label]: ok ($(( $(date +%s) - t0 )) s)" | tee -a "$LOG"
  else echo "VERDICT[$label]: FAILED rc=$rc" | tee -a "$LOG"; fi
  return $rc
}
: > "$LOG"
echo "== multi_gpu_check: $N GPU(s), $(date -u +%Y-%m-%dT%H:%M:%SZ), host $(hostname), dry-run=$DRY" | tee -a "$LOG"
if [ $DRY = 0 ]; then
  /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | sort | uniq -c | tee -a "$LOG"
  python - <<'PY' 2>&1 | tee -a "$LOG"
import ctypes
try:
    r = ctypes.CDLL("/opt/rocm/lib/librccl.so")
    v = ctypes.c_int(0); r.ncclGetVersion(ctypes.byref(v)); print("RCCL version code:", v.value)
except Exception as e:
    print("RCCL: cannot load librccl.so:", e)
try:
    import torch
    print("torch", torch.__version__, "devices", torch.cuda.device_count(), "nccl available", torch.distributed.is_nccl_available())
    for i in range(torch.cuda.device_count()):
        p = torch.cuda.get_device_properties(i)
        print("  rank %d <-> HIP device %d: %s, %d CUs, %.0f GiB" % (i, i, p.name, p.multi_processor_count, p.total_memory / 2**30))
except Exception as e:
    print("torch:", e)
PY
fi
if [ "$N" -lt 1 ]; then echo "VERDICT: no GPU visible" | tee -a "$LOG"; exit 1; fi

# ---- 0. RCCL by itself, one GPU at a time: a one-rank communicator and the gather's grouped send / recv with itself ----
d=0
while [ $d -lt "$N" ]; do
  run "rccl self-loop on device $d (fsea_comm_selftest_rccl, 64 MiB)" python -c "
import ctypes, sys
L = ctypes.CDLL('frequensea_amd/libfsea_rccl.so'); L.fsea_comm_selftest_rccl.argtypes = [ctypes.c_int, ctypes.c_size_t]
L.fsea_comm_last_error.restype = ctypes.c_char_p
rc = L.fsea_comm_selftest_rccl($d, 64 << 20)
print('rc', rc, '' if rc == 0 else L.fsea_comm_last_error().decode()); sys.exit(1 if rc else 0)"
  d=$((d + 1))
done

# ---- 1. the C tool over RCCL ------------------------------------------------------------------------------------
W=${TMPDIR:-/tmp}/fsea_mgc.$$
CAPS=""
NCAP=$(( 2 * N + 3 ))            # ragged: not a multiple of the member count
if [ $DRY = 0 ]; then
  mkdir -p "$W/ref" "$W/sweep"
  python - "$W" "$NCAP" <<'PY'
import sys, numpy as np
w, ncap = sys.argv[1], int(sys.argv[2])
for k in range(ncap):                                   # 123 transfers of 262144 bytes each, one all-zero capture (gate)
    rng = np.random.default_rng(100 + k)
    raw = np.zeros(123 * 262144, np.int8) if k == 2 else rng.normal(0, 20, 123 * 262144).round().clip(-128, 127).astype(np.int8)
    raw.tofile("%s/c%d.raw" % (w, 660 + 5 * k))
PY
fi
for k in $(seq 0 $(( NCAP - 1 ))); do f=$(( 660 + 5 * k )); CAPS="$CAPS $f=$W/c$f.raw"; done
LAST=$(( 660 + 5 * (NCAP - 1) ))
BIN=frequensea_amd/bin
run "tool: fsea-fft-batch on one GPU (the tiles the sweep must reproduce)" bash -c "$BIN/fsea-fft-batch --broad --rows 120 --skip 3 --out $W/ref $CAPS | grep -c Frequency; ls $W/ref | wc -l"
DEV="0"; [ "$N" -gt 1 ] && DEV="0-$(( N - 1 ))"
run "tool: fsea-fft-sweep --devices $DEV (RCCL gather)" bash -c "$BIN/fsea-fft-sweep --broad --devices $DEV --chunk 2 --rows 120 --skip 3 --out $W/sweep $CAPS 2>&1 | grep -E 'Gather backend|fsea_comm|Written.*stitched|rror'"
if [ $DRY = 0 ]; then
  if grep -q "Gather backend: rccl" "$LOG.step"; then echo "VERDICT[gather backend]: rccl -- RCCL initialised over $N devices (ncclCommInitAll) and carried the tiles" | tee -a "$LOG"
  elif [ "$N" -gt 1 ]; then echo "VERDICT[gather backend]: *** NOT rccl on $N GPUs (see the fsea_comm line above): the copy backend carried the tiles ***" | tee -a "$LOG"
  else echo "VERDICT[gather backend]: copy (one GPU: nothing to gather over)" | tee -a "$LOG"; fi
  python - "$W" "$LAST" <<'PY' 2>&1 | tee -a "$LOG"
import sys, glob, numpy as np
from PIL import Image
w, last = sys.argv[1], sys.argv[2]
a = sorted(glob.glob("%s/sweep/broad-stitched-*.png" % w)); tiles = sorted(glob.glob("%s/ref/broad-[0-9]*.png" % w))
if not a:
    print("VERDICT[tool image]: FAILED, the sweep wrote no stitched image")
else:
    img = np.array(Image.open(a[0])); want = np.zeros_like(img)
    for t in tiles:
        f = int(t.split("broad-")[-1].split(".")[0]); k = (f - 660) // 5
        tile = np.array(Image.open(t)); want[:, k * 256:(k + 1) * 256] = np.maximum(want[:, k * 256:(k + 1) * 256], tile)
    same = np.array_equal(img, want)
    print("VERDICT[tool image]: %s (%s, %d tiles of the single-GPU batch run composited)" % ("identical to batch + composite" if same else "DIFFERS", a[0].split("/")[-1], len(tiles)))
PY
fi
if [ "$N" -gt 1 ]; then
  run "tool: the same sweep over the copy backend (FSEA_COMM_BACKEND=copy)" bash -c "FSEA_COMM_BACKEND=copy $BIN/fsea-fft-sweep --broad --devices $DEV --chunk 2 --rows 120 --skip 3 --no-tiles --out $W/sweep $CAPS 2>&1 | grep -E 'Gather backend|rror'"
fi

# ---- 2. bench.py at 1, 2, 4 ... N ranks -------------------------------------------------------------------------
port=29600
n=1
while [ $n -le "$N" ]; do
  for wl in "headline|" "headline-hann|--window hann" "broad-resident|--workload broad --regime resident" "broad-ingest|--workload broad --regime ingest" "stft-stream|--workload stft16384stream" "stft-stream-hann|--workload stft16384stream --window hann"; do
    name=${wl%%|*}; args=${wl#*|}
    port=$((port + 1))
    # a headline step is 50 us: 20 of them is one millisecond on a cold clock, so those two run bench.py's own default length
    case $name in headline*) SW="--steps 2000 --warmup 200";; *) SW="--steps 20 --warmup 5";; esac
    if [ $n -eq 1 ]; then
      run "bench $name x1" python bench.py --gpus 1 $SW --no-extra --no-cpu-baseline $args
    else
      run "bench $name x$n" python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n $SW --no-extra --no-cpu-baseline $args
    fi
    if [ $DRY = 0 ]; then
      python - "$LOG.step" "$name" "$n" <<'PY' 2>&1 | tee -a "$LOG"
import sys, json
lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
if lines:
    d = json.loads(lines[-1])
    print("RESULT[%s x%s]: %.4g %s, %.4f ms/step, roofline frac %.3f, regime: %s" % (sys.argv[2], sys.argv[3], d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["config"].get("regime", d["config"].get("parallelism"))))
else:
    print("RESULT[%s x%s]: no JSON line" % (sys.argv[2], sys.argv[3]))
PY
    fi
  done
  n=$((n * 2))
done
# ---- 3. the driver's OWN command line: bench.py --gpus n --steps 20 --warmup 5 (what SCALE_rNN.json is made from) --------------
# Since round 5 this one line runs, behind the headline's timed steps, the sharded sweep in both regimes and the halo-sharded
# stream with the chunked RCCL gather, and reports the communicator's census, the checksums of what arrived against what the
# members computed and the gather's rate per link in `extra` (bench.py: multi_gpu_leg).  The checksums must be the same at every n.
n=1
while [ $n -le "$N" ]; do
  port=$((port + 1))
  if [ $n -eq 1 ]; then
    run "driver line x1" python bench.py --gpus 1 --steps 20 --warmup 5
  else
    run "driver line x$n" python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n --steps 20 --warmup 5
  fi
  if [ $DRY = 0 ]; then
    python - "$LOG.step" "$n" <<'PY' 2>&1 | tee -a "$LOG"
import sys, json
lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not lines:
    print("RESULT[driver line x%s]: no JSON line" % sys.argv[2])
else:
    d = json.loads(lines[-1]); e = d.get("extra", {})
    print("RESULT[driver line x%s]: value %.4g frames/s (weak), multi_gpu_error %s, backend %s, rccl_world %s, distinct GPUs %s" %
          (sys.argv[2], d["value"], e.get("multi_gpu_error"), e.get("gather_backend"), e.get("rccl_world"), e.get("distinct_gpus")))
    print("RESULT[driver line x%s]: sweep resident %.3f ms, ingest %.3f ms, stream %.3f ms; gather alone %s ms = %s GB/s per link, %s GB/s into the root" %
          (sys.argv[2], e.get("broad_sweep_ms_resident", float("nan")), e.get("broad_sweep_ms_ingest", float("nan")), e.get("stft_stream_ms", float("nan")),
           e.get("gather_only_ms"), e.get("gather_gbps_per_link"), e.get("gather_gbps_into_root")))
    print("RESULT[driver line x%s]: checksums sweep %s / %s (ok %s / %s), stream %s (ok %s)" %
          (sys.argv[2], e.get("broad_sweep_resident_gathered_checksum"), e.get("broad_sweep_ingest_gathered_checksum"),
           e.get("broad_sweep_resident_gathered_checksum_ok"), e.get("broad_sweep_ingest_gathered_checksum_ok"),
           e.get("stft_stream_gathered_checksum"), e.get("stft_stream_gathered_checksum_ok")))
PY
  fi
  n=$((n * 2))
done
[ $DRY = 0 ] && rm -rf "$W" "$LOG.step"
echo "== done; verdicts:" | tee -a "$LOG"
grep -E "^VERDICT|^RESULT" "$LOG" | sort | uniq | tee -a "$LOG.summary" > /dev/null
cat "$LOG.summary" 2>/dev/null; rm -f "$LOG.summary"
