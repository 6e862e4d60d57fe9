#!/bin/bash
# fsea-fft-batch on the reference's own geometry (1024 points, 16384 rows per centre frequency, c/fft-batch.c:14-21), six
# captures: how long each of the three host stages is busy and what the overlapped loop takes (--timing).
# The captures are sparse files: only the first 2 KiB of every 262144-byte transfer is ever read (c/fft-batch.c:62-69).
R=$(cd "$(dirname "$0")/.." && pwd)
W=${TMPDIR:-/tmp}/fsea_tool_timing.$$; mkdir -p $W/out
python - "$W" <<'PY'
import sys, numpy as np
w = sys.argv[1]
for k in range(6):
    rng = np.random.default_rng(k)
    with open("%s/c%d.raw" % (w, k), "wb") as f:
        f.truncate(16394 * 262144)
        for t in range(16394):
            f.seek(t * 262144)
            f.write(rng.integers(-60, 60, 2048, dtype=np.int8).tobytes())
PY
CAPS=""; for k in 0 1 2 3 4 5; do CAPS="$CAPS $((1802 + 2 * k))=$W/c$k.raw"; done
for rep in 1 2; do
  $R/frequensea_amd/bin/fsea-fft-batch --timing --out $W/out $CAPS | grep "Stages busy"
done
ls -la $W/out | head -4
rm -rf $W
