#!/usr/bin/env python3
"""profiles/traffic.json from a PMC summary (scripts/pmc.sh -> scripts/pmc_summary.py lines "kernel COUNTER mean=... n=...").

  python scripts/pmc_to_traffic.py profiles/r05_pmc_fsea_fft8192_u8_mag.txt fsea_fft8192_u8_mag 4096 512 256 73104

The constants bench.py quotes beside its live timing (only while kernel, frames per launch, grid, block and LDS match):
  hbm_bytes_per_launch          FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md: 128-B requests tallied at 64 B) + WRITE_SIZE, KB -> B
  sq_insts_valu_per_launch      vector wave-instructions the launch issues
  sq_lds_idx_active_per_launch  cycles the CUs' LDS units were active, summed over the CUs
  shader_cycles_per_launch      SQ_BUSY_CYCLES / 32 (the counter sums the 32 shader engines: 8 XCDs x 4)
"""
import json
import re
import sys

path, kernel, frames, grid, block, lds = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
vals = {}
for line in open(path):
    m = re.match(r"(\S+) (\S+) mean=(\S+) n=(\d+)", line.strip())
    if m and m.group(1) == kernel:
        vals[m.group(2)] = float(m.group(3))
need = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_BUSY_CYCLES")
missing = [k for k in need if k not in vals]
if missing:
    sys.exit("missing counters for %s in %s: %s" % (kernel, path, missing))
out = {
    "kernel": kernel, "frames": frames, "grid_block_lds": [grid, block, lds],
    "hbm_bytes_per_launch": int(round((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)),
    "fetch_size_kb_raw": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"],
    "sq_insts_valu_per_launch": vals["SQ_INSTS_VALU"],
    "sq_insts_lds_per_launch": vals.get("SQ_INSTS_LDS"),
    "sq_lds_idx_active_per_launch": vals["SQ_LDS_IDX_ACTIVE"],
    "sq_lds_bank_conflict_per_launch": vals.get("SQ_LDS_BANK_CONFLICT"),
    "shader_cycles_per_launch": vals["SQ_BUSY_CYCLES"] / 32.0,
    "compute_units": 256,
    "valu_peak_wave_insts_per_s": 460e9,
    "valu_issue_note": "SQ_INSTS_VALU per launch (this file) / the run's own launch time / 460 G wave-instructions per second: the "
                       "chip-wide issue rate measured for v_pk_fma_f32 -- 87 % of this kernel's vector instructions -- at the kernel's "
                       "two waves per SIMD (profiles/r01_valu_issue_rate_microbench.txt; that microbenchmark itself runs at the "
                       "package power cap)",
    "lds_active_note": "SQ_LDS_IDX_ACTIVE per launch / (256 CUs x SQ_BUSY_CYCLES / 32 shader engines): the share of the launch's "
                       "shader cycles in which a CU's LDS unit was busy",
    "method": "rocprofv3 --kernel-trace --pmc in separate passes (scripts/pmc.sh), per-dispatch mean; FETCH_SIZE doubled per the gfx950 "
              "correction in MI355X_MICROARCH.md (128-B requests tallied at 64 B)",
    "source": "%s (PMC passes on this round's product kernel, scripts/pmc.sh; constants, not measured by the bench run that quotes them)" % path,
}
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("hbm_bytes_per_launch", "sq_insts_valu_per_launch", "sq_lds_idx_active_per_launch", "shader_cycles_per_launch")}))
