"""Regenerates tests/golden/bench_job_checksums.json ON A GPU BOX: runs bench.py's config-4 sweep and config-5 stream
through bench.run_broad / bench.run_stft_stream, compares every row with the oracle (tests/test_gpu_bench_jobs.py:
verify_jobs, which raises on a mismatch), and writes the checksums of the verified image / rows and of their inputs.

  python tests/golden/make_bench_checksums.py [out.json]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from tests.test_gpu_bench_jobs import GOLDEN, verify_jobs
    sums = verify_jobs(torch)
    sums["note"] = ("checksums (frequensea_amd/sweep.py: checksum) of bench.py's config-4 stitched image and config-5 rows on one "
                    "MI355X, written only after every row matched the oracle (orc_rows_mt; tests/parity.py tolerances), and of "
                    "the captures those jobs generate with torch's device generator (seeds 4000000 + f, 5000000 + block)")
    sums["torch"] = torch.__version__
    out = sys.argv[1] if len(sys.argv) > 1 else GOLDEN
    with open(out, "w") as fp:
        json.dump(sums, fp, indent=1, sort_keys=True)
        fp.write("\n")
    print(json.dumps(sums, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
