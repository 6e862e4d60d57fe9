"""CPU tier: the unattended scripts' plumbing, without a GPU."""
import os
import subprocess

from tests.conftest import ROOT


def test_multi_gpu_first_contact_kit_plumbing(tmp_path):
    """scripts/multi_gpu_check.sh --dry-run prints what it would run on an 8-GPU node: the C sweep tool over all eight
    devices (and once more with FSEA_COMM_BACKEND=copy), and bench.py at 1, 2, 4, 8 ranks for the headline, the sweep in
    both regimes and the halo-sharded stream -- under torch.distributed.run with 127.0.0.1 rendezvous for N > 1."""
    log = tmp_path / "mgc.log"
    env = dict(os.environ, FSEA_COMM_BACKEND="copy")
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "multi_gpu_check.sh"), "8", "--dry-run", "--log", str(log)],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert "fsea-fft-sweep --broad --devices 0-7" in out
    assert "FSEA_COMM_BACKEND=copy frequensea_amd/bin/fsea-fft-sweep --broad --devices 0-7" in out
    assert out.count("660=") >= 3 and "750=" in out                     # 2 * 8 + 3 = 19 captures, 660 ... 750 MHz
    for n in (2, 4, 8):
        for wl in ("", " --window hann", " --workload broad --regime resident", " --workload broad --regime ingest",
                   " --workload stft16384stream", " --workload stft16384stream --window hann"):
            want = "--nproc-per-node %d --master-addr 127.0.0.1" % n
            line = [ln for ln in out.splitlines() if want in ln and ln.rstrip().endswith(("--no-cpu-baseline" + wl).strip())]
            assert line, (n, wl)
    assert "python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --workload broad --regime ingest" in out
    assert log.exists() and "dry-run=1" in log.read_text()
    # one GPU: no distributed launch, the tool runs on device 0
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "multi_gpu_check.sh"), "1", "--dry-run", "--log", str(log)],
                       capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert r.returncode == 0 and "torch.distributed.run" not in r.stdout and "--devices 0 " in r.stdout
