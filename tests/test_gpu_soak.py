"""GPU tier: a short run of scripts/soak.py -- thousands of randomly drawn launches (every size and mode, both byte
conventions, three frame distributions, plain and tiled entry points) on four streams, each checked on sampled rows
against numpy.  Long launches under load are what found the store-data hazard of DESIGN.md section 3; the unit tests'
launches are too short to load a CU.  (Run the script itself for minutes: python scripts/soak.py 300.)"""
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", ["11", "random"])
def test_soak_run_finds_no_mismatch(seed):
    """One fixed seed and one drawn per run (scripts/soak.py prints it first, so a failure names the seed to replay)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "soak.py"), "12", seed], capture_output=True,
                       text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "no mismatch, no hang" in r.stdout and "soak: seed" in r.stdout
    print(r.stdout.splitlines()[0])
