"""Randomised end-to-end rounds (tools/stress_parity.py): random k, genome counts, insertion orders, incremental rebuilds and build
options; presence, colour sets, colour rows and sequence queries through host and device calls against Python-dictionary ground truth."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_randomised_rounds(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_parity.py"), "25", str(seed)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "stress OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
