"""A slice of the randomised parity sweeps of tools/fuzz_parity.py inside the GPU suite (the full sweeps -- 120 seeds of each
-- are run by hand): the HIP path against the oracle on random shapes, the streaming Coverage on random
piles, Edits + Genomic Features, and file -> device ingest -> kernels against file -> host ingest -> kernels."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [("--seeds", "8"), ("--seeds", "0", "--sorted", "6"), ("--seeds", "0", "--extra", "4"),
                                   ("--seeds", "0", "--ingest", "10"), ("--seeds", "0", "--genome", "4")])
def test_fuzz_slice(flags):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), *flags], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "all seeds ok" in r.stdout
