"""`python bench.py --gpus N` starts its own ranks (it used to demand an external torch.distributed.run).  Without
a GPU every rank must fail loudly ("no HIP device", exit code 3: there is no CPU fallback) and the launcher must hand that code
back instead of hanging -- which also shows that nothing in the launcher itself needs a device."""
import os
import subprocess
import sys

import pytest

from ngs_amd import ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(ffi.load_library().ngsq_device_count() > 0, reason="checks the no-GPU failure mode of the launcher")
@pytest.mark.parametrize("n", [1, 2])
def test_launcher_without_gpu(n):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-800:])
    assert "no HIP device visible; the hot path has no CPU fallback" in r.stderr
    assert r.stdout.strip() == ""   # no JSON line from a run that measured nothing
