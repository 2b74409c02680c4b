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


def _run_bench(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--records", "2000000",
                           "--cpu-sample", "100000", "--extra-facet-records", "1000000", "--file-big-records", "0",
                           "--file-realistic-records", "400000", "--mixed-records", "2000000", "--mixed-steps", "3",
                           "--all-facets-records", "2000000", "--all-facets-steps", "2", *flags],
                          capture_output=True, text=True, env=env, timeout=1500)


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [("--h2d-batch", "200000", "--file-records", "300000"),
                                   ("--gpus", "2", "--same-gpu", "--file-records", "300000", "--file-write-budget", "5")])
def test_bench_line_contract(flags):
    """What the driver reads: stdout is ONE line of JSON (RCCL's banner and everything else on stderr), with the
    contract's keys; at N > 1 the file legs report a document equal to the one-GPU command's."""
    import json
    r = _run_bench(*flags)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, lines[:3]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "parity_check"):
        assert key in d, key
    assert d["parity_check"].startswith("ok") and d["value"] > 0 and d["steps"] == 2 and d["warmup"] == 1
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    fe = d["file_end_to_end"]
    if d["n_gpus"] == 1:
        assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1
        assert fe["json_equal_device_host_inprocess"] is True and fe["check_total"] == 300000
        # round 4: the aligner-style file, the mixed-CIGAR workload, Edits on an aligner's CIGARs
        assert fe["realistic"]["same_document_as_host_reader"] is True and fe["realistic"]["check_total"] == fe["realistic"]["records"]
        assert fe["realistic"]["inflated_bytes_per_record"] > 330
        assert d["mixed"]["parity_check"].startswith("ok") and d["mixed"]["roofline"]["kernel"] == "k_qual_ragged"
        assert d["extra_facets"]["edits_aligner_cigars"]["avg_ms"] > 0
        assert isinstance(d["roofline"]["traffic_measured_in_this_run"], bool)
        # round 5: three timed loops and the card's state, every scan of every file leg and the median, reads that differ from the
        # reference, all seven facets as one scan with the SEQ column read once
        assert len(d["ms_per_step_each_loop"]) == 3 and "source" in d["gpu_state"] and d["ms_per_step_outside_kernels"] < d["ms_per_step"]
        assert len(fe["in_process_device_ingest"]["seconds_each_scan"]) == 6 and "median" in fe["value_is"]
        assert len(fe["realistic"]["seconds_each_scan"]) == 5 and "median" in fe["realistic"]["value_is"]
        assert "phases_ms" in fe["cli_device_ingest"] and len(fe["cli_device_ingest"]["seconds_each_run"]) == 3
        for leg in ("edits_subst_5pct", "edits_subst_25pct", "edits_iid_reads"):
            assert d["extra_facets"][leg]["avg_ms"] > 0
        assert d["all_facets"]["parity_check"].startswith("ok") and d["all_facets"]["seq_column_reads"].startswith("once")
        assert "gc" not in d["all_facets"]["kernels"] and d["all_facets"]["kernels"]["edits"]["avg_ms"] > 0
    else:
        assert fe["json_equal_sharded_one_gpu"] is True and fe["in_process"]["json_equal_one_gpu"] is True
        assert fe["check_total"] == fe["records"] == fe["in_process"]["check_total"]
