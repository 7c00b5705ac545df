"""``python bench.py --gpus 2`` end to end on the GPU box's one MI355X: byte for byte the code path the driver's SCALE
command takes (bench.py's own rank launcher -> two fresh rank processes -> init_process_group -> the encoder row, the LSTM
leg, the full VAE row with the gradient all-reduce in vae_train.py:78-83's order -> ONE JSON line on stdout), with gloo
as the rehearsal backend and both ranks on cuda:0 (``GGPM_BENCH_ONE_DEVICE=1``).  RCCL with more than one rank needs the
driver's multi-GPU node; what this guards is everything around the collective.

Collected right after test_aa_data_parallel_gpu.py (whose two rank processes have exited by then) and before any test of
this process initialises HIP; the ranks are fresh children of the launcher, nothing is re-executed.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_with_two_ranks_prints_one_line_through_its_own_launcher(tmp_path):
    env = dict(os.environ, GGPM_BENCH_ONE_DEVICE="1", GGPM_LAUNCH_TIMEOUT="420")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, "-u", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "6",
           "--no-cpu-baseline", "--no-configs4"]
    err = tmp_path / "bench_two_ranks.log"
    with open(err, "w") as ef:
        p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=ef, timeout=480)
    log = open(err).read()
    assert p.returncode == 0, "bench.py --gpus 2 exited with %d:\n%s" % (p.returncode, log[-6000:])
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, "stdout must carry exactly one line, got %d:\n%s" % (len(lines), p.stdout.decode()[:2000])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 6
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 64
    assert "gloo all-reduce (rehearsal backend, not RCCL)" in d["config"]["workload"]
    ar = d.get("allreduce")
    assert ar and ar.get("form") in ("single", "bucketed") and ar["ms_per_step_single"] > 0 and ar["ms_per_step_bucketed"] > 0
    # the timed-region protocol is the same on every world size: the record is there with N = 2 as well
    assert d["timed_region"]["attempts"] >= 1 and len(d["timed_region"]["ms_per_step_of_each_attempt"]) == d["timed_region"]["attempts"]
    v = d.get("vae_step")
    assert v and "error" not in v, v
    assert v["n_gpus"] == 2 and v["ms_per_step"] > 0 and v["value"] > 0
    assert "cpu_baseline" not in d and "configs4" not in d
    assert "starting 2 rank processes" in log
