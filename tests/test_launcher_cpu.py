"""The rank launcher behind ``python bench.py --gpus N`` (ggpm_amd/launcher.py), with stub children: no GPU, no torch.

What the driver needs from it: N fresh processes with the torch.distributed.run environment, rank 0's ONE line on
stdout, a non-zero exit as soon as any rank fails, the rest ended, a bounded wait.
"""
import io
import json
import os
import sys
import textwrap
import time

import pytest

from ggpm_amd import launcher

STUB = textwrap.dedent('''
    import json, os, sys, time
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    mode = sys.argv[1]
    marker = sys.argv[2]
    open(os.path.join(marker, "started.%d" % rank), "w").write(str(os.getpid()))
    if mode == "fail" and rank == 1:
        sys.exit(7)
    if mode in ("fail", "hang") and rank != 1:
        time.sleep(60)                       # must be ended by the launcher, not run to completion
        open(os.path.join(marker, "survived.%d" % rank), "w").write("x")
    print(json.dumps({"rank": rank, "argv": sys.argv[1:], "big": "x" * (200000 if rank == 0 else 0),
                      "env": {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE",
                                                              "MASTER_ADDR", "MASTER_PORT", "GPU_MAX_HW_QUEUES",
                                                              "HSA_ENABLE_IPC_MODE_LEGACY")}}))
''')


@pytest.fixture
def stub(tmp_path):
    p = tmp_path / "stub_rank.py"
    p.write_text(STUB)
    return str(p), str(tmp_path)


def test_rank_environment_is_what_torch_distributed_run_would_set():
    cmds = launcher.rank_commands("/x/bench.py", ["--gpus", "4", "--steps", "3"], 4, 29999, python="py", base_env={"A": "b"})
    assert len(cmds) == 4
    for r, (cmd, env) in enumerate(cmds):
        assert cmd == ["py", "/x/bench.py", "--gpus", "4", "--steps", "3"]
        assert env["RANK"] == env["LOCAL_RANK"] == str(r)
        assert env["WORLD_SIZE"] == env["LOCAL_WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29999"
        assert env["GPU_MAX_HW_QUEUES"] == "8" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["A"] == "b"
    # a value the user chose is kept
    _, env = launcher.rank_commands("s", [], 1, 1, base_env={"GPU_MAX_HW_QUEUES": "2"})[0]
    assert env["GPU_MAX_HW_QUEUES"] == "2"


def test_rank_zero_line_is_relayed_and_only_that(stub):
    script, marker = stub
    out, err = io.StringIO(), open(os.path.join(marker, "err.log"), "w")
    rc = launcher.run_ranks(script, ["ok", marker], 3, timeout=60, out=out, err=err)
    err.close()
    assert rc == 0
    lines = out.getvalue().strip().splitlines()
    assert len(lines) == 1                                  # ranks 1, 2 wrote to stderr
    got = json.loads(lines[0])
    assert got["rank"] == 0 and got["argv"] == ["ok", marker] and len(got["big"]) == 200000
    assert got["env"]["WORLD_SIZE"] == "3" and got["env"]["MASTER_ADDR"] == "127.0.0.1"
    others = open(os.path.join(marker, "err.log")).read()
    assert '"rank": 1' in others and '"rank": 2' in others
    assert sorted(f for f in os.listdir(marker) if f.startswith("started.")) == ["started.0", "started.1", "started.2"]


def test_a_failing_rank_fails_the_run_and_ends_the_others(stub):
    script, marker = stub
    out, err = io.StringIO(), open(os.path.join(marker, "err.log"), "w")
    t0 = time.time()
    rc = launcher.run_ranks(script, ["fail", marker], 3, timeout=60, out=out, err=err)
    err.close()
    assert rc == 7 and time.time() - t0 < 30
    assert out.getvalue() == ""                             # no result line from a failed run
    time.sleep(0.2)
    for r in (0, 2):                                        # the sleeping ranks are gone (exact PIDs, by process group)
        pid = int(open(os.path.join(marker, "started.%d" % r)).read())
        with pytest.raises(ProcessLookupError):
            os.kill(pid, 0)
        assert not os.path.exists(os.path.join(marker, "survived.%d" % r))


def test_timeout_ends_every_rank(stub):
    script, marker = stub
    out, err = io.StringIO(), open(os.path.join(marker, "err.log"), "w")
    t0 = time.time()
    rc = launcher.run_ranks(script, ["hang", marker], 2, timeout=1.5, out=out, err=err)
    err.close()
    assert rc == 124 and time.time() - t0 < 20 and out.getvalue() == ""
    assert "did not finish" in open(os.path.join(marker, "err.log")).read()


def test_bench_argument_parsing_and_launch_decision(monkeypatch):
    """bench.py: --gpus N with WORLD_SIZE unset goes to the launcher with the ORIGINAL argv; with WORLD_SIZE set it is a rank."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    a = bench.parse_args(["--gpus", "2", "--steps", "6", "--backend", "gloo"])
    assert (a.gpus, a.steps, a.backend) == (2, 6, "gloo")
    seen = {}

    def fake_run(script, argv, world, timeout=0, **kw):
        seen.update(script=script, argv=list(argv), world=world)
        return 0

    monkeypatch.setattr(launcher, "run_ranks", fake_run)
    monkeypatch.setenv("GGPM_BENCH_ONE_DEVICE", "1")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.launch_ranks(a, ["--gpus", "2", "--steps", "6", "--backend", "gloo"]) == 0
    assert seen["world"] == 2 and seen["argv"] == ["--gpus", "2", "--steps", "6", "--backend", "gloo"]
    assert os.path.samefile(seen["script"], os.path.join(root, "bench.py"))
    monkeypatch.delenv("GGPM_BENCH_ONE_DEVICE")
    with pytest.raises(SystemExit):                         # this container shows no GPU: a clear refusal, not a hang
        bench.launch_ranks(a, [])


def test_gpus_are_counted_from_the_driver_topology_without_hip(tmp_path, monkeypatch):
    """launcher.visible_gpu_count: GPU nodes of the amdkfd topology (simd_count > 0), narrowed by *_VISIBLE_DEVICES; the
    parent of a multi-rank run must not call torch.cuda (device_count() can fall back to hipGetDeviceCount)."""
    topo = tmp_path / "nodes"
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):        # two CPU agents, three GPUs
        d = topo / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (64 if simd == 0 else 0, simd))
    assert launcher.visible_gpu_count({}, str(topo)) == 3
    assert launcher.visible_gpu_count({"HIP_VISIBLE_DEVICES": "0,2"}, str(topo)) == 2
    assert launcher.visible_gpu_count({"ROCR_VISIBLE_DEVICES": "1", "HIP_VISIBLE_DEVICES": "0,1"}, str(topo)) == 1
    assert launcher.visible_gpu_count({"HIP_VISIBLE_DEVICES": ""}, str(topo)) == 0
    assert launcher.visible_gpu_count({}, str(tmp_path / "absent")) == 0
    # bench.launch_ranks asks the launcher, never torch.cuda
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    import torch

    def boom(*a, **k):
        raise AssertionError("torch.cuda touched in the launcher parent")

    monkeypatch.setattr(torch.cuda, "device_count", boom)
    monkeypatch.setattr(torch.cuda, "is_available", boom)
    monkeypatch.setattr(launcher, "visible_gpu_count", lambda *a, **k: 8)
    monkeypatch.setattr(launcher, "run_ranks", lambda script, argv, world, timeout=0, **kw: 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("GGPM_BENCH_ONE_DEVICE", raising=False)
    assert bench.launch_ranks(bench.parse_args(["--gpus", "8"]), ["--gpus", "8"]) == 0


def test_other_ranks_never_write_to_the_parents_stdout(stub, capfd):
    """`err` without a file descriptor (io.StringIO): ranks > 0 go to DEVNULL, not to the inherited stdout."""
    script, marker = stub
    out, err = io.StringIO(), io.StringIO()
    rc = launcher.run_ranks(script, ["ok", marker], 2, timeout=60, out=out, err=err)
    assert rc == 0 and len(out.getvalue().strip().splitlines()) == 1
    assert '"rank": 1' not in capfd.readouterr().out
