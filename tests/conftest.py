import os
import socket
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_DP = {"procs": None}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle's CPU runs: as many threads as this process may use (a GPU box shows 256 CPUs under a quota of 16, and
    # torch's default of one thread per visible core has them fight over that share)
    import torch
    from ggpm_amd.launcher import host_cores
    torch.set_num_threads(min(16, host_cores()))


def _have_gpu():
    import torch
    return torch.cuda.device_count() > 0       # does not initialise HIP in this process (is_available() would)


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_collection_finish(session):
    """Start the two data-parallel rank processes of test_aa_data_parallel_gpu.py while this process has not yet made
    a single HIP call (nothing before this point does: collection only imports modules)."""
    wanted = [it for it in session.items if it.nodeid.startswith("tests/test_aa_data_parallel_gpu.py")
              or "test_aa_data_parallel_gpu" in it.nodeid]
    if not wanted or not _have_gpu() or session.config.option.collectonly:
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    tmp = tempfile.mkdtemp(prefix="ggpm_dp_")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
        log = os.path.join(tmp, "rank%d.log" % rank)
        with open(log, "w") as f:
            procs.append((subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_rank_worker.py")], env=env,
                                           stdout=f, stderr=subprocess.STDOUT, cwd=ROOT), log))
    _DP["procs"] = procs


@pytest.fixture
def dp_rank_processes():
    return _DP["procs"]


def pytest_sessionfinish(session, exitstatus):
    for p, _ in _DP["procs"] or []:
        if p.poll() is None:
            p.kill()
