"""Start one fresh process per GPU for a data-parallel run (one node) and relay rank 0's result line.

The reference is single-device (ggpm/nnutils.py:9-10, vae_train.py:78-83); the data-parallel form runs the same loop
body on every rank (forward, backward, gradient all-reduce in front of clip_grad_norm_ / Adam).  This module is what
``python bench.py --gpus N`` uses when nobody else (torch.distributed.run) has set up the ranks.

Rules it keeps:
  * the parent makes NO HIP call: it may import torch, it only ever counts devices; a process that has initialised the
    GPU must not start other programs on this pool, and nothing here re-executes a running process;
  * every child is a fresh interpreter in its own process group with RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT set (the container hostname may not resolve);
  * rank 0's stdout is relayed verbatim (the ONE JSON line), the other ranks' stdout goes to stderr;
  * any child that exits non-zero fails the run at once (the others are ended by process group, exact PIDs only), and so
    does the timeout.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    """Environment of rank `rank` of `world` on this node (what torch.distributed.run would have set)."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               GROUP_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GGPM_LAUNCHED_BY="ggpm_amd.launcher")
    # the host driver of this pool only supports dmabuf IPC (RCCL / tensor sharing across processes)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # main + high-priority atom stream + second stream + copy stream + the collective's own: more than the default 4
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    return env


def rank_commands(script: str, argv: Sequence[str], world: int, port: int, python: Optional[str] = None,
                  base_env: Optional[Dict[str, str]] = None):
    """[(command list, environment)] for the `world` rank processes: the same script, the same arguments."""
    exe = python or sys.executable
    return [([exe, script] + list(argv), rank_env(r, world, port, base_env)) for r in range(world)]


def _end(procs: List[subprocess.Popen], grace: float = 5.0) -> None:
    """End exactly the processes started here (each leads its own process group)."""
    for sig in (signal.SIGTERM, signal.SIGKILL):
        alive = [p for p in procs if p.poll() is None]
        if not alive:
            return
        for p in alive:
            try:
                os.killpg(p.pid, sig)
            except (ProcessLookupError, PermissionError):
                pass
        t_end = time.time() + grace
        while time.time() < t_end and any(p.poll() is None for p in alive):
            time.sleep(0.05)


def run_ranks(script: str, argv: Sequence[str], world: int, timeout: float = 1500.0, python: Optional[str] = None,
              base_env: Optional[Dict[str, str]] = None, out=None, err=None) -> int:
    """Run `world` ranks of `script argv`; write rank 0's stdout to `out`; -> exit code (0 only if every rank's is 0).

    124 on timeout (like coreutils' timeout), otherwise the first non-zero exit code seen."""
    out = out if out is not None else sys.stdout
    err = err if err is not None else sys.stderr
    port = free_port()
    procs: List[subprocess.Popen] = []
    captured: List[bytes] = []
    try:
        err_fd = err.fileno()
    except Exception:
        err_fd = None
    for rank, (cmd, env) in enumerate(rank_commands(script, argv, world, port, python, base_env)):
        procs.append(subprocess.Popen(cmd, env=env, stdin=subprocess.DEVNULL,
                                      stdout=subprocess.PIPE if rank == 0 else (err_fd if err_fd is not None else None),
                                      stderr=err_fd, start_new_session=True))

    def drain():                              # rank 0's result line can exceed a pipe buffer
        captured.append(procs[0].stdout.read())

    reader = threading.Thread(target=drain, daemon=True)
    reader.start()
    t_end, code = time.time() + timeout, 0
    try:
        while True:
            states = [p.poll() for p in procs]
            bad = [(r, s) for r, s in enumerate(states) if s not in (None, 0)]
            if bad:
                code = bad[0][1] if bad[0][1] > 0 else 128 - bad[0][1]
                print("[launcher] rank %d exited with %d; ending the other ranks" % bad[0], file=err, flush=True)
                break
            if all(s == 0 for s in states):
                break
            if time.time() > t_end:
                code = 124
                print("[launcher] %d ranks did not finish within %.0f s; ending them" % (world, timeout), file=err, flush=True)
                break
            time.sleep(0.05)
    finally:
        _end(procs)
        reader.join(timeout=10.0)
    text = b"".join(captured).decode(errors="replace")
    if code == 0:
        out.write(text)
        out.flush()
    elif text:
        print(text, file=err, flush=True)
    return code


def host_cores():
    """CPU share of this process: affinity, capped by the cgroup quota (a GPU box grants ~16 per GPU)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, int(os.environ.get("GGPM_CPU_THREADS", "64"))))
