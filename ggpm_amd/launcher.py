"""Start one fresh process per GPU for a data-parallel run (one node) and relay rank 0's result line.

The reference is single-device (ggpm/nnutils.py:9-10, vae_train.py:78-83); the data-parallel form runs the same loop
body on every rank (forward, backward, gradient all-reduce in front of clip_grad_norm_ / Adam).  This module is what
``python bench.py --gpus N`` uses when nobody else (torch.distributed.run) has set up the ranks.

Rules it keeps:
  * the parent makes NO HIP call and no torch.cuda call at all: it counts GPUs from the kernel driver's topology files
    (``visible_gpu_count``; torch.cuda.device_count() falls back to hipGetDeviceCount when amdsmi is not usable, which
    starts the runtime and keeps /dev/kfd open in the parent for the whole run).  Starting fresh child processes is always
    fine; what is forbidden on this pool is REPLACING a process that has initialised the GPU (os.exec*), and nothing here
    re-executes a running process;
  * every child is a fresh interpreter in its own process group with RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT set (the container hostname may not resolve);
  * rank 0's stdout is relayed verbatim (the ONE JSON line), the other ranks' stdout goes to stderr (or nowhere when the
    caller's ``err`` has no file descriptor -- never to the parent's stdout);
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


def visible_gpu_count(env: Optional[Dict[str, str]] = None, topology: str = "/sys/class/kfd/kfd/topology/nodes") -> int:
    """GPUs this process would see, WITHOUT touching HIP: the amdkfd topology lists one node per agent, GPU nodes are those
    with ``simd_count > 0``; ``HIP_VISIBLE_DEVICES`` / ``ROCR_VISIBLE_DEVICES`` / ``CUDA_VISIBLE_DEVICES`` (comma lists) narrow
    it.  -> 0 without the driver's topology directory (no amdkfd: no GPU), -1 when it exists but cannot be read (the
    caller then lets rank 0 find out)."""
    env = os.environ if env is None else env
    n = 0
    try:
        nodes = sorted(os.listdir(topology), key=lambda d: int(d) if d.isdigit() else 1 << 30)
    except FileNotFoundError:
        return 0
    except OSError:
        return -1
    for d in nodes:
        try:
            with open(os.path.join(topology, d, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for key in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(key)
        if v is not None:
            ids = [x for x in v.split(",") if x.strip() != ""]
            n = min(n, len(ids))
    return n


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
        # ranks > 0 never inherit the parent's stdout (it carries ONE JSON line): the caller's stderr, or nowhere
        other_out = err_fd if err_fd is not None else subprocess.DEVNULL
        procs.append(subprocess.Popen(cmd, env=env, stdin=subprocess.DEVNULL,
                                      stdout=subprocess.PIPE if rank == 0 else other_out,
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
