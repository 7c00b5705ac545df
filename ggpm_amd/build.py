"""Build the gfx950 shared library in-tree:  python -m ggpm_amd.build

hipcc cross-compiles without a GPU; the resulting ``ggpm_amd/libggpm_hip.so`` is git-ignored but travels
with the tree (it is what the GPU box loads).  No cmake, no JIT cache.
"""
from __future__ import annotations

import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libggpm_hip.so")
SOURCES = ["capi.hip", "graph.hip", "gemm.hip", "gather.hip", "mpn_gru.hip", "mpn_lstm.hip", "encoder.hip", "losses.hip", "decode.hip", "tree_level.hip", "schedule.hip"]
HEADERS = ["common.h", "tile_mma.h", os.path.join("..", "..", "include", "ggpm_hip.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    env = os.environ.get("HIPCC")
    if env:
        return env
    return "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, variant: str = "", defines=()) -> str:
    """Compile every source for gfx950 and link ggpm_amd/libggpm_hip.so.

    ``variant`` / ``defines``: a second build of the same sources with compile-time tuning or ablation switches
    (``-DGGPM_...``), written to ggpm_amd/libggpm_hip.<variant>.so and selected at run time with GGPM_LIB_PATH -- how the
    A/B figures of DESIGN.md were measured without a runtime switch in the product."""
    lib = LIB_PATH if not variant else LIB_PATH.replace(".so", ".%s.so" % variant)
    if not variant and not force and not needs_build():
        return LIB_PATH
    objs, procs = [], []
    bdir = os.path.join(PKG_DIR, "build" if not variant else "build_" + variant)
    os.makedirs(bdir, exist_ok=True)
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17"] + list(defines) + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(obj)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    # python -m ggpm_amd.build [--force] [--variant NAME -DGGPM_X=1 ...]
    name = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    print(build(force="--force" in sys.argv, variant=name, defines=[a for a in sys.argv[1:] if a.startswith("-D")]))
