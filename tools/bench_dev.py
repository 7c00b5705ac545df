#!/usr/bin/env python3
"""bench.py with development settings changed first (ggpm_amd/_dev.py): python tools/bench_dev.py NAME=VALUE [...] -- <bench.py arguments>

    python tools/bench_dev.py DECODE_TABLES=False -- --only-vae
"""
import ast
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ggpm_amd import _dev      # noqa: E402

args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
for item in args[:cut]:
    name, value = item.split("=", 1)
    assert hasattr(_dev, name), name
    setattr(_dev, name, ast.literal_eval(value))
sys.argv = [os.path.join(ROOT, "bench.py")] + args[cut + 1:]
import bench                   # noqa: E402

bench.main()
