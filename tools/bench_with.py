#!/usr/bin/env python3
"""A/B shim: bench.py with module attributes of the package preset.
  python tools/bench_with.py ops.HEAD_SPLIT_OUT=0 [stem.X=1 ...] -- <bench.py arguments>"""
import importlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
cut = sys.argv.index("--")
for word in sys.argv[1:cut]:
    k, v = word.split("=")
    mod, name = k.rsplit(".", 1)
    m = importlib.import_module("videonavqa_amd." + mod)
    setattr(m, name, (bool(int(v)) if isinstance(getattr(m, name), bool) else type(getattr(m, name))(int(v))))
import bench  # noqa: E402

sys.argv = ["bench.py"] + sys.argv[cut + 1:]
bench.main()
