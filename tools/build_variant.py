#!/usr/bin/env python3
"""Build an ALTERNATE copy of the HIP library with extra compiler flags for same-box A/B runs:
    python tools/build_variant.py diag -DVNQA_DIAG_SKIP_DMA
writes videonavqa_amd/lib/libvnqa_diag.so (git-ignored; select it with VNQA_LIB=<path>)."""
import os
import subprocess
import sys

from videonavqa_amd import build as B

name, extra = sys.argv[1], sys.argv[2:]
out_dir = os.path.join("/tmp/build", name)
os.makedirs(out_dir, exist_ok=True)
procs, objs = [], []
for src in B.sources():
    obj = os.path.join(out_dir, os.path.basename(src) + ".o")
    cmd = [B.HIPCC] + extra + B.FLAGS + B.PER_FILE_FLAGS.get(os.path.basename(src), []) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]
    procs.append((src, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
    objs.append(obj)
for src, p in procs:
    if p.wait() != 0:
        raise SystemExit("hipcc failed on %s" % src)
lib = os.path.join(B.LIBDIR, "libvnqa_%s.so" % name)
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
print("built", lib)
