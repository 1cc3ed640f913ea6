#!/usr/bin/env python3
"""ON THE GPU BOX, diagnostic build libvnqa_psd3.so (bash tools/build_one_variant.sh psd3 conv_ps.hip -DVNQA_PS_DIAG=3; bf16 objects): the clock
the chip holds INSIDE the composed 5x5 conv's K loop (MI355X_MICROARCH.md, DVFS item 6): delta s_memtime / delta s_memrealtime x 100 MHz per
workgroup, read after >= 2 s of back-to-back stem passes; and the MFMA pipe's share of those cycles (K-steps x 112 MFMAs x 16 cycles per wave)."""
import argparse
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["VNQA_LIB"] = os.path.join(ROOT, "videonavqa_amd", "lib", "libvnqa_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "psd3"))
os.environ["VNQA_NO_REBUILD"] = "1"
os.environ["VNQA_HALF"] = "bf16"
import bench  # noqa: E402
from videonavqa_amd import _lib as L  # noqa: E402
from videonavqa_amd.models.common import FrameLayout  # noqa: E402

a = argparse.Namespace(batch=8, frames=35, height=224, width=224, precision="bf16", model="film_attn_pt", blocks=1, channels=512,
                       tail_channels=0)
stem = bench.build(a, torch.device("cuda"))[1]
clip = torch.rand(a.batch, 3, a.height, a.width, a.frames, device="cuda")
lay = FrameLayout([a.frames] * a.batch, a.frames, "cuda")
t0 = time.time()
n = 0
while time.time() - t0 < 3.0:
    for _ in range(10):
        stem.forward_clip(clip, lay.img_of, lay.n_img)
    torch.cuda.synchronize()
    n += 10
lib = L.lib()
words = 8 * 16384
buf = (ctypes.c_ulonglong * words)()
rc = lib.vnqa_ps_diag_stamps(buf, words)
s = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
s = s[s[:, 1] > 0]
clk = s[:, 0] / s[:, 1] * 100e6 / 1e9
ksteps = s[0, 3]
cyc = s[:, 0]
med = lambda v: float(np.median(v))
print("passes %d, workgroups stamped %d (rc %d), K-steps per tile %d" % (n, len(s), rc, ksteps))
print("in-kernel clock over the K loop: median %.3f GHz (p10 %.3f, p90 %.3f)" % (med(clk), np.percentile(clk, 10), np.percentile(clk, 90)))
print("K loop: median %.0f shader cycles = %.1f per K-step; the MFMA pipe needs %d per K-step (112 x 16): busy %.3f of the loop's cycles"
      % (med(cyc), med(cyc) / ksteps, 112 * 16, 112 * 16 * ksteps / med(cyc)))
print("phases of a workgroup (median us; s_memrealtime, 10 ns ticks): entry -> K loop %.2f | K loop %.2f (%.3f per K-step) | -> tile staged in LDS %.2f "
      "| -> stored %.2f | lifetime %.2f" % (med(s[:, 2]) / 100, med(s[:, 1]) / 100, med(s[:, 1]) / 100 / ksteps, med(s[:, 6] - s[:, 5]) / 100,
                                          med(s[:, 7] - s[:, 6]) / 100, med(s[:, 7] - s[:, 4]) / 100))
# the launch as a whole: first entry to last exit, against 7840 tiles on 256 CUs
span = (s[:, 7].max() - s[:, 4].min()) / 100
print("launch: %.1f us from the first entry to the last exit = %.2f us per round of 256 tiles; workgroup lifetime %.2f -> %.2f us per tile are "
      "between workgroups (dispatch, LDS / register allocation) or lost to the tail" % (span, span / (len(s) / 256.0), med(s[:, 7] - s[:, 4]) / 100,
                                                                                       span / (len(s) / 256.0) - med(s[:, 7] - s[:, 4]) / 100))
