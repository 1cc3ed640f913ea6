"""Prints CPU model / core count / torch thread settings of the box (context for bench.py's cpu_baseline)."""
import os, time, sys, torch
sys.path.insert(0, "/root/repo")
print("cpu_count", os.cpu_count(), "torch threads", torch.get_num_threads(), flush=True)
import torch.nn.functional as F
for nt in (8, 16, 32, 64):
    if nt > (os.cpu_count() or 1): break
    torch.set_num_threads(nt)
    x = torch.rand(2, 64, 224, 224); w = torch.rand(64, 64, 3, 3)
    F.conv2d(x, w, padding=1)
    t = time.time(); F.conv2d(x, w, padding=1); F.conv2d(x, w, padding=1)
    print("threads", nt, "conv1_2 2 frames: %.3f s" % ((time.time() - t) / 2), flush=True)
