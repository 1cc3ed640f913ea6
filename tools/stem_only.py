#!/usr/bin/env python3
"""The frozen stem alone on the chip (B*T frames, conv1_1 .. conv32), N iterations — the target of a
`rocprofv3 --kernel-trace --stats -- python3 tools/stem_only.py` per-kernel breakdown of the stem.
Prints the event-timed average ms per pass."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from videonavqa_amd.models.common import FrameLayout  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=35)
    ap.add_argument("--height", type=int, default=224)
    ap.add_argument("--width", type=int, default=224)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--precision", default="fp16h")
    ap.add_argument("--set", nargs="*", default=[], help="module attributes of videonavqa_amd.stem preset for an A/B, e.g. MEAN_SHIFT=0 SPLIT_DEPTH=1")
    a = ap.parse_args()
    import videonavqa_amd.stem as S
    for word in a.set:
        k, v = word.split("=")
        setattr(S, k, type(getattr(S, k))(int(v)))
    a.model, a.blocks, a.channels, a.tail_channels = "film_attn_pt", 1, 512, 0
    if a.precision in ("fp16", "fp16h"):
        from videonavqa_amd import _lib as L
        L.set_half("f16")
    stem = bench.build(a, torch.device("cuda"))[1]
    clip = torch.randn(a.batch, 3, a.height, a.width, a.frames, device="cuda")
    lay = FrameLayout([a.frames] * a.batch, a.frames, "cuda")
    for _ in range(3):
        stem.forward_clip(clip, lay.img_of, lay.n_img)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        stem.forward_clip(clip, lay.img_of, lay.n_img)
    e1.record()
    torch.cuda.synchronize()
    print("stem alone: %.3f ms per pass of %d frames" % (e0.elapsed_time(e1) / a.iters, lay.n_img))


if __name__ == "__main__":
    main()
