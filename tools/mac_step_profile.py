#!/usr/bin/env python3
"""Is `bench.py --model mac` bound by the launch thread or by the chip?  Full Trainer steps (stem + MACNetwork + clip + Adam):
(a) wall time per step, (b) pure enqueue time per step when the GPU queue is drained first, (c) the same steps with the
forward/backward repeated on a quarter of the frames (GPU work / 4, host work unchanged), (d) cProfile by own time."""
import argparse, cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench as Bn
from videonavqa_amd.train import Trainer

def main():
    dev = torch.device("cuda", 0)
    for frames in (35, 9):
        a = argparse.Namespace(precision="bf16", model="mac", batch=8, frames=frames, height=224, width=224, blocks=1, channels=512,
                               tail_channels=0)
        model, stem, _, _ = Bn.build(a, dev)
        tr = Trainer(model, stem)
        batches = [Bn.synth_batch(a, 0, dev, i) for i in range(2)]
        def step(i):
            clip, q, vl, ql, y = batches[i % 2]
            return tr.step(clip, q, vl, ql, y)
        for i in range(4): step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10): step(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("frames %2d: %.2f ms/step wall (enqueue of 10 steps %.2f ms/step, drain %.2f ms)" % (frames, (t2 - t0) * 100, (t1 - t0) * 100, (t2 - t1) * 1e3))
        enq = 0.0
        for i in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); step(i); enq += time.perf_counter() - t0
        print("           enqueue of ONE step on an idle queue: %.2f ms" % (enq / 5 * 1e3))
        # the trunk alone (features precomputed) and the stem alone, drained
        clip, q, vl, ql, y = batches[0]
        native, v_sorted, perm = tr.extract_features(clip, vl)
        qd, qld, yd = q[perm.to(dev)], ql[perm], y[perm.to(dev)]
        def trunk():
            model.train()
            loss = tr.loss_fn(model(native, qd, v_sorted, qld), yd)
            loss.backward()
        for _ in range(3): trunk()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): trunk()
        torch.cuda.synchronize(); print("           trunk alone (fwd+bwd, no optimizer): %.2f ms" % ((time.perf_counter() - t0) * 100))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): tr.extract_features(clip, vl)
        torch.cuda.synchronize(); print("           stem alone: %.2f ms" % ((time.perf_counter() - t0) * 100))
        if frames == 35:
            pr = cProfile.Profile(); pr.enable()
            for i in range(3): step(i)
            pr.disable(); torch.cuda.synchronize()
            pstats.Stats(pr).sort_stats("tottime").print_stats(28)

if __name__ == "__main__":
    main()
