#!/usr/bin/env python3
"""Per-step time of the wide-LSTM step kernels (csrc/lstm_wide.hip) at MACNetwork's two shapes: the tail LSTM
(hidden 1536, 35 steps) and the bidirectional question LSTM (hidden 512, 24 steps; one direction per call and both
directions per launch).  GPU box:  PYTHONPATH=$PWD python tools/bench_lstm_wide.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from videonavqa_amd import kernels as K


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3     # us per call


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    for H, T, B in ((1536, 35, 8), (512, 24, 8)):
        lens = sorted([max(1, T - 3 * i) for i in range(B)], reverse=True)
        bsz = [sum(1 for v in lens if v > t) for t in range(T)]
        xg = torch.randn(T, B, 4 * H, device=dev) * 0.1
        w = torch.randn(4 * H, H, device=dev) / H ** 0.5
        wt = w.t().contiguous()
        hs, cs, gates = K.lstm_wide_fwd(xg, w, bsz)
        dhs = torch.randn_like(hs)
        f = timed(lambda: K.lstm_wide_fwd(xg, w, bsz))
        b = timed(lambda: K.lstm_wide_bwd(wt, bsz, gates, cs, dhs))
        print("H %4d T %2d B %d: fwd %6.1f us/step, bwd %6.1f us/step (incl. the zero-fills of the call)" % (H, T, B, f / T, b / T))
        if hasattr(K, "lstm_wide_bidir_fwd"):
            xg2, w2 = xg.clone(), w.clone()
            out = K.lstm_wide_bidir_fwd(xg, xg2, w, w2, bsz)
            f2 = timed(lambda: K.lstm_wide_bidir_fwd(xg, xg2, w, w2, bsz))
            b2 = timed(lambda: K.lstm_wide_bidir_bwd(wt, wt, bsz, out[2][0], out[2][1], out[1][0], out[1][1], dhs, dhs))
            print("              both directions per launch: fwd %6.1f us/step, bwd %6.1f us/step" % (f2 / T, b2 / T))


if __name__ == "__main__":
    main()
