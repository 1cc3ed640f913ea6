#!/usr/bin/env python3
"""Persistent LSTM chain micro-bench: the headline workload's question encoder (8 samples x 25 tokens x 35 repeats,
H = 128), forward and BPTT, µs per cell."""
import torch

from videonavqa_amd import kernels as K

B, H, Lq, R = 8, 128, 25, 35
S = Lq * R
dev = "cuda"
xg = torch.randn(B, Lq, 4 * H, device=dev) * 0.5
w = torch.randn(4 * H, H, device=dev) / H ** 0.5
ql = torch.full((B,), Lq, dtype=torch.int32, device=dev)
h0 = torch.zeros(B, H, device=dev)
c0 = torch.zeros(B, H, device=dev)


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


hs, gates, hN, cN = K.lstm_seq_fwd(xg, w, ql, h0, c0, R, S)
dhs = torch.randn_like(hs) * 0.1
f = timed(lambda: K.lstm_seq_fwd(xg, w, ql, h0, c0, R, S))
b = timed(lambda: K.lstm_seq_bwd(w, ql, c0, gates, dhs, None, None, R))
print("lstm fwd %.3f ms (%.2f us/cell)  bwd %.3f ms (%.2f us/cell)  checksum %.6f" % (f, f * 1e3 / S, b, b * 1e3 / S, float(hs.double().sum())))
