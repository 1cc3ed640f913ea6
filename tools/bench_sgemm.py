#!/usr/bin/env python3
"""vnqa_sgemm on the shapes the library issues (MAC reasoning step, out_linear, FiLM generator) against torch.matmul (rocBLAS):
per-call time of a back-to-back chain of 200 calls.  (the round-2 plain-FMA kernel it replaced was removed in round 4)."""
import torch
from videonavqa_amd import kernels as K

def timed(fn, it=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3

for (m, n, k, form) in ((280, 512, 512, "nt"), (280, 512, 512, "nn"), (512, 512, 280, "tn"), (8, 70, 4480, "nt"), (280, 1024, 128, "nt"),
                        (2800, 512, 300, "nt"), (280, 2048, 512, "nt"), (2240, 2048, 512, "nt")):
    if form == "nt":
        a, b = torch.randn(m, k, device="cuda"), torch.randn(n, k, device="cuda")
        ours = lambda: K.linear_nt(a, b)
        ref = lambda: a @ b.t()
    elif form == "nn":
        a, b = torch.randn(m, k, device="cuda"), torch.randn(k, n, device="cuda")
        ours = lambda: K.matmul_nn(a, b)
        ref = lambda: a @ b
    else:
        a, b = torch.randn(k, m, device="cuda"), torch.randn(k, n, device="cuda")
        ours = lambda: K.matmul_tn(a, b)
        ref = lambda: a.t() @ b
    err = float((ours() - ref()).abs().max() / ref().abs().max())
    print("%-2s m=%5d n=%5d k=%5d  ours %7.1f us  rocBLAS %7.1f us  rel err %.1e" % (form, m, n, k, timed(ours), timed(ref), err))
