"""Micro-benchmark of fc_embed_attn's per-step pieces at the headline size (280 images x 14x14x512 -> 128): weight packs,
forward GEMM (+ split-K reduce), dX by the generic path (transposed pack + igemm) and by vnqa_fc_dx, dW, un-pack."""
import torch
from videonavqa_amd import kernels as K
from videonavqa_amd import _lib as L

L.set_half("bf16")
dt = torch.bfloat16
N, C, h, w, rows = 280, 512, 14, 14, 128
kn = (h + 2) * (w + 2) * C
wt = torch.randn(rows, C * h * w, device="cuda") * 0.01
x = torch.randn(N, kn, device="cuda").to(dt)
dout = torch.randn(N, rows, device="cuda").to(dt)


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


nat, nat_t = K.pack_fc_weight(wt, C, h, w, C, rows, dt)
print("pack nat only          %7.1f us" % t(lambda: K.pack_fc_weight(wt, C, h, w, C, rows, dt, want_t=False)))
print("pack nat + nat_t       %7.1f us" % t(lambda: K.pack_fc_weight(wt, C, h, w, C, rows, dt)))
print("forward gemm_nt        %7.1f us" % t(lambda: K.gemm_nt(x, nat)))
print("dX generic (igemm)     %7.1f us" % t(lambda: K.gemm_nt(dout, nat_t)))
print("dX vnqa_fc_dx          %7.1f us  (%.2f TB/s on nat + dx)" % ((lambda us: (us, (nat.numel() + N * kn) * 2 / us / 1e6))(t(lambda: K.fc_dx(dout, nat)))))
dw = K.gemm_tn(dout, x)
print("dW gemm_tn             %7.1f us" % t(lambda: K.gemm_tn(dout, x)))
print("unpack dW              %7.1f us" % t(lambda: K.unpack_fc_wgrad(dw, rows, C, h, w, C)))
a, b = K.fc_dx(dout, nat).float(), K.gemm_nt(dout, nat_t).float()
print("max |fc_dx - generic| / max|generic| = %.2e" % float((a - b).abs().max() / b.abs().max()))
