"""Micro-benchmark of the fused conv1_1+conv1_2 kernel (280 frames x 224x224): 16x16 vs 32x16 (wide) tile shapes.
argv[1] = value of the `relu` field (1, or the timing-only flags 256 / 512 / 1024 of a -DVNQA_DIAG_SKIP_DMA build)."""
import torch
from videonavqa_amd import kernels as K, _lib as L
import ctypes
B, T, H, W = 8, 35, 224, 224
N = B * T
clip = torch.rand(B, 3, H, W, T, device="cuda")
img_of = torch.arange(N, dtype=torch.int32, device="cuda").view(T, B).t().contiguous().view(-1)
w1 = torch.randn(64, 3, 3, 3, device="cuda") * 0.3; b1 = torch.randn(64, device="cuda") * 0.1
w2 = torch.randn(64, 64, 3, 3, device="cuda") / 24; b2 = torch.randn(64, device="cuda") * 0.1
wt = K.pack_conv_weight(w2, torch.bfloat16, c_out_pad=64, c_in_pad=64)
img4 = K.clip_to_nhwc4(clip, img_of, N)
out = torch.zeros(N, H // 2 + 2, W // 2 + 2, 64, dtype=torch.bfloat16, device="cuda")
import sys
FLAGS = int(sys.argv[1]) if len(sys.argv) > 1 else 1
def run(tile):
    d = L.ConvDesc(L.BF16, N, H, W, 64, 64, 64, 9, 1, 1, FLAGS, 1, tile, 0, 0)
    L.check(L.lib().vnqa_conv_first_c64_fwd(ctypes.byref(d), L.ptr(img4), L.ptr(w1.contiguous()), L.ptr(b1), L.ptr(wt), L.ptr(b2), None, None, L.ptr(out), L.stream()), "x")
for tile in (3, 0, 3, 0):
    run(tile); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run(tile)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("tile %d (%s): %.3f ms  %.0f TFLOP/s" % (tile, "16x16" if tile == 3 else "32x16 wide", ms, 2.0 * N * H * W * 64 * 64 * 9 / ms / 1e9))
