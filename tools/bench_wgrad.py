"""Micro-benchmark of the trunk's weight-gradient launch (280 images x 14x14, 512 -> 512, 3x3, bf16)."""
import torch, json
from videonavqa_amd import kernels as K
N, h, w, C = 280, 14, 14, 512
x = torch.zeros(N, h+2, w+2, C, dtype=torch.bfloat16, device="cuda"); x[:, 1:-1, 1:-1] = torch.randn(N, h, w, C, device="cuda").to(torch.bfloat16)
dy = torch.zeros_like(x); dy[:, 1:-1, 1:-1] = torch.randn(N, h, w, C, device="cuda").to(torch.bfloat16)
for _ in range(3): K.conv2d_wgrad(x, dy, 9)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): K.conv2d_wgrad(x, dy, 9)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("wgrad 512x512 3x3 280 img: %.3f ms  %.0f TFLOP/s (valid pixels)" % (ms, 2.0 * N * h * w * C * C * 9 / ms / 1e9))
