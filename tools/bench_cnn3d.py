#!/usr/bin/env python3
"""Config 2 (BASELINE.json): v_only_cnn3d on 1 MI355X, synthetic 16x3x112x112 clips, bs=32, fwd+bwd+Adam."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from videonavqa_amd.models import VideoOnlyCNN3D

def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    B = 32
    torch.manual_seed(0)
    m = VideoOnlyCNN3D(70, fc6_in_features=128 * 1 * 3 * 3, precision=prec).cuda().train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    x = torch.rand(B, 3, 16, 112, 112, device="cuda")
    y = torch.randint(0, 70, (B,), device="cuda")
    def step():
        loss = torch.nn.functional.cross_entropy(m(x), y, reduction="sum")
        loss.backward(); opt.step(); opt.zero_grad()
        return loss
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): loss = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    flops = B * 3 * (2.0 * 16 * 112 * 112 * 3 * 64 * 27 + 2.0 * 16 * 56 * 56 * 64 * 128 * 27 + 2.0 * 4 * 14 * 14 * 128 * 128 * 27)
    print(json.dumps({"config": "v_only_cnn3d bs=32 16x3x112x112 %s" % prec, "clips_per_s": round(B / dt, 1),
                      "ms_per_step": round(dt * 1e3, 2), "conv_tflops_fwd_bwd": round(flops / dt / 1e12, 1),
                      "loss": round(float(loss), 3)}))

if __name__ == "__main__":
    main()
