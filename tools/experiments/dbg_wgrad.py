import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videonavqa_amd import kernels as K, _lib as L
L.set_half("f16")
dt = L.half_dtype()
for cfg in ((40, 14, 14, 256, 256, 9), (6, 14, 14, 256, 256, 9), (20, 14, 14, 256, 256, 9), (40, 14, 14, 256, 256, 1), (280, 14, 14, 512, 512, 9)):
    N, H, W, Cin, Cout, taps = cfg
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    dy = torch.randn(N, Cout, H, W, generator=g).cuda()
    xp, dyp = K.nchw_to_nhwc(x, dt, c_pad=Cin), K.nchw_to_nhwc(dy, dt, c_pad=Cout)
    a, _ = K.conv2d_wgrad(xp, dyp, taps)
    b, _ = K.conv2d_wgrad(xp, dyp, taps, eight_waves=True)
    d = (a - b).abs()
    print(cfg, "max diff", float(d.max()), "ref max", float(b.abs().max()))
    if float(d.max()) > 1e-3:
        print("  per tap:", [round(float(d[:, t].max()), 3) for t in range(taps)])
        bad = (d > 1e-3)
        print("  bad fraction", float(bad.float().mean()), "co range", bad.any(2).any(1).nonzero().flatten()[[0, -1]].tolist(),
              "ci range", bad.any(1).any(0).nonzero().flatten()[[0, -1]].tolist())
        # which pixels? contribution test: zero all but one image
        for n0 in (0, N // 2, N - 1):
            dy1 = torch.zeros_like(dyp); dy1[n0] = dyp[n0]
            a1, _ = K.conv2d_wgrad(xp, dy1, taps); b1, _ = K.conv2d_wgrad(xp, dy1, taps, eight_waves=True)
            print("   only image", n0, "max diff", float((a1 - b1).abs().max()))
