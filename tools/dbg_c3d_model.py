import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vnqa_oracle as O
from videonavqa_amd.models import VideoOnlyCNN3D
torch.manual_seed(3)
B, D, H, W = 8, 16, 64, 64
m = VideoOnlyCNN3D(7, fc6_in_features=128 * 2 * 2, precision="bf16")
Wd = {k: v.detach().clone() for k, v in m.state_dict().items()}
x = torch.rand(B, 3, D, H, W); y = torch.randint(0, 7, (B,))
names = [k for k, v in Wd.items() if v.is_floating_point() and "running" not in k]
for k in names: Wd[k].requires_grad_(True)
ref = O.video_only_cnn3d_forward(Wd, x, training=True)
gref = dict(zip(names, torch.autograd.grad(F.cross_entropy(ref, y, reduction="sum"), [Wd[k] for k in names])))
m = m.cuda().train()
for mode in ("0", "1"):
    os.environ["VNQA_CNN3D_GENERIC"] = mode
    m.zero_grad()
    out = m(x.cuda()); F.cross_entropy(out, y.cuda(), reduction="sum").backward()
    rep = {}
    for k, p in m.named_parameters():
        a, r = p.grad.cpu().flatten(), gref[k].flatten()
        rep[k] = "%.3f/%.2f" % (float(torch.dot(a, r) / (a.norm() * r.norm() + 1e-12)), float(a.norm() / (r.norm() + 1e-12)))
    print("generic" if mode == "1" else "fused  ", "logits err %.3f" % float((out.detach().cpu() - ref.detach()).abs().max() / ref.abs().max()), rep)
