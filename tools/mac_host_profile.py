"""Host-side (launch thread) cost of one MACNetwork training step: enqueue time of forward / backward + a cProfile."""
import time, torch, cProfile, pstats
from videonavqa_amd.models import MACNetwork
from videonavqa_amd.models.common import FrameLayout, NativeFeatures
from videonavqa_amd import kernels as K
dev = torch.device("cuda")
B, T = 8, 35
model = MACNetwork(n_vocab=134, dim=512, embed_hidden=128, classes=70, precision="bf16").to(dev)
v_lens = torch.full((B,), T); q_lens = torch.randint(5, 26, (B,))
lay = FrameLayout(v_lens, T, dev)
x = torch.randn(lay.n_img, 16, 16, 512, device=dev).to(torch.bfloat16)
x[:, 0] = 0; x[:, -1] = 0; x[:, :, 0] = 0; x[:, :, -1] = 0
native = NativeFeatures(x, lay, 512, 14, 14)
q = torch.randint(1, 134, (B, 56), device=dev); y = torch.randint(0, 70, (B,), device=dev)
model.train()
def step():
    t0 = time.perf_counter()
    out = model(native, q, v_lens, q_lens)
    loss = torch.nn.functional.cross_entropy(out, y, reduction="sum")
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1
for _ in range(3): step()
torch.cuda.synchronize()
f = b = 0
for _ in range(5):
    a, c = step(); f += a; b += c
torch.cuda.synchronize()
print("host enqueue: forward %.2f ms, backward %.2f ms" % (f / 5 * 1e3, b / 5 * 1e3))
pr = cProfile.Profile(); pr.enable(); step(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
