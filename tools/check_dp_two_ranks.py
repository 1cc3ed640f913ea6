#!/usr/bin/env python3
"""GPU diagnostic: two gloo ranks on one GPU, early-hook gradient all-reduce on — prints, per parameter, the reduced
gradient against the sum of the two ranks' own gradients (and against the patterns a double / mismatched all-reduce would
give) plus the slice contents at hook time.  This is what located the double-firing reducer hook of round 2."""
import os, sys, torch
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")

def worker(rank, world, port):
    sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_gpu_trainer import _setup
    from videonavqa_amd.train import Trainer
    import videonavqa_amd.train as T
    model, stem, batches = _setup(seed=10 + rank)
    tr = Trainer(model, stem, lr=1e-3, world_size=world, rank=rank)
    tr.reducer.__init__(tr.fp, world, "sum", early_numel=int(os.environ.get("EARLY", "4096")))
    cap = {}
    orig = T.K.clip_adam_step
    def spy(p, g, m, v, partial, step, lr, clip=1.0, **kw):
        cap["g"] = g.clone()
        return orig(p, g, m, v, partial, step, lr, clip, **kw)
    T.K.clip_adam_step = spy
    b = batches[rank]
    tr.reducer.enabled = False
    w0, m0, v0, sc = tr.fp.flat.clone(), tr.fp.m.clone(), tr.fp.v.clone(), tr.fp.step_count
    tr.step(*b)
    g_local = cap["g"].clone()
    tr.fp.flat.copy_(w0); tr.fp.m.copy_(m0); tr.fp.v.copy_(v0); tr.fp.step_count = sc
    model.bn_init.reset_running_stats()
    tr.reducer.enabled = True
    pre = {}
    orig_hook = tr.reducer._hook
    def hook(p):
        a_, b_ = tr.reducer.early[p]
        torch.cuda.synchronize()
        pre[p] = tr.fp.grad[a_:b_].clone()
        orig_hook(p)
    tr.reducer._hook = hook
    for p in tr.reducer.early:
        sk = getattr(p, "_vnqa_grad_sink", None)
        if sk is not None:
            sk.on_ready = (lambda q=p: hook(q))
    tr.step(*b)
    for p, (a_, b_) in tr.reducer.early.items():
        name = [n for n, q in model.named_parameters() if q is p][0]
        x = pre.get(p)
        print("rank %d %-24s at hook: vs local %.2e  vs 2*local %.2e" % (rank, name,
              float((x - g_local[a_:b_]).abs().max() / g_local[a_:b_].abs().max()),
              float((x - 2 * g_local[a_:b_]).abs().max() / g_local[a_:b_].abs().max())), flush=True)
    g_red = cap["g"]
    parts = [torch.zeros_like(g_local) for _ in range(world)]
    dist.all_gather(parts, g_local)
    ref = parts[0] + parts[1]
    if rank == 0:
        off = 0
        for n, p in model.named_parameters():
            if not p.requires_grad: continue
            k = p.numel()
            a, r = g_red[off:off+k], ref[off:off+k]
            l0, l1 = parts[0][off:off+k], parts[1][off:off+k]
            e = lambda x: float((a - x).abs().max() / (r.abs().max() + 1e-12))
            print("%-26s early=%d  vs ref %.2e | ref+l0 %.2e | ref+l1 %.2e | 2ref %.2e | l0 %.2e | l1 %.2e" % (
                n, p in tr.reducer.early, e(r), e(r + l0), e(r + l1), e(2 * r), e(l0), e(l1)), flush=True)
            off += k
    dist.destroy_process_group()

if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29811), nprocs=2, join=True)
