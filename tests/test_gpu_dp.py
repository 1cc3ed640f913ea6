"""GPU: the Trainer's data-parallel path with REAL device tensors — two ranks sharing cuda:0 over gloo
(NCCL needs one GPU per rank; the collective semantics exercised here are backend-independent):
replica broadcast incl. the unregistered conv1x1 layers, hook-launched async all-reduce of the big
fc_embed_attn slice, finish(), fused clip+Adam.  Checks: replicas stay identical, and the reduced
gradient equals the sum of the two ranks' individually computed gradients."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_gpu_trainer import _setup
    from videonavqa_amd.train import Trainer
    import videonavqa_amd.train as T
    model, stem, batches = _setup(seed=10 + rank)          # DIFFERENT initial weights per rank on purpose
    tr = Trainer(model, stem, lr=1e-3, world_size=world, rank=rank)
    tr.reducer.__init__(tr.fp, world, "sum", early_numel=4096)   # make fc_embed_attn.weight take the hook path
    assert len(tr.reducer.early) >= 1
    # 1) replicas identical after the start-up broadcast (incl. the frozen unregistered conv1x1 layers)
    flat0 = [torch.zeros_like(tr.fp.flat) for _ in range(world)]
    dist.all_gather(flat0, tr.fp.flat)
    assert torch.equal(flat0[0], flat0[1])
    c1 = torch.cat([t.reshape(-1) for t in model.extra_state_tensors().values()])
    g1 = [torch.zeros_like(c1) for _ in range(world)]
    dist.all_gather(g1, c1)
    assert torch.equal(g1[0], g1[1])
    # 2) reduced gradient == sum of per-rank gradients (capture the flat grad right before the update)
    captured = {}
    orig = T.K.clip_adam_step

    def spy(p, g, m, v, partial, step, lr, clip=1.0, **kw):
        captured["g"] = g.clone()
        return orig(p, g, m, v, partial, step, lr, clip, **kw)

    T.K.clip_adam_step = spy
    b = batches[rank]                                       # each rank its own minibatch
    # local gradient without any communication, from the same weights
    tr.reducer.enabled = False
    w_before = tr.fp.flat.clone()
    m_before, v_before, sc = tr.fp.m.clone(), tr.fp.v.clone(), tr.fp.step_count
    tr.step(*b)
    g_local = captured["g"].clone()
    tr.fp.flat.copy_(w_before); tr.fp.m.copy_(m_before); tr.fp.v.copy_(v_before); tr.fp.step_count = sc
    model.bn_init.reset_running_stats()
    tr.reducer.enabled = True
    loss, _ = tr.step(*b)
    g_red = captured["g"]
    parts = [torch.zeros_like(g_local) for _ in range(world)]
    dist.all_gather(parts, g_local)
    ref = parts[0] + parts[1]
    err = float((g_red - ref).abs().max() / (ref.abs().max() + 1e-12))
    assert err < 1e-4, err
    # 3) replicas still identical after the update
    flat1 = [torch.zeros_like(tr.fp.flat) for _ in range(world)]
    dist.all_gather(flat1, tr.fp.flat)
    assert torch.equal(flat1[0], flat1[1])
    assert bool(torch.isfinite(tr.fp.flat).all())
    open(os.path.join(out_dir, "ok%d" % rank), "w").write("%g" % err)
    dist.destroy_process_group()


def test_two_rank_dp_on_gpu_tensors(tmp_path):
    port = 29700 + (os.getpid() % 1500)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
