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
    # 4) BatchNorm running statistics are RANK-LOCAL (the reference has no SyncBN: every rank normalises with its own
    #    minibatch): after a step on different minibatches they differ between ranks, nothing reduces them, and
    #    the checkpoint policy is "rank 0's copy" — model.state_dict() on rank 0 holds exactly rank 0's buffers
    rm = model.bn_init.running_mean.detach().clone()
    rms = [torch.zeros_like(rm) for _ in range(world)]
    dist.all_gather(rms, rm)
    assert not torch.equal(rms[0], rms[1])
    if rank == 0:
        assert torch.equal(model.state_dict()["bn_init.running_mean"], rms[0])
    open(os.path.join(out_dir, "ok%d" % rank), "w").write("%g" % err)
    dist.destroy_process_group()


def test_two_rank_dp_on_gpu_tensors(tmp_path):
    port = 29700 + (os.getpid() % 1500)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def _run_bench(extra_env, argv, timeout=900):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, (r.stdout[-400:], r.stderr[-1500:])
    lines = [x for x in r.stdout.strip().splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 prints ONE JSON line
    return json.loads(lines[0])


SMALL = ["--steps", "2", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline", "--no-parity", "--batch", "2",
         "--frames", "4", "--height", "64", "--width", "96"]


def test_bench_gpus2_spawns_its_own_ranks_single_device_gloo():
    """`bench.py --gpus 2` with no launcher: the parent starts both ranks itself (here on ONE GPU over gloo — the test
    hook for 1-GPU boxes), rank 0 prints the single JSON line with n_gpus = 2 and the comm block."""
    out = _run_bench({"VNQA_DIST_BACKEND": "gloo", "VNQA_SINGLE_DEVICE": "1"}, ["--gpus", "2"] + SMALL)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    c = out["comm"]
    assert c["ranks"] == 2 and c["backend"] == "gloo" and c["allreduce_alone_ms"] > 0
    assert c["ms_per_step_without_collectives"] > 0 and "exposed_comm_ms_per_step" in c
    assert out["value"] > 0 and out["scaling"] == "weak"


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL, one rank per GPU)")
def test_bench_gpus2_rccl_two_gpus():
    """Two ranks on two MI355X over RCCL through the self-spawning entry point; also A/Bs the persistent conv kernels'
    CU reservation (VNQA_STEM_RESERVE_CUS: a CU-masked stem stream, the persistent conv kernels sized for it), which exists for exactly
    this co-scheduling with RCCL's kernels."""
    base = _run_bench({}, ["--gpus", "2", "--steps", "5", "--warmup", "2", "--repeats", "1", "--no-cpu-baseline", "--no-parity"])
    assert base["n_gpus"] == 2 and base["comm"]["backend"] == "rccl" and base["comm"]["ranks"] == 2
    resv = _run_bench({"VNQA_STEM_RESERVE_CUS": "32"}, ["--gpus", "2", "--steps", "5", "--warmup", "2", "--repeats", "1",
                                                       "--no-cpu-baseline", "--no-parity"])
    print("dp2 clips/s: reserve 0 -> %.1f, reserve 32 -> %.1f; exposed comm %.3f / %.3f ms"
          % (base["value"], resv["value"], base["comm"]["exposed_comm_ms_per_step"], resv["comm"]["exposed_comm_ms_per_step"]))
    assert resv["n_gpus"] == 2


def _val_worker(rank, world, port, out_dir):
    """val_epoch on `world` gloo ranks sharing cuda:0: every rank evaluates its share of the split (ShardedBatchSampler), the
    shards are merged by one gather; the returned average loss and the merged predictions must equal the single-process run's."""
    import io
    import contextlib
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch.nn as nn
    from torch.utils.data import DataLoader
    from videonavqa_amd.eval import q_and_v_eval as E
    from videonavqa_amd.eval.dataset import SyntheticVNQADataset
    from videonavqa_amd.models import FiLMAttnPretrainedStem, ObjDetectCNN
    from videonavqa_amd.stem import FrozenStem, VGGFront
    from videonavqa_amd.train import Trainer
    torch.manual_seed(0)
    B, H, W, T = 2, 64, 96, 35
    vgg, od = VGGFront("fp32"), ObjDetectCNN(5, 64, 8, 0, True, True, precision="fp32")
    with torch.no_grad():
        for conv in vgg.features.values():
            nn.init.kaiming_uniform_(conv.weight, a=1.0)
    model = FiLMAttnPretrainedStem(B, 16, 70, num_input_channels=64, num_res_block_channels=64, hidden_size=16, at_hidden_size=16,
                                   max_num_frames=T, vocab_size=134, spatial_size=(H // 16) * (W // 16), precision="fp32").cuda()
    stem = FrozenStem(vgg.cuda().eval(), od.cuda().eval(), "fp32")
    tr = Trainer(model, stem, world_size=world, rank=rank, feature_channels=64, collectives=False)
    data = SyntheticVNQADataset(11, H, W, seed=99)                        # 5 full batches of 2 + one short batch (dropped)
    loader = DataLoader(data, num_workers=0, batch_sampler=E.ShardedBatchSampler(len(data), B, rank, world))
    args = E.build_parser().parse_args(["--model", "film_attn_pt", "--batch_size", str(B), "--num_classes", "70"])
    captured = {}
    real = E.gather_eval_shards

    def spy(*a, **k):
        out = real(*a, **k)
        captured["merged"] = out
        return out
    E.gather_eval_shards = spy
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        avg = E.val_epoch(args, tr, loader, torch.device("cuda", 0), rank, world)
    (y_t, y_p), loss, n = captured["merged"]
    torch.save({"avg": avg, "y_t": y_t, "y_p": y_p, "loss": loss, "n": n, "own": len(loader), "printed": buf.getvalue()},
               os.path.join(out_dir, "val_w%d_r%d.pt" % (world, rank)))
    if world > 1:
        dist.destroy_process_group()


def test_sharded_validation_on_two_ranks_equals_the_single_process_run(tmp_path):
    import numpy as np
    port = 29900 + (os.getpid() % 1000)
    _val_worker(0, 1, port, str(tmp_path))                                 # the single-process reference, in this process
    mp.spawn(_val_worker, args=(2, port + 1, str(tmp_path)), nprocs=2, join=True)
    one = torch.load(tmp_path / "val_w1_r0.pt", weights_only=False)
    assert one["n"] == 10 and one["own"] == 5 and "Validation:" in one["printed"]
    r0, r1 = (torch.load(tmp_path / ("val_w2_r%d.pt" % r), weights_only=False) for r in (0, 1))
    assert (r0["own"], r1["own"]) == (3, 2)                                # batches 0, 2, 4 / 1, 3: nobody evaluates the whole split
    for r in (r0, r1):
        assert r["n"] == 10 and np.array_equal(r["y_t"], one["y_t"]) and np.array_equal(r["y_p"], one["y_p"])
        assert abs(r["avg"] - one["avg"]) < 1e-5 * max(1.0, abs(one["avg"]))
    assert "Validation:" in r0["printed"] and "Validation:" not in r1["printed"]      # rank 0 prints the epoch line
