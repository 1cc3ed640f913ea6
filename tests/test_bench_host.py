"""CPU: host-side pieces of bench.py — FLOP accounting, the CPU-baseline child, the self-spawning `--gpus N` guard."""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
ENV = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")


def test_flop_accounting_matches_survey_8d():
    sys.path.insert(0, ROOT)
    import bench as Bn
    assert abs(Bn.stem_flops_per_frame(224, 224) / 1e9 - 37.167) < 2e-3
    assert abs(Bn.stem_flops_per_frame(160, 208) / 1e9 - 24.652) < 2e-3
    fwd, fb = Bn.trunk_flops_per_frame(196, 512, 512, 1, 128)
    assert abs(fwd / 1e9 - 1.978) < 2e-3
    clip = 35 * (Bn.stem_flops_per_frame(224, 224) + fb)
    assert abs(clip / 1e9 - 1476.2) < 0.2
    # the composed conv11 . conv12 pair executes fewer FLOPs than the reference formulation, never more
    assert Bn.stem_executed_flops_per_frame(224, 224, True) < Bn.stem_flops_per_frame(224, 224)
    assert Bn.stem_executed_flops_per_frame(224, 224, False) == Bn.stem_flops_per_frame(224, 224)


def test_cpu_baseline_child_prints_cumulative_lines():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline-only", "--cpu-batch", "2",
                        "--frames", "3", "--height", "32", "--width", "48", "--cpu-steps", "2"], env=ENV,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-500:]
    lines = [json.loads(x) for x in r.stdout.strip().splitlines() if x.startswith("{")]
    assert len(lines) == 2
    for i, leg in enumerate(lines, 1):
        assert leg["kind"] == "port" and leg["unit"] == "clips/s" and leg["value"] > 0 and leg["cores"] >= 1
        assert "%d timed step(s) after 1 warm-up" % i in leg["sample"] and "2 clips x 3 frames 32x48" in leg["sample"]


def test_gpus_n_without_launcher_refuses_when_gpus_are_missing():
    """`bench.py --gpus 2` starts its own ranks; with fewer GPUs than ranks it must say so instead of silently
    running world=1 and printing n_gpus: 1 (missing #3)."""
    env = {k: v for k, v in ENV.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "only 0 GPU(s) visible" in r.stderr and "{" not in r.stdout


def test_cpu_child_dumps_the_oracle_logits_of_a_reproducible_workload(tmp_path):
    """Round 4: the CPU leg writes the oracle's forward logits of its minibatch; the GPU parent regenerates the SAME
    weights and minibatch from the seed (bench.oracle_workload, no oracle import) to run the HIP path on them."""
    import torch
    sys.path.insert(0, ROOT)
    import argparse
    import bench as Bn
    out = str(tmp_path / "logits.pt")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline-only", "--cpu-batch", "2", "--frames", "3",
                        "--height", "32", "--width", "48", "--cpu-steps", "1", "--cpu-logits-out", out], env=ENV,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-500:]
    d = torch.load(out)
    assert d["logits"].shape == (2, 70) and d["batch"] == 2 and sorted(d["perm"].tolist()) == [0, 1]
    assert bool(torch.isfinite(d["logits"]).all())
    a = argparse.Namespace(height=32, width=48, frames=3, channels=512, blocks=1)
    w1 = Bn.oracle_workload(a, 2)
    w2 = Bn.oracle_workload(a, 2)
    for x, y in zip(w1[:3], w2[:3]):                       # the three weight dictionaries, bit for bit
        assert x.keys() == y.keys() and all(torch.equal(x[k], y[k]) for k in x)
    assert all(torch.equal(p, q) for p, q in zip(w1[3], w2[3]))
    assert w1[2]["fc_embed_attn.weight"].shape == (128, (32 // 16) * (48 // 16) * 512)
    # the oracle run on the regenerated workload reproduces the child's logits (same process-independent arithmetic)
    from oracle import vnqa_oracle as O
    W_vgg, W_od, W, (clip, q, v_lens, q_lens, y) = w1
    with torch.no_grad():
        feats = O.stem_forward(clip, W_vgg, W_od)
        v2, q2, vl2, ql2, _, perm = O.sort_batch(feats, q, v_lens, q_lens, y)
        ref = O.film_attn_forward({k: v.clone() for k, v in W.items()}, v2, q2, vl2, ql2, training=True)
    assert torch.equal(perm, d["perm"]) and float((ref - d["logits"]).abs().max()) < 1e-5 * float(ref.abs().max())


def test_bench_parser_knows_the_round4_modes():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], env=ENV, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    for token in ("--mode", "fp16h", "--plumbing", "--clip-dtype", "--no-eval-leg", "--no-robustness"):
        assert token in r.stdout, token


def test_gpus_8_plumbing_run_spawns_eight_ranks_and_prints_one_line():
    """`bench.py --gpus 8`'s own spawn path with 8 ranks — no 8-GPU node is within reach, so the host side of the
    multi-rank path (fresh child interpreters, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* contract, gloo rendezvous at 127.0.0.1,
    replica broadcast, the Trainer's overlapped gradient reducer, barrier-bracketed timed region with MAX over ranks, the `comm`
    block, rank 0's single JSON line relayed by the parent) runs on CPU tensors with a stand-in module."""
    env = {k: v for k, v in ENV.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--plumbing", "--steps", "4", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    lines = [x for x in r.stdout.strip().splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["global_batch"] == 64 and d["config"]["parallelism"] == "dp8"
    c = d["comm"]
    assert c["ranks"] == 8 and c["backend"] == "gloo" and c["allreduce_alone_ms"] > 0 and c["early_reduced_parameters"] >= 1
    assert "exposed_comm_ms_per_step" in c and "ms_per_step_without_collectives" in c
    assert d["replicas_identical"] is True
    # a rank whose environment disagrees with --gpus refuses instead of running a smaller world
    bad = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--plumbing"], env=bad, capture_output=True, text=True,
                        timeout=120)
    assert r2.returncode != 0 and "{" not in r2.stdout
