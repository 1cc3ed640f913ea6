#!/usr/bin/env python3
"""bench.py — FiLM-attn video-QA training throughput (clips/s) on N MI355X, weak scaling.

One "step" = the hot path on one per-GPU minibatch of synthetic clips:
  frozen VGG-16[:10] + ObjDetectCNN(512) stem forward over all B*T frames, FiLMAttnPretrainedStem
  forward + backward, gradient all-reduce (N>1), global-norm clip, Adam  (eval/q_and_v_eval.py:84-139).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: this process starts the N
                                                                 ranks itself, one per GPU, and relays rank 0's line)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

The JSON line also carries
  roofline      dominant kernel (the frozen stem's composed 5x5 conv): executed FLOPs / HIP-event launch time vs the dense bf16 MFMA peak
  cpu_baseline  the CPU oracle (a port of the reference) on this host's cores: B=8 clips, 1 warm-up + 3 timed steps
  parity        bf16 benchmark precision vs the exact-f32 parity precision on the SAME weights and batches at this very
                workload: max logits error relative to max |logit|, argmax agreement, loss error, fp32-mode clips/s
  repeats       the timed K-step region is run `--repeats` times (default 3); `value` is the MEDIAN region
  comm          (N > 1) ranks, all-reduce payload / time, and the exposed (non-overlapped) communication per step
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "video-QA clips/sec (35×224² frames) FiLM-attn fwd+bwd, 1→8 MI355X"
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3

# Algorithmic FLOPs (2*MAC) per frame of the stem igemm layers with Cout=512, at 224x224
# conv11 3.6994 + conv12 14.7975 + conv21 3.6994 + conv22 3.6994 + conv31 0.9248 + conv32 0.9248
def stem512_flops_per_frame(H, W):
    px = lambda d: (H // d) * (W // d)
    f = lambda p, ci, co: 2.0 * p * ci * co * 9
    return (f(px(4), 128, 512) + f(px(4), 512, 512) + 2 * f(px(8), 512, 512) + 2 * f(px(16), 512, 512))


def stem_flops_per_frame(H, W):
    px = lambda d: (H // d) * (W // d)
    f = lambda p, ci, co: 2.0 * p * ci * co * 9
    return (f(px(1), 3, 64) + f(px(1), 64, 64) + f(px(2), 64, 128) + f(px(2), 128, 128)) + stem512_flops_per_frame(H, W)


def stem_executed_flops_per_frame(H, W, composed):
    """FLOPs the stem actually executes per frame.  With the conv11 . conv12 pair evaluated as one composed 5x5 conv
    (stem.py: FrozenStem._compose_pair) that pair costs 25*128*512 MACs per pixel plus the border-ring GEMMs instead of
    (9*128 + 9*512)*512: the ALGORITHMIC figure (reference formulation) stays the work unit of the metric."""
    if not composed:
        return stem_flops_per_frame(H, W)
    h, w = H // 4, W // 4
    pair = 2.0 * h * w * 512 * (9 * 128 + 9 * 512)
    ring = 2.0 * (2 * (w + 2) + 2 * h) * 512 * 9 * 128 + 2.0 * (2 * w + 2 * h) * 512 * 3 * 512
    return stem_flops_per_frame(H, W) - pair + 2.0 * h * w * 128 * 512 * 25 + ring


def trunk_flops_per_frame(S, C_in, C, blocks, at):
    conv_init = 2.0 * S * C_in * C * 9
    block = 2.0 * S * C * C * (1 + 9)
    fc = 2.0 * S * C * at
    fwd = conv_init + blocks * block + fc
    return fwd, 3 * fwd - conv_init       # fwd, fwd+bwd (no dgrad through conv_init's input)


COMPOSED_STEM = [True]     # set by build(): whether the frozen stem evaluates conv11 . conv12 as one composed 5x5 conv


def build(args, device):
    from videonavqa_amd.models import (FiLMAttnPretrainedStem, FiLMGlobalPoolingPretrainedStem, ObjDetectCNN,
                                       TimeMultiHopFiLMPretrainedStem)
    from videonavqa_amd.stem import FrozenStem, VGGFront
    import torch.nn as nn
    torch.manual_seed(int(getattr(args, "seed", 0)))     # identical replicas on every rank (tools/error_budget.py --seed: other weights)
    prec = args.precision
    mprec = prec
    vgg = VGGFront(prec)
    od = ObjDetectCNN(27, 512, 1024, 0, True, True, precision=prec)   # eval/utils.py:43-48
    with torch.no_grad():
        for conv in vgg.features.values():
            nn.init.kaiming_uniform_(conv.weight, a=1.0)
            conv.bias.normal_(0, 0.02)
        for m in od.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1.0)
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.8, 1.2)
    S = (args.height // 16) * (args.width // 16)
    if args.model == "film_attn_pt":
        model = FiLMAttnPretrainedStem(args.batch, 128, 70, num_res_blocks=args.blocks,
                                       num_res_block_channels=args.channels, max_num_frames=args.frames,
                                       spatial_size=S, precision=mprec)
    elif args.model == "mac":              # MACNetwork on the same stem (eval/q_and_v_eval.py:288-293)
        from videonavqa_amd.models import MACNetwork
        model = MACNetwork(n_vocab=134, dim=args.channels, embed_hidden=128, classes=70, max_num_frames=args.frames,
                           precision=mprec)
    elif args.model == "film_gp_pt":       # BASELINE.json config 3
        model = FiLMGlobalPoolingPretrainedStem(args.batch, 128, 70, num_res_blocks=args.blocks,
                                                num_res_block_channels=args.channels, spatial_size=S, precision=mprec,
                                                **({"num_tail_channels": args.tail_channels} if getattr(args, "tail_channels", 0) else {}))
    else:                                  # BASELINE.json config 5 (use --frames 70)
        model = TimeMultiHopFiLMPretrainedStem(args.batch, 128, 70, num_res_blocks=args.blocks,
                                               num_res_block_channels=args.channels, spatial_size=S, precision=mprec,
                                               **({"num_tail_channels": args.tail_channels} if getattr(args, "tail_channels", 0) else {}))
    vgg, od, model = vgg.to(device).eval(), od.to(device).eval(), model.to(device)
    # (args.calibration: frames [N, 3, H, W] of the deployment's kind for the stem's weight rounding and channel means — tests / tools)
    stem = FrozenStem(vgg, od, prec, calibration=getattr(args, "calibration", "auto"), split_features=args.model != "mac",
                      split_depth=getattr(model, "stem_split_depth", None))      # (pooling heads ask for split depth 3)
    COMPOSED_STEM[0] = stem.composed is not None
    return model, stem, vgg, od


def synth_batch(args, rank, device, index=0):
    """Synthetic minibatch `index` of rank `rank` (SURVEY 8d inputs; index 0 keeps the round-1/2 seed 1234 + rank)."""
    g = torch.Generator(device="cpu").manual_seed(1234 + rank + 7919 * index)
    B, T = args.batch, args.frames
    if getattr(args, "clip_dtype", "f32") == "u8":
        # raw 8-bit pixels k, valued float32(k / 255.0) like the reference's loader forms them (eval/dataset.py:91): what
        # VNQADataset(uint8_video=True) delivers — a quarter of the bytes per clip
        clip = torch.randint(0, 256, (B, 3, args.height, args.width, T), generator=g, dtype=torch.uint8)
    else:
        clip = torch.rand(B, 3, args.height, args.width, T, generator=g)
    q_lens = torch.randint(5, 26, (B,), generator=g)
    q = torch.randint(1, 134, (B, 56), generator=g)
    q = q * (torch.arange(56).unsqueeze(0) < q_lens.unsqueeze(1)).long()
    v_lens = torch.full((B,), T, dtype=torch.long)
    y = torch.randint(0, 70, (B,), generator=g)
    return clip.to(device), q.to(device), v_lens, q_lens, y.to(device)


def oracle_workload(args, Bs):
    """Weights (reference GPU-flavour state_dict names, fp32, seeded) and ONE minibatch of the metric's shape for the CPU leg.
    Pure tensor generation (no oracle import): the GPU parent regenerates the identical dictionaries from the same seed to
    run its exact-f32 precision on them (`parity.oracle_full_size_*`), the CPU child feeds them to the oracle."""
    S = (args.height // 16) * (args.width // 16)
    C = args.channels
    g = torch.Generator(device="cpu").manual_seed(99)

    def rnd(*shape, fan=None):
        return torch.randn(*shape, generator=g) / (fan or shape[-1]) ** 0.5

    W_vgg = {}
    for idx, (ci, co) in zip((0, 2, 5, 7), ((3, 64), (64, 64), (64, 128), (128, 128))):
        W_vgg["features.%d.weight" % idx] = rnd(co, ci, 3, 3, fan=ci * 9)
        W_vgg["features.%d.bias" % idx] = torch.zeros(co)
    W_od = {}
    for name, ci in (("conv11", 128), ("conv12", 512), ("conv21", 512), ("conv22", 512), ("conv31", 512), ("conv32", 512)):
        W_od[name + ".weight"] = rnd(512, ci, 3, 3, fan=ci * 9)
        W_od[name + ".bias"] = torch.zeros(512)
    for name, c in (("bn_input", 128), ("bn1", 512), ("bn2", 512), ("bn3", 512)):
        W_od[name + ".weight"], W_od[name + ".bias"] = torch.ones(c), torch.zeros(c)
        W_od[name + ".running_mean"], W_od[name + ".running_var"] = torch.zeros(c), torch.ones(c)
    at, Hq, E = 128, 128, 128
    W = {"embed.weight": rnd(134, E, fan=1), "conv_init.weight": rnd(C, 512, 3, 3, fan=4608),
         "conv_init.bias": torch.zeros(C), "bn_init.weight": torch.ones(C), "bn_init.bias": torch.zeros(C),
         "bn_init.running_mean": torch.zeros(C), "bn_init.running_var": torch.ones(C),
         "bn_init.num_batches_tracked": torch.tensor(0),
         "film_layer.0.weight_ih_l0": rnd(4 * Hq, E), "film_layer.0.weight_hh_l0": rnd(4 * Hq, Hq),
         "film_layer.0.bias_ih_l0": torch.zeros(4 * Hq), "film_layer.0.bias_hh_l0": torch.zeros(4 * Hq),
         "film_layer.1.weight": rnd(2 * C * args.blocks, Hq), "film_layer.1.bias": torch.ones(2 * C * args.blocks) * 0.5,
         "fc_embed_attn.weight": rnd(at, S * C), "fc_embed_attn.bias": torch.zeros(at),
         "fc_attn_1.weight": rnd(1, at), "fc_attn_1.bias": torch.zeros(1),
         "fc_hidden_attn.weight": rnd(1, at), "fc_hidden_attn.bias": torch.zeros(1),
         "lstm_attn.weight_ih": rnd(4 * at, at), "lstm_attn.weight_hh": rnd(4 * at, at),
         "lstm_attn.bias_ih": torch.zeros(4 * at), "lstm_attn.bias_hh": torch.zeros(4 * at),
         "out_linear.weight": rnd(70, args.frames * at), "out_linear.bias": torch.zeros(70)}
    for k in range(args.blocks):
        W["film_pipeline.%d.weight" % k] = rnd(C, C, 3, 3, fan=C * 9)
        W["film_pipeline.%d.bias" % k] = torch.zeros(C)
        W["conv1x1_layers.%d.weight" % k] = rnd(C, C, 1, 1, fan=C)
        W["conv1x1_layers.%d.bias" % k] = torch.zeros(C)
    clip = torch.rand(Bs, 3, args.height, args.width, args.frames, generator=g)
    q_lens = torch.randint(5, 26, (Bs,), generator=g)
    q = torch.randint(1, 134, (Bs, 56), generator=g)
    v_lens = torch.full((Bs,), args.frames, dtype=torch.long)
    y = torch.randint(0, 70, (Bs,), generator=g)
    return W_vgg, W_od, W, (clip, q, v_lens, q_lens, y)


def cpu_baseline_child(args):
    """Runs in a CPU-only child process (`bench.py --cpu-baseline-only`): the oracle — a CPU
    restatement of the reference, kind 'port' — timed on this host's cores on the metric's own minibatch
    (B = 8 clips x T frames, 1 warm-up + 3 timed full steps: stem fwd + FiLM-attn fwd/bwd + clip + Adam).
    A cumulative JSON line is printed after every timed step, so a parent that runs out of patience still has a
    measurement of the steps completed so far.  With --cpu-logits-out the oracle's train-mode forward logits on the
    initial weights (the first thing the warm-up step computes anyway) are written there for the parent's
    HIP-vs-oracle comparison at the benchmark's own size."""
    from oracle import vnqa_oracle as O
    torch.manual_seed(0)
    nthreads = torch.get_num_threads()
    Bs = args.cpu_batch
    W_vgg, W_od, W, (clip, q, v_lens, q_lens, y) = oracle_workload(args, Bs)
    adam = O.AdamState(list(W))

    if args.cpu_logits_out:
        with torch.no_grad():
            feats = O.stem_forward(clip, W_vgg, W_od)
            v2, q2, vl2, ql2, y2, _perm = O.sort_batch(feats, q, v_lens, q_lens, y)
            W0 = {k: v.clone() for k, v in W.items()}          # (train-mode BN advances the running statistics in place)
            logits = O.film_attn_forward(W0, v2, q2, vl2, ql2, training=True)
        torch.save({"logits": logits.detach().clone(), "batch": Bs, "perm": _perm.clone()}, args.cpu_logits_out)
        del feats, v2, W0

    def one():
        feats = O.stem_forward(clip, W_vgg, W_od)
        v2, q2, vl2, ql2, y2, _ = O.sort_batch(feats, q, v_lens, q_lens, y)
        O.train_step("film_attn_pt", W, v2, q2, vl2, ql2, y2, adam, 1e-4)

    one()
    t0 = time.time()
    for n in range(1, args.cpu_steps + 1):
        one()
        dt = (time.time() - t0) / n
        print(json.dumps({"value": round(Bs / dt, 4), "unit": "clips/s", "cores": nthreads, "kind": "port",
                          "sample": "%d clips x %d frames %dx%d, full step (stem fwd + FiLM-attn fwd/bwd + clip + Adam), "
                                    "%d timed step(s) after 1 warm-up, torch CPU fp32, %d threads"
                                    % (Bs, args.frames, args.height, args.width, n, nthreads)}), flush=True)


def cpu_baseline(argv, limit_s=420, logits_out=None):
    """Launch the CPU leg as a child process BEFORE this process touches the GPU; bounded by a timeout.  The child
    prints a cumulative line after every timed step: on a timeout the last complete line is what is reported.
    `logits_out`: file the child writes the oracle's forward logits of its minibatch to (parity.oracle_full_size_*)."""
    import subprocess
    env = dict(os.environ)
    env["HIP_VISIBLE_DEVICES"] = ""
    env["CUDA_VISIBLE_DEVICES"] = ""
    extra = ["--cpu-logits-out", logits_out] if logits_out else []
    proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"] + argv + extra, env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    note = ""
    try:
        out, err = proc.communicate(timeout=limit_s)
    except subprocess.TimeoutExpired:
        proc.kill()                                   # exactly the child started above
        out, err = proc.communicate()
        note = " (stopped at the %d s budget)" % limit_s
    for line in reversed((out or "").strip().splitlines()):
        if line.startswith("{"):
            leg = json.loads(line)
            leg["sample"] += note
            return leg
    return {"value": None, "unit": "clips/s", "cores": None, "kind": "port",
            "sample": "cpu leg produced no measurement%s: %s" % (note, (err or "").strip()[-300:])}


def fp16_leg(args, precision="fp16"):
    """The fp16-storage precision (libvnqa_hip_f16.so: the same kernels with IEEE fp16 as the 16-bit format, loss-scaled
    backward) measured on the same workload in a CHILD process — one 16-bit storage format per process — after this
    process's own measurement: throughput of a short run and its parity block against the exact-f32 precision."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--precision", precision, "--no-cpu-baseline", "--no-fp16-leg", "--no-eval-leg", "--no-robustness",
           "--repeats", "1",
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--batch", str(args.batch), "--frames", str(args.frames),
           "--height", str(args.height), "--width", str(args.width), "--blocks", str(args.blocks), "--channels", str(args.channels)]
    env = {k: v for k, v in os.environ.items() if k not in ("VNQA_HALF",)}
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
        line = [x for x in r.stdout.strip().splitlines() if x.startswith("{")]
        if not line:
            return {"error": (r.stderr or "no output").strip()[-300:]}
        d = json.loads(line[-1])
        p = d.get("parity", {})
        what = ("bench.py --precision fp16 (plain fp16 storage, fp32 accumulate, dynamic loss scale from 2^10; stem weights second-order "
                "rounded and activations mean-shifted like in every 16-bit precision; no split operands in the trunk): inside 1e-3 on the "
                "measured kinds with a thin margin — 0.65-0.94e-3 worst of twelve by weight seed and kind of clip "
                "(profiles/r06_error_by_kind.txt); %d timed steps, child process") % d["steps"]
        if precision == "bf16":
            what = ("bench.py --precision bf16: BASELINE.json's storage dtype (bf16 storage, fp32 accumulate; stem weights second-order "
                    "rounded and activations mean-shifted like in every 16-bit precision) — the round 1-4 headline, NOT tolerance-compliant "
                    "(logits ~4e-3 of exact fp32 in round 6, 7e-3 before); %d timed steps, child process") % d["steps"]
        key = precision + "_logits_rel_err"
        return {"what": what, "precision": precision,
                # the precision's own full line: the same fields the top-level line carries, measured the same way
                "metric": d["metric"], "value": d["value"], "unit": d["unit"], "dtype": d["dtype"], "steps": d["steps"],
                "warmup": d["warmup"], "ms_per_step": d["ms_per_step"], "roofline": d["roofline"],
                "stem_alone_ms": d["config"].get("stem_alone_ms"), "stem_alone_mfma_util": d["config"].get("stem_alone_mfma_util"),
                "clips_per_s": d["value"], "roofline_frac": d["roofline"]["frac"],
                key: p.get(key), key + "_per_batch": p.get(key + "_per_batch"),
                precision + "_logits_rel_l2_err_per_batch": p.get(precision + "_logits_rel_l2_err_per_batch"),
                "argmax_equal_at_init": p.get("argmax_equal_at_init"), "twelve_minibatches": p.get("twelve_minibatches"),
                "loss_rel_err": p.get("loss_rel_err"),
                "grad_rel_l2_err": p.get("grad_rel_l2_err"), "after_fit": p.get("after_fit")}
    except subprocess.TimeoutExpired:
        return {"error": "%s leg exceeded 420 s" % precision}


def bench_cnn3d(args):
    """BASELINE config 2 (`--model v_only_cnn3d`): VideoOnlyCNN3D on 1 MI355X, synthetic 16x3x112x112 clips, bs = 32 unless --batch is
    given explicitly, forward + backward + fused clip+Adam over the flat parameter buffer; one JSON line in the same schema.
    Single GPU only; `roofline` prices the whole step's algorithmic conv FLOPs (per-kernel figures: profiles/r03_cnn3d.md)."""
    from videonavqa_amd import kernels as K, ops
    from videonavqa_amd.models import VideoOnlyCNN3D
    from videonavqa_amd.train import FlatParams
    assert args.gpus == 1, "config 2 is a single-GPU ladder rung"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    B = args.batch
    D, H, W = 16, 112, 112
    torch.manual_seed(0)
    model = VideoOnlyCNN3D(70, fc6_in_features=128 * (D // 16) * (H // 32) * (W // 32), precision=args.precision).to(dev).train()
    fp = FlatParams(model.parameters())
    g = torch.Generator().manual_seed(7)
    batches = [(torch.rand(B, 3, D, H, W, generator=g).to(dev), torch.randint(0, 70, (B,), generator=g).to(dev))
               for _ in range(max(args.minibatches, 1))]

    def step(i):
        x, y = batches[i % len(batches)]
        loss = ops.cross_entropy(model(x), y, reduction="sum")
        loss.backward()
        fp.clip_adam_step(1e-4, 1e30)
        return loss
    for i in range(args.warmup):
        step(i)
    regions = []
    for _ in range(args.repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss = step(i)
        torch.cuda.synchronize()
        regions.append(time.perf_counter() - t0)
    dt = sorted(regions)[len(regions) // 2]
    conv = 3 * B * (2.0 * D * H * W * 3 * 64 * 27 + 2.0 * D * (H // 2) * (W // 2) * 64 * 128 * 27
                    + 2.0 * (D // 4) * (H // 8) * (W // 8) * 128 * 128 * 27)
    peak = PEAK_BF16_TFLOPS if args.precision in ("bf16", "fp16") else PEAK_F32_TFLOPS
    tf = conv * args.steps / dt / 1e12
    print(json.dumps({
        "metric": "v_only_cnn3d clips/sec (16x3x112x112 clips) fwd+bwd+Adam, 1 MI355X (BASELINE config 2)", "value": round(B * args.steps / dt, 1),
        "unit": "clips/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"bf16": "bf16", "fp16": "f16", "fp32": "f32"}[args.precision], "data": "synthetic",
        "repeats": {"n": len(regions), "value_is": "median region", "clips_per_s": [round(B * args.steps / r, 1) for r in regions]},
        "config": {"workload": "VideoOnlyCNN3D training step (bn_input, 3 x [Conv3d, ReLU, MaxPool3d, BatchNorm3d], 3 FC + 2 BatchNorm1d), "
                               "bs=%d, %dx3x%dx%d clips; %s" % (B, D, H, W, "fused 16-bit path (csrc/cnn3d.hip)" if model._fast_ok(batches[0][0])
                                                                 else "generic path"),
                   "global_batch": B, "minibatches_rotated": len(batches), "final_loss": round(float(loss), 4),
                   "gflop_per_clip": round(conv / B / 1e9, 1)},
        "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4), "traffic": None,
                     "kernel": "whole step: algorithmic FLOPs of the three Conv3d layers (forward + dgrad + wgrad) / step time; the three "
                               "conv2 GEMMs are the MFMA-bound part, the rest is HBM-bound passes (per-kernel split: profiles/r03_cnn3d.md)"},
        "cpu_baseline": None}), flush=True)


def plumbing_rank(args):
    """`--plumbing` (never a benchmark): ONE rank of the multi-rank path on CPU tensors over gloo — everything of `--gpus N` that is
    not a kernel: the launcher's environment contract, process-group start-up at 127.0.0.1, replica broadcast, the Trainer's own
    flat buffers and overlapped gradient reducer (early slice + finish()), the timed-region protocol (barrier on both sides, MAX
    over ranks), the `comm` block and rank 0's ONE JSON line.  A small torch module stands in for the trunk (the product has no CPU
    kernels, by design); the update is plain SGD on the flat buffer.  Lets the 8-rank plumbing run where no 8-GPU node exists."""
    import torch.distributed as dist
    import torch.nn as nn
    from videonavqa_amd.train import FlatParams, OverlappedGradReducer, sync_replicas
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    assert world == args.gpus and 0 <= int(os.environ["LOCAL_RANK"]) < world and os.environ["MASTER_ADDR"] == "127.0.0.1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1000 + rank)                     # replicas start DIFFERENT: the broadcast has to make them equal
    model = nn.Sequential(nn.Linear(64, 512), nn.Tanh(), nn.Linear(512, 70))
    sync_replicas(list(model.state_dict().values()))
    fp = FlatParams(model.parameters())
    reducer = OverlappedGradReducer(fp, world, "sum", early_numel=4096)
    g = torch.Generator().manual_seed(1234 + rank)     # own minibatch per rank, as synth_batch
    x, y = torch.randn(args.batch, 64, generator=g), torch.randint(0, 70, (args.batch,), generator=g)
    loss_fn = nn.CrossEntropyLoss(reduction="sum")

    def run_step():
        loss = loss_fn(model(x), y)
        loss.backward()
        reducer.finish()
        fp.flat.sub_(1e-3 * fp.grad)
        fp.zero_grad()
        return loss

    def timed():
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = run_step()
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), loss
    for _ in range(args.warmup):
        run_step()
    dt, loss = timed()
    for _ in range(2):
        dist.all_reduce(fp.grad)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(5):
        dist.all_reduce(fp.grad)
    dist.barrier()
    ar_ms = (time.perf_counter() - t0) / 5 * 1e3
    fp.zero_grad()
    reducer.enabled = False
    dt_nc, _ = timed()
    reducer.enabled = True
    sync_replicas([fp.flat])                            # (the collective-free region let the replicas drift: re-join, then check)
    run_step()
    gathered = [torch.zeros_like(fp.flat) for _ in range(world)]
    dist.all_gather(gathered, fp.flat)
    same = all(torch.equal(gathered[0], t) for t in gathered[1:])
    if rank == 0:
        print(json.dumps({"metric": "PLUMBING (CPU tensors over gloo, a stand-in module; not a benchmark)", "value": round(args.batch * world * args.steps / dt, 1),
                          "unit": "stand-in samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "plumbing", "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                                     "final_loss": round(float(loss), 4)},
                          "comm": {"ranks": world, "backend": "gloo", "allreduce_payload_mb": round(fp.n * 4 / 1e6, 3),
                                   "allreduce_alone_ms": round(ar_ms, 3), "ms_per_step_without_collectives": round(dt_nc / args.steps * 1e3, 3),
                                   "exposed_comm_ms_per_step": round((dt - dt_nc) / args.steps * 1e3, 3), "early_reduced_parameters": len(reducer.early)},
                          "replicas_identical": bool(same)}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not same:
        raise SystemExit("plumbing: replicas differ after the reduced step")


def spawn_ranks(n, argv, plumbing=False):
    """`bench.py --gpus N` without a launcher: start N rank processes (one per GPU) from THIS process, which has not
    touched the GPU, with the torchrun environment contract; rank 0's stdout (the one JSON line) is relayed.
    Never re-executes a GPU-initialised process: children are fresh interpreters."""
    import socket
    import subprocess
    have = torch.cuda.device_count()          # counting devices does not initialise the GPU
    single = os.environ.get("VNQA_SINGLE_DEVICE") == "1" or plumbing
    if have < n and not single:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node — no multi-GPU number can be "
                         "measured here (the scaling curve needs the driver's 8-GPU node)" % (n, have))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode() if out else "")
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit("bench.py: rank exit codes %s" % rcs)


def parity_batches(args, device, first=0, count=3):
    """Seeded minibatches of the benchmark's shape: number 0 has all clips full length, the others ragged video lengths
    (3..T frames, the real data's range after 1-in-4 subsampling) and ragged question lengths.  The parity block uses 0..2; the
    tolerance mode is also checked on 3..11 (`twelve_minibatches`)."""
    out = []
    for i in range(first, first + count):
        g = torch.Generator(device="cpu").manual_seed(777 + i)
        B, T = args.batch, args.frames
        clip = torch.rand(B, 3, args.height, args.width, T, generator=g)
        q_lens = torch.randint(5, 26, (B,), generator=g)
        q = torch.randint(1, 134, (B, 56), generator=g)
        q = q * (torch.arange(56).unsqueeze(0) < q_lens.unsqueeze(1)).long()
        v_lens = torch.full((B,), T, dtype=torch.long) if i == 0 else torch.randint(3, T + 1, (B,), generator=g)
        if i > 0:
            v_lens[0] = T
            clip = clip * (torch.arange(T).view(1, 1, 1, 1, T) < v_lens.view(B, 1, 1, 1, 1)).float()   # zero past the end
        y = torch.randint(0, 70, (B,), generator=g)
        out.append((clip.to(device), q.to(device), v_lens, q_lens, y.to(device)))
    return out


def oracle_full_size(args, device, dump_path):
    """HIP path vs the CPU oracle at the BENCHMARK'S OWN SIZE: the exact-f32 precision of the product (stem +
    FiLM-attn, train-mode forward through the C ABI) on the very weights and minibatch the CPU leg ran
    (`oracle_workload`, regenerated from the same seed), against the logits the oracle computed there.  The oracle runs
    only in the CPU child; this function reads its dumped logits."""
    import copy
    import torch.nn as nn
    from videonavqa_amd.models import FiLMAttnPretrainedStem, ObjDetectCNN
    from videonavqa_amd.models.common import FrameLayout, NativeFeatures
    from videonavqa_amd.stem import FrozenStem, VGGFront
    if not dump_path or not os.path.exists(dump_path) or args.model != "film_attn_pt":
        return None
    d = torch.load(dump_path)
    ref, Bs, perm_o = d["logits"].float(), int(d["batch"]), d["perm"].long()
    W_vgg, W_od, W, (clip, q, v_lens, q_lens, y) = oracle_workload(args, Bs)
    S = (args.height // 16) * (args.width // 16)

    def run(prec):
        vgg = VGGFront(prec)
        od = ObjDetectCNN(27, 512, 1024, 0, True, True, precision=prec)
        vgg.load_state_dict(W_vgg)
        od.load_state_dict(W_od, strict=False)          # (the unused classifier tail / num_batches_tracked are not in the dict)
        model = FiLMAttnPretrainedStem(Bs, 128, 70, num_res_blocks=args.blocks, num_res_block_channels=args.channels,
                                       max_num_frames=args.frames, spatial_size=S, precision=prec)
        vgg, od, model = vgg.to(device).eval(), od.to(device).eval(), model.to(device)
        model.load_reference_tensors(W)
        stem = FrozenStem(vgg, od, prec)
        v_sorted, perm = torch.sort(v_lens, dim=0, descending=True, stable=True)
        lay = FrameLayout(v_sorted, args.frames, device, perm=perm)
        feats = stem.forward_clip(clip.to(device), lay.img_of, lay.n_img)
        native = NativeFeatures(feats, lay, 512, args.height // 16, args.width // 16, segs=stem.feature_segs, shift=stem.feature_shift)
        model.train()
        model.init_hidden()
        with torch.no_grad():
            out = model(native, q.to(device)[perm.to(device)], v_sorted, q_lens[perm]).float().cpu()
        # rows: product row j = original sample perm[j]; oracle row i = original sample perm_o[i]
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(len(perm))
        out_o = out[inv[perm_o]]
        del model, stem, vgg, od, feats
        torch.cuda.empty_cache()
        return (float((out_o - ref).abs().max() / ref.abs().max()), float((out_o - ref).norm() / ref.norm()),
                bool((out_o.argmax(1) == ref.argmax(1)).all()))

    err, l2, same = run("fp32")
    res = {"oracle_full_size_rel_err": round(err, 9), "oracle_full_size_rel_l2_err": round(l2, 9),
           "oracle_full_size_argmax_equal": same,
           "oracle_full_size_what": "precision='fp32' HIP path (stem + FiLM-attn train-mode forward) vs oracle/vnqa_oracle.py "
                                    "(stem_forward + film_attn_forward, run in the CPU child) on identical weights and the CPU "
                                    "leg's minibatch: %d clips x %d frames %dx%d; max|d logit| / max|logit|"
                                    % (Bs, args.frames, args.height, args.width)}
    if args.precision in ("fp16h", "fp16", "bf16"):
        # ... and the HEADLINE precision against the oracle DIRECTLY (not through the product's fp32 mode)
        e16, l16, same16 = run(args.precision)
        res.update({"oracle_full_size_%s_rel_err" % args.precision: round(e16, 9),
                    "oracle_full_size_%s_rel_l2_err" % args.precision: round(l16, 9),
                    "oracle_full_size_%s_argmax_equal" % args.precision: same16,
                    "oracle_full_size_%s_within_1e-3" % args.precision: bool(e16 <= 1e-3)})
    return res


def smooth_batches(args, device, first=0, count=12):
    """Minibatches whose pixels do NOT look like the stem's calibration frames (seeded uniform noise): 14 x 14 noise per frame
    upsampled 16 x, a per-clip brightness and a slow drift over the frames — large flat regions, other channel means."""
    import torch.nn.functional as F
    out = []
    for i in range(first, first + count):
        g = torch.Generator(device="cpu").manual_seed(777 + i)
        B, T = args.batch, args.frames
        low = torch.rand(B * T, 3, args.height // 16, args.width // 16, generator=g)
        up = F.interpolate(low, size=(args.height, args.width), mode="bilinear", align_corners=False).view(B, T, 3, args.height, args.width)
        gain = 0.3 + 0.7 * torch.rand(B, 1, 1, 1, 1, generator=g)
        drift = torch.linspace(0, 0.2, T).view(1, T, 1, 1, 1) * torch.rand(B, 1, 1, 1, 1, generator=g)
        clip = (up * gain + drift).clamp_(0, 1).permute(0, 2, 3, 4, 1).contiguous()
        q_lens = torch.randint(5, 26, (B,), generator=g)
        q = torch.randint(1, 134, (B, 56), generator=g)
        q = q * (torch.arange(56).unsqueeze(0) < q_lens.unsqueeze(1)).long()
        v_lens = torch.full((B,), T, dtype=torch.long) if i == 0 else torch.randint(3, T + 1, (B,), generator=g)
        if i > 0:
            v_lens[0] = T
            clip = clip * (torch.arange(T).view(1, 1, 1, 1, T) < v_lens.view(B, 1, 1, 1, 1)).float()
        y = torch.randint(0, 70, (B,), generator=g)
        out.append((clip.to(device), q.to(device), v_lens, q_lens, y.to(device)))
    return out


def blocks_clip(B, T, H, W, g, n_rect=24):
    """[B, 3, H, W, T]: a random background colour per clip and 24 axis-aligned rectangles of random colour, each drifting up to two
    pixels per frame — flat regions and sharp edges, unlike uniform noise and unlike smooth gradients."""
    img = torch.rand(B, 3, 1, 1, 1, generator=g).expand(B, 3, H, W, T).clone()
    for _ in range(n_rect):
        y0, x0 = torch.randint(0, H - 8, (B,), generator=g), torch.randint(0, W - 8, (B,), generator=g)
        hh, ww = torch.randint(8, H // 2, (B,), generator=g), torch.randint(8, W // 2, (B,), generator=g)
        col = torch.rand(B, 3, generator=g)
        dx = torch.randint(-2, 3, (B,), generator=g)
        for b in range(B):
            for t in range(T):
                xs = int(min(max(int(x0[b]) + int(dx[b]) * t, 0), W - 8))
                img[b, :, int(y0[b]):int(y0[b] + hh[b]), xs:xs + int(ww[b]), t] = col[b].view(3, 1, 1)
    return img


def textured_clip(B, T, H, W, g):
    """[B, 3, H, W, T]: blocks_clip's flat regions MODULATED by a static low-frequency texture per clip (28 x 28 noise upsampled 8 x, gain
    0.55 .. 1) and a slow global illumination ramp over the frames — flat regions + texture + temporal coherence, the fourth held-out
    kind (rendered interiors: textured walls under changing light)."""
    import torch.nn.functional as F
    img = blocks_clip(B, T, H, W, g, n_rect=12)
    low = torch.rand(B, 1, H // 8, W // 8, generator=g)
    tex = 0.55 + 0.45 * F.interpolate(low, size=(H, W), mode="bilinear", align_corners=False)
    light = 1.0 - 0.25 * torch.rand(B, 1, 1, 1, 1, generator=g) * torch.linspace(0, 1, T).view(1, 1, 1, 1, T)
    return (img * tex.unsqueeze(-1) * light).clamp_(0, 1)


def blocks_batches(args, device, first=0, count=12, kind="blocks"):
    """Minibatches of a kind the stem's calibration frames (half uniform noise, half smooth) contain nothing of: piecewise-constant
    images — a background colour and 24 drifting rectangles per clip (blocks_clip); kind 'textured' = textured_clip."""
    out = []
    for i in range(first, first + count):
        g = torch.Generator(device="cpu").manual_seed(777 + i)
        B, T = args.batch, args.frames
        clip = (textured_clip if kind == "textured" else blocks_clip)(B, T, args.height, args.width, g)
        q_lens = torch.randint(5, 26, (B,), generator=g)
        q = torch.randint(1, 134, (B, 56), generator=g)
        q = q * (torch.arange(56).unsqueeze(0) < q_lens.unsqueeze(1)).long()
        v_lens = torch.full((B,), T, dtype=torch.long) if i == 0 else torch.randint(3, T + 1, (B,), generator=g)
        if i > 0:
            v_lens[0] = T
            clip = clip * (torch.arange(T).view(1, 1, 1, 1, T) < v_lens.view(B, 1, 1, 1, 1)).float()
        y = torch.randint(0, 70, (B,), generator=g)
        out.append((clip.to(device), q.to(device), v_lens, q_lens, y.to(device)))
    return out


def tolerance_sweep(args, device, seeds=(1, 2, 3), count=12):
    """How ROBUST the tolerance mode's 1e-3 is: the headline precision against the exact-f32 precision, train-mode
    forward at this workload's size, on `count` seeded minibatches for OTHER random weights (weight seeds 1..3; seed 0 is the
    parity block's own twelve_minibatches) and — seed 0 — on smooth clips and on 'blocks' clips, a kind the stem's calibration frames
    contain nothing of."""
    import copy
    from videonavqa_amd.train import Trainer

    def logits(prec, seed, data):
        a = copy.copy(args)
        a.precision, a.seed = prec, seed
        model, stem, _, _ = build(a, device)
        tr = Trainer(model, stem, lr=1e-4, clip=1.0, loss_reduction="sum")
        model.train()
        res = []
        with torch.no_grad():
            for clip, q, v_lens, q_lens, _ in data:
                native, v_sorted, perm = tr.extract_features(clip, v_lens)
                model.init_hidden()
                res.append(model(native, q[perm.to(device)], v_sorted, q_lens[perm]).float().cpu())
        del tr, model, stem
        torch.cuda.empty_cache()
        return res

    def compare(seed, data):
        ref, got = logits("fp32", seed, data), logits(args.precision, seed, data)
        rel = [float((g - r).abs().max() / r.abs().max()) for g, r in zip(got, ref)]
        same = sum(int((g.argmax(1) == r.argmax(1)).sum()) for g, r in zip(got, ref))
        return {"max": round(max(rel), 8), "rms": round((sum(x * x for x in rel) / len(rel)) ** 0.5, 8),
                "argmax_equal": "%d/%d" % (same, sum(r.shape[0] for r in ref))}

    out = {"what": "max |d logit| / max |logit| of precision '%s' against precision 'fp32' (identical weights and inputs, train-mode forward, "
                   "%d clips x %d frames %dx%d) on %d seeded minibatches (one full-length, the rest ragged) per entry: other random "
                   "weights (torch.manual_seed(s) before the models are built) smooth clips (14 x 14 noise upsampled 16 x + brightness "
                   "+ drift) and — stress kinds, counted in max_incl_stress — blocks clips (piecewise-constant images, near-identical frames: a kind "
                   "the stem's calibration frames, half uniform noise and half smooth, contain nothing of; flat regions and repeated frames make "
                   "the activation roundings coherent over pixels AND frames) and textured clips (flat regions x static texture x light ramp)"
                   % (args.precision, args.batch, args.frames, args.height, args.width, count)}
    noise = parity_batches(args, device, first=0, count=count)
    for s in seeds:
        out["weight_seed_%d" % s] = compare(s, noise)
    del noise
    out["smooth_clips_weight_seed_0"] = compare(0, smooth_batches(args, device, count=count))
    # held-out kinds (nothing like them in the calibration frames), temporally COHERENT clips: the stress side of the claim
    blocks = blocks_batches(args, device, count=count)
    out["blocks_clips_weight_seed_0"] = compare(0, blocks)
    out["blocks_clips_weight_seed_3"] = compare(3, blocks)
    del blocks
    out["textured_clips_weight_seed_3"] = compare(3, blocks_batches(args, device, count=count, kind="textured"))
    return out


def precision_parity(args, device, speed_steps=5, fit_steps=12):
    """bf16 benchmark precision against the exact-f32 parity precision (itself pinned <= 1e-3 to the reference goldens,
    tests/test_gpu_models.py) on IDENTICAL fp32 master weights and inputs at this workload's full size:
      * at initialisation: train-mode forward (per-frame batch-statistics BN, as the timed step runs it) on three
        minibatches incl. ragged ones, and one backward on the first (flat-gradient error);
      * after the fp32 mode has FIT the first minibatch for `fit_steps` optimisation steps (an untrained net's answer
        logits are near-ties — top-2 gaps below 1e-3 of the logit range — so argmax agreement is only meaningful on
        weights that separate the classes): the same weights loaded into the bf16 mode, logits / argmax compared.
    Returns the measured errors and the fp32 mode's own throughput."""
    import copy
    from videonavqa_amd.train import Trainer
    batches = parity_batches(args, device)
    logits, losses, grads, speed, fit, names, gp, more = {}, {}, {}, {}, {}, {}, {}, {}
    trained = None

    def forward(tr, batch, grad=False):
        clip, q, v_lens, q_lens, y = batch
        native, v_sorted, perm = tr.extract_features(clip, v_lens)
        perm_d = perm.to(device)
        tr.model.init_hidden()
        with torch.set_grad_enabled(grad):
            out = tr.model(native, q[perm_d], v_sorted, q_lens[perm])
            loss = tr.loss_fn(out, y[perm_d])
        return out, loss

    low = args.precision if args.precision in ("bf16", "fp16", "fp16h") else "bf16"      # the 16-bit precision under test
    from videonavqa_amd import _lib as L
    L.set_half("f16" if low in ("fp16", "fp16h") else "bf16")       # one 16-bit storage format per process: fix it before the fp32 build
    for prec in ("fp32", low):
        a = copy.copy(args)
        a.precision = prec
        model, stem, _, _ = build(a, device)
        tr = Trainer(model, stem, lr=1e-4, clip=1.0, loss_reduction="sum")
        model.train()
        lg, ls = [], []
        for bi, batch in enumerate(batches):
            out, loss = forward(tr, batch, grad=(bi == 0))
            if bi == 0:
                loss.backward()
                grads[prec] = tr.fp.grad.clone()
                tr.fp.zero_grad()
                names[prec] = [(n, p.numel()) for n, p in model.named_parameters() if p.requires_grad]
                am = getattr(model, "_gp_argmax", None)       # pooling heads: which frame supplied every pooled feature
                if am is not None:
                    if prec == "fp32":
                        gp["ref_argmax"] = am.clone()
                    else:
                        # the same backward with the gradient ROUTED by the fp32 run's arg-max frames: what is left of the
                        # gradient error once max-over-frames picks the same frames
                        ref_am = gp["ref_argmax"]
                        live = ref_am >= 0
                        gp["flip_frac"] = float(((am != ref_am) & live).sum()) / max(float(live.sum()), 1.0)
                        model._gp_route = ref_am
                        out2, loss2 = forward(tr, batch, grad=True)
                        loss2.backward()
                        gp["routed_grad"] = tr.fp.grad.clone()
                        tr.fp.zero_grad()
                        model._gp_route = None
            lg.append(out.detach().float().cpu())
            ls.append(float(loss.detach()))
        logits[prec], losses[prec] = lg, ls
        if low == "fp16h":      # the tolerance mode: nine more minibatches, forward only (three under-sample the maximum)
            for j in range(3):
                for batch in parity_batches(args, device, first=3 + 3 * j, count=3):
                    more.setdefault(prec, []).append(forward(tr, batch)[0].detach().float().cpu())
        b0 = batches[0]
        if prec == "fp32":
            # fit the first minibatch (this also measures the parity precision's own throughput on the workload)
            nxt = dict(next_clip=b0[0], next_v_lens_cpu=b0[2])
            model.bn_init.reset_running_stats()
            for _ in range(2):
                tr.step(*b0, **nxt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(speed_steps):
                tr.step(*b0, **nxt)
            torch.cuda.synchronize()
            speed[prec] = args.batch * speed_steps / (time.perf_counter() - t0)
            for _ in range(max(fit_steps - speed_steps - 2, 0)):
                tr.step(*b0, **nxt)
            trained = {k: v.detach().clone() for k, v in model.state_dict().items()}
        else:
            model.load_state_dict(trained)          # in-place copies into the flat parameter buffer
        out, loss = forward(tr, b0)
        fit[prec] = (out.detach().float().cpu(), float(loss))
        del tr, model, stem
        torch.cuda.empty_cache()
    rel = [float((b - f).abs().max() / f.abs().max()) for b, f in zip(logits[low], logits["fp32"])]
    rel_l2 = [float((b - f).norm() / f.norm()) for b, f in zip(logits[low], logits["fp32"])]
    same = sum(int((b.argmax(1) == f.argmax(1)).sum()) for b, f in zip(logits[low], logits["fp32"]))
    total = sum(f.shape[0] for f in logits["fp32"])

    def top2_gap(f):     # how decisive the fp32 prediction is: top-1 / top-2 logit gap relative to max |logit|
        t = f.topk(2, 1)[0]
        return (t[:, 0] - t[:, 1]) / f.abs().max()

    # samples whose bf16 argmax differs: their fp32 top-2 gap (a flip needs gap < 2 x the logits error)
    flipped = [round(float(g), 6) for b, f in zip(logits[low], logits["fp32"])
               for g, eq in zip(top2_gap(f), b.argmax(1) == f.argmax(1)) if not bool(eq)]
    gf, gb = grads["fp32"], grads[low]
    ff, fb = fit["fp32"][0], fit[low][0]

    def per_param(g_low):
        """the parameters that carry the flat-gradient error: (name, own relative L2 error, share of the squared error)"""
        rows, off, tot = [], 0, float((g_low - gf).pow(2).sum())
        for n, k in names["fp32"]:
            d, r = g_low[off:off + k] - gf[off:off + k], gf[off:off + k]
            rows.append((n, float(d.norm() / (r.norm() + 1e-30)), float(d.pow(2).sum()) / max(tot, 1e-30)))
            off += k
        rows.sort(key=lambda t: -t[2])
        return [{"param": n, "rel_l2_err": round(e, 5), "share_of_sq_err": round(sh, 4)} for n, e, sh in rows[:5]]

    pooling = None
    if "routed_grad" in gp:
        gr = gp["routed_grad"]
        pooling = {"argmax_frame_flip_frac": round(gp["flip_frac"], 5),
                   "grad_rel_l2_err_routed_by_fp32_argmax": round(float((gr - gf).norm() / gf.norm()), 6),
                   "grad_err_by_param_routed": per_param(gr),
                   "note": "max over frames hands each pooled feature's WHOLE gradient to ONE frame: where two frames are within "
                           "rounding of each other the 16-bit run picks the other one (argmax_frame_flip_frac) and that feature's "
                           "gradient moves to another image's activations — a discontinuity of the model, not a kernel error. "
                           "Routing the 16-bit backward by the fp32 run's frames removes that part; what remains sits in "
                           "conv_init.weight, whose gradient is a sum over the SPARSE set of (frame, pixel) positions the max "
                           "selected (no averaging over frames as in the attention model) and scales with the storage "
                           "format's rounding (bf16 -> fp16: 2.7x smaller)"}
    twelve = None
    if more:
        allb = [(b, f) for b, f in zip(logits[low] + more[low], logits["fp32"] + more["fp32"])]
        r12 = [float((b - f).abs().max() / f.abs().max()) for b, f in allb]
        twelve = {"what": "the same comparison on twelve seeded minibatches (the three above + nine more ragged ones), train-mode forward",
                  "logits_rel_err_per_batch": [round(r, 8) for r in r12], "max": round(max(r12), 8),
                  "rms": round((sum(r * r for r in r12) / len(r12)) ** 0.5, 8),
                  "argmax_equal": "%d/%d" % (sum(int((b.argmax(1) == f.argmax(1)).sum()) for b, f in allb), sum(f.shape[0] for _, f in allb))}
    robust = None
    if low == "fp16h" and args.model == "film_attn_pt" and not getattr(args, "no_robustness", False):
        robust = tolerance_sweep(args, device)
        if twelve is not None:
            robust["weight_seed_0"] = {"max": twelve["max"], "rms": twelve["rms"], "argmax_equal": twelve["argmax_equal"]}
            # (the tolerance is stated for the benchmark's clips: the blocks entry — a stress kind in which large flat regions make the
            # stored activations' rounding errors COHERENT over pixels — is reported beside max_over_all, not inside it)
            stress = ("blocks_clips_weight_seed_0", "blocks_clips_weight_seed_3", "textured_clips_weight_seed_3")
            mx = max([twelve["max"]] + [v["max"] for k, v in robust.items() if isinstance(v, dict) and "max" in v
                                       and k not in ("weight_seed_0",) + stress])
            robust["max_over_all"] = round(mx, 8)
            robust["within_1e-3"] = bool(mx <= 1e-3)
            # ... and WITH the held-out stress kinds: the compliance flag must not depend on which kinds are counted
            ms = max([mx] + [robust[k]["max"] for k in stress if k in robust])
            robust["max_incl_stress"] = round(ms, 8)
            robust["within_1e-3_incl_stress"] = bool(ms <= 1e-3)
    return {"reference": "precision='fp32' (exact-f32 MFMA kernels; pinned <= 1e-3 to the reference goldens by tests/test_gpu_models.py)",
            "robustness": robust,
            "batches": "3 x (%d clips x %d frames %dx%d): full length, ragged, ragged; train-mode forward"
                       % (args.batch, args.frames, args.height, args.width),
            "precision": low,
            "%s_logits_rel_err" % low: round(max(rel), 8), "%s_logits_rel_err_per_batch" % low: [round(r, 8) for r in rel],
            # the same comparison as a relative L2 norm over the minibatch's logits (the max-norm figure above is the strict one)
            "%s_logits_rel_l2_err_per_batch" % low: [round(r, 8) for r in rel_l2],
            "argmax_equal_at_init": "%d/%d" % (same, total), "fp32_top2_gap_rel_of_flipped_at_init": flipped,
            "twelve_minibatches": twelve,
            "loss_rel_err": round(max(abs(b - f) / max(abs(f), 1e-9) for b, f in zip(losses[low], losses["fp32"])), 6),
            "grad_rel_l2_err": round(float((gb - gf).norm() / gf.norm()), 6),
            "grad_err_by_param": per_param(gb), "pooling_head": pooling,
            "after_fit": {"fit_steps_fp32": fit_steps, "fp32_loss": round(fit["fp32"][1], 4), "%s_loss" % low: round(fit[low][1], 4),
                          "%s_logits_rel_err" % low: round(float((fb - ff).abs().max() / ff.abs().max()), 6),
                          "argmax_equal": bool((fb.argmax(1) == ff.argmax(1)).all()),
                          "fp32_min_top2_gap_rel": round(float(top2_gap(ff).min()), 6)},
            "argmax_equal": bool((fb.argmax(1) == ff.argmax(1)).all()),
            "fp32_mode_clips_per_s": round(speed["fp32"], 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=3, help="the timed K-step region is run this many times back to back; "
                    "the reported value / ms_per_step are the MEDIAN region's, all regions are listed in `repeats`")
    ap.add_argument("--precision", default="fp16h", choices=["bf16", "fp16", "fp16h", "fp32"],
                    help="fp16h (default, the headline; also the default of every model constructor and CLI): the TOLERANCE-COMPLIANT precision — "
                         "fp16 storage and fp16 MFMA products with fp32 accumulation, stem weights second-order rounded on calibration frames, "
                         "stem activations and features stored mean-shifted, conv_init as two products against split weights with its output "
                         "split into its BatchNorm, the frozen 1x1 conv and fc_embed_attn with split weights: logits within north star's 1e-3 of "
                         "exact fp32 on 4 weight seeds x 12 minibatches at 224x224 AND at the reference's 160x208, and on held-out kinds of clip; "
                         "fp16: the same without the trunk's split operands (a leg); bf16: BASELINE.json's storage dtype (4e-3, a leg); "
                         "fp32: the exact-f32 parity precision")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU minibatch; default 8 (the metric's), 32 for --model v_only_cnn3d "
                    "(BASELINE config 2)")
    ap.add_argument("--frames", type=int, default=35)
    ap.add_argument("--height", type=int, default=224)
    ap.add_argument("--width", type=int, default=224)
    ap.add_argument("--blocks", type=int, default=1)
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--tail-channels", type=int, default=0, help="num_tail_channels of the pooling models (0 = the "
                    "constructor default: 16 / 32; eval.sh passes 32 for film_gp_pt and 64 for time_multi_hop)")
    ap.add_argument("--model", default="film_attn_pt", choices=["film_attn_pt", "film_gp_pt", "time_multi_hop", "mac", "v_only_cnn3d"],
                    help="film_attn_pt is the metric's model; film_gp_pt / time_multi_hop are BASELINE.json's ladder "
                         "configs 3 and 5, mac is the remaining stem-consuming model of the same CLI")
    ap.add_argument("--h2d", action="store_true", help="PCIe-inclusive variant: clips start in pinned host memory "
                    "and are copied to the GPU every step (on the stem stream); never the headline value")
    ap.add_argument("--h2d-ablation", default=None, choices=["pinonly", "copyonly"],
                    help="timing diagnostics of the PCIe-inclusive rate (without --h2d): the resident-input loop while the pinned staging clips "
                    "merely exist / while one H2D copy per step runs into scratch buffers nobody reads (profiles/r05_h2d.txt)")
    ap.add_argument("--clip-dtype", default="f32", choices=["f32", "u8"], help="u8: synthetic clips are RAW 8-bit pixels (value "
                    "k / 255 as eval/dataset.py:91 forms it; VNQADataset(uint8_video=True)) — with --h2d a quarter of the PCIe bytes")
    ap.add_argument("--minibatches", type=int, default=4, help="distinct HBM-resident minibatches (own clips, questions, "
                    "labels) the steps rotate through, so that the timed region does not fit ONE batch to loss 1e-3 and "
                    "run its backward kernels on collapsed gradients")
    ap.add_argument("--mode", default="train", choices=["train", "eval"], help="eval: time the INFERENCE path (Trainer.eval_step: "
                    "val_epoch / test of the reference, forward only) instead of the training step; never the headline metric")
    ap.add_argument("--no-eval-leg", action="store_true", help="skip the inference path's region (`eval_mode` block) after the training "
                    "regions — for kernel traces of the training step alone")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the bf16-vs-fp32 parity block (adds ~10 s)")
    ap.add_argument("--parity-only", action="store_true", help="print only the parity block of --precision (no timing)")
    ap.add_argument("--no-fp16-leg", action="store_true", help="skip the other storage precisions' own short runs (bf16_mode / fp16_mode, child processes)")
    ap.add_argument("--no-robustness", action="store_true", help="skip the tolerance mode's sweep over weight seeds 1..3 and smooth clips (parity.robustness, ~40 s)")
    ap.add_argument("--no-overlap", action="store_true", help="run the stem on the main stream (no side-stream pipeline)")
    ap.add_argument("--feature-slots", type=int, default=2, help=argparse.SUPPRESS)       # A/B hook: 3 = the stem never waits for the trunk
    ap.add_argument("--cpu-batch", type=int, default=8, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-steps", type=int, default=3, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--plumbing", action="store_true", help="NOT a benchmark: run the multi-rank path's host side (spawned ranks, environment "
                    "contract, replica broadcast, the Trainer's gradient reducer, timed-region protocol, comm block, rank-0 line) on CPU tensors "
                    "over gloo with a stand-in module — `--gpus 8 --plumbing` works without any GPU")
    ap.add_argument("--cpu-logits-out", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 32 if args.model == "v_only_cnn3d" else 8
    if args.cpu_baseline_only:
        cpu_baseline_child(args)
        return
    if args.plumbing:          # the multi-rank path's host side on CPU tensors (tests/test_bench_host.py: 8 ranks without a GPU)
        if "WORLD_SIZE" not in os.environ:
            spawn_ranks(args.gpus, sys.argv[1:], plumbing=True)
        else:
            plumbing_rank(args)
        return

    if args.precision in ("fp16", "fp16h"):      # the fp16-storage build of the library (one 16-bit format per process)
        from videonavqa_amd import _lib as L
        L.set_half("f16")
    if args.model == "v_only_cnn3d":
        if args.precision == "fp16h":     # (the split-activation pieces of 'fp16h' belong to the FiLM path: config 2 runs as fp16 storage)
            args.precision = "fp16"
        bench_cnn3d(args)
        return
    if args.parity_only:
        torch.cuda.set_device(0)
        print(json.dumps(precision_parity(args, torch.device("cuda", 0))), flush=True)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus, sys.argv[1:])          # this process never touches the GPU
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    cpu_leg, oracle_dump = None, None
    if world == 1 and not args.no_cpu_baseline:
        import tempfile
        fwd = [a for a in sys.argv[1:] if a not in ("--no-cpu-baseline",)]
        if args.model == "film_attn_pt" and not args.no_parity:
            oracle_dump = os.path.join(tempfile.mkdtemp(prefix="vnqa_bench_"), "oracle_logits.pt")
        cpu_leg = cpu_baseline(fwd, logits_out=oracle_dump)          # child process, before any GPU initialisation here
    import torch.distributed as dist
    # test hook: VNQA_DIST_BACKEND=gloo VNQA_SINGLE_DEVICE=1 runs N ranks on ONE GPU to exercise the multi-rank
    # code path where only one GPU exists (never a benchmark configuration)
    if os.environ.get("VNQA_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("VNQA_DIST_BACKEND", "nccl")       # "nccl" is RCCL on ROCm
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            # RCCL's kernels on a high-priority stream: they are dispatched ahead of the co-running stem's workgroups, like the trunk's
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, pg_options=opts)
            except (AttributeError, TypeError):
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from videonavqa_amd.train import Trainer
    model, stem, vgg, od = build(args, device)
    trainer = Trainer(model, stem, lr=1e-4, clip=1.0, loss_reduction="sum", world_size=world, rank=rank, feature_slots=args.feature_slots)
    NB = max(args.minibatches, 1)
    batches = [synth_batch(args, rank, device, i) for i in range(NB)]
    if args.h2d:
        batches = [(b[0].cpu().pin_memory(),) + tuple(b[1:]) for b in batches]
    # diagnostics of the PCIe-inclusive rate (round 5's h2d ablation, docs/history): the resident-input loop with ONE ingredient of --h2d added
    pinned_copies, scratch, scratch_stream = None, None, None
    if args.h2d_ablation and not args.h2d:
        pinned_copies = [b[0].cpu().pin_memory() for b in batches]       # 'pinonly': the pinned staging clips exist, nothing reads them
        if args.h2d_ablation == "copyonly":                                # + one H2D copy per step into scratch buffers nobody reads
            scratch = [torch.empty_like(batches[0][0]) for _ in range(3)]
            scratch_stream = torch.cuda.Stream()
    batch = batches[0]
    step_no = [0]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Steady-state software pipeline: every step runs its own trunk pass AND the frozen stem of the
    # following minibatch (side stream).  The timed region therefore contains exactly K stem passes
    # and K trunk passes: it starts with one stem already in flight from warm-up and ends having
    # produced one for the step after the region.  Steps rotate through the NB resident minibatches.
    if args.h2d:      # 3-stage input pipeline: H2D(i+2) on the copy engine | stem(i+1) | trunk(i)
        queue = [trainer.upload(batches[0][0]), trainer.upload(batches[1 % NB][0])]

        def run_step():
            i = step_no[0]
            step_no[0] += 1
            cur, nx = queue
            b, bn = batches[i % NB], batches[(i + 1) % NB]
            out = trainer.step(cur, *b[1:], next_clip=nx, next_v_lens_cpu=bn[2])
            # the copy of clip i+2 is enqueued AFTER step i (its consumer, stem(i+2), is enqueued by step i+1): the launch thread
            # hands the GPU this step's kernels first
            queue[0], queue[1] = nx, trainer.upload(batches[(i + 2) % NB][0])
            return out
    else:
        def run_step():
            i = step_no[0]
            step_no[0] += 1
            b, bn = batches[i % NB], batches[(i + 1) % NB]
            if args.no_overlap:
                return trainer.step(*b)
            out = trainer.step(*b, next_clip=bn[0], next_v_lens_cpu=bn[2])
            if scratch is not None:
                with torch.cuda.stream(scratch_stream):
                    scratch[i % 3].copy_(pinned_copies[(i + 2) % NB], non_blocking=True)
            return out
    for _ in range(3):        # priming (lazy HIP attribute calls, allocator growth, pinned staging): not part of --warmup
        run_step()
    for _ in range(args.warmup):
        run_step()
    import gc
    gc.collect()
    if os.environ.get("VNQA_BENCH_GC", "0") != "1":
        gc.disable()          # the launch thread must not stall in the cyclic collector mid-step

    def timed_region():
        """EXACTLY --steps steps between barrier + synchronize on both sides; max over ranks."""
        stem.timing = []          # (start, end) HIP events around every stem-tagged igemm launch
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss, _ = run_step()
        t_enq = time.perf_counter() - t0      # host time to ENQUEUE the K steps (launch-thread cost, no device wait)
        barrier()
        dt = time.perf_counter() - t0
        ev = stem.timing
        stem.timing = None
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, t_enq, ev, loss

    if args.mode == "eval":
        # inference (val_epoch / test, eval/q_and_v_eval.py:159-224, eval/q_and_v_test.py:64-142): the SAME timed-region protocol
        # with Trainer.eval_step — forward-only fused trunk, stem of the next minibatch on the stem stream
        def run_step():       # noqa: F811
            i = step_no[0]
            step_no[0] += 1
            b, bn = batches[i % NB], batches[(i + 1) % NB]
            if args.no_overlap:
                return trainer.eval_step(*b)[:2]
            return trainer.eval_step(*b, next_clip=bn[0], next_v_lens_cpu=bn[2])[:2]
        for _ in range(3 + args.warmup):
            run_step()
    regions = [timed_region() for _ in range(max(args.repeats, 1))]
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    dt, t_enqueue, events, loss = regions[order[len(order) // 2]]          # the MEDIAN region is the reported one
    # Host cost of ONE step's enqueue on an IDLE queue (the in-region figure includes back-pressure — a launch
    # thread that is ahead of the GPU blocks in the runtime once the hardware queues are full, which looks like launch cost):
    # synchronize, enqueue one step, stop the clock before waiting for it; median of five.  Not part of the timed region.
    idle = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_step()
        idle.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    t_enqueue_idle = sorted(idle)[len(idle) // 2]

    # the inference path's throughput next to the training headline (same protocol, same resident minibatches, one region)
    eval_mode = None
    if args.mode == "train" and world == 1 and not args.h2d and args.model != "mac" and not args.no_eval_leg:
        train_step = run_step

        def run_step():       # noqa: F811
            i = step_no[0]
            step_no[0] += 1
            b, bn = batches[i % NB], batches[(i + 1) % NB]
            return trainer.eval_step(*b, next_clip=bn[0], next_v_lens_cpu=bn[2])[:2]
        for _ in range(3):
            run_step()
        e_dt = timed_region()[0]
        eval_mode = {"what": "Trainer.eval_step (model.eval() under no_grad: forward-only fused trunk, BatchNorm running statistics "
                             "folded into conv_init's epilogue, stem of the next minibatch on the stem stream), %d timed steps, same "
                             "resident minibatches; `bench.py --mode eval` prints this as its own line" % args.steps,
                     "value": round(args.batch * args.steps / e_dt, 1), "unit": "clips/s", "ms_per_step": round(e_dt / args.steps * 1e3, 3)}
        run_step = train_step

    # (N > 1) what the gradient all-reduce costs: the flat buffer's all-reduce alone, and the step WITHOUT collectives
    # (replicas diverge from here on: after the measurement) -> exposed communication per step
    comm = None
    if world > 1:
        g = trainer.fp.grad
        for _ in range(2):
            dist.all_reduce(g)
        barrier()
        t0 = time.perf_counter()
        for _ in range(5):
            dist.all_reduce(g)
        barrier()
        ar_ms = (time.perf_counter() - t0) / 5 * 1e3
        g.zero_()
        trainer.reducer.enabled = False
        dt_nc = timed_region()[0]
        trainer.reducer.enabled = True
        rccl_ver = None
        try:
            rccl_ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            pass
        comm = {"ranks": world, "backend": "rccl" if backend == "nccl" else backend, "rccl_version": rccl_ver,
                # RCCL picks algorithm / protocol per call from its tuner unless pinned by the environment; what this run used:
                "rccl_algo": os.environ.get("NCCL_ALGO", "tuner default (NCCL_ALGO unset)"),
                "rccl_proto": os.environ.get("NCCL_PROTO", "tuner default (NCCL_PROTO unset)"),
                "allreduce_payload_mb": round(trainer.fp.n * 4 / 1e6, 1), "allreduce_alone_ms": round(ar_ms, 3),
                "allreduce_alone_busbw_gbs": round(2 * (world - 1) / world * trainer.fp.n * 4 / (ar_ms * 1e-3) / 1e9, 1),
                "ms_per_step_without_collectives": round(dt_nc / args.steps * 1e3, 3),
                "exposed_comm_ms_per_step": round((dt - dt_nc) / args.steps * 1e3, 3)}
    gc.enable()
    # north-star side metric, outside the timed region: the frozen stem ALONE on the chip (all B*T frames,
    # conv1_1 .. conv32), as a fraction of the dense bf16 MFMA peak
    stem_ms = None
    if rank == 0:
        dev_clip = batch[0].to(device)
        lay = trainer.extract_features(dev_clip, batch[2])[0].layout
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            stem.forward_clip(dev_clip, lay.img_of, lay.n_img)
        e1.record()
        torch.cuda.synchronize()
        stem_ms = e0.elapsed_time(e1) / 5
        # ... and the same launches event-timed with the chip to themselves (no trunk stream co-running)
        stem.timing = []
        for _ in range(5):
            stem.forward_clip(dev_clip, lay.img_of, lay.n_img)
        torch.cuda.synchronize()
        alone_events, stem.timing = stem.timing, None
    parity = None
    # (single-GPU runs only: at N > 1 the other ranks would sit in the barrier below for the minute this takes)
    if world == 1 and not args.no_parity and args.model != "mac" and args.precision in ("bf16", "fp16", "fp16h"):
        loss = loss.clone()
        del trainer, model, stem
        torch.cuda.empty_cache()
        parity = precision_parity(args, device)
        full = oracle_full_size(args, device, oracle_dump)
        if full is not None:
            parity.update(full)
    if world > 1:
        dist.barrier()

    if rank == 0:
        ms = dt / args.steps * 1e3
        clips = args.batch * world * args.steps / dt
        all_clips = [args.batch * world * args.steps / r[0] for r in regions]
        H, W, T, B = args.height, args.width, args.frames, args.batch
        n_frames = B * T
        # HIP events around every launch of the frozen stem's C_out = 512 layers, grouped by the kernel that served them: the
        # DOMINANT kernel (most time in the timed region) is the roofline's subject, the other is listed beside it
        by_kernel = {}
        for ev in events:
            k = ev[3] if len(ev) > 3 else "conv_igemm_kernel"
            by_kernel.setdefault(k, []).append((ev[0].elapsed_time(ev[1]), ev[2]))
        kstats = {}
        for k, lst in by_kernel.items():
            tot_ms = sum(d for d, _ in lst)
            kstats[k] = {"launches_per_step": max(len(lst) // args.steps, 1), "avg_launch_ms": round(tot_ms / len(lst), 4),
                         "gflop_per_launch": round(sum(f for _, f in lst) / len(lst) / 1e9, 1),
                         "achieved_tflops": round(sum(f for _, f in lst) / (tot_ms * 1e-3) / 1e12, 1) if tot_ms > 0 else 0.0,
                         "ms_per_step": round(tot_ms / args.steps, 4)}
        def peak_for(kname):
            """MFMA peak a kernel entry is priced against, on its ALGORITHMIC FLOPs: the 16-bit dense peak for one-product kernels
            (bf16 and fp16 run at the same rate; the three-product layers of 'fp16h' count the FLOPs they execute), the exact-f32 MFMA
            peak for precision 'fp32'."""
            return PEAK_F32_TFLOPS if args.precision == "fp32" else PEAK_BF16_TFLOPS
        alone = {}
        for ev in alone_events:
            k = ev[3] if len(ev) > 3 else "conv_igemm_kernel"
            a = alone.setdefault(k, [0.0, 0.0, 0])
            a[0] += ev[0].elapsed_time(ev[1]); a[1] += ev[2]; a[2] += 1
        for k, (tot_ms, fl, n) in alone.items():
            if k in kstats and tot_ms > 0:
                kstats[k]["chip_to_itself"] = {"avg_launch_ms": round(tot_ms / n, 4), "achieved_tflops": round(fl / (tot_ms * 1e-3) / 1e12, 1),
                                               "frac": round(fl / (tot_ms * 1e-3) / 1e12 / peak_for(k), 4)}
        dom = max(kstats, key=lambda k: kstats[k]["ms_per_step"]) if kstats else "conv_igemm_kernel"
        dstat = kstats.get(dom, {"launches_per_step": 0, "avg_launch_ms": 0.0, "gflop_per_launch": 0.0, "achieved_tflops": 0.0})
        avg_ms, launches_per_step = dstat["avg_launch_ms"], dstat["launches_per_step"]
        flops_per_launch = dstat["gflop_per_launch"] * 1e9
        achieved = dstat["achieved_tflops"]
        peak = peak_for(dom)
        # HBM bytes per launch of the same kernel from the PMC passes (FETCH_SIZE / WRITE_SIZE cannot be read
        # live; collected with rocprofv3 --pmc in separate runs, corrected per the microarch guide) — only
        # valid for the default workload the passes were taken on
        traffic, traffic_src = None, None
        default_cfg = (args.precision in ("bf16", "fp16", "fp16h") and (B, T, H, W) == (8, 35, 224, 224))     # (the same kernel, shapes and bytes in both 16-bit formats)
        for tname in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            tfile = os.path.join(ROOT, "profiles", tname)
            if default_cfg and os.path.exists(tfile):
                tj = json.load(open(tfile))
                traffic = (tj.get("kernels", {}).get(dom) or {}).get("hbm_bytes_per_launch", tj.get("hbm_bytes_per_launch") if dom == "conv_igemm_kernel" else None)
                if traffic is None:
                    continue
                traffic_src = "profiles/%s (separate rocprofv3 --pmc passes over this workload's stem; not measured live)" % tname
                break
        S = (H // 16) * (W // 16)
        _, trunk_fb = trunk_flops_per_frame(S, 512, args.channels, args.blocks, 128)
        if args.model == "mac":     # three 3x3 convs (fwd + wgrad + dgrad except the first's) + the position-wise GEMM
            d = args.channels
            c1, c2 = 2.0 * S * 512 * d * 9, 2.0 * S * d * d * 9
            trunk_fb = 3 * (c1 + 2 * c2) - c1 + 3 * 2.0 * S * d * d
        flops_clip = T * (stem_flops_per_frame(H, W) + trunk_fb)
        composed = COMPOSED_STEM[0]
        flops_clip_exec = T * (stem_executed_flops_per_frame(H, W, composed) + trunk_fb)
        out = {
            "metric": METRIC, "value": round(clips, 3), "unit": "clips/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp16": "f16", "fp16h": "f16 (fp16 storage, fp16 MFMA products, fp32 accumulate; frozen stem weights second-order rounded on calibration frames; the stem's activations and features stored mean-shifted — value minus the calibration channel mean, the consumer's bias absorbs the mean —; conv_init as two products against split weights with its output split into its BatchNorm, 1x1 / fc_embed_attn with split weights)", "fp32": "f32"}[args.precision], "data": "synthetic",
            "repeats": {"n": len(regions), "value_is": "median region", "clips_per_s": [round(c, 1) for c in all_clips],
                        "spread_rel": round((max(all_clips) - min(all_clips)) / clips, 4)},
            "config": {"workload": "%s training step: VGG-16[:10]+ObjDetectCNN(512) frozen stem + "
                                   "FiLM trunk (%d block(s), C=%d, spatial %d) fwd+bwd + clip + Adam; "
                                   "bs=%d/GPU, %d-frame %dx%d clips, data-parallel"
                                   % (args.model, args.blocks, args.channels, S, B, T, H, W),
                       "global_batch": B * world, "frames": T, "parallelism": "dp%d" % world,
                       "gflop_per_clip": round(flops_clip / 1e9, 1),
                       # ALGORITHMIC work (the reference formulation's FLOPs) per second, and what the chip
                       # actually EXECUTES per second (the composed conv11.conv12 pair does fewer FLOPs for the same result)
                       "whole_step_tflops_algorithmic": round(clips * flops_clip / 1e12, 1),
                       "gflop_per_clip_executed": round(flops_clip_exec / 1e9, 1),
                       "whole_step_tflops_executed": round(clips * flops_clip_exec / 1e12, 1),
                       "final_loss": round(float(loss), 4), "minibatches_rotated": NB, "inputs": ("pinned host memory, H2D every step" if args.h2d else "resident in HBM") +
                                 (", raw uint8 pixels (k / 255 formed on the device)" if args.clip_dtype == "u8" else ""),
                       "host_enqueue_ms_per_step": round(t_enqueue / args.steps * 1e3, 3),
                       # the same step enqueued on an IDLE queue (no back-pressure from a busy GPU): the launch thread's own cost
                       "host_enqueue_idle_queue_ms_per_step": round(t_enqueue_idle * 1e3, 3),
                       "stem_alone_ms": round(stem_ms, 3),
                       # whole frozen stem alone on the chip: EXECUTED FLOPs / time / peak (hardware utilisation) and the
                       # same with the reference formulation's algorithmic FLOPs (37.167 GF/frame at 224x224)
                       "stem_formulation": ("conv11.conv12 composed into one 5x5 conv + exact border correction"
                                            if composed else "layer by layer"),
                       "stem_alone_mfma_util": round(n_frames * stem_executed_flops_per_frame(H, W, composed)
                                                     / (stem_ms * 1e-3) / 1e12 / peak, 4),
                       "stem_alone_mfma_util_algorithmic": round(n_frames * stem_flops_per_frame(H, W) / (stem_ms * 1e-3)
                                                                 / 1e12 / peak, 4)},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": {"conv_igemm_kernel": "conv_igemm_kernel<%s,256,256,2,4,TAG=1> (frozen-stem implicit GEMM on v_mfma_f32_16x16x32: "
                                                         "the composed 5x5 conv11.conv12 and any C_out=512 layer the patch-stationary "
                                                         "kernel does not serve; FLOPs = those its launches execute)"
                                                         % ("bf16" if args.precision == "bf16" else "f16"),
                                    "conv_ps_kernel<28,5x5>": "conv_ps_kernel<28,2,TAG=1> (frozen-stem patch-stationary conv on v_mfma_f32_16x16x32, 5x5 "
                                                              "instantiation: the composed conv11.conv12 on 8 x 28-pixel 2-D tiles, border correction, ReLU "
                                                              "floor and 2x2 pool fused; FLOPs = those its launches execute)",
                                    "conv_ps_kernel<28>": "conv_ps_kernel<28,1,TAG=1> (patch-stationary 3x3 conv, 4 waves x 512 registers: conv21, "
                                                          "conv22 on 28x28 maps; FLOPs = those its launches execute)",
                                    "conv_ps_kernel<14>": "conv_ps_kernel<14,1,TAG=1> (patch-stationary 3x3 conv: conv31, conv32 on 14x14 maps)"}.get(dom, dom),
                         "measured": "HIP events around every launch of this kernel in the timed region, while the trunk stream "
                                     "co-runs on the same chip (see stem_alone_* for the kernel with the chip to itself)",
                         # the same kernel's launches with no other stream on the chip (stem-alone pass after the timed region):
                         # the timed-region figure above is lower because the previous step's trunk kernels share the CUs
                         "chip_to_itself": dstat.get("chip_to_itself"),
                         "other_stem_kernels": {k: v for k, v in kstats.items() if k != dom},
                         "launches_per_step": launches_per_step, "avg_launch_ms": round(avg_ms, 4),
                         "gflop_per_launch": round(flops_per_launch / 1e9, 1)},
        }
        if args.mode == "eval":
            out["metric"] = METRIC.replace("fwd+bwd", "forward only (inference: val_epoch / test)")
            out["config"]["workload"] = out["config"]["workload"].replace("training step", "INFERENCE step (eval mode, no backward / optimizer)")
        if eval_mode is not None:
            out["eval_mode"] = eval_mode
        if comm is not None:
            out["comm"] = comm
        if parity is not None:
            out["parity"] = parity
        if world == 1 and args.precision == "fp16h" and not args.no_fp16_leg and not args.no_parity and args.model == "film_attn_pt":
            # the other storage precisions' own lines (child processes: one 16-bit format per process) — NOT tolerance-compliant:
            # bf16 is BASELINE.json's storage dtype (logits ~7e-3 of exact fp32), fp16 the plain fp16 storage (0.6-1.0e-3 rms by seed)
            out["bf16_mode"] = fp16_leg(args, "bf16")
            out["fp16_mode"] = fp16_leg(args, "fp16")
        if cpu_leg is not None:
            out["cpu_baseline"] = cpu_leg
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
