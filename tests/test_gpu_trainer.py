"""GPU: the training step driver — stem prefetch pipeline on a side stream must not change results;
fused clip+Adam kernel vs torch.optim.Adam + clip_grad_norm_."""
import pytest
import torch
import torch.nn as nn

from helpers import LOW, LOW_DTYPE

pytestmark = pytest.mark.gpu


def _setup(precision="fp32", seed=0):
    from videonavqa_amd.models import FiLMAttnPretrainedStem, ObjDetectCNN
    from videonavqa_amd.stem import FrozenStem, VGGFront
    torch.manual_seed(seed)
    B, T, H, W, NF = 3, 4, 64, 96, 64
    vgg = VGGFront(precision)
    od = ObjDetectCNN(5, NF, 8, 0, True, True, precision=precision)
    with torch.no_grad():
        for conv in vgg.features.values():
            nn.init.kaiming_uniform_(conv.weight, a=1.0)
        for m in od.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1.0)
    model = FiLMAttnPretrainedStem(B, 16, 7, num_input_channels=NF, num_res_block_channels=64, num_res_blocks=2,
                                   hidden_size=16, at_hidden_size=16, max_num_frames=T, vocab_size=20,
                                   spatial_size=(H // 16) * (W // 16), precision=precision)
    vgg, od, model = vgg.cuda().eval(), od.cuda().eval(), model.cuda()
    stem = FrozenStem(vgg, od, precision)
    g = torch.Generator().manual_seed(5)
    batches = []
    for i in range(3):
        clip = torch.rand(B, 3, H, W, T, generator=g).cuda()
        q = torch.randint(1, 20, (B, 9), generator=g).cuda()
        v_lens = torch.tensor([[4, 2, 3], [4, 4, 4], [1, 4, 2]][i])
        q_lens = torch.randint(2, 10, (B,), generator=g)
        y = torch.randint(0, 7, (B,), generator=g).cuda()
        batches.append((clip, q, v_lens, q_lens, y))
    return model, stem, batches


def _run(overlap):
    from videonavqa_amd.train import Trainer
    model, stem, batches = _setup()
    tr = Trainer(model, stem, lr=1e-3)
    losses = []
    for i in range(6):
        b = batches[i % 3]
        nb = batches[(i + 1) % 3]
        kw = dict(next_clip=nb[0], next_v_lens_cpu=nb[2]) if overlap else {}
        loss, logits = tr.step(*b, **kw)
        losses.append(float(loss))
    torch.cuda.synchronize()
    return losses, tr.fp.flat.clone()


def test_stem_prefetch_pipeline_is_transparent():
    l0, w0 = _run(False)
    l1, w1 = _run(True)
    # torch's scatter/index_put/embedding backward use float atomics, so two runs of the SAME
    # configuration already differ in the last bits; a race in the pipeline would be O(1) off
    assert all(abs(a - b) <= 1e-5 * max(1.0, abs(a)) for a, b in zip(l0, l1)), (l0, l1)
    # Adam turns a noise-level gradient into a +-lr step, so single elements may differ by O(lr);
    # the bulk must agree tightly
    d = (w0 - w1).abs()
    assert float(d.max()) < 6e-3            # half of the maximum possible Adam travel 2 * lr * steps
    assert float(torch.quantile(d[:1000000], 0.999)) < 1e-5
    assert l0[-1] < l0[0] * 1.5  # finite, sane


def test_fused_clip_adam_matches_torch():
    from videonavqa_amd import kernels as K
    torch.manual_seed(1)
    n = 100003 + 1  # multiple of 4
    p = torch.randn(n, device="cuda")
    p_ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=1e-3)
    m = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    partial = torch.zeros(1024, device="cuda")
    for step in range(1, 4):
        g = torch.randn(n, device="cuda") * (3.0 if step == 2 else 0.001)
        p_ref.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([p_ref], 1.0)
        opt.step()
        gg = g.clone()
        K.clip_adam_step(p, gg, m, v, partial, step, 1e-3, 1.0)
        assert float(gg.abs().max()) == 0.0          # zero_grad fused
        assert float((p - p_ref.detach()).abs().max()) < 2e-6


def test_bf16_and_fp32_paths_train_alike():
    """40 optimisation steps on one fixed minibatch (whole pipeline: stem -> FiLM-attn trunk -> clip -> Adam) in the
    benchmark precision (bf16 storage, bf16-operand weight-gradient GEMMs, hardware exp2/rcp LSTM activations) and in
    the exact-f32 parity precision: both must fit the batch and their loss curves must stay close."""
    curves = {}
    from videonavqa_amd.train import Trainer
    for prec in ("fp32", LOW):
        model, stem, batches = _setup(prec, seed=3)
        trainer = Trainer(model, stem, lr=3e-4)
        losses = []
        for _ in range(40):
            loss, _ = trainer.step(*batches[0])
            losses.append(float(loss))
        assert all(l == l and l < 1e4 for l in losses), losses        # finite
        curves[prec] = losses
    f, b = curves["fp32"], curves[LOW]
    assert f[-1] < 0.6 * f[0] and b[-1] < 0.6 * b[0], (f[0], f[-1], b[0], b[-1])
    assert abs(b[0] - f[0]) < 0.05 * abs(f[0]) + 0.05
    rel = [abs(x - y) / max(abs(y), 1e-3) for x, y in zip(b[:20], f[:20])]
    assert max(rel) < 0.25, rel


def test_overlapped_reducer_on_one_rank_rccl_group():
    """The N>1 gradient path (hook-driven async all-reduces on RCCL's stream, finish(), clip+Adam on the reduced
    buffer) exercised on the one GPU of this box through a ONE-rank RCCL group: it must reproduce the
    non-distributed trainer, including with the stem pipeline running on its side stream."""
    import socket
    import torch.distributed as dist
    from videonavqa_amd.train import OverlappedGradReducer, Trainer
    l0, w0 = _run(True)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        model, stem, batches = _setup()
        tr = Trainer(model, stem, lr=1e-3, collectives=True)
        tr.reducer = OverlappedGradReducer(tr.fp, 1, "mean", early_numel=256, active=True)
        assert len(tr.reducer.early) >= 3          # several parameters take the early (hook) path
        losses = []
        for i in range(6):
            b, nb = batches[i % 3], batches[(i + 1) % 3]
            loss, _ = tr.step(*b, next_clip=nb[0], next_v_lens_cpu=nb[2])
            losses.append(float(loss))
        torch.cuda.synchronize()
        w1 = tr.fp.flat.clone()
    finally:
        dist.destroy_process_group()
    assert all(abs(a - b) <= 1e-5 * max(1.0, abs(a)) for a, b in zip(l0, losses)), (l0, losses)
    d = (w0 - w1).abs()                      # same tolerances as the pipeline test above (Adam on noise-level gradients)
    assert float(d.max()) < 6e-3
    assert float(torch.quantile(d[:1000000], 0.999)) < 1e-5


def _two_backwards(direct):
    """flat gradient after TWO backward passes (different minibatches) without an optimizer step in between"""
    import os
    from videonavqa_amd.train import Trainer
    from videonavqa_amd import ops
    old = os.environ.get("VNQA_DIRECT_GRADS")
    os.environ["VNQA_DIRECT_GRADS"] = "1" if direct else "0"
    try:
        model, stem, batches = _setup(seed=3)
        tr = Trainer(model, stem, lr=1e-3)
    finally:
        if old is None:
            os.environ.pop("VNQA_DIRECT_GRADS", None)
        else:
            os.environ["VNQA_DIRECT_GRADS"] = old
    model.train()
    grads = []
    for b in batches[:2]:
        clip, q, v_lens, q_lens, y = b
        native, v_sorted, perm = tr.extract_features(clip, v_lens)
        perm_d = perm.cuda()
        model.init_hidden()
        logits = model(native, q.index_select(0, perm_d), v_sorted, q_lens[perm])
        loss = ops.cross_entropy(logits, y, row_perm=perm_d.to(torch.int32), reduction="sum")
        loss.backward()
        grads.append(tr.fp.grad.clone())
    torch.cuda.synchronize()
    return grads


def test_grad_sinks_accumulate_over_two_backward_passes():
    """a second backward before the buffer is zeroed must ADD to the flat gradient (as p.grad does), not
    overwrite the first pass's slice — the sink hands its slice out once per zeroing, later producers go through
    AccumulateGrad."""
    g_ref = _two_backwards(False)
    g_dir = _two_backwards(True)
    for a, b in zip(g_ref, g_dir):
        scale = float(a.abs().max())
        assert scale > 0
        # (two separately built runs: float atomics in torch's scatter / index_put backward and split-K slab orders give
        # noise up to ~1e-3 of the largest gradient; an overwrite instead of an accumulation would be O(1) off)
        assert float((a - b).abs().max()) <= 5e-3 * scale, float((a - b).abs().max()) / scale
    # and the second pass really added something
    assert float((g_dir[1] - g_dir[0]).abs().max()) > 1e-6 * float(g_dir[0].abs().max())


def _first_steps(env_prio, monkeypatch, direct_prefetch=False):
    """Losses of steps 0..3 where step 0's clip was NOT prefetched (its stem runs inline on the trunk stream) and step 0
    immediately prefetches step 1's clip on the stem stream: the two stem passes share every intermediate buffer."""
    from videonavqa_amd.train import Trainer
    if env_prio is None:
        monkeypatch.delenv("VNQA_TRUNK_PRIO", raising=False)
    else:
        monkeypatch.setenv("VNQA_TRUNK_PRIO", env_prio)
    model, stem, batches = _setup()
    tr = Trainer(model, stem, lr=1e-3)
    losses = []
    for i in range(4):
        b, nb = batches[i % 3], batches[(i + 1) % 3]
        if direct_prefetch and i == 2:
            # a caller that prefetches by itself between steps, then hands step() a DIFFERENT clip: the stale prefetch is
            # discarded and the inline stem must wait for it
            tr.prefetch(batches[0][0], batches[0][2])
        loss, _ = tr.step(*b, next_clip=nb[0], next_v_lens_cpu=nb[2])
        losses.append(float(loss))
    torch.cuda.synchronize()
    return losses


@pytest.mark.parametrize("direct", [False, True])
def test_inline_first_step_then_prefetch_shares_no_buffers_in_flight(monkeypatch, direct):
    """ADVICE r3 (train.py: two stems at once on shared intermediate buffers): with the trunk on its own stream the stem
    stream used to wait only for the caller's stream, not for the stem that step 0 runs INLINE — so step 1's prefetched stem
    overwrote step 0's intermediates.  Pipelined losses must equal the single-stream run's."""
    ref = _first_steps("none", monkeypatch, direct)
    for _ in range(3):          # a race shows up intermittently: repeat
        got = _first_steps(None, monkeypatch, direct)
        assert all(abs(a - b) <= 1e-5 * max(1.0, abs(a)) for a, b in zip(ref, got)), (ref, got)


def test_clip_adam_skips_non_finite_gradients_and_counts_them():
    """Loss-scaled (fp16 storage) training: with an overflow counter the fused kernel skips an update whose gradient norm is
    not finite — parameters and moments untouched, gradients zeroed, counter incremented — and behaves as before otherwise."""
    from videonavqa_amd import kernels as K
    torch.manual_seed(2)
    n = 50000
    p, g = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
    m, v, partial = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(1024, device="cuda")
    count = torch.zeros(1, dtype=torch.int32, device="cuda")
    p0 = p.clone()
    for bad in (float("inf"), float("nan")):
        g.normal_()
        g[1234] = bad
        K.clip_adam_step(p, g, m, v, partial, 1, 1e-3, 1.0, overflow_count=count)
        assert torch.equal(p, p0) and float(m.abs().max()) == 0 and float(v.abs().max()) == 0
        assert float(g.abs().max()) == 0
    assert int(count) == 2
    g.normal_()
    ref_p, ref_g, ref_m, ref_v = p.clone(), g.clone(), m.clone(), v.clone()
    K.clip_adam_step(ref_p, ref_g, ref_m, ref_v, partial, 1, 1e-3, 1.0)               # no counter: the plain kernel
    K.clip_adam_step(p, g, m, v, partial, 1, 1e-3, 1.0, overflow_count=count)
    assert torch.equal(p, ref_p) and torch.equal(m, ref_m) and torch.equal(v, ref_v) and int(count) == 2


def test_dynamic_loss_scale_halves_after_overflow_and_grows_when_clean():
    from videonavqa_amd.models import common as C
    from videonavqa_amd.train import DynamicLossScale
    try:
        s = DynamicLossScale("cuda", init=1024.0, growth_interval=3)
        assert C.grad_scale_of(torch.float16) == 1024.0 and C.grad_scale_of(torch.bfloat16) == 1.0
        s.count += 2                       # two overflowed steps inside one observation window: ONE halving
        seen = [s.after_step() for _ in range(4)]
        # fixed lag: the copy started at step k is consumed at step k + LAG — never earlier, whatever has arrived
        assert seen == [0] * s.LAG + [2] + [0] * (3 - s.LAG) and s.skipped_steps == 2
        assert s.scale == 512.0 and C.grad_scale_of(torch.float16) == 512.0
        for _ in range(3):
            s.after_step()
        assert s.scale == 1024.0           # growth_interval clean observations double it
        d = s.state_dict()
        t = DynamicLossScale("cuda", init=64.0)
        t.load_state_dict(d)
        assert t.scale == s.scale and int(t.count) == 0 and d["skipped_steps"] == 2
    finally:
        C.set_fp16_loss_scale(C.FP16_GRAD_SCALE)


def test_eval_step_pipeline_matches_plain_per_batch_evaluation(monkeypatch):
    """Trainer.eval_step (inference path: forward-only fused trunk, next batch's stem prefetched on the stem stream, loss on the
    device) against the plain evaluation of every batch by itself on the op-by-op eval graph (VNQA_FUSED_EVAL=0, no
    pipeline): same logits (fp32: 1e-5), same losses, n_real slicing as q_and_v_test.py:123."""
    from videonavqa_amd.train import Trainer
    model, stem, batches = _setup()
    tr = Trainer(model, stem)
    model.bn_init.running_mean.normal_(0, 0.3)           # non-trivial running statistics to fold
    model.bn_init.running_var.uniform_(0.5, 1.5)
    ref = []
    monkeypatch.setenv("VNQA_FUSED_EVAL", "0")
    model.eval()
    with torch.no_grad():
        for clip, q, v_lens, q_lens, y in batches:
            native, v_sorted, perm = tr.extract_features(clip, v_lens)
            pd = perm.cuda()
            model.init_hidden()
            out = model(native, q[pd], v_sorted, q_lens[perm])
            ref.append((out.clone(), float(nn.CrossEntropyLoss(reduction="sum")(out, y[pd])),
                        float(nn.CrossEntropyLoss(reduction="sum")(out[:2], y[pd][:2]))))
    monkeypatch.setenv("VNQA_FUSED_EVAL", "1")
    for rounds in range(2):
        for i, (clip, q, v_lens, q_lens, y) in enumerate(batches):
            nb = batches[(i + 1) % 3]
            loss, out, perm_d = tr.eval_step(clip, q, v_lens, q_lens, y, next_clip=nb[0], next_v_lens_cpu=nb[2])
            assert not model.training and out.shape == ref[i][0].shape
            assert float((out - ref[i][0]).abs().max()) < 1e-5 * max(1.0, float(ref[i][0].abs().max()))
            assert abs(float(loss) - ref[i][1]) < 1e-4 * max(1.0, abs(ref[i][1]))
            assert torch.equal(perm_d.long().cpu(), torch.sort(v_lens, descending=True, stable=True)[1])
    loss2, _, _ = tr.eval_step(*batches[0], n_real=2)        # padded last batch: loss over the first 2 sorted rows only
    assert abs(float(loss2) - ref[0][2]) < 1e-4 * max(1.0, abs(ref[0][2]))
    # training still works after evaluation on the same Trainer (model back in train mode, prefetch state consistent)
    l0, _ = tr.step(*batches[0], next_clip=batches[1][0], next_v_lens_cpu=batches[1][2])
    assert model.training and float(l0) == float(l0)
