"""One training step of the video-question fusion path, data-parallel over GPUs.

Restates the inner loop of train_epoch (eval/q_and_v_eval.py:84-139):
  frozen stem under no_grad (:102-110) -> sort by video length (:113-116) -> init_hidden (:120)
  -> forward (:121) -> CrossEntropyLoss (:124, reduction per --loss_reduction) -> backward (:136)
  -> clip_grad_norm 1.0 (:137) -> Adam (:138) -> zero_grad (:139)
with the MI355X-native pieces: the stem runs on all frames of the minibatch at once and hands
its output to the model in kernel-native layout; parameters and gradients live in ONE flat
fp32 buffer each, so the data-parallel gradient sum is a single RCCL all-reduce on that
buffer and clip+Adam+zero_grad are two HIP launches.

Data parallelism (new; the reference is single-GPU): one process per GPU, every rank holds a
full replica and its own minibatch; gradients are SUMMED across ranks (loss reduction 'sum',
eval.sh:16), the global-norm clip then applies to the reduced gradient, and every rank performs
the identical Adam update.  BatchNorm statistics stay rank-local (no SyncBN upstream).
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib as L
from . import kernels as K
from . import ops
from .models.common import FrameLayout, NativeFeatures


class FlatParams(object):
    """Re-homes a model's trainable parameters (and their .grad) into flat fp32 buffers."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        n_pad = (n + 3) // 4 * 4
        self.n = n_pad
        self.flat = torch.zeros(n_pad, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n_pad, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n_pad, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n_pad, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.grad[off:off + k].view_as(p)
            # the HIP ops' backward writes this parameter's gradient straight into its slice (ops.GradSink): valid because
            # the buffer is zeroed by the fused clip+Adam kernel every step and each parameter has one gradient producer
            if os.environ.get("VNQA_DIRECT_GRADS", "1") != "0":
                p._vnqa_grad_sink = ops.GradSink(p.grad)
            off += k
        self.partial = torch.zeros(1024, dtype=torch.float32, device=dev)
        self.step_count = 0

    def zero_grad(self):
        self.grad.zero_()
        self.mark_zeroed()

    def clip_adam_step(self, lr, clip=1.0, overflow_count=None):
        """Global-norm clip + Adam + zero_grad over the flat buffers (two HIP launches) AND the gradient sinks' reset that must
        go with every zeroing of the gradient buffer — one call, so that no training loop can forget the second half
        (a loop calling K.clip_adam_step alone left the sinks `written`, and every later backward silently took
        the temporary + AccumulateGrad route)."""
        self.step_count += 1
        K.clip_adam_step(self.flat, self.grad, self.m, self.v, self.partial, self.step_count, lr, clip,
                         overflow_count=overflow_count)
        self.mark_zeroed()

    def mark_zeroed(self):
        """The flat gradient buffer has just been zeroed (fused clip+Adam kernel, zero_grad): the next gradient producer of
        every parameter may write its slice in place again (ops.GradSink)."""
        for p in self.params:
            s = getattr(p, "_vnqa_grad_sink", None)
            if s is not None:
                s.written = False
                s._handed = False


class phase(object):
    """roctx range around a phase of the step (stem / trunk_fwd / backward / allreduce / optimizer) on the launch thread: shows up as a
    marker row in `rocprofv3 --marker-trace` timelines (torch.cuda.nvtx is roctx on ROCm builds).  Host-side only, ~1 us per range."""
    _nvtx = None

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if phase._nvtx is None:
            try:
                phase._nvtx = torch.cuda.nvtx if torch.cuda.is_available() else False
            except Exception:
                phase._nvtx = False
        if phase._nvtx:
            try:
                phase._nvtx.range_push(self.name)
            except Exception:
                phase._nvtx = False
        return self

    def __exit__(self, *exc):
        if phase._nvtx:
            phase._nvtx.range_pop()
        return False


class DynamicLossScale(object):
    """Loss scale of the fp16-storage precision, adjusted like torch.amp.GradScaler but WITHOUT a host sync in the step and
    DETERMINISTICALLY: the fused clip+Adam kernel itself skips an update whose gradient norm is not finite and counts
    it in a device int32 (`count`; the same kernel derives Adam's bias correction from launches - count ON THE DEVICE, so no host
    read-back enters an update).  After every step the host starts an asynchronous copy of the counter into one of LAG + 1 pinned
    words and consumes the copy started exactly LAG steps earlier (`event.synchronize()` on a copy that old returns at once) —
    never "whichever copy happens to have arrived": data-parallel replicas see identical reduced gradients, hence identical
    counters, hence take every scale decision at the same step whatever their host timing.  One overflow EPISODE halves the scale
    once: after a halving, the overflows of the LAG steps that were already launched with the old scale (observed over the next LAG
    calls) are counted but do not halve again (they cost scale / 8 and ~600 clean steps to regrow);
    `growth_interval` clean observations double it (cap 2^24).
    Powers of two only: scaling is exact.  With `pinned_scale` the scale never moves (the counter still counts).
    scale = 1 and growth off is the form every other precision uses (overflow skip + statistics only)."""
    LAG = 2

    def __init__(self, device, init=None, growth_interval=200, max_scale=2.0 ** 24, min_scale=1.0, applies=True):
        from .models import common as C
        self._C = C
        self.applies = bool(applies)         # False: a counter only (precisions without a loss scale)
        self.scale = float(C.FP16_GRAD_SCALE if init is None else init) if self.applies else 1.0
        self.growth_interval, self.max_scale, self.min_scale = int(growth_interval), float(max_scale), float(min_scale)
        self.count = torch.zeros(1, dtype=torch.int32, device=device)
        self._cuda = torch.device(device).type == "cuda"
        self._host = [torch.zeros(1, dtype=torch.int32) for _ in range(self.LAG + 1)]
        if self._cuda:
            self._host = [t.pin_memory() for t in self._host]
        self._events = [None] * (self.LAG + 1)
        self._steps, self._seen, self._clean = 0, 0, 0
        self._ignore_until = 0               # observations at calls < this index report steps launched before the last halving took effect
        self._skipped_before = 0             # skipped steps of the run(s) before the last load_state_dict (the device counter restarts at 0)
        self.skipped_steps = 0
        if self.applies:
            C.set_fp16_loss_scale(self.scale)

    def after_step(self):
        """Call right after the clip+Adam launch: start this step's read-back, consume the one started LAG steps ago.
        Returns the number of skipped steps newly observed (statistics; the device keeps Adam's step count itself)."""
        new = 0
        slot = self._steps % (self.LAG + 1)
        old = (self._steps + 1) % (self.LAG + 1)          # the slot written LAG steps ago (and re-used next step)
        self._host[slot].copy_(self.count, non_blocking=True)
        if self._cuda:
            ev = torch.cuda.Event()
            ev.record()
        else:                                             # (host tensors — the CPU tests of the host logic: the copy is done)
            ev = True
        self._events[slot] = ev
        if self._events[old] is not None:
            if self._cuda:
                self._events[old].synchronize()
            seen = int(self._host[old][0])
            new = seen - self._seen
            self._seen = seen
        call = self._steps
        self._steps += 1
        if new > 0:
            self.skipped_steps += new
            self._clean = 0
            if self.applies and call >= self._ignore_until:
                self.scale = max(self.scale * 0.5, self.min_scale)
                # the new scale reaches the step launched after this call; the LAG steps before it ran with the old one and are
                # observed at the next LAG calls
                self._ignore_until = call + 1 + self.LAG
        else:
            self._clean += 1
            if self.applies and self._clean >= self.growth_interval:
                self.scale = min(self.scale * 2.0, self.max_scale)
                self._clean = 0
        if self.applies:
            self._C.set_fp16_loss_scale(self.scale)
        return new

    def skipped_now(self):
        """The device counter, read synchronously (checkpoint time)."""
        return int(self.count.item())

    def state_dict(self):
        return {"scale": self.scale, "clean_steps": self._clean, "skipped_steps": self._skipped_before + self.skipped_now()}

    def load_state_dict(self, d):
        """Restores the scale and the growth counter; the device counter restarts at zero together with the launch count
        (Trainer.load_checkpoint sets the launch count to the checkpoint's APPLIED steps)."""
        if self.applies:
            self.scale = float(d.get("scale", self.scale))
            self._C.set_fp16_loss_scale(self.scale)
        self._clean = int(d.get("clean_steps", 0))
        self._skipped_before = self.skipped_steps = int(d.get("skipped_steps", 0))      # cumulative over resumes (saved as before + device count)
        self.count.zero_()
        self._events = [None] * (self.LAG + 1)
        self._steps, self._seen, self._ignore_until = 0, 0, 0


def sync_replicas(tensors, src=0):
    """Make every rank hold rank `src`'s tensors (weights, buffers AND the frozen unregistered
    conv1x1 layers, which state_dict() does not carry — SURVEY §0.5)."""
    for t in tensors:
        dist.broadcast(t, src=src)


class OverlappedGradReducer(object):
    """SUM all-reduce of the flat gradient buffer, overlapped with backward.

    Big parameters (>= `early_numel` elements; for FiLM-attn at 224x224 that is fc_embed_attn.weight = 51 MB of
    the 73 MB payload, whose gradient is complete early in backward, and the two 9.4 MB conv weights) are reduced
    asynchronously as soon as their gradient kernel has been enqueued (ops.GradSink notification / post-accumulate-grad
    hook), each as its own contiguous slice of the flat buffer; `finish()` reduces the remaining ranges (~1.5 MB of
    biases, LSTM and classifier weights) and waits for the early ones.  xGMI rings are per-link bound, so few large
    transfers beat many small buckets here."""

    def __init__(self, fp, world_size, loss_reduction="sum", early_numel=1 << 20, active=None):
        self.fp, self.world, self.loss_reduction = fp, world_size, loss_reduction
        self.early, self.pending, self.done_ranges, self._fired = {}, [], [], set()
        # active=True with world_size 1 runs the collectives on a one-rank group (RCCL code-path test on a 1-GPU box)
        self.enabled = world_size > 1 if active is None else bool(active)
        off = 0
        for p in fp.params:
            k = p.numel()
            if self.enabled and k >= early_numel:
                self.early[p] = (off, off + k)
                p.register_post_accumulate_grad_hook(self._hook)
                sink = getattr(p, "_vnqa_grad_sink", None)
                if sink is not None:      # gradient written in place by its kernel: no AccumulateGrad node fires the hook
                    sink.on_ready = (lambda q=p: self._hook(q))
            off += k

    def _hook(self, p):
        # Fires once per parameter and step whichever way the gradient arrived: from autograd's AccumulateGrad node (which
        # runs its post-accumulate hooks even when the producing node returned None because it wrote the gradient in place)
        # or from the producing kernel's GradSink.
        if not self.enabled or p in self._fired:
            return
        self._fired.add(p)
        a, b = self.early[p]
        self.pending.append(dist.all_reduce(self.fp.grad[a:b], op=dist.ReduceOp.SUM, async_op=True))
        self.done_ranges.append((a, b))

    def finish(self):
        """Call after backward: reduce what the hooks did not cover, wait for everything."""
        if not self.enabled:
            self._fired.clear()
            return
        cur = 0
        for a, b in sorted(self.done_ranges) + [(self.fp.n, self.fp.n)]:
            if a > cur:
                self.pending.append(dist.all_reduce(self.fp.grad[cur:a], op=dist.ReduceOp.SUM, async_op=True))
            cur = max(cur, b)
        for w in self.pending:
            w.wait()
        self.pending, self.done_ranges = [], []
        self._fired.clear()
        if self.loss_reduction != "sum":
            self.fp.grad.div_(self.world)


def allreduce_gradients(flat_grad, world_size, loss_reduction="sum"):
    """The one data-path collective of a step: SUM the flat gradient buffer over ranks (RCCL on GPUs).
    With reduction='sum' (eval.sh:16) the summed gradient IS the gradient of the global-batch loss;
    'mean' losses are averaged over ranks."""
    if world_size > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        if loss_reduction != "sum":
            flat_grad.div_(world_size)


class Trainer(object):
    def __init__(self, model, stem, lr=1e-4, clip=1.0, loss_reduction="sum", class_weights=None,
                 world_size=1, rank=0, feature_channels=512, collectives=None, feature_slots=2):
        self.model, self.stem = model, stem
        self.lr, self.clip = lr, clip
        self.world_size, self.rank = world_size, rank
        self.loss_fn = nn.CrossEntropyLoss(weight=class_weights, reduction=loss_reduction)    # (eval loops use this form)
        self.class_weights = None if class_weights is None else class_weights.detach().float().contiguous()
        self.loss_reduction = loss_reduction
        self.feature_channels = feature_channels
        collectives = world_size > 1 if collectives is None else bool(collectives)
        if collectives:
            self.sync_replicas()
        self.fp = FlatParams(model.parameters())
        # software pipeline: the frozen stem of the NEXT minibatch runs on a side stream while this
        # minibatch's trunk forward/backward runs on the main stream (two output slots)
        self.reducer = OverlappedGradReducer(self.fp, world_size, loss_reduction, active=collectives)
        self.stem_device = self.fp.flat.device
        self.copy_stream = None
        prio = 0          # (raising the STEM's priority instead of the trunk's costs 9 %: DESIGN 5)
        # CU partition: a model whose trunk is a long dependent chain of SMALL kernels (MACNetwork: ~1 000 launches per step)
        # cannot overlap with full-chip stem kernels that own every CU's LDS — each small kernel would wait for stem workgroups to
        # retire.  Such a model asks for `stem_reserve_cus` CUs the stem stream never touches (its own attribute; env
        # VNQA_STEM_RESERVE_CUS overrides, 0 = off): the chain runs there at once, the stem on the rest.
        reserve = os.environ.get("VNQA_STEM_RESERVE_CUS")
        reserve = int(reserve) if reserve is not None else int(getattr(model, "stem_reserve_cus", 0))
        self.stem_reserve_cus = reserve if (stem is not None and self.fp.flat.is_cuda) else 0
        if self.stem_reserve_cus > 0:
            try:
                self.stem_stream = L.reserved_stream(self.stem_reserve_cus, self.stem_device)
            except L.VnqaError as e:          # (a device whose CU count the mask layout was not measured on: run un-partitioned)
                import warnings
                warnings.warn("stem CU reservation unavailable (%s): the stem runs on an ordinary stream" % e)
                self.stem_reserve_cus = 0
        if self.stem_reserve_cus == 0:
            # (host tensors — the gloo tests of the data-parallel host logic, tests/test_dp_gloo.py — have no streams: the step's
            # device-independent half, _reduce_and_update, is all such a Trainer runs)
            self.stem_stream = torch.cuda.Stream(priority=prio) if self.fp.flat.is_cuda else None
        elif stem is not None:
            # the persistent conv kernels size their grids for the masked stream's CUs: a per-call descriptor field of THIS stem
            stem.reserve_cus = max(int(getattr(stem, "reserve_cus", 0)), self.stem_reserve_cus)
        # The trunk (the step's dependent chain: question LSTMs, FiLM blocks, attention tail, backward, Adam) runs on its own
        # HIGH-priority stream by default: its kernels are dispatched ahead of the co-running stem's whenever both have
        # workgroups pending, so the chain finishes sooner and the stem fills what is left (same-box A/B, 4 rounds each:
        # 846 -> 876 clips/s, +3.5 %).  VNQA_TRUNK_PRIO=none: the caller's stream, as before; any integer: that priority.
        tprio = os.environ.get("VNQA_TRUNK_PRIO", "-1")
        use_ts = tprio.lower() != "none" and self.fp.flat.is_cuda
        self.trunk_stream = torch.cuda.Stream(priority=int(tprio)) if use_ts else None
        # (the FiLM generator's parameters accumulate their gradients on its side stream by design)
        if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        self._prefetched = None          # (key, NativeFeatures, v_sorted, perm, done_event)
        # Feature slots: the stem of minibatch i+1 writes slot (i+1) % n and may start as soon as the trunk pass that last read that
        # slot is done.  With two slots that is the trunk of minibatch i-1 (stem and trunk start in lockstep); a third slot
        # (`feature_slots=3`: the stem runs up to two minibatches ahead and never waits for the trunk, one more feature buffer)
        # measured equal (931 vs 934 clips/s, three interleaved rounds) once the question chain runs on its side stream: the step
        # is then bound by summed kernel work, not by who waits for whom — so the default stays 2.
        self._n_slots = max(2, int(feature_slots))
        self._slot = 0
        self._trunk_done = [None] * self._n_slots  # event per slot: last trunk pass that read that slot
        self._inputs_ready = None        # recorded on the CALLER's stream at step() entry: clips / labels produced so far
        # fp16 storage: dynamic loss scale (VNQA_FP16_LOSS_SCALE=<value> pins it: no adjustment, the round-2/3 behaviour)
        # Every precision gets the device-side skip of a non-finite step and its counter; only fp16 STORAGE has a loss scale.
        pinned = os.environ.get("VNQA_FP16_LOSS_SCALE")
        self.loss_scaler = DynamicLossScale(self.fp.flat.device, init=float(pinned) if pinned else None,
                                            growth_interval=(1 << 62) if pinned else 200,
                                            applies=getattr(model, "compute_dtype", None) == torch.float16)
        self._in_step = False            # prefetch() called from inside step() (current stream = the trunk stream) or by the caller
        self._inline_stem_done = None    # event after a stem pass that ran INLINE on the trunk / caller's stream (shared buffers)

    def sync_replicas(self):
        """Rank 0's weights everywhere, INCLUDING the frozen unregistered conv1x1 layers
        (which state_dict() does not carry)."""
        sync_replicas(list(self.model.state_dict().values()) + list(self.model.extra_state_tensors().values()))
        plan = getattr(self.stem, "packed_tensors", None)       # the frozen stem's rounded 16-bit weights (stem.FrozenStem)
        if plan is not None:
            # EVERY plan tensor (non-contiguous ones used to be skipped silently — replica identity is the point)
            for t in plan():
                if t.is_contiguous():
                    dist.broadcast(t, src=0)
                else:
                    c = t.contiguous()
                    dist.broadcast(c, src=0)
                    t.copy_(c)

    # ---- checkpoint interface with torch.optim.Adam's state_dict layout (eval/q_and_v_eval.py:148-156,344-345) --
    def optimizer_state_dict(self):
        """Same structure as torch.optim.Adam(model.parameters()).state_dict(): per-parameter
        step / exp_avg / exp_avg_sq, one param group."""
        state, off = {}, 0
        # torch's `step` = updates APPLIED = clip+Adam launches minus the launches the device skipped (non-finite gradients)
        applied = self.fp.step_count - (self.loss_scaler.skipped_now() if self.loss_scaler is not None else 0)
        for i, p in enumerate(self.fp.params):
            k = p.numel()
            state[i] = {"step": torch.tensor(float(applied)),
                        "exp_avg": self.fp.m[off:off + k].view_as(p).clone(),
                        "exp_avg_sq": self.fp.v[off:off + k].view_as(p).clone()}
            off += k
        group = {"lr": self.lr, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False,
                 "params": list(range(len(self.fp.params)))}
        return {"state": state, "param_groups": [group]}

    def extra_state_dict(self):
        """CPU copies of the tensors state_dict() does not carry (frozen conv1x1_layers) for the checkpoint's
        'extra_state' key."""
        fn = getattr(self.model, "extra_state_tensors", None)
        out = {k: v.detach().cpu().clone() for k, v in fn().items()} if fn else {}
        if self.loss_scaler is not None and self.loss_scaler.applies:
            # the dynamic loss scale travels with the checkpoint (a resume restarted at 2^10)
            out["_loss_scale"] = dict(self.loss_scaler.state_dict())
        calib = getattr(self.stem, "calib", None)
        if calib is not None:
            # what the frozen stem's 16-bit weights were rounded against (stem.second_order_round / coherent_round): the calibration
            # FRAMES ("noise" = the seeded default, else the tensor of frames) and the input means measured on them — the test-time
            # stem redoes the rounding from them, so a model is tested behind the very stem weights it was trained behind
            out["_stem_calibration"] = {k: (v if isinstance(v, str) else v.detach().cpu().clone()) for k, v in calib.items()}
            if hasattr(self.stem, "packs_checksum"):
                # ... and a checksum of the rounded packs themselves: the re-rounding is bit-reproducible on the same GPU / ROCm / torch
                # build only (a different reduction order can flip a tie) — the loader verifies and says so
                if getattr(self, "_stem_sha", None) is None:
                    self._stem_sha = self.stem.packs_checksum()
                out["_stem_packs_sha256"] = self._stem_sha
        return out

    def load_checkpoint(self, ckpt):
        """Restore model + optimizer from a reference-schema checkpoint dict."""
        self.model.load_state_dict(ckpt["state_dict"])      # in-place copies: parameters stay views of the flat buffer
        if ckpt.get("extra_state") and hasattr(self.model, "load_reference_tensors"):
            # the frozen conv1x1_layers the reference leaves out of state_dict(); upstream loaders ignore the key
            self.model.load_reference_tensors(ckpt["extra_state"])
        opt = ckpt.get("optimizer")
        if opt and opt.get("param_groups"):
            # optimizer.load_state_dict restores the saved learning rate upstream too (eval/q_and_v_eval.py:345)
            self.lr = float(opt["param_groups"][0].get("lr", self.lr))
        if opt and opt.get("state"):
            off = 0
            for i, p in enumerate(self.fp.params):
                k = p.numel()
                st = opt["state"].get(i)
                if st is not None:
                    self.fp.m[off:off + k].copy_(st["exp_avg"].reshape(-1))
                    self.fp.v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                    self.fp.step_count = int(st["step"])
                off += k
        if self.loss_scaler is not None:       # launch count := applied steps, device skip counter := 0, saved scale restored
            self.loss_scaler.load_state_dict((ckpt.get("extra_state") or {}).get("_loss_scale") or {})

    def to_device_async(self, t):
        return L.to_device_async(t, self.stem_device)

    def extract_features(self, clip, v_lens_cpu, slot=0):
        """Stem + batch sort on the CURRENT stream.  clip fp32 [B,3,H,W,T] on the GPU."""
        B, _, H, W, T = clip.shape
        v_sorted, perm = torch.sort(v_lens_cpu, dim=0, descending=True, stable=True)
        lay = FrameLayout(v_sorted, T, clip.device, perm=perm)
        with phase("stem"):
            feats = self.stem.forward_clip(clip, lay.img_of, lay.n_img, slot=slot)
        segs = int(getattr(self.stem, "feature_segs", 1))       # stated by the producer (FrozenStem), checked against the shape
        return NativeFeatures(feats, lay, self.feature_channels, H // 16, W // 16, segs=segs,
                              shift=getattr(self.stem, "feature_shift", None)), v_sorted, perm

    def upload(self, clip_host):
        """Start the H2D copy of a (pinned) host clip on the copy stream into one of three rotating device
        buffers and return that device tensor; consumers wait on its event.  With upload(i+2) issued while
        stem(i+1) and trunk(i) run, the copy engine, the stem and the trunk form a 3-stage pipeline."""
        if self.copy_stream is None:
            self.copy_stream = torch.cuda.Stream()
            self._up_bufs, self._up_idx, self._up_events, self._up_read_done = [None, None, None], 0, {}, {}
        i = self._up_idx
        self._up_idx = (i + 1) % 3
        if self._up_bufs[i] is None or self._up_bufs[i].shape != clip_host.shape:
            self._up_bufs[i] = torch.empty(clip_host.shape, dtype=clip_host.dtype, device=self.stem_device)
        buf = self._up_bufs[i]
        # The copy may overwrite this buffer once the stem pass that last READ it has finished — an event recorded right after
        # that pass on whichever stream ran it (_mark_clip_read).  Rounds 1-3 made the copy wait for the whole stem stream
        # instead, i.e. for stem(i+1), which had just been enqueued: the copy of clip i+2 then started only when stem(i+1)
        # ended and stem(i+2) waited for the copy — the stem stream, the pipeline's bottleneck, idled for the length of every
        # copy (-4 % with fp32 clips AND with uint8 clips a quarter the size: profiles/r04_h2d.txt).
        ev = self._up_read_done.get(buf.data_ptr())
        if ev is not None:
            self.copy_stream.wait_event(ev)
        with torch.cuda.stream(self.copy_stream):
            buf.copy_(clip_host, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self._up_events[buf.data_ptr()] = ev
        return buf

    def _wait_upload(self, clip):
        ev = self._up_events.get(clip.data_ptr()) if self.copy_stream is not None else None
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def _mark_clip_read(self, clip):
        """Right after a stem pass over `clip` on the current stream: if it is one of upload()'s rotating buffers, record that its
        last reader is done (the next copy into that buffer waits for exactly this)."""
        if self.copy_stream is not None and clip.is_cuda and clip.data_ptr() in self._up_events:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._up_read_done[clip.data_ptr()] = ev

    def prefetch(self, clip, v_lens_cpu):
        """Start the stem of an upcoming minibatch on the side stream (returns immediately)."""
        slot = (self._slot + 1) % self._n_slots
        main = torch.cuda.current_stream()
        if self._in_step and self._inputs_ready is not None and self.trunk_stream is not None:
            self.stem_stream.wait_event(self._inputs_ready)  # the caller's stream up to this step() call (NOT the trunk stream: the
        else:                                                # stem must not wait for the previous minibatch's trunk)
            self.stem_stream.wait_stream(main)               # clip / layout uploads issued so far (also: prefetch() called directly)
        if self._inline_stem_done is not None:
            # A stem pass ran inline on the trunk stream (a step whose clip had not been prefetched: first step of an epoch,
            # bench priming).  Only the last layer's output is per slot — every other stem buffer is shared — so this pass
            # must not start before that one has finished.
            self.stem_stream.wait_event(self._inline_stem_done)
            self._inline_stem_done = None
        if self._trunk_done[slot] is not None:
            self.stem_stream.wait_event(self._trunk_done[slot])   # that slot's previous reader
        with torch.cuda.stream(self.stem_stream):
            key = clip.data_ptr()
            if not clip.is_cuda:     # host (pinned) clip: the H2D copy rides the stem stream (see upload() for a deeper pipeline)
                clip = clip.to(self.stem_device, non_blocking=True)
            self._wait_upload(clip)
            native, v_sorted, perm = self.extract_features(clip, v_lens_cpu, slot=slot)
            self._mark_clip_read(clip)
            done = torch.cuda.Event()
            done.record(self.stem_stream)
        self._prefetched = (key, native, v_sorted, perm, done, slot)

    def _on_trunk_stream(self, fn, *args, **kw):
        """Run fn on the trunk's own high-priority stream, ordered after the caller's stream and joined back to it."""
        if self.trunk_stream is None:
            return fn(*args, **kw)
        outer = torch.cuda.current_stream()
        self._inputs_ready = torch.cuda.Event()
        self._inputs_ready.record(outer)
        self.trunk_stream.wait_stream(outer)
        with torch.cuda.stream(self.trunk_stream):
            self._in_step = True
            try:
                out = fn(*args, **kw)
            finally:
                self._in_step = False
        outer.wait_stream(self.trunk_stream)
        for t in out:                 # (allocated on the trunk stream, consumed by the caller on its own)
            if torch.is_tensor(t):
                t.record_stream(outer)
        return out

    def step(self, clip, q_input, v_lens_cpu, q_lens_cpu, ys, next_clip=None, next_v_lens_cpu=None):
        return self._on_trunk_stream(self._step, clip, q_input, v_lens_cpu, q_lens_cpu, ys, next_clip, next_v_lens_cpu)

    def eval_step(self, clip, q_input, v_lens_cpu, q_lens_cpu, ys, next_clip=None, next_v_lens_cpu=None, n_real=None):
        """Forward-only counterpart of step() for val_epoch / test (eval/q_and_v_eval.py:188-213, eval/q_and_v_test.py:
        100-131): model.eval() under no_grad on the inference path (fused forward-only trunk), with the SAME software pipeline
        as training — the frozen stem of `next_clip` runs on the stem stream beside this minibatch's trunk — and nothing
        read back: returns (loss, logits, perm_d) as device tensors (logits rows in length-sorted order, perm_d the sort
        permutation; loss over the first `n_real` sorted rows — q_and_v_test.py:123 slices a padded last batch that way)."""
        return self._on_trunk_stream(self._eval_step, clip, q_input, v_lens_cpu, q_lens_cpu, ys, next_clip, next_v_lens_cpu,
                                     n_real)

    def _features_for(self, clip, v_lens_cpu, main):
        """This minibatch's stem output: the prefetched one when `clip` is what prefetch() was given, else computed inline."""
        if self._prefetched is not None and self._prefetched[0] == clip.data_ptr():
            _, native, v_sorted, perm, done, slot = self._prefetched
            main.wait_event(done)
            native.layout.record_stream(main)
            self._slot = slot
        else:
            if not clip.is_cuda:
                clip = clip.to(self.stem_device, non_blocking=True)
            self._wait_upload(clip)
            # inline stem: a discarded / stale prefetch may still be running on the stem stream and shares every intermediate
            # buffer with this pass — wait for it, and make the next prefetch wait for this pass (event below)
            main.wait_stream(self.stem_stream)
            native, v_sorted, perm = self.extract_features(clip, v_lens_cpu, slot=self._slot)
            self._mark_clip_read(clip)
            self._inline_stem_done = torch.cuda.Event()
            self._inline_stem_done.record(main)
        self._prefetched = None
        return native, v_sorted, perm

    def _eval_step(self, clip, q_input, v_lens_cpu, q_lens_cpu, ys, next_clip=None, next_v_lens_cpu=None, n_real=None):
        self.model.eval()
        main = torch.cuda.current_stream()
        with torch.no_grad():
            native, v_sorted, perm = self._features_for(clip, v_lens_cpu, main)
            if next_clip is not None:
                self.prefetch(next_clip, next_v_lens_cpu if next_v_lens_cpu is not None else v_lens_cpu)
            perm_d = L.to_device_async(perm.to(torch.int32), self.stem_device)
            if hasattr(self.model, "init_hidden"):
                self.model.init_hidden()
            logits = self.model(native, q_input.index_select(0, perm_d), v_sorted, q_lens_cpu[perm])
            n = logits.shape[0] if n_real is None else int(n_real)
            loss = ops.cross_entropy(logits[:n], ys, row_perm=perm_d[:n], weight=self.class_weights,
                                     reduction=self.loss_reduction)
        ev = torch.cuda.Event()
        ev.record(main)
        self._trunk_done[self._slot] = ev
        return loss.detach(), logits, perm_d

    def _reduce_and_update(self):
        """The device-independent second half of a step, after backward: finish the gradient all-reduce (the early slices are already
        in flight), clamp (MACNetwork), global-norm clip + Adam + zero_grad with the device-side skip of a non-finite step, then the
        lagged loss-scale observation.  Runs unchanged on host tensors over gloo (tests/test_dp_gloo.py: 4 and 8 ranks) with
        FlatParams.clip_adam_step's HIP launch replaced by its restatement."""
        with phase("allreduce"):
            self.reducer.finish()
        with phase("optimizer"):
            clamp = getattr(self.model, "grad_clamp", None)
            if clamp:    # MACNetwork: per-parameter gradient clamp hooks (eval/q_and_v_eval.py:348-351), on the reduced gradient
                self.fp.grad.clamp_(-clamp, clamp)
            self.fp.clip_adam_step(self.lr, self.clip,        # (the kernel zeroes the gradient buffer; the sinks are reset with it)
                                   overflow_count=None if self.loss_scaler is None else self.loss_scaler.count)
        if self.loss_scaler is not None:      # (updates skipped on the device are taken off Adam's step count ON the device)
            self.loss_scaler.after_step()

    def _step(self, clip, q_input, v_lens_cpu, q_lens_cpu, ys, next_clip=None, next_v_lens_cpu=None):
        """One optimisation step.  v_lens_cpu / q_lens_cpu are host int64 tensors (as a DataLoader
        delivers them); clip, q_input, ys are on the GPU.  If `next_clip` is given, its stem is
        launched on the side stream so that it overlaps this step's trunk.  Returns (loss, logits) —
        logits rows in length-sorted order like the reference (:121-130)."""
        self.model.train()
        main = torch.cuda.current_stream()
        native, v_sorted, perm = self._features_for(clip, v_lens_cpu, main)
        if next_clip is not None:
            self.prefetch(next_clip, next_v_lens_cpu if next_v_lens_cpu is not None else v_lens_cpu)
        perm_d = L.to_device_async(perm.to(torch.int32), self.stem_device)
        if hasattr(self.model, "init_hidden"):     # `--model mac` has none (eval/q_and_v_eval.py:119-120)
            self.model.init_hidden()
        with phase("trunk_fwd"):
            logits = self.model(native, q_input.index_select(0, perm_d), v_sorted, q_lens_cpu[perm])
            # CrossEntropyLoss forward + d logits as one HIP launch; the targets are read through the batch-sort permutation
            loss = ops.cross_entropy(logits, ys, row_perm=perm_d, weight=self.class_weights, reduction=self.loss_reduction)
        with phase("backward"):
            loss.backward()
        self._reduce_and_update()
        ev = torch.cuda.Event()
        ev.record(main)
        self._trunk_done[self._slot] = ev
        return loss.detach(), logits.detach()
