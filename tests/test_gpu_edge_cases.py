"""GPU parity on edge-case batches the goldens do not hold: single-frame videos, single-token and maximum-length
(56) questions, all videos shorter than max_num_frames.  Checker: the oracle (itself pinned to the reference goldens
in test_oracle_golden.py) on the same seeded inputs and the product's own randomly initialised weights; fp32 path,
logits within 1e-3 relative (the north-star bound) with equal argmax, train-mode gradients within 2e-3."""
import numpy as np
import pytest
import torch

from oracle import vnqa_oracle as O
from helpers import rel_err

pytestmark = pytest.mark.gpu

T, L, V, K, CIN, C = 6, 56, 20, 7, 8, 8


def _batch(seed, v_lens, q_lens, h=10, w=13, cin=CIN):
    g = torch.Generator().manual_seed(seed)
    B = len(v_lens)
    v = torch.rand(B, cin, h, w, T, generator=g) * 1.5
    q = torch.zeros(B, L, dtype=torch.long)
    for b, n in enumerate(q_lens):
        q[b, :n] = torch.randint(1, V, (n,), generator=g)
    y = torch.randint(0, K, (B,), generator=g)
    return v, q, torch.tensor(v_lens), torch.tensor(q_lens), y


def _weights(model):
    W = {k: t.detach().cpu().clone() for k, t in model.state_dict().items()}
    W.update({k: t.detach().cpu().clone() for k, t in model.extra_state_tensors().items()})
    return W


EDGE = [
    ("single_frame_videos", [1, 1, 1], [3, 1, 2]),
    ("one_long_rest_single", [6, 1, 1], [56, 1, 1]),
    ("max_len_questions", [5, 4, 4], [56, 56, 55]),
    ("short_videos_single_tokens", [2, 2, 1], [1, 1, 1]),
]


@pytest.mark.parametrize("name,v_lens,q_lens", EDGE)
@pytest.mark.parametrize("model_name", ["film_attn_pt", "film_gp_pt", "time_multi_hop"])
def test_film_models_edge_batches_vs_oracle(model_name, name, v_lens, q_lens):
    import videonavqa_amd.models as M
    torch.manual_seed(7)
    B = len(v_lens)
    common = dict(batch_size=B, q_embedding_size=12, nb_classes=K, num_input_channels=CIN, num_res_block_channels=C,
                  num_res_blocks=2, hidden_size=16, vocab_size=V, spatial_size=130, precision="fp32")
    if model_name == "film_attn_pt":
        model = M.FiLMAttnPretrainedStem(at_hidden_size=16, max_num_frames=T, **common)
    elif model_name == "film_gp_pt":
        model = M.FiLMGlobalPoolingPretrainedStem(num_tail_channels=4, **common)
    else:
        model = M.TimeMultiHopFiLMPretrainedStem(num_tail_channels=4, **common)
    with torch.no_grad():            # keep the FiLM generator alive so the blocks see non-trivial gamma/beta
        for n, p in model.named_parameters():
            if n.endswith("film_layer.1.bias"):
                p.add_(0.5)
    model = model.cuda()
    W = _weights(model)
    v, q, vl, ql, y = _batch(11, v_lens, q_lens)

    for training in (False, True):
        model.train(training)
        model.init_hidden()
        if training:
            logits = model(v.cuda(), q.cuda(), vl, ql)
            loss = torch.nn.functional.cross_entropy(logits, y.cuda(), reduction="sum")
            loss.backward()
        else:
            with torch.no_grad():
                logits = model(v.cuda(), q.cuda(), vl, ql)
        Wo = {k: t.clone() for k, t in W.items()}
        names = [k for k in Wo if Wo[k].is_floating_point() and O.is_trainable(k)]
        if training:
            for k in names:
                Wo[k].requires_grad_(True)
        ref = O.FORWARDS[model_name](Wo, v, q, vl, ql, training=training, aux={})
        got = logits.detach().cpu()
        assert torch.isfinite(got).all()
        assert rel_err(got.numpy(), ref.detach().numpy()) < 1e-3, (name, training)
        assert (got.argmax(1) == ref.detach().argmax(1)).all()
        if training:
            grads = torch.autograd.grad(O.cross_entropy_sum(ref, y), [Wo[k] for k in names], allow_unused=True)
            gref = dict(zip(names, grads))
            checked = 0
            for k, p in model.named_parameters():
                if k not in gref or gref[k] is None:
                    continue
                a = np.zeros_like(gref[k].numpy()) if p.grad is None else p.grad.cpu().numpy()
                b = gref[k].numpy()
                assert np.abs(a - b).max() <= 2e-3 * np.abs(b).max() + 2e-6, (name, k)
                checked += 1
            assert checked >= 8


@pytest.mark.parametrize("name,v_lens,q_lens", EDGE)
def test_mac_edge_batches_vs_oracle(name, v_lens, q_lens):
    import videonavqa_amd.models as M
    torch.manual_seed(9)
    model = M.MACNetwork(n_vocab=V, dim=16, embed_hidden=12, max_step=3, classes=K, max_num_frames=T,
                         self_attention=True, memory_gate=True, precision="fp32").cuda()
    W = _weights(model)
    v, q, vl, ql, y = _batch(13, v_lens, q_lens, h=4, w=5, cin=512)
    model.eval()
    with torch.no_grad():
        got = model(v.cuda(), q.cuda(), vl, ql).cpu()
        ref = O.mac_forward(W, v, q, vl, ql, 3, T, True, True)
    assert torch.isfinite(got).all()
    assert rel_err(got.numpy(), ref.numpy()) < 1e-3, name
    assert (got.argmax(1) == ref.argmax(1)).all()


def test_carried_question_state_follows_the_sorted_order_between_batches():
    """two consecutive forwards WITHOUT init_hidden() whose q_len sort permutations differ.  Upstream carries
    film_hidden in q_len-sorted order, so sorted slot i of the new batch inherits sorted slot i of the old one
    (film_attn_pt_stem.py:150,160).  The lazily kept sample-order state must give exactly what the materialised
    sorted-order state (the `film_hidden` getter) gives."""
    import videonavqa_amd.models as M
    torch.manual_seed(3)
    B = 3
    model = M.FiLMAttnPretrainedStem(batch_size=B, q_embedding_size=12, nb_classes=K, num_input_channels=CIN,
                                     num_res_block_channels=C, num_res_blocks=1, hidden_size=16, at_hidden_size=16,
                                     max_num_frames=T, vocab_size=V, spatial_size=130, precision="fp32").cuda().eval()
    b1 = _batch(21, [6, 5, 4], [9, 2, 5])          # sorted order of the questions: samples 0, 2, 1
    b2 = _batch(22, [6, 6, 3], [3, 8, 4])          # ... then 1, 2, 0
    outs = []
    for materialise in (False, True):
        model.init_hidden()
        with torch.no_grad():
            model(b1[0].cuda(), b1[1].cuda(), b1[2], b1[3])
            if materialise:
                fh = model.film_hidden              # sorted-order form, drops the lazily kept sample-order state
                assert fh[0].shape == (1, B, 16)
            outs.append(model(b2[0].cuda(), b2[1].cuda(), b2[2], b2[3]).cpu())
    assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())
    # and the state really is carried: a reset in between changes the second batch's logits
    model.init_hidden()
    with torch.no_grad():
        fresh = model(b2[0].cuda(), b2[1].cuda(), b2[2], b2[3]).cpu()
    assert float((fresh - outs[0]).abs().max()) > 1e-6


def test_ce_loss_ignores_out_of_range_targets_like_ignore_index():
    """vnqa_ce_loss must not index out of bounds on a target outside [0, K); such rows are ignored (zero weight,
    zero gradient), which is what nn.CrossEntropyLoss does for ignore_index = -100."""
    from videonavqa_amd import ops
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(6, K, generator=g).cuda().requires_grad_(True)
    ys = torch.tensor([1, -100, 3, 0, K - 1, 2]).cuda()
    w = (torch.rand(K, generator=g) + 0.5).cuda()
    for reduction in ("sum", "mean"):
        for weight in (None, w):
            lg = logits.detach().clone().requires_grad_(True)
            loss = ops.cross_entropy(lg, ys, weight=weight, reduction=reduction)
            loss.backward()
            ref_l = logits.detach().clone().requires_grad_(True)
            ref = torch.nn.functional.cross_entropy(ref_l, ys, weight=weight, reduction=reduction, ignore_index=-100)
            ref.backward()
            assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
            assert float((lg.grad - ref_l.grad).abs().max()) <= 1e-6
            assert float(lg.grad[1].abs().max()) == 0.0


def test_uint8_clip_gives_bit_identical_stem_input_and_features():
    """raw 8-bit pixels + the 256-entry table float32(k / 255.0) (division in float64, eval/dataset.py:91)
    give the stem bit for bit what the fp32 clip gives — the 4-channel image list of the fused first conv, and the features
    of the whole stem (16-bit fused path and exact-f32 path)."""
    import torch.nn as nn
    from videonavqa_amd import kernels as K
    from videonavqa_amd.models import ObjDetectCNN
    from videonavqa_amd.models.common import FrameLayout
    from videonavqa_amd.stem import FrozenStem, VGGFront
    from helpers import LOW
    g = torch.Generator().manual_seed(3)
    B, T, H, W = 2, 5, 32, 48
    u8 = torch.randint(0, 256, (B, 3, H, W, T), generator=g, dtype=torch.uint8)
    u8[0, 0, 0, :, 0] = torch.arange(W, dtype=torch.uint8)
    f32 = (u8.double() / 255.0).float()                      # the reference loader's values (dataset.py:91, q_and_v_eval.py:92)
    lay = FrameLayout([T, 3], T, "cuda")
    a = K.clip_to_nhwc4(u8.cuda(), lay.img_of, lay.n_img)
    b = K.clip_to_nhwc4(f32.cuda(), lay.img_of, lay.n_img)
    assert torch.equal(a, b)
    assert torch.equal(K.expand_u8_clip(u8.cuda()).cpu(), f32)
    for prec in (LOW, "fp32"):
        torch.manual_seed(0)
        vgg, od = VGGFront(prec), ObjDetectCNN(5, 64, 8, 0, True, True, precision=prec)
        with torch.no_grad():
            for conv in vgg.features.values():
                nn.init.kaiming_uniform_(conv.weight, a=1.0)
        stem = FrozenStem(vgg.cuda().eval(), od.cuda().eval(), prec)
        fa = stem.forward_clip(u8.cuda(), lay.img_of, lay.n_img).clone()
        fb = stem.forward_clip(f32.cuda(), lay.img_of, lay.n_img)
        assert torch.equal(fa, fb), prec


def test_frame_layout_tables_written_on_the_device_equal_the_host_built_ones(monkeypatch):
    """vnqa_frame_layout (lengths and sort permutation as kernel arguments, no host-to-device copy) against the host-built tables,
    for full, ragged, single-sample, zero-length and 70-frame batches with and without a sort permutation
    (film_attn_pt_stem.py:201-208: frame t is processed for the samples that have it)."""
    import torch
    from videonavqa_amd.models.common import FrameLayout
    g = torch.Generator().manual_seed(3)
    cases = [([35] * 8, 35), ([35, 30, 30, 12, 7, 3, 3, 1], 35), ([5], 35), ([70, 64, 3], 70), ([4, 0, 0], 6), ([0, 0], 4),
             (sorted(torch.randint(1, 36, (32,), generator=g).tolist(), reverse=True), 35)]
    for vl, T in cases:
        for use_perm in (False, True):
            perm = torch.randperm(len(vl), generator=g) if use_perm else None
            dev = FrameLayout(vl, T, "cuda", perm=perm)
            host = FrameLayout(vl, T, "cuda", perm=perm, on_device=False)
            for name in ("img_of", "frame_of_i32", "sample_of_i32", "frame_off_i32"):
                a, b = getattr(dev, name).cpu(), getattr(host, name).cpu()
                assert a.shape == b.shape and torch.equal(a, b), (vl, T, use_perm, name)
            assert dev.n_img == host.n_img == sum(vl) and dev.n_frames == host.n_frames


def test_fused_conv1_dynamic_tile_schedule_is_bit_identical_and_resets_itself(monkeypatch):
    """vnqa_conv_first_c64_fwd_sched: the persistent workgroups draw their tiles from a device counter (so that a workgroup held up by
    another stream's kernels draws fewer) — the same bits as the static stride, on repeated launches (the schedule words are left
    zero), with a kernel of another stream in the way, and for an image count that gives fewer tiles than workgroups."""
    import torch
    from videonavqa_amd import kernels as K
    from videonavqa_amd import _lib as L
    g = torch.Generator().manual_seed(21)
    half = L.half_dtype()
    w1 = (torch.randn(64, 3, 3, 3, generator=g) / 5).cuda()
    b1 = (torch.randn(64, generator=g) / 10).cuda()
    w2 = (torch.randn(64, 64, 3, 3, generator=g) / 24).cuda()
    wt = K.pack_conv_weight(w2, half, c_out_pad=64, c_in_pad=64)
    bias = torch.randn(64, generator=g).cuda()
    for n_img, H, W in ((3, 64, 96), (37, 96, 128), (1, 32, 32)):
        clip = torch.rand(1, 3, H, W, n_img, generator=g).cuda()
        img_of = torch.arange(n_img, dtype=torch.int32, device="cuda")
        img4 = torch.zeros(n_img, H + 4, W + 4, 4, dtype=half, device="cuda")
        K.clip_to_nhwc4(clip, img_of, n_img, out=img4)
        ref = K.conv_first_c64(img4, w1, b1, wt, bias=bias, relu=True, pool2=True)
        sched = torch.zeros(2, dtype=torch.int32, device="cuda")
        side = torch.cuda.Stream()
        busy = torch.randn(4096, 4096, device="cuda")
        for rep in range(4):
            if rep >= 2:              # something else on the chip while the persistent kernel runs
                with torch.cuda.stream(side):
                    for _ in range(3):
                        busy = torch.tanh(busy)
            got = K.conv_first_c64(img4, w1, b1, wt, bias=bias, relu=True, pool2=True, sched=sched)
            torch.cuda.synchronize()
            assert torch.equal(got, ref), (n_img, H, W, rep)
            assert sched.tolist() == [0, 0], (sched.tolist(), rep)
