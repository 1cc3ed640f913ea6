"""importing the reference's pretrained files into the frozen stem.

`vgg16_caffe.pth` (eval.sh:21 -> demo.get_frcnn_feature_extractor(path), eval/q_and_v_eval.py:308) is a full VGG-16
state dict (13 convs `features.N.*` + `classifier.*`); `obj_detect.pt` (eval/utils.py:14,49) is a checkpoint dict whose
'state_dict' entry holds ObjDetectCNN's parameters under the names of models/obj_detector.py:22-41.  Neither file
exists in this image, so synthetic files with exactly those key names / shapes are written and loaded through the
product's loaders; the GPU test then runs the loaded stem against the oracle evaluated on the very same tensors."""
import numpy as np
import pytest
import torch

from helpers import LOW, LOW_DTYPE

# torchvision VGG-16 'D' conv positions inside `features` and their (c_out, c_in)
VGG16_CONVS = {0: (64, 3), 2: (64, 64), 5: (128, 64), 7: (128, 128), 10: (256, 128), 12: (256, 256), 14: (256, 256),
               17: (512, 256), 19: (512, 512), 21: (512, 512), 24: (512, 512), 26: (512, 512), 28: (512, 512)}

# ObjDetectCNN(27, 512, 1024, 0, True, True).state_dict() of the reference, in order (models/obj_detector.py:22-41)
OBJDET_KEYS = (
    [("bn_input." + s, (128,)) for s in ("weight", "bias", "running_mean", "running_var")] + [("bn_input.num_batches_tracked", ())]
    + [("conv11.weight", (512, 128, 3, 3)), ("conv11.bias", (512,)), ("conv12.weight", (512, 512, 3, 3)), ("conv12.bias", (512,))]
    + [("bn1." + s, (512,)) for s in ("weight", "bias", "running_mean", "running_var")] + [("bn1.num_batches_tracked", ())]
    + [("conv21.weight", (512, 512, 3, 3)), ("conv21.bias", (512,)), ("conv22.weight", (512, 512, 3, 3)), ("conv22.bias", (512,))]
    + [("bn2." + s, (512,)) for s in ("weight", "bias", "running_mean", "running_var")] + [("bn2.num_batches_tracked", ())]
    + [("conv31.weight", (512, 512, 3, 3)), ("conv31.bias", (512,)), ("conv32.weight", (512, 512, 3, 3)), ("conv32.bias", (512,))]
    + [("bn3." + s, (512,)) for s in ("weight", "bias", "running_mean", "running_var")] + [("bn3.num_batches_tracked", ())]
    + [("fc_tail1.weight", (1024, 15360)), ("fc_tail1.bias", (1024,))]
    + [("bn_tail1." + s, (1024,)) for s in ("weight", "bias", "running_mean", "running_var")] + [("bn_tail1.num_batches_tracked", ())]
    + [("fc_tail2.weight", (27, 1024)), ("fc_tail2.bias", (27,))])


def write_vgg16_file(path, seed=11):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for idx, (co, ci) in VGG16_CONVS.items():
        sd["features.%d.weight" % idx] = torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (ci * 9)) ** 0.5
        sd["features.%d.bias" % idx] = torch.randn(co, generator=g) * 0.05
    sd["classifier.0.weight"], sd["classifier.0.bias"] = torch.zeros(8, 8), torch.zeros(8)      # ignored by the loader
    torch.save(sd, path)
    return sd


def write_obj_detect_file(path, seed=12):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape in OBJDET_KEYS:
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(7)
        elif k.endswith("running_var"):
            sd[k] = torch.rand(shape, generator=g) * 0.8 + 0.6
        elif k.startswith("bn") and k.endswith(".weight"):
            sd[k] = torch.rand(shape, generator=g) * 0.8 + 0.6
        elif len(shape) == 4:
            sd[k] = torch.randn(shape, generator=g) * (1.0 / (shape[1] * 9)) ** 0.5
        elif k.startswith("fc_tail1"):
            sd[k] = torch.zeros(shape)          # 63 MB of zeros: the detector's own classifier tail, unused on this path
        else:
            sd[k] = torch.randn(shape, generator=g) * 0.1
    torch.save({"epoch": 3, "state_dict": sd, "val_acc": 0.5}, path)         # checkpoint dict as written upstream
    return sd


def test_loaders_read_reference_style_files(tmp_path):
    from videonavqa_amd.eval.utils import get_object_detector
    from videonavqa_amd.stem import get_frcnn_feature_extractor
    vsd = write_vgg16_file(tmp_path / "vgg16_caffe.pth")
    osd = write_obj_detect_file(tmp_path / "obj_detect.pt")
    vgg = get_frcnn_feature_extractor(str(tmp_path / "vgg16_caffe.pth"), "fp32")
    assert not vgg.training and all(not p.requires_grad for p in vgg.parameters())
    for idx in (0, 2, 5, 7):                       # only features[0:10] belong to the front
        assert torch.equal(vgg.features[str(idx)].weight, vsd["features.%d.weight" % idx])
        assert torch.equal(vgg.features[str(idx)].bias, vsd["features.%d.bias" % idx])
    assert sorted(vgg.state_dict()) == sorted("features.%d.%s" % (i, s) for i in (0, 2, 5, 7) for s in ("weight", "bias"))
    od = get_object_detector(str(tmp_path / "obj_detect.pt"), "fp32")
    assert not od.training
    got = od.state_dict()
    assert list(got) == [k for k, _ in OBJDET_KEYS]
    for k, _ in OBJDET_KEYS:
        assert torch.equal(got[k], osd[k]), k
    # a checkpoint with a missing / misnamed tensor must fail loudly, as nn.Module.load_state_dict does upstream
    bad = dict(osd)
    bad["conv13.weight"] = bad.pop("conv12.weight")
    torch.save({"state_dict": bad}, tmp_path / "bad.pt")
    with pytest.raises(RuntimeError):
        get_object_detector(str(tmp_path / "bad.pt"), "fp32")


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", LOW])
def test_stem_from_imported_files_vs_oracle(tmp_path, precision):
    """Files -> loaders -> FrozenStem (composed conv11.conv12, folded BN, 512 filters) on the reference's own 160x208
    clip geometry (10x13 maps) against the oracle's per-frame loop on the same tensors."""
    from oracle import vnqa_oracle as O
    from videonavqa_amd import kernels as K
    from videonavqa_amd.eval.utils import get_object_detector
    from videonavqa_amd.models.common import FrameLayout
    from videonavqa_amd.stem import FrozenStem, get_frcnn_feature_extractor
    vsd = write_vgg16_file(tmp_path / "vgg16_caffe.pth")
    osd = write_obj_detect_file(tmp_path / "obj_detect.pt")
    vgg = get_frcnn_feature_extractor(str(tmp_path / "vgg16_caffe.pth"), precision).cuda()
    od = get_object_detector(str(tmp_path / "obj_detect.pt"), precision).cuda()
    stem = FrozenStem(vgg, od, precision)
    B, T, H, W = 2, 2, 160, 208
    clip = torch.rand(B, 3, H, W, T, generator=torch.Generator().manual_seed(3))
    lay = FrameLayout([2, 1], T, "cuda")
    feats = stem.forward_clip(clip.cuda(), lay.img_of, lay.n_img)
    assert feats.shape == (3, 12, 15, 512)
    got = stem.plain_features(feats)[:, 1:-1, 1:-1, :512].permute(0, 3, 1, 2).cpu()
    W_vgg = {k: v for k, v in vsd.items() if k.startswith("features.")}
    ref = O.stem_forward(clip, W_vgg, {k: v.float() for k, v in osd.items()})          # [B,512,10,13,T]
    tol = 1e-4 if precision == "fp32" else 4e-2
    for n in range(lay.n_img):
        t, b = int(lay.frame_of[n]), int(lay.sample_of[n])
        r = ref[b, :, :, :, t]
        assert float((got[n] - r).abs().max() / (r.abs().max() + 1e-9)) < tol, (n, t, b)
