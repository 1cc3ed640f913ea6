"""The q_and_v_eval entry point: flag surface (CPU) and an end-to-end synthetic run with checkpoint
save / resume on the GPU."""
import os

import pytest
import torch


def test_parser_matches_reference_flags_and_defaults():
    """Flag names and defaults of eval/q_and_v_eval.py:32-64."""
    from videonavqa_amd.eval.q_and_v_eval import build_parser
    a = build_parser().parse_args(["--model", "film_attn_pt"])
    expect = dict(num_classes=70, q_encoder="lstm", use_obj_detector=True, use_visual_features=True,
                  vocab_size=134, embed_size=128, hidden_size=128, at_hidden_size=128, num_res_blocks=1,
                  num_res_block_channels=512, num_input_channels=512, num_tail_channels=16, mac_dim=512,
                  mac_max_step=12, batch_size=8, clip_value=1.0, l_rate=1e-4, num_epochs=1,
                  use_class_weights=False, checkpoint_path=None, frcnn_pretrained_path=None, num_workers=4,
                  stats_after_every=400, val_only=False)
    for k, v in expect.items():
        assert getattr(a, k) == v, k
    # eval.sh:44-59 command line parses (incl. its stray --best_acc)
    b = build_parser().parse_args("--model time_multi_hop --num_classes 70 --vocab_size 134 --num_res_blocks 3 "
                                  "--num_res_block_channels 1024 --num_tail_channels 64 --at_hidden_size 128 "
                                  "--hidden_size 128 --batch_size 16 --loss_reduction sum --l_rate 0.00005 "
                                  "--num_epochs 1 --best_acc 0 --frcnn_pretrained_path ../vgg16_caffe.pth "
                                  "--checkpoint_path tmh.pt --stats_after_every 500".split())
    assert b.num_res_block_channels == 1024 and b.batch_size == 16 and b.loss_reduction == "sum"


def test_constants_match_reference():
    from videonavqa_amd.eval import utils as U
    assert (U.DROP_EVERY_N_FRAMES, U.MAX_ALLOWED_NUM_FRAMES_DROPPING, U.MAX_NUM_VIDEO_FRAMES, U.MAX_Q_LEN,
            U.NUM_CLASSES, U.VID_HEIGHT, U.VID_WIDTH) == (4, 35, 400, 56, 70, 160, 208)
    import numpy as np
    acc = U.per_class_accuracies(np.array([0, 0, 1, 2]), np.array([0, 1, 1, 0]), 4)
    assert acc.tolist() == [0.5, 1.0, 0.0, 0.0]


def test_synthetic_dataset_contract():
    from videonavqa_amd.eval.dataset import SyntheticVNQADataset
    ds = SyntheticVNQADataset(5, 32, 48, seed=3)
    X, y = ds[2]
    assert X["video"].shape == (3, 32, 48, 35) and X["question"].shape == (56,) and X["question"].dtype == torch.int64
    assert 3 <= X["v_len"] <= 35 and 5 <= X["q_len"] <= 25 and 0 <= y < 70
    assert float(X["video"][..., X["v_len"]:].abs().max() if X["v_len"] < 35 else 0.0) == 0.0
    assert int((X["question"][X["q_len"]:] != 0).sum()) == 0 and int((X["question"][:X["q_len"]] == 0).sum()) == 0
    X2, y2 = ds[2]
    assert torch.equal(X["video"], X2["video"]) and y == y2


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["film_attn_pt", "film_gp_pt", "time_multi_hop", "mac"])
def test_cli_synthetic_train_val_checkpoint_resume(model, tmp_path, capsys):
    from videonavqa_amd.eval import q_and_v_eval as E
    os.chdir(tmp_path)
    argv = ["--model", model, "--synthetic", "6", "--batch_size", "2", "--num_workers", "0", "--height", "64",
            "--width", "96", "--num_res_block_channels", "64", "--hidden_size", "16", "--at_hidden_size", "16",
            "--embed_size", "16", "--precision", "fp32", "--checkpoint_path", "ck.pt", "--stats_after_every", "1",
            "--mac_dim", "64", "--mac_max_step", "3"]
    E.main(argv)
    out = capsys.readouterr().out
    if model == "mac":      # eval/q_and_v_eval.py:357-363: after epoch 0 the rate drops to l_rate/10
        assert "learning rate 0.00001" in out
    assert "Train Epoch: 0" in out and "Validation:" in out and "Average loss after 1 iterations in epoch 1" in out
    ck = torch.load(tmp_path / "e0_ck.pt", map_location="cpu")
    # reference schema (eval/q_and_v_eval.py:148-156) + 'extra_state' (the frozen conv1x1_layers state_dict() leaves out)
    assert set(ck) == {"epoch", "model", "state_dict", "train_f1w", "train_f1micro", "optimizer", "extra_state"}
    if model != "mac":
        assert set(ck["extra_state"]) == {"conv1x1_layers.0.weight", "conv1x1_layers.0.bias"}
    assert ck["model"] == model and ck["epoch"] == 0
    # the optimizer entry is a genuine torch.optim.Adam state_dict
    params = [torch.nn.Parameter(torch.zeros_like(s["exp_avg"])) for s in ck["optimizer"]["state"].values()]
    torch.optim.Adam(params, lr=1e-4).load_state_dict(ck["optimizer"])
    # resume: reference convention = start from --checkpoint_path itself (q_and_v_eval.py:337-346)
    os.replace(tmp_path / "e0_ck.pt", tmp_path / "ck.pt")
    E.main(argv)
    out = capsys.readouterr().out
    assert "Restored checkpoint ck.pt (epoch 1)" in out and "Train Epoch: 1" in out
    assert (tmp_path / "e1_ck.pt").exists()
    if model == "film_attn_pt":
        # test-split script on the same checkpoint: 5 items with batch 2 -> the last batch is padded
        from videonavqa_amd.eval import q_and_v_test as T
        targv = [a for a in argv]
        targv[targv.index("--synthetic") + 1] = "5"
        T.main(targv)
        out = capsys.readouterr().out
        assert "Testing:" in out and "Accuracy:" in out
        import numpy as np
        t, p = np.load(tmp_path / "t_ck.pt.npy"), np.load(tmp_path / "p_ck.pt.npy")
        assert t.shape == (5,) and p.shape == (5,)


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["film_attn_pt", "film_gp_pt"])
def test_cli_bag_of_words_question_encoder(model, tmp_path, capsys):
    """--q_encoder bow (eval/q_and_v_eval.py:35; film_attn_pt_stem.py:75-77): one epoch, checkpoint, resume.  The encoder
    is an nn.Linear, so the checkpoint carries film_layer.0.weight / .bias instead of the LSTM's four tensors."""
    from videonavqa_amd.eval import q_and_v_eval as E
    os.chdir(tmp_path)
    argv = ["--model", model, "--q_encoder", "bow", "--synthetic", "4", "--batch_size", "2", "--num_workers", "0",
            "--height", "64", "--width", "96", "--num_res_block_channels", "64", "--hidden_size", "16", "--at_hidden_size", "16",
            "--embed_size", "16", "--checkpoint_path", "bow.pt", "--stats_after_every", "1"]
    E.main(argv)
    out = capsys.readouterr().out
    assert "Train Epoch: 0" in out and "Validation:" in out
    ck = torch.load(tmp_path / "e0_bow.pt", map_location="cpu")
    keys = set(ck["state_dict"])
    assert {"film_layer.0.weight", "film_layer.0.bias", "film_layer.1.weight"} <= keys
    assert not any(k.startswith("film_layer.0.weight_hh") for k in keys)
    os.replace(tmp_path / "e0_bow.pt", tmp_path / "bow.pt")
    E.main(argv)
    assert "Restored checkpoint bow.pt (epoch 1)" in capsys.readouterr().out


@pytest.mark.gpu
def test_checkpoint_restores_frozen_conv1x1_layers(tmp_path, capsys):
    """A checkpoint written from a model whose frozen conv1x1_layers do NOT come from seed 0 must evaluate identically
    after q_and_v_test restores it: the 1x1 convs travel in the 'extra_state' key."""
    import numpy as np
    from videonavqa_amd.eval import q_and_v_eval as E, q_and_v_test as T
    os.chdir(tmp_path)
    argv = ["--model", "film_attn_pt", "--synthetic", "4", "--batch_size", "2", "--num_workers", "0", "--height", "64",
            "--width", "96", "--num_res_block_channels", "64", "--hidden_size", "16", "--at_hidden_size", "16",
            "--embed_size", "16", "--precision", "fp32", "--checkpoint_path", "ck.pt"]
    E.main(argv)
    capsys.readouterr()
    ck = torch.load(tmp_path / "e0_ck.pt", map_location="cpu")
    T.main(argv[:-1] + ["e0_ck.pt"])
    base = capsys.readouterr().out
    p0 = np.load(tmp_path / "p_e0_ck.pt.npy")
    # same checkpoint with DIFFERENT 1x1 convs: predictions / loss must follow the stored tensors, not the seed
    ck2 = dict(ck)
    ck2["extra_state"] = {k: torch.randn_like(v) * 3 for k, v in ck["extra_state"].items()}
    torch.save(ck2, tmp_path / "other.pt")
    T.main(argv[:-1] + ["other.pt"])
    other = capsys.readouterr().out
    loss = lambda out: float(out.split("Average loss: ")[1].split(",")[0])
    assert loss(base) != loss(other)
    # and without the key (an upstream-written checkpoint) the seed-0 construction reproduces the training-time convs
    ck3 = {k: v for k, v in ck.items() if k != "extra_state"}
    torch.save(ck3, tmp_path / "plain.pt")
    T.main(argv[:-1] + ["plain.pt"])
    plain = capsys.readouterr().out
    assert loss(plain) == loss(base) and (np.load(tmp_path / "p_plain.pt.npy") == p0).all()


def test_single_modality_parsers_match_reference_defaults():
    """Flag names / defaults of eval/q_only_eval.py:20-44 and eval/v_only_cnn3d_eval.py:21-38."""
    from videonavqa_amd.eval import q_only_eval as Q, v_only_cnn3d_eval as V3
    a = Q.build_parser().parse_args([])
    assert (a.embed_size, a.hidden_size, a.num_classes, a.vocab_size, a.batch_size, a.l_rate, a.num_epochs,
            a.stats_after_every, a.use_class_weights, a.num_workers, a.model) == (128, 128, 70, 134, 1024, 1e-5, 1000, 50,
                                                                                  True, 4, 'lstm')
    b = V3.build_parser().parse_args([])
    assert (b.num_classes, b.use_class_weights, b.batch_size, b.l_rate, b.num_epochs, b.num_workers, b.stats_after_every,
            b.val_only, b.loss_reduction) == (70, False, 8, 1e-4, 1, 4, 1000, False, None)


@pytest.mark.gpu
def test_q_only_cli_synthetic(tmp_path, capsys):
    """Ladder config 1: 1k synthetic encoded questions through the q_only entry point."""
    from videonavqa_amd.eval import q_only_eval as Q
    os.chdir(tmp_path)
    Q.main(["--synthetic", "1000", "--batch_size", "250", "--num_epochs", "2", "--stats_after_every", "1",
            "--num_workers", "0", "--l_rate", "1e-3", "--checkpoint_path", "q.pt"])
    out = capsys.readouterr().out
    assert "1000 train examples" in out and out.count("Train Epoch:") == 2 and out.count("Validation:") == 2
    ck = torch.load(tmp_path / "q.pt", map_location="cpu")
    assert set(ck) == {"epoch", "model", "state_dict", "val_acc", "optimizer"} and ck["model"] == "lstm"


@pytest.mark.gpu
def test_v_only_cnn3d_cli_synthetic(tmp_path, capsys):
    """Ladder config 2 geometry (16x112x112 clips), tiny run: train, checkpoint, resume."""
    from videonavqa_amd.eval import v_only_cnn3d_eval as V3
    os.chdir(tmp_path)
    argv = ["--synthetic", "8", "--batch_size", "4", "--num_workers", "0", "--checkpoint_path", "c.pt"]
    V3.main(argv)
    out = capsys.readouterr().out
    assert "Train Epoch: 0" in out and "Validation:" in out and (tmp_path / "e0_c.pt").exists()
    os.replace(tmp_path / "e0_c.pt", tmp_path / "c.pt")
    V3.main(argv)
    assert "Restored checkpoint c.pt (epoch 1)" in capsys.readouterr().out


def test_vnqa_dataset_contract_with_stubbed_decoder(tmp_path, monkeypatch):
    """VNQADataset item layout (eval/dataset.py:57-106) without OpenCV: the mp4 reader is replaced by a stub that returns
    numbered BGR frames, so the 1-in-4 subsampling, the 35-frame cap, the [3,H,W,35] frames-last layout, the /255
    scaling and the zero padding can be checked exactly."""
    import json
    import numpy as np
    from videonavqa_amd.eval import dataset as D, utils as U
    qd, vd = tmp_path / "q", tmp_path / "v"
    qd.mkdir()
    vd.mkdir()
    np.save(qd / "clip0.npy", np.array([5, 9, 2, 7], dtype=np.int64))
    np.save(qd / "clip1.npy", np.arange(1, 31, dtype=np.int64))

    def fake_frames(path):
        n = 23 if path.endswith("clip0.mp4") else 200          # 200 raw frames -> 50 windows -> capped at 35
        return [np.full((U.VID_HEIGHT, U.VID_WIDTH, 3), i % 256, np.uint8) for i in range(n)]
    monkeypatch.setattr(D, "_read_frames", fake_frames)
    ds = D.VNQADataset(q_dir=str(qd), v_dir=str(vd), filenames=["clip0", "clip1"], labels={"clip0": 3, "clip1": 60})
    X, y = ds[0]
    assert y == 3 and X["q_len"] == 4 and X["question"].tolist()[:5] == [5, 9, 2, 7, 0] and X["question"].shape == (U.MAX_Q_LEN,)
    assert X["video"].shape == (3, U.VID_HEIGHT, U.VID_WIDTH, U.MAX_ALLOWED_NUM_FRAMES_DROPPING)
    assert X["v_len"] == 6                                      # ceil(23 / 4) windows
    picked = (X["video"][0, 0, 0, :6] * 255).round().long().tolist()
    assert all(4 * k <= f <= min(4 * k + 3, 22) for k, f in enumerate(picked)), picked   # one frame out of every window of 4
    assert float(X["video"][..., 6:].abs().max()) == 0.0 and float(X["video"].max()) <= 1.0
    X1, y1 = ds[1]
    assert X1["v_len"] == U.MAX_ALLOWED_NUM_FRAMES_DROPPING and y1 == 60 and X1["q_len"] == 30
    w = ds.get_class_weights()
    assert w.shape == (U.NUM_CLASSES,) and w[3] == 1.0 and w[60] == 1.0 and np.isinf(w[0])
    # question-only / video-only modes (q_only_eval.py, v_only_cnn3d_eval.py)
    qs = D.VNQADataset(q_dir=str(qd), v_dir=str(vd), filenames=["clip0"], labels={"clip0": 3}, q_only=True)[0][0]
    vs = D.VNQADataset(q_dir=str(qd), v_dir=str(vd), filenames=["clip0"], labels={"clip0": 3}, v_only=True)[0][0]
    assert set(qs) == {"question", "q_len"} and set(vs) == {"video", "v_len"}


def test_uint8_video_option_is_bit_identical_to_the_float64_path(tmp_path, monkeypatch):
    """VNQADataset(uint8_video=True) hands out the RAW 8-bit clip; the pixel table the stem applies to it
    (kernels.pixel_lut: float32(k / 255.0), division in float64) reproduces the default path's `clip / 255.0` on float64
    followed by the training loop's `.float()` (eval/dataset.py:91, eval/q_and_v_eval.py:92) BIT FOR BIT, for all 256 pixel
    values — so uploading a quarter of the bytes changes nothing downstream."""
    import random
    import numpy as np
    from videonavqa_amd import kernels as K
    from videonavqa_amd.eval import dataset as D, utils as U
    qd, vd = tmp_path / "q", tmp_path / "v"
    qd.mkdir()
    vd.mkdir()
    np.save(qd / "c.npy", np.array([5, 9, 2], dtype=np.int64))
    rng = np.random.RandomState(0)
    frames = [rng.randint(0, 256, (U.VID_HEIGHT, U.VID_WIDTH, 3)).astype(np.uint8) for _ in range(37)]
    frames[0][:2, :128, 0] = np.arange(256).reshape(2, 128)            # every pixel value occurs
    monkeypatch.setattr(D, "_read_frames", lambda path: frames)
    kw = dict(q_dir=str(qd), v_dir=str(vd), filenames=["c"], labels={"c": 1})
    random.seed(11)
    Xf, _ = D.VNQADataset(**kw)[0]
    random.seed(11)                                    # the same 1-in-4 frame draw
    Xu, _ = D.VNQADataset(uint8_video=True, **kw)[0]
    assert Xu["video"].dtype == torch.uint8 and Xu["video"].shape == Xf["video"].shape and Xu["v_len"] == Xf["v_len"]
    lut = K.pixel_lut("cpu")
    assert lut.dtype == torch.float32 and lut.shape == (256,)
    assert torch.equal(lut[Xu["video"].long()], Xf["video"].float())       # bit-exact, padding frames included
    assert torch.equal(lut, (torch.arange(256, dtype=torch.float64) / 255.0).float())


def test_stem_calibration_frames_from_the_dataset():
    """--stem_calibration: 'noise' / 'off' / 'data' -> FrozenStem's calibration argument; 'data' takes the first valid frames of the
    first items (the same on every rank), uint8 videos as k / 255."""
    import argparse
    from videonavqa_amd.eval.dataset import SyntheticVNQADataset
    from videonavqa_amd.eval.q_and_v_eval import build_parser, stem_calibration
    assert build_parser().parse_args(["--model", "film_attn_pt"]).stem_calibration == "noise"
    ds = SyntheticVNQADataset(5, 32, 48, seed=7)
    ns = lambda m: argparse.Namespace(stem_calibration=m)
    assert stem_calibration(ns("noise"), ds) == "noise" and stem_calibration(ns("off"), ds) is None
    assert stem_calibration(ns("data"), None) == "noise"
    fr = stem_calibration(ns("data"), ds, n_frames=8)
    assert fr.shape == (8, 3, 32, 48) and fr.dtype == torch.float32 and float(fr.min()) >= 0 and float(fr.max()) < 1
    v0 = ds[0][0]["video"]
    assert torch.equal(fr[0], v0[:, :, :, 0]) and torch.equal(fr[1], v0[:, :, :, 1])      # two frames per item from five items
    assert torch.equal(fr[2], ds[1][0]["video"][:, :, :, 0])

    class U8(object):
        def __len__(self):
            return 1

        def __getitem__(self, i):
            return {"video": torch.full((3, 4, 4, 6), 51, dtype=torch.uint8), "v_len": 6}, 0
    fr = stem_calibration(ns("data"), U8(), n_frames=4)
    assert fr.shape == (4, 3, 4, 4) and float((fr - 0.2).abs().max()) < 1e-7


@pytest.mark.gpu
def test_cli_runs_with_the_stem_calibrated_on_data_and_with_calibration_off(tmp_path, capsys):
    from videonavqa_amd.eval import q_and_v_eval as E
    os.chdir(tmp_path)
    for mode in ("data", "off"):
        argv = ["--model", "film_attn_pt", "--synthetic", "4", "--batch_size", "2", "--num_workers", "0", "--height", "64",
                "--width", "96", "--num_res_block_channels", "64", "--hidden_size", "16", "--at_hidden_size", "16",
                "--embed_size", "16", "--stats_after_every", "1", "--stem_calibration", mode]
        E.main(argv)
        out = capsys.readouterr().out
        assert "Train Epoch: 0" in out and "Validation:" in out
