"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# The 16-bit precision the GPU tests exercise next to fp32: 'fp16' (libvnqa_hip_f16.so: the storage format of the headline precision
# 'fp16h' and of every constructor default since round 6) by default; VNQA_TEST_LOW_PRECISION=bf16 re-runs the very same tests on the
# bf16-storage build (libvnqa_hip.so, BASELINE.json's storage dtype) — one 16-bit storage format per process, so that run is a
# separate pytest process (tests/test_gpu_bf16.py starts it).
LOW = os.environ.get("VNQA_TEST_LOW_PRECISION", "fp16")
LOW_DTYPE = torch.float16 if LOW == "fp16" else torch.bfloat16
os.environ.setdefault("VNQA_HALF", "f16" if LOW == "fp16" else "bf16")      # read by videonavqa_amd._lib when it is first imported


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def weights_from(g, prefix):
    """Tensors stored as '<prefix>/<name>' -> {name: torch tensor}."""
    out = {}
    for k, v in g.items():
        if k.startswith(prefix + "/"):
            out[k[len(prefix) + 1:]] = torch.from_numpy(np.array(v))
    return out


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


MODEL_OF = {"film_attn": "film_attn_pt", "film_gp": "film_gp_pt", "tmh": "time_multi_hop"}


def model_of_case(case):
    for k, v in MODEL_OF.items():
        if case.startswith(k):
            return v
    raise KeyError(case)


QV_CASES = ["film_attn_full", "film_attn_ragged", "film_attn_short", "film_attn_s196", "film_attn_b5", "film_attn_bow",
            "film_gp_full", "film_gp_ragged", "film_gp_bow", "tmh_full", "tmh_ragged"]


# constructor arguments used by tools/capture_goldens.py for each golden case
_B, _CIN, _C, _T, _L = 3, 8, 8, 6, 9
ATTN_KW = dict(batch_size=_B, q_embedding_size=12, nb_classes=7, num_input_channels=_CIN,
               num_res_block_channels=_C, num_res_blocks=2, hidden_size=16, at_hidden_size=16,
               max_num_frames=_T, q_encoder="lstm", vocab_size=20)
GP_KW = dict(batch_size=_B, q_embedding_size=12, nb_classes=7, num_input_channels=_CIN,
             num_res_block_channels=_C, num_tail_channels=4, num_res_blocks=2, hidden_size=16,
             q_encoder="lstm", vocab_size=20)
TMH_KW = dict(batch_size=_B, q_embedding_size=12, nb_classes=7, num_input_channels=_CIN,
              num_res_block_channels=_C, num_res_blocks=2, num_tail_channels=4, hidden_size=16,
              vocab_size=20)


def build_product_model(case, precision):
    """The product (HIP) model configured like the golden case, on the GPU, golden weights loaded."""
    import videonavqa_amd.models as M
    g = load_golden(case)
    if case.startswith("film_attn"):
        kw = dict(ATTN_KW)
        spatial = 130
        if case == "film_attn_s196":
            kw["num_res_blocks"] = 1
            spatial = 196
        if case == "film_attn_b5":          # eval.sh's depth: 5 FiLM blocks, 14x14 maps
            kw["num_res_blocks"] = 5
            spatial = 196
        if case == "film_attn_bow":         # q_encoder='bow' (film_attn_pt_stem.py:75-77,171-177)
            kw["q_encoder"] = "bow"
        model = M.FiLMAttnPretrainedStem(spatial_size=spatial, precision=precision, **kw)
    elif case.startswith("film_gp"):
        kw = dict(GP_KW, q_encoder="bow") if case == "film_gp_bow" else GP_KW
        model = M.FiLMGlobalPoolingPretrainedStem(spatial_size=130, precision=precision, **kw)
    else:
        model = M.TimeMultiHopFiLMPretrainedStem(spatial_size=130, precision=precision, **TMH_KW)
    model = model.cuda()
    model.load_reference_tensors(weights_from(g, "w0"))
    return model, g


MAC_CASES = ["mac_plain", "mac_sa_gate"]


def mac_case(name):
    """Golden MAC case -> (g, cfg dict, fp32 weights, inputs) as tools/capture_goldens.py:run_mac_case wrote them."""
    g = load_golden(name)
    dim, E, steps, K, V, T, sa, mg = [int(x) for x in g["cfg"]]
    cfg = dict(dim=dim, embed_hidden=E, max_step=steps, classes=K, n_vocab=V, max_num_frames=T,
               self_attention=bool(sa), memory_gate=bool(mg))
    W = {k: v.float() for k, v in weights_from(g, "w0").items()}
    inputs = (torch.from_numpy(g["v"].astype(np.float32)), torch.from_numpy(g["q"]), torch.from_numpy(g["v_lens"]),
              torch.from_numpy(g["q_lens"]), torch.from_numpy(g["y"]))
    n_frames = int(g["v_lens"][0])
    masks = [(torch.from_numpy(g["drop_mask/%d/control" % i]), torch.from_numpy(g["drop_mask/%d/memory" % i]))
             for i in range(n_frames)]
    return g, cfg, W, inputs, masks
