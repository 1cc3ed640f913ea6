"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


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


QV_CASES = ["film_attn_full", "film_attn_ragged", "film_attn_short", "film_attn_s196",
            "film_gp_full", "film_gp_ragged", "tmh_full", "tmh_ragged"]
