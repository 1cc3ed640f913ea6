#!/usr/bin/env python3
"""Measured 16-bit errors of the small golden-pinned nets (what tests/test_gpu_models.py bounds): worst logits error, per-tensor
gradient relative L2 and worst element (as a fraction of the tensor's max |grad|) over all Q+V cases."""
import os, sys
import numpy as np, torch, torch.nn as nn
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from helpers import QV_CASES, build_product_model, rel_err, LOW
PREC = sys.argv[1] if len(sys.argv) > 1 else LOW          # e.g. fp16h (fp16 build: the default), fp16; bf16 needs VNQA_TEST_LOW_PRECISION=bf16
LOW = PREC
worst = {"eval_logits": 0, "train_logits": 0, "l2": (0, ""), "elem": (0, ""), "whole_grad_l2": (0, "")}
for case in QV_CASES:
    model, g = build_product_model(case, LOW)
    v, q, vl, ql, y = (torch.from_numpy(g[k]).cuda() for k in ("v", "q", "v_lens", "q_lens", "y"))
    model.eval()
    with torch.no_grad():
        model.init_hidden(); lg = model(v, q, vl, ql)
    worst["eval_logits"] = max(worst["eval_logits"], rel_err(lg.float().cpu().numpy(), g["eval_logits"]))
    model, g = build_product_model(case, LOW)
    model.train(); model.init_hidden()
    logits = model(v, q, vl, ql)
    nn.CrossEntropyLoss(reduction="sum")(logits, y).backward()
    worst["train_logits"] = max(worst["train_logits"], rel_err(logits.detach().float().cpu().numpy(), g["train_logits"]))
    num = den = 0.0
    for name, p in model.named_parameters():
        if "grad/" + name in g and p.grad is not None:
            num += float(((p.grad.float().cpu() - torch.from_numpy(g["grad/" + name])) ** 2).sum())
            den += float((torch.from_numpy(g["grad/" + name]) ** 2).sum())
    if den > 0 and (num / den) ** 0.5 > worst["whole_grad_l2"][0]:
        worst["whole_grad_l2"] = (round((num / den) ** 0.5, 4), case)
    for name, p in model.named_parameters():
        ref = g["grad/" + name]
        got = np.zeros_like(ref) if p.grad is None else p.grad.float().cpu().numpy()
        if np.linalg.norm(ref) < 1e-5: continue
        l2 = float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-9))
        el = float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12))
        if l2 > worst["l2"][0]: worst["l2"] = (round(l2, 4), case + ":" + name)
        if el > worst["elem"][0]: worst["elem"] = (round(el, 4), case + ":" + name)
print(LOW, worst)
