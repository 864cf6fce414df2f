#!/usr/bin/env python
"""Measured max-abs errors of the pipeline against the CPU oracle on one K21 frame for the BEV conv variants
(Winograd F(4x4) / fused F(2x2) / direct) -- where the fp32 noise of the final boxes comes from."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sassd  # noqa: E402
from sassd import synth  # noqa: E402
from sassd.pipeline import InferencePlan  # noqa: E402
import helpers as H  # noqa: E402
import test_gpu_pipeline as T  # noqa: E402

dev = torch.device("cuda:0")
model, c = T._model()
sd = {k: v.clone() for k, v in model.state_dict().items()}
an, bv = T._anchors()
clouds = [H.frame("k21", 0)]
ref, rpn_thr, score_thr = H.oracle_forward_safe(sd, clouds, an, bv, T.CFG)
for name, wino in (("F(4x4)", True), ("F(2x2)", 2), ("direct", 0)):
    plan = InferencePlan(sd, batch_size=1, anchors=an, anchors_bv=bv, device=dev, rpn_thr=rpn_thr, score_thr=score_thr,
                         winograd=wino)
    plan.run_from_points([torch.from_numpy(p).to(dev) for p in clouds])
    torch.cuda.synchronize()
    k = int(plan.df["counts"][0].item())
    gb = ref["guided"][0][0].numpy()
    out = {"bev_x": (plan.x.cpu() - ref["x"]).abs().max().item(),
           "bev_conv6": (plan.conv6.cpu() - ref["conv6"]).abs().max().item(),
           "k": (k, len(gb))}
    if k == len(gb):
        d = np.abs(plan.df["guided"][0, :k].cpu().numpy().astype(np.float64) - gb)
        out["guided_by_field"] = ["%.1e" % v for v in d.max(0)]
        out["logits"] = float(np.abs(plan.logits[0, :k].cpu().numpy() - ref["logits"][0].numpy()).max())
    print(name, out)
