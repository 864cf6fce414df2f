#!/usr/bin/env python
"""How much of the fp32 gradient of the bench's training workload (car_cfg, two K21 frames: tests/golden/train_k21_ref.npz)
is decided by the SUMMATION ORDER of the sparse convolutions?  CPU only: the oracle's step (oracle/train_ref.py) is run
with its sparse convolution summing the 27 offsets (a) in ascending order -- the stored golden -- and (b) in another order
(--order rev: descending; --order split3: three interleaved partial sums added at the end, the order of the round-1
register-stationary kernel), same inputs, same weights, same thresholds.  Both are legitimate fp32 evaluations of the same
function; the relative L2 distance of their gradients, per stored tensor, is the floor under any kernel-vs-oracle bar.

    python tests/analysis/train_order_sensitivity.py [--order rev|split3] [--only-cin 4]     (~40 CPU-seconds per step)"""
import argparse
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--order", default="rev", choices=["rev", "split3"])
ap.add_argument("--only-cin", type=int, default=0, help="re-order only the sparse layers with this many input channels "
                "(4 = the input layer alone); 0 = all of them")
args = ap.parse_args()
gdir = os.path.join(ROOT, "tests", "golden")
spec = importlib.util.spec_from_file_location("make_golden_train_k21", os.path.join(gdir, "make_golden_train_k21.py"))
MG = importlib.util.module_from_spec(spec)
spec.loader.exec_module(MG)
from oracle import clib, nets as onets, train_ref  # noqa: E402

G = np.load(os.path.join(gdir, "train_k21_ref.npz"))
model, c, w, clouds, gts = MG.build()
cal = w["cal"]
sd = {k: v.clone() for k, v in model.state_dict().items()}
feats, coors, masks = [], [], []
for b, p in enumerate(clouds):
    v, co, n = clib.points_to_voxel(p, cal["voxel_size"], cal["pc_range"], cal["max_points"], True, cal["max_voxels"])
    feats.append(clib.voxel_mean(v, n))
    coors.append(np.concatenate([np.full((len(co), 1), b, np.int32), co], 1))
    masks.append(onets.anchors_mask(co, w["anchors_bv"], cal["voxel_size"], cal["pc_range"], cal["grid_xyz"], 1))
feats, coors, m = np.concatenate(feats, 0), np.concatenate(coors, 0), np.stack(masks, 0)
types = [np.array(["Car"] * MG.NGT) for _ in range(MG.B)]
a = c.train_cfg.rpn.assigner["Car"]
an = np.broadcast_to(w["anchors"][None], (MG.B,) + w["anchors"].shape).copy()
shape = tuple(model.neck.sparse_shape) if hasattr(model.neck, "sparse_shape") else (41, 1600, 1408)
step_args = (sd, feats, coors, MG.B, shape, gts, types, ["Car"], {"Car": an}, {"Car": m}, {"Car": (a.pos_iou_thr, a.neg_iou_thr)})
thr = float(G["anchor_thr"])

plain = train_ref.gather_conv


def reordered(x, nbr, w):
    """train_ref.gather_conv with another order of the 27 offset terms"""
    if args.only_cin and w.shape[1] != args.only_cin:
        return plain(x, nbr, w)
    n, K = x.shape[0], nbr.shape[1]
    xp = torch.cat([x, x.new_zeros(1, x.shape[1])], 0)
    idx = torch.as_tensor(np.where(nbr < 0, n, nbr), dtype=torch.int64)

    def part(ks):
        y = x.new_zeros(nbr.shape[0], w.shape[2])
        for k in ks:
            if (nbr[:, k] >= 0).any():
                y = y + xp[idx[:, k]] @ w[k]
        return y
    if args.order == "rev":
        return part(range(K - 1, -1, -1))
    return (part(range(0, K, 3)) + part(range(1, K, 3))) + part(range(2, K, 3))


def grads_of(fn):
    train_ref.gather_conv = fn
    try:
        losses, grads, _ = train_ref.train_step(*step_args, anchor_thr=thr)
    finally:
        train_ref.gather_conv = plain
    return losses, {k: g for k, g in grads.items() if g is not None}


l1, g1 = grads_of(reordered)
stored = {k.split(":", 1)[1]: G[k] for k in G.files if k.startswith("grad:")}
rows = []
for k, ref in stored.items():
    ref = torch.from_numpy(ref).double()
    if float(ref.norm()) > 1e-7 and k in g1:
        rows.append((float((g1[k].double() - ref).norm() / ref.norm()), k))
rows.sort(reverse=True)
num = sum(float((g1[k].double() - torch.from_numpy(v).double()).pow(2).sum()) for k, v in stored.items() if k in g1)
den = sum(float(torch.from_numpy(v).double().pow(2).sum()) for k, v in stored.items() if k in g1)
print("oracle, offsets summed in order '%s'%s vs the stored golden (ascending): losses" % (
    args.order, " (layers with %d input channels only)" % args.only_cin if args.only_cin else ""),
    {k: round(float(v), 5) for k, v in l1.items()})
print("stored-layer gradients: worst rel L2 %.2e over %d tensors, taken together %.2e" % (rows[0][0], len(rows), (num / den) ** 0.5))
print("largest:", [(round(v, 4), k) for v, k in rows[:10]])
