#!/usr/bin/env python
"""Golden vector of ONE training step on the workload bench.py's `train` record measures (BASELINE configs[2]): car_cfg on
its own full grid, batch 2 of synthetic lidar64 K21 frames (21 500 points each), 8 car boxes per frame on occupied voxels
(bench.synth_gt_on_points), random-init weights of synth.build_detector_for(seed 0, cls_bias -3) -- computed by the CPU
oracle oracle/train_ref.py::train_step (torch-CPU autograd over oracle rulebooks; a few CPU-minutes, hence a fixture).
Stored like tests/golden/make_golden_waymo_train.py: the six loss terms, the threshold-safe guided-anchor threshold, label /
candidate counts, the full gradient of a subset of layers, and for EVERY parameter its gradient norm and a seeded projection.

    python tests/golden/make_golden_train_k21.py        # writes tests/golden/train_k21_ref.npz

tests/test_gpu_train.py::test_training_step_k21_vs_oracle rebuilds the same model / frames / boxes, runs forward_train +
backward through the HIP kernels on the batch the product's own device_batch builds (fp32: strict bar; bf16 BEV convs:
the stated whole-model bar), and compares."""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sassd  # noqa: E402,F401
import bench  # noqa: E402
from sassd import synth  # noqa: E402

_spec = importlib.util.spec_from_file_location("make_golden_waymo_train", os.path.join(HERE, "make_golden_waymo_train.py"))
MGW = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(MGW)
projection, FULL, SLICED = MGW.projection, MGW.FULL, MGW.SLICED
B, NGT = 2, 8


def build():
    """(model on CPU in train mode, config, workload dict, clouds [B x [21500,4]], gts [B x [8,7]]) -- exactly what
    bench.train_measure feeds rank 0 in its first step."""
    w = synth.workload("car")
    model, cfg = synth.build_detector_for(w, 0, train=True, cls_bias=-3.0)
    clouds = [w["frame"](i) for i in range(B)]
    gts = [bench.synth_gt_on_points(p, i, NGT, "car") for i, p in enumerate(clouds)]
    return model, cfg, w, clouds, gts


def main():
    import helpers as H
    from oracle import clib, nets as onets, train_ref
    model, c, w, clouds, gts = build()
    cal = w["cal"]
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    feats, coors, masks = [], [], []
    for b, p in enumerate(clouds):
        v, co, n = clib.points_to_voxel(p, cal["voxel_size"], cal["pc_range"], cal["max_points"], True, cal["max_voxels"])
        feats.append(clib.voxel_mean(v, n))
        coors.append(np.concatenate([np.full((len(co), 1), b, np.int32), co], 1))
        masks.append(onets.anchors_mask(co, w["anchors_bv"], cal["voxel_size"], cal["pc_range"], cal["grid_xyz"], 1))
    feats, coors, m = np.concatenate(feats, 0), np.concatenate(coors, 0), np.stack(masks, 0)
    types = [np.array(["Car"] * NGT) for _ in range(B)]
    a = c.train_cfg.rpn.assigner["Car"]
    an = np.broadcast_to(w["anchors"][None], (B,) + w["anchors"].shape).copy()
    shape = tuple(model.neck.sparse_shape) if hasattr(model.neck, "sparse_shape") else (41, 1600, 1408)
    args = (sd, feats, coors, B, shape, gts, types, ["Car"], {"Car": an}, {"Car": m}, {"Car": (a.pos_iou_thr, a.neg_iou_thr)})
    losses, grads, ex = train_ref.train_step(*args)
    top = torch.sigmoid(ex["cls"]).reshape(B, -1)[torch.from_numpy(m)].numpy()
    thr = H.safe_threshold(0.1, top, margin=1e-4, step=2.5e-4)
    if abs(thr - 0.1) > 1e-9:
        losses, grads, ex = train_ref.train_step(*args, anchor_thr=thr)
    out = dict(anchor_thr=np.float64(thr), n_voxels=np.int64(len(coors)), n_masked=np.int64(m.sum()),
               n_pos=np.int64((ex["labels"] > 0).sum()), n_ext_pos=np.int64((ex["ext_labels"] > 0).sum()),
               n_guided=np.int64(sum(len(g) for g in ex["guided"])), loss_names=np.array(sorted(losses)),
               losses=np.array([losses[k] for k in sorted(losses)], np.float64))
    names, norms, projs = [], [], []
    for k, g in grads.items():
        if g is None:
            continue
        gd = g.double().reshape(-1)
        names.append(k)
        norms.append(float(gd.norm()))
        projs.append(float(torch.dot(gd, projection(k, gd.numel()))))
        if k in FULL or ".bn" in k or k.split(".")[-2].isdigit() and g.dim() == 1:
            out["grad:" + k] = g.numpy().astype(np.float32)
        elif k in SLICED:
            out["grad8:" + k] = g[:8].numpy().astype(np.float32)
    out.update(grad_names=np.array(names), grad_norms=np.array(norms), grad_projs=np.array(projs))
    # the same step with the rescoring head's loss_cls left out of the differentiated sum: the part of the objective that
    # does not depend on WHICH anchors pass the guided-anchor threshold (the bf16 comparison uses it: bf16 moves the
    # classification scores by ~1e-2, far more than any threshold margin, and ~2000 candidates sit near the threshold)
    _, grads_x, _ = train_ref.train_step(*args, anchor_thr=thr, grad_exclude=("loss_cls",))
    xn, xnorm = [], []
    for k, g in grads_x.items():
        if g is None:
            continue
        xn.append(k)
        xnorm.append(float(g.double().norm()))
        if k in FULL or ".bn" in k or k.split(".")[-2].isdigit() and g.dim() == 1:
            out["gradx:" + k] = g.numpy().astype(np.float32)
        elif k in SLICED:
            out["gradx8:" + k] = g[:8].numpy().astype(np.float32)
    out.update(gradx_names=np.array(xn), gradx_norms=np.array(xnorm))
    path = os.path.join(HERE, "train_k21_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: round(v, 5) for k, v in losses.items()},
          "thr", thr, "voxels", len(coors), "pos", int(out["n_pos"]), "ext_pos", int(out["n_ext_pos"]), "guided",
          int(out["n_guided"]))


if __name__ == "__main__":
    main()
