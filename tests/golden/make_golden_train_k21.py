#!/usr/bin/env python
"""Golden vector of ONE training step on the workload bench.py's `train` record measures (BASELINE configs[2]): car_cfg on
its own full grid, batch 2 of synthetic lidar64 K21 frames (21 500 points each), 8 car boxes per frame on occupied voxels
(bench.synth_gt_on_points), random-init weights of synth.build_detector_for(seed 0, cls_bias -3) -- computed by the CPU
oracle oracle/train_ref.py::train_step (torch-CPU autograd over oracle rulebooks; a few CPU-minutes, hence a fixture).
Stored like tests/golden/make_golden_waymo_train.py: the six loss terms, the threshold-safe guided-anchor threshold, label /
candidate counts, the full gradient of a subset of layers, and for EVERY parameter its gradient norm and a seeded projection
-- in FOUR variants of the step (round 5): the fp32 oracle (keys without prefix), its float64 arbiter ("f64/"), and the
rounded-operand step of BASELINE configs[2] (dense-conv operands rounded to bf16 where the HIP kernels round them) in fp32
("b32/") and float64 ("b64/"); all four differentiate the full objective over the candidate set the fp32 oracle selects
("sel0", "sel1").  "grad_dist" / "b32/grad_dist" hold every parameter's distance ||g_fp32 - g_float64||: the floor of this
workload that the GPU bars are stated in multiples of.

    python tests/golden/make_golden_train_k21.py        # writes tests/golden/train_k21_ref.npz
    python tests/golden/make_golden_train_k21.py --only-rounded      # re-derive "b32/" / "b64/" after a rounding-rule change

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
projection, FULL, SLICED, stored, pack = MGW.projection, MGW.FULL, MGW.SLICED, MGW.stored, MGW.pack
B, NGT = 2, 8


def build():
    """(model on CPU in train mode, config, workload dict, clouds [B x [21500,4]], gts [B x [8,7]]) -- exactly what
    bench.train_measure feeds rank 0 in its first step."""
    w = synth.workload("car")
    model, cfg = synth.build_detector_for(w, 0, train=True, cls_bias=-3.0)
    clouds = [w["frame"](i) for i in range(B)]
    gts = [bench.synth_gt_on_points(p, i, NGT, "car") for i, p in enumerate(clouds)]
    return model, cfg, w, clouds, gts


def step_args(model, c, w, clouds, gts):
    """positional arguments of oracle.train_ref.train_step for this workload, plus the [B, A] anchor masks"""
    from oracle import clib, nets as onets
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
    return (sd, feats, coors, B, shape, gts, types, ["Car"], {"Car": an}, {"Car": m}, {"Car": (a.pos_iou_thr, a.neg_iou_thr)}), m


def refresh_rounded():
    """--only-rounded: recompute the two ROUNDED variants ("b32/", "b64/") under the oracle's current rounding rule
    (oracle.train_ref.bf16_conv_rule -- round 6 added the 1x1 layers) on the stored candidate set and threshold; the fp32
    oracle and its float64 arbiter stay byte for byte what the file holds."""
    from oracle import train_ref
    path = os.path.join(HERE, "train_k21_ref.npz")
    old = dict(np.load(path))
    model, c, w, clouds, gts = build()
    args, m = step_args(model, c, w, clouds, gts)
    thr = float(old["anchor_thr"])
    sel = [np.searchsorted(np.nonzero(m[b])[0], old["sel%d" % b]) for b in range(B)]
    for b in range(B):
        assert np.array_equal(np.nonzero(m[b])[0][sel[b]], old["sel%d" % b])
    lb32, gb32, _ = train_ref.train_step(*args, anchor_thr=thr, guided_sel=sel, bf16=("bev",))
    lb64, gb64, _ = train_ref.train_step(*args, anchor_thr=thr, guided_sel=sel, bf16=("bev",), dtype=torch.float64)
    out = {k: v for k, v in old.items() if not (k.startswith("b32/") or k.startswith("b64/"))}
    out.update(pack("b32/", lb32, gb32, gb64))
    out.update(pack("b64/", lb64, gb64))
    assert set(out) == set(old), set(out) ^ set(old)
    np.savez_compressed(path, **out)
    ks = [k for k in gb32 if gb32[k] is not None and float(gb64[k].norm()) > 1e-7]
    num = sum(float((gb32[k].double() - gb64[k].double()).pow(2).sum()) for k in ks)
    den = sum(float(gb64[k].double().pow(2).sum()) for k in ks)
    print("wrote", path, os.path.getsize(path), "bytes; rounded fp32 vs rounded float64: whole-model gradient %.2e"
          % (num / den) ** 0.5, {k: round(float(v), 5) for k, v in lb64.items()})


def main():
    import time
    import helpers as H
    from oracle import train_ref
    if "--only-rounded" in sys.argv:
        return refresh_rounded()
    model, c, w, clouds, gts = build()
    args, m = step_args(model, c, w, clouds, gts)
    losses, grads, ex = train_ref.train_step(*args)
    top = torch.sigmoid(ex["cls"]).reshape(B, -1)[torch.from_numpy(m)].numpy()
    thr = H.safe_threshold(0.1, top, margin=1e-4, step=2.5e-4)
    if abs(thr - 0.1) > 1e-9:
        losses, grads, ex = train_ref.train_step(*args, anchor_thr=thr)
    sel = ex["guided_sel"]                          # per sample: ranks among the masked anchors
    out = dict(anchor_thr=np.float64(thr), n_voxels=np.int64(len(args[2])), n_masked=np.int64(m.sum()),
               n_pos=np.int64((ex["labels"] > 0).sum()), n_ext_pos=np.int64((ex["ext_labels"] > 0).sum()),
               n_guided=np.int64(sum(len(g) for g in ex["guided"])), loss_names=np.array(sorted(losses)))
    for b in range(B):                              # the selection as indices into ALL anchors (what sassd_guided_select emits)
        out["sel%d" % b] = np.nonzero(m[b])[0][sel[b]].astype(np.int32)
    # the three other variants differentiate the SAME candidate set (guided_sel): the float64 arbiter of the fp32 step, and
    # the rounded-operand step (BASELINE configs[2], bf16 dense convs) in fp32 and in float64
    t0 = time.time()
    l64, g64, _ = train_ref.train_step(*args, anchor_thr=thr, guided_sel=sel, dtype=torch.float64)
    print("f64 %.0f s" % (time.time() - t0), flush=True)
    lb32, gb32, _ = train_ref.train_step(*args, anchor_thr=thr, guided_sel=sel, bf16=("bev",))
    lb64, gb64, _ = train_ref.train_step(*args, anchor_thr=thr, guided_sel=sel, bf16=("bev",), dtype=torch.float64)
    out.update(pack("", losses, grads, g64))
    out.update(pack("f64/", l64, g64))
    out.update(pack("b32/", lb32, gb32, gb64))
    out.update(pack("b64/", lb64, gb64))
    path = os.path.join(HERE, "train_k21_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: round(v, 5) for k, v in losses.items()},
          "thr", thr, "voxels", len(args[2]), "pos", int(out["n_pos"]), "ext_pos", int(out["n_ext_pos"]), "guided",
          int(out["n_guided"]))
    for tag, a, b in (("fp32 vs float64", grads, g64), ("rounded fp32 vs rounded float64", gb32, gb64),
                      ("rounded float64 vs float64", gb64, g64)):
        ks = [k for k in a if a[k] is not None and float(b[k].norm()) > 1e-7]
        num = sum(float((a[k].double() - b[k].double()).pow(2).sum()) for k in ks)
        den = sum(float(b[k].double().pow(2).sum()) for k in ks)
        per = sorted(((float((a[k].double() - b[k].double()).norm() / b[k].double().norm()), k) for k in ks), reverse=True)
        print("%s: whole-model gradient %.2e; worst tensors %s" % (tag, (num / den) ** 0.5, [(round(v, 5), k) for v, k in per[:5]]))


if __name__ == "__main__":
    main()
