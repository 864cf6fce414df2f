#!/usr/bin/env python
"""Golden vector of ONE training step at the BASELINE configs[4] shape (Waymo-scale synthetic frame: 180 000 points,
0.1 x 0.1 x 0.15 m voxels, grid 40 x 1504 x 1504, 79 302 active voxels, BEV 188 x 188) computed by the CPU oracle
oracle/train_ref.py::train_step (torch-CPU autograd over oracle rulebooks; ~2.5 minutes on 8 cores -- too slow to run
inside the GPU test, hence this fixture).  Stored: the six loss terms, the threshold-safe guided-anchor threshold, label /
candidate counts, the full gradient of a subset of layers (first / last sparse convs, every BatchNorm, the heads, the aux
linears, a slice of the big BEV convs), and for EVERY parameter its gradient norm and a seeded random projection -- for the
fp32 oracle (keys without prefix) and, since round 5, for its float64 arbiter ("f64/"), plus every parameter's distance
||g_fp32 - g_float64|| ("grad_dist") that the GPU bars are stated in multiples of.

    python tests/golden/make_golden_waymo_train.py        # writes tests/golden/waymo_train_ref.npz

tests/test_gpu_train.py::test_training_step_waymo_vs_oracle rebuilds the same model / frame / boxes (seeds below), runs
forward_train + backward through the HIP kernels with the batch built by the product's own device_batch, and compares."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sassd  # noqa: E402,F401
from sassd import synth, anchors as A  # noqa: E402
from sassd.config import Config  # noqa: E402
from sassd.detector import build_detector  # noqa: E402

SEED_MODEL, SEED_FRAME, SEED_GT, NGT = 7, 0, 3, 12
FULL = ("neck.backbone.conv0.0.weight", "neck.backbone.conv0.3.weight", "neck.backbone.down0.0.weight",
        "neck.backbone.conv3.6.weight", "neck.backbone.extra_conv.0.weight", "neck.point_fc.weight",
        "neck.point_cls.weight", "neck.point_reg.weight", "rpn_head.conv_box.weight", "rpn_head.conv_box.bias",
        "rpn_head.conv_cls.weight", "rpn_head.conv_cls.bias", "rpn_head.conv_dir_cls.weight",
        "rpn_head.conv_dir_cls.bias", "extra_head.convs.0.weight", "extra_head.convs.3.weight")
SLICED = ("neck.fcn.conv0.weight", "neck.fcn.conv3.weight", "neck.fcn.conv6.weight", "neck.fcn.conv7.weight")   # first 8 couts


def build():
    """(model on CPU in train mode, config, anchors [A,7], anchors_bv [A,4], cloud [180000,4], gt [12,7])."""
    c = Config.fromfile(os.path.join(ROOT, "configs", "car_cfg.py"))
    mcfg = dict(c.model)
    mcfg["neck"] = dict(mcfg["neck"], output_shape=[40, 1504, 1504], aux_offset=synth.WAYMO_RANGE[:3],
                        aux_voxel_size=synth.WAYMO_VOXEL)
    mcfg["extra_head"] = dict(mcfg["extra_head"], grid_offsets=(75.2, 75.2), featmap_stride=0.8)
    model = synth.randomize_detector(build_detector(mcfg, c.train_cfg, c.test_cfg), SEED_MODEL, cls_bias=-3.0,
                                     sparse_fan_div=1)
    an = A.AnchorGeneratorStride(sizes=[1.6, 3.9, 1.56], anchor_strides=[.8, .8, 1.], anchor_offsets=[-74.8, -74.8, -1.0],
                                 rotations=[0, 1.57])([1, 188, 188]).reshape(-1, 7)
    bv = A.rbbox2d_to_near_bbox(an[:, [0, 1, 3, 4, 6]]).astype(np.float32)
    p = synth.waymo_synth(SEED_FRAME)[:180000]
    r = np.random.default_rng(SEED_GT)
    cc = p[r.choice(len(p), NGT, replace=False), :3]
    gt = np.zeros((NGT, 7), np.float32)
    gt[:, 0], gt[:, 1] = np.clip(cc[:, 0], -70, 70), np.clip(cc[:, 1], -70, 70)
    gt[:, 2] = r.uniform(-1.9, -1.5, NGT)
    gt[:, 3], gt[:, 4], gt[:, 5] = r.uniform(1.5, 1.8, NGT), r.uniform(3.5, 4.4, NGT), r.uniform(1.4, 1.7, NGT)
    gt[:, 6] = r.uniform(-3.1, 3.1, NGT)
    return model, c, an.astype(np.float32), bv, p, gt


def projection(name, numel):
    """Seeded unit-variance direction for the random-projection check of parameter `name`."""
    g = torch.Generator().manual_seed(abs(hash_name(name)) % (2 ** 31))
    return torch.randn(numel, generator=g, dtype=torch.float64)


def hash_name(name):
    h = 0
    for ch in name:
        h = (h * 131 + ord(ch)) % 1000000007
    return h


def stored(k, g):
    """-> (key prefix, array) of the slice of gradient `k` a training fixture keeps elementwise, or None"""
    if k in FULL or ".bn" in k or k.split(".")[-2].isdigit() and g.dim() == 1:
        return "grad:", g.numpy().astype(np.float32)
    if k in SLICED:
        return "grad8:", g[:8].numpy().astype(np.float32)
    return None


def pack(tag, losses, grads, arbiter=None):
    """one variant of a step under the key prefix `tag` ("" = the fp32 oracle): loss terms, stored-layer gradients (float32
    storage), norm + seeded projection of EVERY parameter's gradient, and -- when `arbiter` (the float64 gradients of the same
    arithmetic) is given -- every parameter's distance to it (`grad_dist`)."""
    out = {tag + "losses": np.array([losses[k] for k in sorted(losses)], np.float64)}
    names, norms, projs, dn = [], [], [], []
    for k, g in grads.items():
        if g is None:
            continue
        gd = g.double().reshape(-1)
        names.append(k)
        norms.append(float(gd.norm()))
        projs.append(float(torch.dot(gd, projection(k, gd.numel()))))
        if arbiter is not None:
            dn.append(float((gd - arbiter[k].double().reshape(-1)).norm()))
        st = stored(k, g)
        if st is not None:
            out[tag + st[0] + k] = st[1]
    out.update({tag + "grad_names": np.array(names), tag + "grad_norms": np.array(norms), tag + "grad_projs": np.array(projs)})
    if arbiter is not None:
        out[tag + "grad_dist"] = np.array(dn)
    return out


def main():
    import helpers as H
    from oracle import clib, nets as onets, train_ref
    model, c, an, bv, p, gt = build()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    v, co, n = clib.points_to_voxel(p, synth.WAYMO_VOXEL, synth.WAYMO_RANGE, 5, True, 150000)
    feats = clib.voxel_mean(v, n)
    coors = np.concatenate([np.zeros((len(co), 1), np.int32), co], 1)
    m = onets.anchors_mask(co, bv, synth.WAYMO_VOXEL, synth.WAYMO_RANGE, (1504, 1504, 40), 1)
    types = [np.array(["Car"] * NGT)]
    a = c.train_cfg.rpn.assigner["Car"]
    args = (sd, feats, coors, 1, (40, 1504, 1504), [gt], types, ["Car"], {"Car": an[None]}, {"Car": m[None]},
            {"Car": (a.pos_iou_thr, a.neg_iou_thr)})
    kw = dict(grid_offsets=(75.2, 75.2), featmap_stride=0.8, aux_offset=synth.WAYMO_RANGE[:3],
              aux_voxel_size=synth.WAYMO_VOXEL)
    losses, grads, ex = train_ref.train_step(*args, **kw)
    top = torch.sigmoid(ex["cls"]).reshape(1, -1)[torch.from_numpy(m[None])].numpy()
    thr = H.safe_threshold(0.1, top, margin=1e-4, step=2.5e-4)
    if abs(thr - 0.1) > 1e-9:
        losses, grads, ex = train_ref.train_step(*args, anchor_thr=thr, **kw)
    out = dict(anchor_thr=np.float64(thr), n_voxels=np.int64(len(co)), n_masked=np.int64(m.sum()),
               n_pos=np.int64((ex["labels"] > 0).sum()), n_ext_pos=np.int64((ex["ext_labels"] > 0).sum()),
               n_guided=np.int64(len(ex["guided"][0])), loss_names=np.array(sorted(losses)))
    # round 5: the float64 arbiter of the same step on the same candidate set ("f64/" keys) and every parameter's distance
    # ||g_fp32 - g_float64|| ("grad_dist"): the GPU bars of test_training_step_waymo_vs_oracle are multiples of it
    import time
    t0 = time.time()
    l64, g64, _ = train_ref.train_step(*args, anchor_thr=thr, guided_sel=ex["guided_sel"], dtype=torch.float64, **kw)
    print("float64 step: %.0f s" % (time.time() - t0), flush=True)
    out.update(pack("", losses, grads, g64))
    out.update(pack("f64/", l64, g64))
    path = os.path.join(HERE, "waymo_train_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: round(v, 5) for k, v in losses.items()},
          "thr", thr, "pos", int(out["n_pos"]), "ext_pos", int(out["n_ext_pos"]), "guided", int(out["n_guided"]))


if __name__ == "__main__":
    main()
