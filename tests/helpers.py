"""Shared test inputs: seeded synthetic frames and random network parameters (same on CPU oracle and GPU)."""
import os

import numpy as np
import torch

import sassd
from sassd import synth
from oracle import nets as onets


def frame(name="k21", seed=0):
    if name == "k21":
        return synth.k21(seed)
    if name == "k17":
        return synth.k17(seed)
    if name == "small":
        return synth.lidar64(seed)[:3000]
    raise KeyError(name)


def random_bn(c, g):
    return dict(weight=torch.rand(c, generator=g) * 0.5 + 0.75, bias=torch.randn(c, generator=g) * 0.1,
                running_mean=torch.randn(c, generator=g) * 0.1, running_var=torch.rand(c, generator=g) + 0.5)


def vxnet_params(seed=0):
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, kind, cin, cout, key in onets.VXNET_LAYERS:
        k = 1 if kind == "1x1" else 27
        w = torch.randn(k, cin, cout, generator=g) * (2.0 / (cin * min(k, 8))) ** 0.5
        p[name] = dict(weight=w, bn=random_bn(cout, g))
    return p


def bev_params(seed=1, cin=320, c=256):
    g = torch.Generator().manual_seed(seed)
    p = {}
    for i in range(8):
        ci = cin if i == 0 else c
        ks = 1 if i == 7 else 3
        w = torch.randn(c, ci, ks, ks, generator=g) * (2.0 / (ci * ks * ks)) ** 0.5
        p["conv%d" % i] = dict(weight=w, bn=random_bn(c, g))
    return p


def head_params(seed=2, c=256, num_class=1, a=2):
    g = torch.Generator().manual_seed(seed)
    na = a * num_class
    p = {
        "conv_box": dict(weight=torch.randn(na * 7, c, 1, 1, generator=g) * 0.02, bias=torch.randn(na * 7, generator=g) * 0.05),
        "conv_cls": dict(weight=torch.randn(na * num_class, c, 1, 1, generator=g) * 0.02,
                         bias=torch.full((na * num_class,), -2.0) + torch.randn(na * num_class, generator=g) * 0.1),
        "conv_dir_cls": dict(weight=torch.randn(na * 2, c, 1, 1, generator=g) * 0.02, bias=torch.randn(na * 2, generator=g) * 0.05),
    }
    return p


def pswarp_params(seed=3, c=256, parts=28):
    g = torch.Generator().manual_seed(seed)
    return {
        "conv0": dict(weight=torch.randn(parts, c, 3, 3, generator=g) * (2.0 / (c * 9)) ** 0.5, bn=random_bn(parts, g)),
        "conv1": dict(weight=torch.randn(parts, parts, 1, 1, generator=g) * (1.0 / parts) ** 0.5),
    }


def fold_bn(bn, eps=1e-3):
    scale = bn["weight"] / torch.sqrt(bn["running_var"] + eps)
    shift = bn["bias"] - bn["running_mean"] * scale
    return scale.contiguous(), shift.contiguous()


def rand_bev_boxes(rng, k, spread=12.0):
    x = rng.uniform(0, spread, k); y = rng.uniform(0, spread, k)
    w = rng.uniform(1.2, 2.2, k); l = rng.uniform(3, 5, k); a = rng.uniform(-3.3, 3.3, k)
    return np.stack([x - w / 2, y - l / 2, x + w / 2, y + l / 2, a], 1).astype(np.float32)


randomize_detector = synth.randomize_detector          # product-side helper (bench.py must not import this module)


def oracle_params(sd):
    """detector state_dict -> the parameter dicts oracle.nets expects."""
    from sassd.pipeline import VXNET
    bn = lambda p: dict(weight=sd[p + ".weight"], bias=sd[p + ".bias"], running_mean=sd[p + ".running_mean"],   # noqa: E731
                        running_var=sd[p + ".running_var"])
    vx = {}
    for (wname, bnname, kind, cin, cout, key), (oname, *_r) in zip(VXNET, onets.VXNET_LAYERS):
        k = 1 if kind == "1x1" else 27
        vx[oname] = dict(weight=sd["neck.backbone.%s.weight" % wname].reshape(k, cin, cout),
                         bn=bn("neck.backbone.%s" % bnname))
    bev = {"conv%d" % i: dict(weight=sd["neck.fcn.conv%d.weight" % i], bn=bn("neck.fcn.bn%d" % i)) for i in range(8)}
    head = {n: dict(weight=sd["rpn_head.%s.weight" % n], bias=sd["rpn_head.%s.bias" % n])
            for n in ("conv_box", "conv_cls", "conv_dir_cls")}
    ps = {"conv0": dict(weight=sd["extra_head.convs.0.weight"], bn=bn("extra_head.convs.1")),
          "conv1": dict(weight=sd["extra_head.convs.3.weight"])}
    return vx, bev, head, ps


def oracle_features(sd, clouds, anchors, anchors_bv, cfg, num_class=1):
    """The threshold-free part of the path on the CPU oracle: voxels -> sparse backbone -> BEV net -> head maps, anchor
    masks.  clouds: list of numpy [N,4]."""
    vx, bev, head, ps = oracle_params(sd)
    feats, coors, coors3 = [], [], []
    for b, pts in enumerate(clouds):
        v, c, n = clib.points_to_voxel(pts, cfg["voxel_size"], cfg["pc_range"], cfg["max_points"], True,
                                       cfg["max_voxels"])
        feats.append(clib.voxel_mean(v, n))
        coors3.append(c)
        coors.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))
    feats, coors = np.concatenate(feats), np.concatenate(coors)
    B = len(clouds)
    x3, idx3, shape3, middle, books, acts = onets.vxnet_forward(feats, coors, cfg["sparse_shape"], B, vx, True)
    dense = onets.densify(x3, idx3, shape3, B)
    x, conv6 = onets.bevnet_forward(dense, bev)
    box, cls, dirp = onets.ssd_head_forward(x, head, num_class)
    masks = np.stack([onets.anchors_mask(c, anchors_bv, cfg["voxel_size"], cfg["pc_range"], cfg["grid_xyz"], 1)
                      for c in coors3])
    bcls = torch.sigmoid(cls.reshape(B, -1, num_class)).max(-1)[0]
    masked_scores = torch.cat([bcls[b][torch.from_numpy(masks[b])] for b in range(B)])
    return dict(feats=feats, coors=coors, x3=x3, idx3=idx3, acts=acts, dense=dense, x=x, conv6=conv6, box=box, cls=cls,
                dirp=dirp, masks=masks, books=books, ps=ps, anchors=anchors, num_class=num_class, B=B,
                masked_scores=masked_scores, grid_offsets=cfg.get("grid_offsets", (0.0, 40.0)),
                featmap_stride=cfg.get("featmap_stride", 0.4))


def oracle_select(ft, rpn_thr=0.1, score_thr=None, iou_thr=0.1):
    """The threshold-dependent tail: guided anchors (sigmoid > rpn_thr), PSWarp logits, rescoring + rotated NMS
    (skipped when score_thr is None)."""
    B, nc = ft["B"], ft["num_class"]
    an = torch.from_numpy(ft["anchors"]).view(1, -1, 7).expand(B, -1, -1)
    guided = onets.guided_anchors(ft["box"], ft["cls"], ft["dirp"], an, torch.from_numpy(ft["masks"]), nc, rpn_thr)
    logits, psfeat = onets.pswarp_forward(ft["conv6"], ft["ps"], [g[0] for g in guided], ft["grid_offsets"],
                                          ft["featmap_stride"])
    out = dict(guided=guided, logits=logits, psfeat=psfeat)
    if score_thr is not None:
        out["dets"] = [onets.rescore(g[0], lg, g[1], score_thr, iou_thr) for g, lg in zip(guided, logits)]
    return out


def safe_threshold(base, values, margin=1e-4, step=2.5e-4):
    """A threshold near `base` that no value approaches within `margin`: selections made with it are identical for any
    two implementations whose values agree to `margin` (GPU expf vs libm, fp32 sums in another order), so a parity
    test never has to skip a sample because a candidate sits on the threshold."""
    v = np.asarray(values, np.float64).ravel()
    for k in range(400):
        t = base + ((k + 1) // 2) * step * (1 if k % 2 else -1) if k else base
        if v.size == 0 or np.abs(v - t).min() > margin:
            return float(np.float32(t))
    raise AssertionError("no safe threshold near %g" % base)


def widest_gap_threshold(base, values, span=2e-2):
    """The threshold within base +- span that stays farthest from every value (the middle of the widest gap between
    neighbouring values): -> (threshold, distance to the nearest value).  For dense score sets, where safe_threshold
    would have to fall back to a margin of the size of the fp32 noise between two implementations."""
    v = np.sort(np.asarray(values, np.float64).ravel())
    v = v[(v > base - span) & (v < base + span)]
    edges = np.concatenate([[base - span], v, [base + span]])
    i = int(np.argmax(np.diff(edges)))
    t = float(np.float32(0.5 * (edges[i] + edges[i + 1])))
    near = float(np.abs(v - t).min()) if v.size else span
    return t, near


def oracle_forward_safe(sd, clouds, anchors, anchors_bv, cfg, num_class=1, rpn_thr=0.1, score_thr=0.3):
    """Whole path on the CPU oracle with thresholds nudged (by multiples of 2.5e-4) away from every candidate score.
    Returns (results dict, rpn_thr, score_thr) -- hand the two thresholds to the plan under test."""
    ft = oracle_features(sd, clouds, anchors, anchors_bv, cfg, num_class)
    rpn = safe_threshold(rpn_thr, ft["masked_scores"].numpy())
    sel = oracle_select(ft, rpn)
    sc = safe_threshold(score_thr, torch.sigmoid(torch.cat([l.reshape(-1) for l in sel["logits"]])).numpy()
                        if len(sel["logits"]) else np.zeros(0))
    sel["dets"] = [onets.rescore(g[0], lg, g[1], sc, 0.1) for g, lg in zip(sel["guided"], sel["logits"])]
    ft.update(sel)
    return ft, rpn, sc


def oracle_forward(sd, clouds, anchors, anchors_bv, cfg, num_class=1, keep=None):
    """Whole path on the CPU oracle with the configured thresholds (0.1 / cfg score_thr)."""
    ft = oracle_features(sd, clouds, anchors, anchors_bv, cfg, num_class)
    ft.update(oracle_select(ft, 0.1, cfg.get("score_thr", 0.3)))
    return ft


from oracle import clib  # noqa: E402


def calibrate_cls_head(model, cloud, anchors_bv, cfg, target_count=400, target_std=0.45, **_unused):
    """Rescale rpn_head.conv_cls (weights and bias) with the CPU oracle so that masked-anchor logits have std
    `target_std` and about `target_count` anchors pass sigmoid > 0.1 on `cloud` (SURVEY.md 8d asks for K ~ 10^2-10^3)
    instead of tens of thousands with raw random weights."""
    sd = model.state_dict()
    vx, bev, head, ps = oracle_params(sd)
    v, c, n = clib.points_to_voxel(cloud, cfg["voxel_size"], cfg["pc_range"], cfg["max_points"], True, cfg["max_voxels"])
    coors = np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)
    x3, idx3, shape3, *_ = onets.vxnet_forward(clib.voxel_mean(v, n), coors, cfg["sparse_shape"], 1, vx)
    x, _ = onets.bevnet_forward(onets.densify(x3, idx3, shape3, 1), bev)
    ncls = model.rpn_head._num_class
    _, cls, _ = onets.ssd_head_forward(x, head, ncls)
    m = onets.anchors_mask(c, anchors_bv, cfg["voxel_size"], cfg["pc_range"], cfg["grid_xyz"], 1)
    lg = cls.reshape(-1, ncls)[torch.from_numpy(m)].max(-1)[0]
    s = target_std / max(lg.std().item(), 1e-6)
    frac = min(0.5, target_count / max(lg.numel(), 1))
    with torch.no_grad():
        b_old = model.rpn_head.conv_cls.bias.clone()
        # logit' = s * (logit - b_old) + b_new; pick b_new so the (1 - frac) quantile lands on logit(0.1).
        # (per-channel biases keep their scaled spread; the common shift is applied to all of them)
        scaled = (lg.double() - b_old.double().mean()) * s
        qv = torch.quantile(scaled, 1.0 - frac).item()
        thr_logit = float(np.log(0.1 / 0.9))
        model.rpn_head.conv_cls.weight.mul_(s)
        model.rpn_head.conv_cls.bias.copy_((b_old - b_old.mean()) * s + (thr_logit - qv))
    return model


# ---- training-step parity: selection tap + arbiter bars (round 5) --------------------------------------------------------
class GuidedSelectionTap:
    """Context manager around sassd.kernels.guided_select (the one discrete decision of a training step that depends on
    network outputs).  Records the selection the HIP kernel makes -- `.selected[b]`: ascending anchor indices of sample b --
    or, with `force` (per sample an ascending int array of anchor indices), replaces the kernel's selection by the given one
    (the kernel still runs; its result is recorded in `.own`), so that two implementations whose scores differ by more than
    any threshold margin (bf16) differentiate the same candidate set."""

    def __init__(self, force=None):
        self.force, self.selected, self.own = force, None, None

    def __enter__(self):
        import torch
        from sassd import kernels as K
        self._K, self._orig = K, K.guided_select

        def tapped(cls_preds, anchor_mask, score_thr, cap, overflow):
            sel, cnt = self._orig(cls_preds, anchor_mask, score_thr, cap, overflow)
            c = cnt.cpu().numpy()
            self.own = [sel[b, :int(c[b])].cpu().numpy() for b in range(sel.shape[0])]
            self.selected = self.own
            if self.force is not None:
                assert len(self.force) == sel.shape[0]
                sel = torch.arange(cap, dtype=torch.int64, device=sel.device).repeat(sel.shape[0], 1)   # padding: sel[b][p] = p
                for b, f in enumerate(self.force):
                    assert len(f) <= cap
                    sel[b, :len(f)] = torch.as_tensor(np.asarray(f, np.int64), device=sel.device)
                cnt = torch.tensor([len(f) for f in self.force], dtype=torch.int32, device=sel.device)
                self.selected = [np.asarray(f, np.int64) for f in self.force]
            return sel, cnt
        K.guided_select = tapped
        return self

    def __exit__(self, *exc):
        self._K.guided_select = self._orig


def selection_as_mask_ranks(selected, mask):
    """anchor indices (into all anchors of a sample) -> ranks among the sample's masked anchors (oracle guided_sel form)"""
    rank = np.cumsum(np.asarray(mask).astype(np.int64)) - 1
    sel = np.asarray(selected, np.int64)
    assert np.asarray(mask)[sel].all(), "a selected anchor lies outside the anchor mask"
    return rank[sel]


def arbiter_report(got, arb, floor32, factor=3.0, rel_floor=2e-4, whole_factor=2.0, whole_floor=5e-5):
    """The round-5 gradient bar.  got / arb / floor32: {name: tensor} -- the GPU gradient, the ARBITER (the same arithmetic
    in float64) and the CPU oracle's own fp32 evaluation of it.  A tensor passes when the GPU is no farther from the arbiter
    than `factor` x the CPU oracle is (or than rel_floor x ||arbiter||, for tensors the two CPU evaluations happen to agree
    on); the whole model (all tensors concatenated) likewise with whole_factor / whole_floor.
    -> (bad {name: (e_gpu, e_cpu)} relative, whole (E_gpu, E_cpu) relative, rows sorted by e_gpu)."""
    rows, bad = [], {}
    ng = nc = den = 0.0
    for k, a in arb.items():
        a = a.detach().double().cpu().reshape(-1)
        n = float(a.norm())
        if n < 1e-7 or k not in got or got[k] is None:
            continue
        eg = float((got[k].detach().double().cpu().reshape(-1) - a).norm())
        ec = float((floor32[k].detach().double().cpu().reshape(-1) - a).norm())
        ng, nc, den = ng + eg * eg, nc + ec * ec, den + n * n
        rows.append((eg / n, ec / n, k))
        if not eg <= max(factor * ec, rel_floor * n):
            bad[k] = (eg / n, ec / n)
    rows.sort(reverse=True)
    whole = ((ng / den) ** 0.5, (nc / den) ** 0.5)
    whole_ok = whole[0] <= max(whole_factor * whole[1], whole_floor)
    return bad, whole, whole_ok, rows


def dump_rows(title, rows):
    """SASSD_PARITY_DUMP=<file>: append the full per-tensor table of an arbiter comparison (calibration runs on the GPU box)"""
    path = os.environ.get("SASSD_PARITY_DUMP")
    if path:
        with open(path, "a") as f:
            f.write("## %s\n" % title)
            for eg, ec, k in rows:
                f.write("%.3e %.3e %6.1f %s\n" % (eg, ec, eg / max(ec, 1e-30), k))
