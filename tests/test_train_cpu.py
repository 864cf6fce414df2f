"""Training-side host logic (sassd.train_ops + the loss / target methods of the heads) against golden vectors produced
by the reference's own Python (tests/golden/make_golden_train.py -> train_fns.npz).  CPU only: these are torch
elementwise / indexing mirrors; the HIP kernels they sit on are covered by the -m gpu tests."""
import os

import numpy as np
import pytest
import torch

import sassd  # noqa: F401
from sassd import train_ops as T
from sassd.config import ConfigDict
from sassd.detector import PSWarpHead, SpMiddleFHD, SSDRotateHead

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_fns.npz"))


def t(name):
    return torch.from_numpy(G[name])


def close(a, b, tol=1e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), np.abs(a - b).max()


def test_losses():
    close(T.weighted_sigmoid_focal_loss(t("fl_pred"), t("fl_target"), t("fl_weight"), avg_factor=1.), G["fl_avg1"])
    close(T.weighted_sigmoid_focal_loss(t("fl_pred"), t("fl_target"), t("fl_weight")), G["fl_default"])
    close(T.weighted_smoothl1(t("sl_pred"), t("sl_target"), t("sl_weight"), beta=1 / 9., avg_factor=1.),
          G["sl_beta9"])
    close(T.weighted_smoothl1(t("sl_pred"), t("sl_target"), t("sl_weight").expand(200, 7).contiguous()),
          G["sl_default"])
    close(T.weighted_cross_entropy(t("ce_logits"), t("ce_labels"), t("ce_weight"), avg_factor=1.), G["ce_avg1"])
    close(T.weighted_cross_entropy(t("ce_logits"), t("ce_labels"), t("ce_weight")), G["ce_default"])


def test_similarity_and_encode():
    anchors, gt = t("anchors"), t("gt")
    close(T.NearestIouSimilarity()(anchors[::37], gt), G["near_iou"], 1e-6)
    enc = T.second_box_encode(gt[torch.arange(400) % 9], anchors[1000:1400])
    close(enc, G["encode"], 1e-6)
    close(T.second_box_decode(enc, anchors[1000:1400]), gt[torch.arange(400) % 9].numpy(), 1e-5)   # round trip


@pytest.mark.parametrize("case", ["masked", "nomask", "nogt"])
def test_create_target(case):
    anchors, gt = t("anchors"), t("gt")
    gmask = torch.tensor([1, 1, 0, 1, 1, 1, 0, 1, 1], dtype=torch.bool)
    am, gm, gb = dict(masked=(t("anchor_mask"), gmask, gt), nomask=(None, None, gt),
                      nogt=(t("anchor_mask"), None, gt[:0]))[case]
    lab, tar, mx = T.create_target_torch(anchors, am, gb, torch.ones(len(gb), dtype=torch.int64), gm,
                                         similarity_fn=T.NearestIouSimilarity(), box_encoding_fn=T.second_box_encode,
                                         matched_threshold=0.6, unmatched_threshold=0.45, box_code_size=7)
    assert np.array_equal(lab.numpy(), G["ct_%s_labels" % case])          # integer work: exact
    close(tar, G["ct_%s_targets" % case], 1e-6)
    close(mx[am] if am is not None else mx, G["ct_%s_max" % case], 1e-6)   # ours is full-size, -1 at masked-out rows
    if am is not None:
        assert bool((mx[~am] == -1).all()) and bool((lab[~am] == -1).all())
    if case != "nogt":
        assert (lab > 0).sum() > 0


def _rpn_inputs():
    anc = dict(Car=torch.stack([t("anchors"), t("a2")]))
    msk = dict(Car=torch.stack([t("anchor_mask"), t("m2")]))
    gtb = [t("gt"), t("gt2")]
    gtl = [torch.ones(9, dtype=torch.int64), torch.ones(6, dtype=torch.int64)]
    gtt = [np.array(["Car"] * 9), np.array(["Car"] * 5 + ["Van"])]
    return anc, msk, gtb, gtl, gtt


def test_rpn_loss_and_gradients():
    head = SSDRotateHead(num_class=1, num_output_filters=8, num_anchor_per_loc=2, box_code_size=7)
    anc, msk, gtb, gtl, gtt = _rpn_inputs()
    box, cls, dr = (t(k).clone().requires_grad_() for k in ("rpn_box", "rpn_cls", "rpn_dir"))
    cfg = ConfigDict(assigner=ConfigDict(Car=ConfigDict(pos_iou_thr=0.6, neg_iou_thr=0.45, min_pos_iou=0.45),
                                         ignore_iof_thr=-1, similarity_fn="NearestIouSimilarity"), anchor_thr=0.1)
    ls = head.loss(box, cls, dr, gtb, gtl, gtt, anc, msk, cfg)
    for k in ("rpn_loc_loss", "rpn_cls_loss", "rpn_dir_loss"):
        close(ls[k].detach(), G[k], 2e-6)
    tot = ls["rpn_loc_loss"] + ls["rpn_cls_loss"] + ls["rpn_dir_loss"]
    gb, gc, gd = torch.autograd.grad(tot.sum(), [box, cls, dr])
    close(gb, G["rpn_gbox"], 1e-6)
    close(gc, G["rpn_gcls"], 1e-6)
    close(gd, G["rpn_gdir"], 1e-6)


def test_guided_anchors_train_mode():
    head = SSDRotateHead(num_class=1, num_output_filters=8, num_anchor_per_loc=2, box_code_size=7)
    anc, msk, gtb, gtl, _ = _rpn_inputs()
    guided, labels = head.get_guided_anchors(t("rpn_box"), t("rpn_cls"), t("rpn_dir"), anc, msk, gtb, gtl, thr=0.1)
    for i in range(2):
        assert guided[i].shape == G["guided%d" % i].shape
        close(guided[i], G["guided%d" % i], 1e-6)
        assert np.array_equal(labels[i].numpy(), G["guided_labels%d" % i])
        close(guided[i][:len(gtb[i])], gtb[i].numpy(), 0)                  # ground truth is prepended


def test_aux_loss_arithmetic():
    neck = SpMiddleFHD.__new__(SpMiddleFHD)
    neck.build_aux_target = lambda pts, gt: (t("aux_labels"), t("aux_offsets"))
    ls = SpMiddleFHD.aux_loss(neck, None, t("aux_cls"), t("aux_reg"), [None, None])
    close(ls["aux_loss_cls"], G["aux_loss_cls"], 2e-6)
    close(ls["aux_loss_reg"], G["aux_loss_reg"], 2e-6)


def test_rescoring_loss_with_reference_iou():
    """PSWarpHead.loss arithmetic on CPU: the rotated 3-D IoU is an input here (taken from the CPU oracle); the HIP
    overlap kernel itself is checked in the -m gpu suite."""
    from oracle import clib
    import sassd.iou3d_utils as iu

    class OracleIou3d:
        def __call__(self, a, b):
            ov = torch.from_numpy(clib.boxes_overlap_bev(iu.boxes3d_to_bev_torch(a).numpy(),
                                                         iu.boxes3d_to_bev_torch(b).numpy()))
            oh = torch.clamp(torch.min((a[:, 2] + a[:, 5]).view(-1, 1), (b[:, 2] + b[:, 5]).view(1, -1))
                             - torch.max(a[:, 2].view(-1, 1), b[:, 2].view(1, -1)), min=0)
            o3 = ov * oh
            va = (a[:, 3] * a[:, 4] * a[:, 5]).view(-1, 1)
            vb = (b[:, 3] * b[:, 4] * b[:, 5]).view(1, -1)
            return o3 / torch.clamp(va + vb - o3, min=1e-7)
    close(OracleIou3d()(t("anchors")[::37].contiguous(), t("gt")), G["rot_iou3d"], 1e-5)
    T.OracleIou3d = OracleIou3d
    try:
        ext = PSWarpHead(grid_offsets=(0., 40.), featmap_stride=.4, in_channels=8, num_class=1, num_parts=28)
        gsub = [t("guided0")[:600].contiguous(), t("guided1")[:500].contiguous()]
        score = t("ext_score").clone().requires_grad_()
        cfg = ConfigDict(assigner=ConfigDict(pos_iou_thr=0.7, neg_iou_thr=0.7, similarity_fn="OracleIou3d"))
        loss = ext.loss(score, [t("gt"), t("gt2")], None, gsub, cfg)["loss_cls"]
        close(loss.detach(), G["ext_loss"], 1e-5)
        close(torch.autograd.grad(loss.sum(), score)[0], G["ext_gscore"], 1e-5)
    finally:
        del T.OracleIou3d


# ---- the CPU training oracle's own restatements, pinned on the same reference-generated vectors -------------------
def test_train_ref_pieces_pinned():
    from oracle import train_ref as R
    anchors, gt = G["anchors"], G["gt"]
    close(R.nearest_iou(anchors[::37], gt), G["near_iou"], 1e-6)
    close(R.rotated_iou3d(anchors[::37], gt), G["rot_iou3d"], 1e-5)
    close(R.box_encode(gt[np.arange(400) % 9], anchors[1000:1400]), G["encode"], 1e-6)
    gmask = np.array([1, 1, 0, 1, 1, 1, 0, 1, 1], bool)
    for case, (am, g_) in dict(masked=(G["anchor_mask"], gt[gmask]), nomask=(None, gt),
                               nogt=(G["anchor_mask"], gt[:0])).items():
        lab, tar, mx = R.assign(anchors, am, g_, np.ones(len(g_), np.int64), 0.6, 0.45, R.nearest_iou)
        assert np.array_equal(lab, G["ct_%s_labels" % case]), case
        close(tar, G["ct_%s_targets" % case], 1e-6)
        close(mx, G["ct_%s_max" % case], 1e-6)
    close(R.focal_sum(t("fl_pred"), t("fl_target"), t("fl_weight")).view(1), G["fl_avg1"], 2e-6)
    close(R.smooth_l1_sum(t("sl_pred"), t("sl_target"), t("sl_weight"), 1 / 9.).view(1), G["sl_beta9"], 2e-6)
    enc = torch.from_numpy(G["encode"])
    close(R.decode(enc, torch.from_numpy(anchors[1000:1400])), gt[np.arange(400) % 9], 1e-5)


def test_onecycle_schedule_matches_reference():
    from sassd import train
    O = np.load(os.path.join(os.path.dirname(__file__), "golden", "optim_ref.npz"))

    class Opt:
        lr = mom = None
    o = Opt()
    s = train.OneCycle(o, 20, 0.003, [0.95, 0.85], 10, 0.4)
    for it in range(20):
        s.step(it)
        assert abs(o.lr - O["lr"][it]) < 1e-12 and abs(o.mom - O["mom"][it]) < 1e-12, it


def test_flat_params_views():
    from sassd import train
    m = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3, bias=False))
    before = {k: v.clone() for k, v in m.state_dict().items()}
    f = train.FlatParams(m)
    assert f.numel % 4 == 0 and all(o % 4 == 0 for o in f.offsets)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    m(torch.randn(4, 6)).sum().backward()
    assert f.grad.abs().sum() > 0 and m[0].weight.grad.data_ptr() == f.grad.data_ptr()
    f.data.mul_(2.0)
    assert torch.equal(m[0].weight, before["0.weight"] * 2)
    f.zero_grad()                                   # .grad = None: the next backward's gradients are stolen, not added
    assert m[2].weight.grad is None
    assert float(f.grad.abs().sum()) == 0           # the flat buffer collects lazily: no gradient -> zeros
    assert m[2].weight.grad.data_ptr() == f.grad.data_ptr() + 4 * f.offsets[2 + 2]       # and .grad is a view again
    m(torch.randn(4, 6)).sum().backward()           # gradients arrive in their own tensors ...
    assert m[0].weight.grad.data_ptr() == f.grad.data_ptr() and float(f.grad.abs().sum()) > 0    # ... one collect later


def test_rpn_loss_with_an_empty_sample():
    """A frame without ground truth (all anchors inside the mask become negatives) must not break the batch."""
    head = SSDRotateHead(num_class=1, num_output_filters=8, num_anchor_per_loc=2, box_code_size=7)
    anc, msk, gtb, gtl, gtt = _rpn_inputs()
    gtb[1], gtl[1], gtt[1] = gtb[1][:0], gtl[1][:0], gtt[1][:0]
    box, cls, dr = (t(k).clone().requires_grad_() for k in ("rpn_box", "rpn_cls", "rpn_dir"))
    cfg = ConfigDict(assigner=ConfigDict(Car=ConfigDict(pos_iou_thr=0.6, neg_iou_thr=0.45, min_pos_iou=0.45),
                                         ignore_iof_thr=-1, similarity_fn="NearestIouSimilarity"), anchor_thr=0.1)
    ls = head.loss(box, cls, dr, gtb, gtl, gtt, anc, msk, cfg)
    tot = sum(v.sum() for v in ls.values())
    assert torch.isfinite(tot)
    tot.backward()
    assert torch.isfinite(box.grad).all() and float(box.grad[1].abs().sum()) == 0.0     # no positives -> no box loss
    guided, labels = head.get_guided_anchors(box.detach(), cls.detach(), dr.detach(), anc, msk, gtb, gtl, thr=0.1)
    assert guided[1].shape[1] == 7 and len(labels[1]) == len(guided[1])


def test_side_stream_prefetch_without_cuda_is_a_passthrough():
    """sassd.train.SideStreamPrefetch on a machine without a GPU just calls the builder (the stream logic is covered by
    bench.py --mode train on the GPU box)."""
    from sassd import train
    calls = []

    def build(i, k=0):
        calls.append((i, k))
        return dict(a=torch.zeros(2), b=[torch.ones(1), (torch.ones(1), None)])
    pf = train.SideStreamPrefetch(build)
    if not torch.cuda.is_available():
        assert pf.stream is None
    out = pf(3, k=4)
    assert calls == [(3, 4)] and set(out) == {"a", "b"}
    train._record_stream(out, None)            # CPU tensors: nothing to record, nested containers are walked


def test_flat_params_collects_lazily():
    """zero_grad() -> None gradients; backward leaves per-parameter tensors; `flat.grad` gathers them (one multi-tensor
    copy), re-points .grad at the views, and a parameter without a gradient reads as zeros."""
    from sassd import train
    m = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    f = train.FlatParams(m)
    f.zero_grad()
    m[0](torch.ones(5, 4)).sum().backward()    # only the first layer receives gradients
    own = m[0].weight.grad
    assert own.data_ptr() != f._grad.data_ptr()
    g = f.grad
    assert m[0].weight.grad.data_ptr() == g.data_ptr() and torch.equal(m[0].weight.grad, own)
    assert float(m[1].weight.grad.abs().sum()) == 0 and float(g[f.offsets[2]:].abs().sum()) == 0
    f.collect()                                 # idempotent
    assert torch.equal(m[0].weight.grad, own)


def test_waymo_train_golden_is_complete():
    """tests/golden/waymo_train_ref.npz (the CPU oracle's training step at the BASELINE configs[4] shape, made by
    tests/golden/make_golden_waymo_train.py): six non-zero loss terms, the 79 302-voxel frame, positives in both
    assignments, gradients for every parameter (norm + projection) and elementwise for the stored layers."""
    import os
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "waymo_train_ref.npz"))
    assert sorted(str(k) for k in G["loss_names"]) == ["aux_loss_cls", "aux_loss_reg", "loss_cls", "rpn_cls_loss",
                                                      "rpn_dir_loss", "rpn_loc_loss"]
    assert np.all(G["losses"] > 0) and int(G["n_voxels"]) == 79302 and int(G["n_pos"]) > 0 and int(G["n_ext_pos"]) >= 12
    assert len(G["grad_names"]) == len(G["grad_norms"]) == len(G["grad_projs"]) >= 75
    stored = [k for k in G.files if k.startswith("grad:") or k.startswith("grad8:")]
    assert len(stored) >= 40 and all(np.isfinite(G[k]).all() for k in stored)
    # round 5: the float64 arbiter of the same step rides along, with every parameter's distance to it
    assert all(("f64/" + k) in G.files for k in stored) and len(G["grad_dist"]) == len(G["grad_names"]) == len(G["f64/grad_names"])
    assert (np.abs(G["losses"] - G["f64/losses"]) <= 1e-5 * np.abs(G["f64/losses"])).all()
    num = sum(float(((G[k].astype(np.float64) - G["f64/" + k]) ** 2).sum()) for k in stored)
    den = sum(float((G["f64/" + k].astype(np.float64) ** 2).sum()) for k in stored)
    assert 1e-6 < (num / den) ** 0.5 < 3e-3, (num / den) ** 0.5      # the fp32 floor of this workload


def test_deferred_bn_batch_counters_land_in_any_state_dict():
    """count_bn_batch keeps `num_batches_tracked += 1` on the host; every state_dict() that includes the layer -- its own,
    a parent's -- sees the flushed counter (sassd.autograd.count_bn_batch / flush_bn_counters)."""
    import torch
    from sassd import autograd as AG
    bn_a, bn_b = torch.nn.BatchNorm2d(4), torch.nn.BatchNorm1d(3)
    parent = torch.nn.Sequential(torch.nn.Conv2d(4, 4, 1), bn_a)
    for _ in range(3):
        AG.count_bn_batch(bn_a)
    AG.count_bn_batch(bn_b)
    assert int(bn_a.num_batches_tracked) == 0                      # nothing launched yet
    assert int(parent.state_dict()["1.num_batches_tracked"]) == 3
    assert int(bn_b.state_dict()["num_batches_tracked"]) == 1
    AG.count_bn_batch(bn_a)
    AG.flush_bn_counters()
    assert int(bn_a.num_batches_tracked) == 4 and int(bn_b.num_batches_tracked) == 1
    AG.flush_bn_counters()                                          # idempotent
    assert int(bn_a.num_batches_tracked) == 4
