"""Generate tests/golden/train_fns.npz by IMPORTING the reference's training-side Python (run in the build container
only; /root/reference does not exist on the GPU box):

  mmdet/core/loss/losses.py                  weighted_sigmoid_focal_loss / weighted_smoothl1 / weighted_cross_entropy
  mmdet/core/bbox3d/target_ops.py            create_target_torch
  mmdet/ops/iou3d/iou3d_utils.py             NearestIouSimilarity, RotateIou3dSimilarity (boxes_iou3d_gpu)
  mmdet/models/single_stage_heads/ssd_rotate_head.py   second_box_encode, SSDRotateHead.loss / get_guided_anchors,
                                             PSWarpHead.loss
  mmdet/models/necks/cmn.py                  SpMiddleFHD.aux_loss (arithmetic only; the point-in-box targets are inputs)

The reference modules are imported unchanged; packages that need compiled extensions are replaced by stubs, and the
`iou3d_cuda.boxes_overlap_bev_gpu` extension call is served by oracle/_ref (the reference's own iou3d device
functions compiled for the host).  Nothing here is copied into the repo; only the numeric results are saved.

    python tests/golden/make_golden_train.py
"""
import ast
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def _pkg(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    parent, _, leaf = name.rpartition(".")
    if parent in sys.modules:
        setattr(sys.modules[parent], leaf, m)
    return m


def _functions(rel, names, g):
    tree = ast.parse(open(os.path.join(REF, rel)).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), rel, "exec"), g)
    return g


def import_reference():
    from oracle import clib
    assert clib.ref() is not None, "oracle/_ref must be built (reference iou3d sources)"
    for p in ("mmdet", "mmdet.models", "mmdet.ops", "mmdet.ops.iou3d", "mmdet.core", "mmdet.core.loss",
              "mmdet.core.utils", "mmdet.core.bbox3d", "mmdet.core.post_processing"):
        _pkg(p)
    utils = types.ModuleType("mmdet.models.utils")
    utils.torch = torch
    _functions("mmdet/models/utils/__init__.py", {"one_hot"}, utils.__dict__)
    sys.modules["mmdet.models.utils"] = utils
    cu = types.ModuleType("mmdet.ops.iou3d.iou3d_cuda")

    def boxes_overlap_bev_gpu(a, b, out):
        out.copy_(torch.from_numpy(clib.boxes_overlap_bev(a.detach().numpy(), b.detach().numpy(), use_ref=True)))
    cu.boxes_overlap_bev_gpu = boxes_overlap_bev_gpu
    sys.modules["mmdet.ops.iou3d.iou3d_cuda"] = cu
    sys.modules["mmdet.ops.iou3d"].iou3d_cuda = cu
    torch.cuda.FloatTensor = torch.FloatTensor                 # the reference allocates its output with this
    _load("mmdet.ops.iou3d.iou3d_utils", "mmdet/ops/iou3d/iou3d_utils.py")
    _load("mmdet.core.loss.losses", "mmdet/core/loss/losses.py")
    sys.modules["mmcv"] = types.ModuleType("mmcv")
    _load("mmdet.core.utils.misc", "mmdet/core/utils/misc.py")
    _load("mmdet.core.bbox3d.target_ops", "mmdet/core/bbox3d/target_ops.py")
    bc = types.ModuleType("mmdet.core.bbox3d.box_coders")
    bc.GroundBox3dCoder = type("GroundBox3dCoder", (), {})
    sys.modules["mmdet.core.bbox3d.box_coders"] = bc
    sys.modules["mmdet.core.bbox3d"].box_coders = bc
    # nms: the reference's rotate_nms_torch / iou3d_utils.nms_gpu Python run unchanged; only the compiled
    # iou3d_cuda.nms_gpu(boxes_sorted, keep, thresh) -> count is served here: greedy pass in score order
    # (iou3d.cpp:100-116) over the reference's own iou_bev device function (oracle/_ref)
    import ctypes as C

    def nms_gpu_ext(boxes, keep, thresh):
        b = np.ascontiguousarray(boxes.detach().numpy(), np.float32)
        ref = clib.ref()
        n, kept, dead = len(b), [], np.zeros(len(b), bool)
        for i in range(n):
            if dead[i]:
                continue
            kept.append(i)
            pi = b[i:i + 1].ctypes.data_as(C.c_void_p)
            for jj in range(i + 1, n):
                if not dead[jj] and ref.ref_iou_bev(pi, b[jj:jj + 1].ctypes.data_as(C.c_void_p)) > thresh:
                    dead[jj] = True
        keep[:len(kept)] = torch.tensor(kept, dtype=torch.int64)
        return len(kept)
    cu.nms_gpu = nms_gpu_ext
    torch.Tensor.cuda = lambda self, *a, **k: self              # the reference moves `keep` to the GPU
    _load("mmdet.core.post_processing.bbox_nms", "mmdet/core/post_processing/bbox_nms.py")
    return _load("mmdet.models.single_stage_heads_ssd_rotate_head",
                 "mmdet/models/single_stage_heads/ssd_rotate_head.py")


class AttrDict(dict):
    __getattr__ = dict.__getitem__


def scene(seed, n_gt, dev="cpu"):
    """Anchor sub-grid (88 x 88 cells x 2 rotations, the reference's stride / sizes) + ground-truth cars."""
    g = torch.Generator().manual_seed(seed)
    xs = torch.arange(88) * 0.4 + 0.2
    ys = torch.arange(88) * 0.4 - 17.4
    yy, xx = torch.meshgrid(ys, xs, indexing="ij")
    a = torch.zeros(88, 88, 2, 7)
    a[..., 0], a[..., 1], a[..., 2] = xx[..., None], yy[..., None], -1.78
    a[..., 3], a[..., 4], a[..., 5] = 1.6, 3.9, 1.56
    a[..., 1, 6] = 1.57
    gt = torch.zeros(n_gt, 7)
    gt[:, 0] = torch.rand(n_gt, generator=g) * 30 + 2
    gt[:, 1] = torch.rand(n_gt, generator=g) * 30 - 15
    gt[:, 2] = -1.7 + torch.randn(n_gt, generator=g) * 0.1
    gt[:, 3] = 1.6 + torch.randn(n_gt, generator=g) * 0.1
    gt[:, 4] = 3.9 + torch.randn(n_gt, generator=g) * 0.3
    gt[:, 5] = 1.5 + torch.randn(n_gt, generator=g) * 0.1
    gt[:, 6] = (torch.rand(n_gt, generator=g) - 0.5) * 6.2
    mask = torch.rand(88 * 88 * 2, generator=g) > 0.3
    return a.view(-1, 7), mask, gt


def main():
    head = import_reference()
    losses = sys.modules["mmdet.core.loss.losses"]
    iou = sys.modules["mmdet.ops.iou3d.iou3d_utils"]
    tgt = sys.modules["mmdet.core.bbox3d.target_ops"]
    out = {}
    g = torch.Generator().manual_seed(0)

    # ---- plain losses -------------------------------------------------------------------------------------------
    pred = torch.randn(300, 3, generator=g)
    t01 = (torch.rand(300, 3, generator=g) > 0.8).float()
    w = torch.rand(300, 3, generator=g) * (torch.rand(300, 3, generator=g) > 0.2).float()
    out["fl_pred"], out["fl_target"], out["fl_weight"] = pred, t01, w
    out["fl_avg1"] = losses.weighted_sigmoid_focal_loss(pred, t01, w, avg_factor=1.)
    out["fl_default"] = losses.weighted_sigmoid_focal_loss(pred, t01, w)
    p7, t7 = torch.randn(200, 7, generator=g) * 0.3, torch.randn(200, 7, generator=g) * 0.3
    w7 = (torch.rand(200, 1, generator=g) > 0.5).float() / 17
    out["sl_pred"], out["sl_target"], out["sl_weight"] = p7, t7, w7
    out["sl_beta9"] = losses.weighted_smoothl1(p7, t7, w7, beta=1 / 9., avg_factor=1.)
    out["sl_default"] = losses.weighted_smoothl1(p7, t7, w7.expand(200, 7).contiguous())
    lg2, lab2, w2 = torch.randn(150, 2, generator=g), (torch.rand(150, generator=g) > 0.5).long(), \
        torch.rand(150, generator=g)
    out["ce_logits"], out["ce_labels"], out["ce_weight"] = lg2, lab2, w2
    out["ce_avg1"] = losses.weighted_cross_entropy(lg2, lab2, w2, avg_factor=1.)
    out["ce_default"] = losses.weighted_cross_entropy(lg2, lab2, w2)

    # ---- similarity, encode, target assignment --------------------------------------------------------------------
    anchors, amask, gt = scene(1, 9)
    out["anchors"], out["anchor_mask"], out["gt"] = anchors, amask, gt
    out["near_iou"] = iou.NearestIouSimilarity()(anchors[::37], gt)
    out["rot_iou3d"] = iou.RotateIou3dSimilarity()(anchors[::37].contiguous(), gt)
    out["encode"] = head.second_box_encode(gt[torch.arange(400) % 9], anchors[1000:1400])
    gt_cls = torch.ones(9, dtype=torch.int64)
    gmask = torch.tensor([1, 1, 0, 1, 1, 1, 0, 1, 1], dtype=torch.bool)
    for name, (am, gm, gb) in dict(masked=(amask, gmask, gt), nomask=(None, None, gt),
                                   nogt=(amask, None, gt[:0])).items():
        lab, tar, mx = tgt.create_target_torch(anchors, am, gb, gt_cls[:len(gb)], gm,
                                               similarity_fn=iou.NearestIouSimilarity(),
                                               box_encoding_fn=head.second_box_encode, matched_threshold=0.6,
                                               unmatched_threshold=0.45, box_code_size=7)
        out["ct_%s_labels" % name], out["ct_%s_targets" % name], out["ct_%s_max" % name] = lab, tar, mx

    # ---- SSDRotateHead.loss / get_guided_anchors (B = 2, one class) ----------------------------------------------
    rpn = head.SSDRotateHead(num_class=1, num_output_filters=8, num_anchor_per_loc=2, box_code_size=7)
    a2, m2, gt2 = scene(2, 6)
    anc = dict(Car=torch.stack([anchors, a2]))
    msk = dict(Car=torch.stack([amask, m2]))
    gtb, gtl = [gt, gt2], [torch.ones(9, dtype=torch.int64), torch.ones(6, dtype=torch.int64)]
    gtt = [np.array(["Car"] * 9), np.array(["Car"] * 5 + ["Van"])]
    box = (torch.randn(2, 1, 88, 88, 14, generator=g) * 0.2).requires_grad_()
    cls = (torch.randn(2, 1, 88, 88, 2, generator=g) - 2.5).requires_grad_()
    dr = torch.randn(2, 1, 88, 88, 4, generator=g).requires_grad_()
    cfg = AttrDict(assigner=AttrDict(Car=AttrDict(pos_iou_thr=0.6, neg_iou_thr=0.45, min_pos_iou=0.45),
                                     ignore_iof_thr=-1, similarity_fn="NearestIouSimilarity"), anchor_thr=0.1)
    ls = rpn.loss(box, cls, dr, gtb, gtl, gtt, anc, msk, cfg)
    total = ls["rpn_loc_loss"] + ls["rpn_cls_loss"] + ls["rpn_dir_loss"]
    gb, gc, gd = torch.autograd.grad(total.sum(), [box, cls, dr])
    out.update(a2=a2, m2=m2, gt2=gt2, rpn_box=box.detach(), rpn_cls=cls.detach(), rpn_dir=dr.detach(),
               rpn_loc_loss=ls["rpn_loc_loss"].detach(), rpn_cls_loss=ls["rpn_cls_loss"].detach(),
               rpn_dir_loss=ls["rpn_dir_loss"].detach(), rpn_gbox=gb, rpn_gcls=gc, rpn_gdir=gd)
    guided, glabels = rpn.get_guided_anchors(box.detach(), cls.detach(), dr.detach(), anc, msk, gtb, gtl, thr=0.1)
    for i in range(2):
        out["guided%d" % i], out["guided_labels%d" % i] = guided[i], glabels[i]
    # the inference call of the same method (single_stage.py:122: no ground truth, thr 0.1) -- pins oracle/nets.py
    tguided, tlabels = rpn.get_guided_anchors(box.detach(), cls.detach(), dr.detach(), anc, msk, None, None, thr=0.1)
    for i in range(2):
        out["test_guided%d" % i], out["test_guided_labels%d" % i] = tguided[i], tlabels[i]
    # three-class variant (multi_cfg): max over classes picks the label
    rpn3 = head.SSDRotateHead(num_class=3, num_output_filters=8, num_anchor_per_loc=2, box_code_size=7)
    box3 = torch.randn(1, 3, 20, 22, 14, generator=g) * 0.2
    cls3 = torch.randn(1, 3, 20, 22, 6, generator=g) - 1.5
    dr3 = torch.randn(1, 3, 20, 22, 4, generator=g)
    anc3 = torch.stack([anchors[:20 * 22 * 2], a2[:20 * 22 * 2], anchors[100:100 + 20 * 22 * 2]], 0).view(1, -1, 7)
    msk3 = (torch.rand(1, 3 * 20 * 22 * 2, generator=g) > 0.3)
    g3, l3 = rpn3.get_guided_anchors(box3, cls3, dr3, anc3, msk3, None, None, thr=0.1)
    out.update(mc_box=box3, mc_cls=cls3, mc_dir=dr3, mc_anchors=anc3, mc_mask=msk3, mc_guided=g3[0], mc_labels=l3[0])
    # PSWarpHead.get_rescore_bboxes (ssd_rotate_head.py:487-533): sigmoid, score threshold, BEV boxes, rotated NMS
    ext0 = head.PSWarpHead(grid_offsets=(0., 40.), featmap_stride=.4, in_channels=8, num_class=1, num_parts=28)
    rs_boxes = [tguided[0][:900].contiguous(), tguided[1][:700].contiguous(), tguided[0][:0]]
    rs_logits = [torch.randn(900, generator=g) * 1.5, torch.randn(700, generator=g) * 1.5 - 6.0, torch.randn(0)]
    rs_labels = [tlabels[0][:900], tlabels[1][:700], tlabels[0][:0]]
    db, ds, dl = ext0.get_rescore_bboxes(rs_boxes, rs_logits, rs_labels, [None] * 3,
                                         AttrDict(score_thr=0.3, nms=AttrDict(iou_thr=0.1)))
    assert db[1] is None and db[2] is None and db[0] is not None       # sample 1: nothing passes 0.3; sample 2: empty
    out.update(rs_boxes0=rs_boxes[0], rs_logits0=rs_logits[0], rs_labels0=rs_labels[0], rs_logits1=rs_logits[1],
               rs_det_boxes=db[0], rs_det_scores=ds[0], rs_det_labels=dl[0])

    # ---- PSWarpHead.loss (rotated 3-D IoU assignment) -------------------------------------------------------------
    ext = head.PSWarpHead(grid_offsets=(0., 40.), featmap_stride=.4, in_channels=8, num_class=1, num_parts=28)
    gsub = [guided[0][:600].contiguous(), guided[1][:500].contiguous()]
    score = torch.randn(1100, generator=g).requires_grad_()
    cfg2 = AttrDict(assigner=AttrDict(pos_iou_thr=0.7, neg_iou_thr=0.7, min_pos_iou=0.7, ignore_iof_thr=-1,
                                      similarity_fn="RotateIou3dSimilarity"))
    l2 = ext.loss(score, gtb, gtl, gsub, cfg2)["loss_cls"]
    out.update(ext_score=score.detach(), ext_loss=l2.detach(), ext_gscore=torch.autograd.grad(l2.sum(), score)[0])

    # ---- SpMiddleFHD.aux_loss arithmetic ----------------------------------------------------------------------------
    fake = types.SimpleNamespace()
    n = 500
    plab = (torch.rand(n, generator=g) > 0.7).to(torch.uint8)
    poff = torch.randn(n, 3, generator=g) * plab[:, None].float()
    fake.build_aux_target = lambda pts, gtbx: (plab, poff)
    gl = {"torch": torch, "weighted_smoothl1": losses.weighted_smoothl1,
          "weighted_sigmoid_focal_loss": losses.weighted_sigmoid_focal_loss}
    _functions("mmdet/models/necks/cmn.py", {"aux_loss"}, gl)
    pc, pr = torch.randn(n, 1, generator=g), torch.randn(n, 3, generator=g)
    al = gl["aux_loss"](fake, None, pc, pr, [None, None])
    out.update(aux_labels=plab, aux_offsets=poff, aux_cls=pc, aux_reg=pr, aux_loss_cls=al["aux_loss_cls"],
               aux_loss_reg=al["aux_loss_reg"])

    np.savez_compressed(os.path.join(HERE, "train_fns.npz"),
                        **{k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
    print("train_fns.npz:", len(out), "arrays; rpn losses", ls["rpn_loc_loss"].item(), ls["rpn_cls_loss"].item(),
          ls["rpn_dir_loss"].item(), "ext", l2.item(), "guided", [len(x) for x in guided],
          "positives", int((out["ct_masked_labels"] > 0).sum()))


if __name__ == "__main__":
    main()
