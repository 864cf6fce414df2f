"""Host-side (torch) mirrors of the reference's training-only glue for the hot path.  These are small
elementwise / indexing computations that the reference also runs as plain torch ops; the heavy kernels they call
(rotated overlap, point-in-box, 3-NN, sparse / dense conv backward) are HIP (kernels.py).

  losses                 mmdet/core/loss/losses.py:13-23, 54-62, 92-96 (+ sigmoid_focal_loss :35-52, smooth_l1 :72-89)
  second_box_encode/decode  mmdet/models/single_stage_heads/ssd_rotate_head.py:15-91
  NearestIouSimilarity   mmdet/ops/iou3d/iou3d_utils.py:9-45,163-183
  create_target_torch    mmdet/core/bbox3d/target_ops.py:139-277
  one_hot / multi_apply  mmdet/models/utils/__init__.py:56-60, mmdet/core/utils/misc.py:21
"""
import math

import torch
import torch.nn.functional as F


def multi_apply(func, *args, **kwargs):
    results = [func(*a, **kwargs) for a in zip(*args)]
    return tuple(map(list, zip(*results)))


def one_hot(t, depth, dim=-1, on_value=1.0, dtype=torch.float32):
    out = torch.zeros(*t.shape, depth, dtype=dtype, device=t.device)
    return out.scatter_(dim, t.unsqueeze(dim).long(), on_value)


# ---- losses ------------------------------------------------------------------------------------------------------
def sigmoid_focal_loss_sum(pred, target, weight, gamma=2.0, alpha=0.25):
    p = pred.sigmoid()
    target = target.type_as(pred)
    pt = (1 - p) * target + p * (1 - target)
    w = (alpha * target + (1 - alpha) * (1 - target)) * weight * pt.pow(gamma)
    return (F.binary_cross_entropy_with_logits(pred, target, reduction='none') * w).sum()


def weighted_sigmoid_focal_loss(pred, target, weight, gamma=2.0, alpha=0.25, avg_factor=None, num_classes=80):
    if avg_factor is None:
        avg_factor = torch.sum(weight > 0).float().item() / num_classes + 1e-6
    return sigmoid_focal_loss_sum(pred, target, weight, gamma, alpha)[None] / avg_factor


def weighted_smoothl1(pred, target, weight, beta=1.0, avg_factor=None):
    if avg_factor is None:
        avg_factor = torch.sum(weight > 0).float().item() / 4 + 1e-6
    d = torch.abs(pred - target)
    loss = torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta)
    return torch.sum(loss * weight)[None] / avg_factor


def weighted_cross_entropy(pred, label, weight, avg_factor=None):
    if avg_factor is None:
        avg_factor = max(torch.sum(weight > 0).float().item(), 1.)
    return torch.sum(F.cross_entropy(pred, label, reduction='none') * weight)[None] / avg_factor


# ---- box coding ----------------------------------------------------------------------------------------------------
def second_box_encode(boxes, anchors):
    """(x,y,z,w,l,h,r) ground truth vs anchors -> regression targets (log sizes, plain angle difference)."""
    xa, ya, za, wa, la, ha, ra = torch.split(anchors, 1, dim=-1)
    xg, yg, zg, wg, lg, hg, rg = torch.split(boxes, 1, dim=-1)
    zg = zg + hg / 2
    za = za + ha / 2
    diag = torch.sqrt(la ** 2 + wa ** 2)
    return torch.cat([(xg - xa) / diag, (yg - ya) / diag, (zg - za) / ha, torch.log(wg / wa), torch.log(lg / la),
                      torch.log(hg / ha), rg - ra], dim=-1)


def second_box_decode(enc, anchors):
    """ssd_rotate_head.py:52-91 (plain residual coding, exp sizes, angle sum)."""
    xa, ya, za, wa, la, ha, ra = torch.split(anchors, 1, dim=-1)
    xt, yt, zt, wt, lt, ht, rt = torch.split(enc, 1, dim=-1)
    za = za + ha / 2
    diag = torch.sqrt(la ** 2 + wa ** 2)
    xg, yg, zg = xt * diag + xa, yt * diag + ya, zt * ha + za
    lg, wg, hg = torch.exp(lt) * la, torch.exp(wt) * wa, torch.exp(ht) * ha
    return torch.cat([xg, yg, zg - hg / 2, wg, lg, hg, rt + ra], dim=-1)


# ---- similarity ------------------------------------------------------------------------------------------------------
def _limit_period(val, offset=0.5, period=math.pi):
    return val - torch.floor(val / period + offset) * period


def boxes3d_to_near_torch(boxes3d):
    """rotated (x,y,w,l,r) -> nearest axis-aligned (xmin,ymin,xmax,ymax)."""
    xy, w, l = boxes3d[:, 0:2], boxes3d[:, 3:4], boxes3d[:, 4:5]          # column slices: no index-tensor uploads
    swap = (torch.abs(_limit_period(boxes3d[:, 6], 0.5, math.pi)) > math.pi / 4)[..., None]
    cen = torch.cat([xy, torch.where(swap, l, w), torch.where(swap, w, l)], dim=1)
    return torch.cat([cen[:, :2] - cen[:, 2:] / 2, cen[:, :2] + cen[:, 2:] / 2], dim=-1)


def boxes_iou(b1, b2, eps=0.0):
    rows, cols = b1.size(0), b2.size(0)
    if rows * cols == 0:
        return b1.new(rows, cols)
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt + eps).clamp(min=0)
    ov = wh[:, :, 0] * wh[:, :, 1]
    a1 = (b1[:, 2] - b1[:, 0] + eps) * (b1[:, 3] - b1[:, 1] + eps)
    a2 = (b2[:, 2] - b2[:, 0] + eps) * (b2[:, 3] - b2[:, 1] + eps)
    return ov / (a1[:, None] + a2 - ov)


class NearestIouSimilarity:
    def __call__(self, boxes1, boxes2):
        return boxes_iou(boxes3d_to_near_torch(boxes1), boxes3d_to_near_torch(boxes2))


# ---- anchor <-> ground-truth assignment --------------------------------------------------------------------------
def create_target_torch(all_anchors, anchor_mask, gt_boxes, gt_classes, gt_mask, similarity_fn, box_encoding_fn,
                        matched_threshold=0.6, unmatched_threshold=0.45, box_code_size=7):
    """Labels (1.. positive class, 0 negative, -1 ignore), regression targets and best overlap per anchor, with the
    reference's semantics (target_ops.py:139-277) but FIXED-SHAPE: anchors outside `anchor_mask` and ground truths
    outside `gt_mask` are neutralised by masking the overlap matrix instead of being compacted away with boolean
    indexing, so the whole assignment is a handful of dense kernels over [A, G] with no host synchronisation.
    (The reference's optional positive/negative subsampling branch, never enabled by the configs, is omitted; the
    third return value is full-size [A] with -1 at masked-out anchors, the reference returns the masked rows.)"""
    total = all_anchors.shape[0]
    dev = all_anchors.device
    if gt_classes is None:
        gt_classes = torch.ones([gt_boxes.shape[0]], dtype=torch.int64, device=dev)
    row_ok = anchor_mask.bool() if anchor_mask is not None else torch.ones(total, dtype=torch.bool, device=dev)
    labels = torch.full((total,), -1, dtype=torch.int64, device=dev)
    targets = torch.zeros((total, box_code_size), dtype=all_anchors.dtype, device=dev)
    if gt_boxes.shape[0] == 0 or total == 0:
        labels = torch.where(row_ok, torch.zeros_like(labels), labels)
        return labels, targets, torch.where(row_ok, 0.0, -1.0).type_as(all_anchors)
    col_ok = gt_mask.bool() if gt_mask is not None else torch.ones(gt_boxes.shape[0], dtype=torch.bool, device=dev)
    ok = row_ok[:, None] & col_ok[None, :]
    ov = torch.where(ok, similarity_fn(all_anchors, gt_boxes), torch.full((), -1.0, device=dev))     # [A, G]
    a2g_max, a2g_arg = ov.max(dim=1)
    g2a_max = ov.max(dim=0)[0]
    forced = ((ov == g2a_max[None, :]) & ok & (g2a_max > 0)[None, :]).any(dim=1)   # anchors tying a gt's best overlap
    cls_of_arg = gt_classes[a2g_arg]
    labels = torch.where(a2g_max >= matched_threshold, cls_of_arg, labels)
    labels = torch.where(a2g_max < unmatched_threshold, torch.zeros_like(labels), labels)
    labels = torch.where(forced, cls_of_arg, labels)
    labels = torch.where(row_ok, labels, torch.full_like(labels, -1))
    fg = labels > 0
    enc = box_encoding_fn(gt_boxes[a2g_arg], all_anchors)
    targets = torch.where(fg[:, None], enc, targets)
    return labels, targets, torch.where(row_ok, a2g_max, torch.full_like(a2g_max, -1.0))
