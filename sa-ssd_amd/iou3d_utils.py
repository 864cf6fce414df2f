"""Mirror of mmdet/ops/iou3d/iou3d_utils.py (reference :47-183) and core/post_processing/bbox_nms.py:4-26 on the
HIP kernels (heads.hip).  Same function names / argument meaning; tensors stay on the device (the reference
round-trips `keep` through a CPU LongTensor, iou3d_utils.py:126-128)."""
import math

import torch

from . import kernels as K


def limit_period(val, offset=0.5, period=math.pi):
    return val - torch.floor(val / period + offset) * period


def boxes3d_to_bev_torch(boxes3d):
    """(N,7) [x,y,z,w,l,h,ry] -> (N,5) [x1,y1,x2,y2,ry] using cols 0,1,3,4,6 (reference :47-60)."""
    out = boxes3d.new_empty((boxes3d.shape[0], 5))
    hx, hy = boxes3d[:, 3] / 2, boxes3d[:, 4] / 2
    out[:, 0], out[:, 1] = boxes3d[:, 0] - hx, boxes3d[:, 1] - hy
    out[:, 2], out[:, 3] = boxes3d[:, 0] + hx, boxes3d[:, 1] + hy
    out[:, 4] = boxes3d[:, 6]
    return out


def boxes_iou_bev(boxes_a, boxes_b):
    return K.boxes_iou_bev(boxes3d_to_bev_torch(boxes_a).contiguous(), boxes3d_to_bev_torch(boxes_b).contiguous())


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N,7),(M,7) -> 3-D IoU [N,M] (reference :79-111)."""
    ov = K.boxes_overlap_bev(boxes3d_to_bev_torch(boxes_a).contiguous(), boxes3d_to_bev_torch(boxes_b).contiguous())
    a_max, a_min = (boxes_a[:, 2] + boxes_a[:, 5]).view(-1, 1), boxes_a[:, 2].view(-1, 1)
    b_max, b_min = (boxes_b[:, 2] + boxes_b[:, 5]).view(1, -1), boxes_b[:, 2].view(1, -1)
    oh = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    o3 = ov * oh
    va = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vb = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return o3 / torch.clamp(va + vb - o3, min=1e-7)


def nms_gpu(boxes, scores, thresh):
    """(N,5) bev boxes, (N) scores -> indices of kept boxes, descending score (reference :114-128)."""
    order = scores.sort(dim=0, descending=True, stable=True)[1]
    keep, num = K.nms_gpu(boxes[order].contiguous(), thresh)
    return order[keep[:int(num.item())]].contiguous()


def nms_normal_gpu(boxes, scores, thresh):
    """(N,5) bev boxes, (N) scores -> kept indices by descending score, axis-aligned IoU (reference :130-144)."""
    order = scores.sort(dim=0, descending=True, stable=True)[1]
    keep, num = K.nms_gpu(boxes[order].contiguous(), thresh, normal=True)
    return order[keep[:int(num.item())]].contiguous()


def rotate_nms_torch(rbboxes, scores, pre_max_size=None, post_max_size=None, iou_threshold=0.5):
    """bbox_nms.py:4-26."""
    indices = None
    if pre_max_size is not None:
        scores, indices = torch.topk(scores, k=min(scores.shape[0], pre_max_size))
        rbboxes = rbboxes[indices]
    if len(rbboxes) == 0:
        return None
    keep = nms_gpu(rbboxes, scores, iou_threshold)[:post_max_size]
    if keep.shape[0] == 0:
        return None
    return indices[keep] if indices is not None else keep


class RotateIou2dSimilarity:
    def __call__(self, boxes1, boxes2):
        return boxes_iou_bev(boxes1, boxes2)


class RotateIou3dSimilarity:
    def __call__(self, boxes1, boxes2):
        return boxes_iou3d_gpu(boxes1, boxes2)
