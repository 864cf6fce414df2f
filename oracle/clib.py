"""ctypes front-end of oracle/sassd_oracle.c and oracle/_ref (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os

import numpy as np

from . import build as _b

_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_b.build_oracle())
        _lib.orc_points_to_voxel.restype = C.c_int
        _lib.orc_box_overlap.restype = C.c_float
        _lib.orc_iou_bev.restype = C.c_float
        _lib.orc_nms_rotated.restype = C.c_int
    return _lib


def ref():
    """The reference's own iou3d device functions built for the host, or None when unavailable."""
    global _ref
    if _ref is None:
        p = _b.build_ref()
        if p is None or not os.path.exists(p):
            return None
        _ref = C.CDLL(p)
        _ref.ref_box_overlap.restype = C.c_float
        _ref.ref_iou_bev.restype = C.c_float
    return _ref


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


def points_to_voxel(points, voxel_size, coors_range, max_points=35, reverse_index=True, max_voxels=20000):
    """Same signature/returns as mmdet/ops/points_op/points_ops.py:104 (reverse_index=True only)."""
    assert reverse_index, "only the zyx kernel (the one the configs use) is restated"
    points = np.ascontiguousarray(points, np.float32)
    n, ndim = points.shape
    vs = np.asarray(voxel_size, np.float32)
    cr = np.asarray(coors_range, np.float32)
    voxels = np.empty((max_voxels, max_points, ndim), np.float32)
    coors = np.empty((max_voxels, 3), np.int32)
    num = np.empty((max_voxels,), np.int32)
    m = lib().orc_points_to_voxel(_p(points), n, ndim, _p(vs), _p(cr), int(max_points), int(max_voxels),
                                  _p(voxels), _p(coors), _p(num))
    assert m >= 0
    return voxels[:m], coors[:m], num[:m]


def voxel_mean(voxels, num_points, nfeat=4):
    voxels = np.ascontiguousarray(voxels, np.float32)
    num_points = np.ascontiguousarray(num_points, np.int32)
    m, t, ndim = voxels.shape
    out = np.empty((m, nfeat), np.float32)
    lib().orc_voxel_mean(_p(voxels), _p(num_points), m, t, ndim, nfeat, _p(out))
    return out


def boxes_overlap_bev(a, b, use_ref=False):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.empty((len(a), len(b)), np.float32)
    if use_ref:
        r = ref()
        for i in range(len(a)):
            for j in range(len(b)):
                out[i, j] = r.ref_box_overlap(_p(a[i:i + 1]), _p(b[j:j + 1]))
    else:
        lib().orc_boxes_overlap_bev(_p(a), len(a), _p(b), len(b), _p(out))
    return out


def boxes_iou_bev(a, b, use_ref=False):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.empty((len(a), len(b)), np.float32)
    if use_ref:
        r = ref()
        for i in range(len(a)):
            for j in range(len(b)):
                out[i, j] = r.ref_iou_bev(_p(a[i:i + 1]), _p(b[j:j + 1]))
    else:
        lib().orc_boxes_iou_bev(_p(a), len(a), _p(b), len(b), _p(out))
    return out


def nms_rotated(boxes_sorted, thr, return_mask=False):
    """iou3d.cpp:73-120 on boxes already sorted by descending score. Returns kept indices (int64)."""
    b = np.ascontiguousarray(boxes_sorted, np.float32)
    n = len(b)
    keep = np.empty((max(n, 1),), np.int64)
    cb = (n + 63) // 64
    mask = np.zeros((max(n, 1), max(cb, 1)), np.uint64)
    k = lib().orc_nms_rotated(_p(b), n, C.c_float(thr), _p(keep), _p(mask))
    return (keep[:k], mask[:n, :cb]) if return_mask else keep[:k]


def three_nn(unknown, known):
    u = np.ascontiguousarray(unknown, np.float32)
    k = np.ascontiguousarray(known, np.float32)
    d = np.empty((len(u), 3), np.float32)
    i = np.empty((len(u), 3), np.int32)
    lib().orc_three_nn(len(u), len(k), _p(u), _p(k), _p(d), _p(i))
    return d, i


def three_interpolate(points, idx, weight):
    pts = np.ascontiguousarray(points, np.float32)
    idx = np.ascontiguousarray(idx, np.int32)
    w = np.ascontiguousarray(weight, np.float32)
    out = np.empty((len(idx), pts.shape[1]), np.float32)
    lib().orc_three_interpolate(pts.shape[1], len(pts), len(idx), _p(pts), _p(idx), _p(w), _p(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    g = np.ascontiguousarray(grad_out, np.float32)
    idx = np.ascontiguousarray(idx, np.int32)
    w = np.ascontiguousarray(weight, np.float32)
    gp = np.zeros((m, g.shape[1]), np.float32)
    lib().orc_three_interpolate_grad(g.shape[1], len(g), m, _p(g), _p(idx), _p(w), _p(gp))
    return gp


def pts_in_boxes3d(pts, boxes):
    p = np.ascontiguousarray(pts, np.float32)
    b = np.ascontiguousarray(boxes, np.float32)
    flag = np.zeros((len(b), len(p)), np.int32)
    reg = np.zeros((len(p), 3), np.float32)
    lib().orc_pts_in_boxes3d(_p(p), len(p), _p(b), len(b), _p(flag), _p(reg))
    return flag, reg


def rotate_iou_eval(boxes, query_boxes, criterion=-1):
    b = np.ascontiguousarray(boxes, np.float32)
    q = np.ascontiguousarray(query_boxes, np.float32)
    out = np.zeros((len(b), len(q)), np.float32)
    lib().orc_rotate_iou_eval(_p(b), len(b), _p(q), len(q), int(criterion), _p(out))
    return out
