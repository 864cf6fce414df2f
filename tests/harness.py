"""ctypes access to oracle/_build/libharness.so (TEST INFRASTRUCTURE): the product's __host__ __device__ per-point
functions (sa-ssd_amd/csrc/augment_core.h) looped on the CPU, plus drop-in replacements for the four kernel entry points
of sassd.geometry / sassd.point_augmentor that work on CPU tensors -- so whole augmentation frames can be checked against
the reference-generated vectors without a GPU.  The GPU tests run the real kernels against the same vectors."""
import ctypes as C

import numpy as np
import torch

from oracle import build as ob

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(ob.build_harness())
    return _lib


def _p(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr() if torch.is_tensor(t) else t.ctypes.data)


def points_in_polytopes(points, planes, f32_math, device=None):
    was_numpy = not torch.is_tensor(points)
    pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)) if was_numpy else points
    assert pts.dtype == torch.float32 and pts.stride(-1) == 1 and not pts.is_cuda
    planes = np.ascontiguousarray(planes, dtype=np.float64)
    n, m = pts.shape[0], planes.shape[0]
    mask = torch.zeros((n, m), dtype=torch.uint8)
    if n and m:
        lib().hst_points_in_polytopes(_p(pts), n, pts.stride(0), _p(planes), m, int(f32_math), _p(mask))
    mask = mask.bool()
    return mask.numpy() if was_numpy else mask


def paste_objects(db, src_start, out_start, n_obj, n_out, shift, lower, out):
    lib().hst_paste_objects(_p(db), _p(src_start), _p(out_start), n_obj, C.c_int64(n_out), _p(shift), _p(lower), _p(out))


def points_transform(points, mask8, valid, centers, rot_sin, rot_cos, loc):
    lib().hst_points_transform(_p(points), points.shape[0], points.stride(0), _p(mask8), mask8.shape[1], _p(valid),
                               _p(centers), _p(rot_sin), _p(rot_cos), _p(loc))


def points_global(points, flip, rot_sin, rot_cos, scale):
    lib().hst_global_transform(_p(points), points.shape[0], points.stride(0), int(flip), C.c_float(float(rot_sin)),
                               C.c_float(float(rot_cos)), C.c_float(float(scale)))


def patch(monkeypatch):
    """Route the four kernel entry points to the CPU harness (tests only)."""
    from sassd import geometry, point_augmentor
    monkeypatch.setattr(geometry, "points_in_polytopes", points_in_polytopes)
    monkeypatch.setattr(point_augmentor, "_paste_objects", paste_objects)
    monkeypatch.setattr(point_augmentor, "_points_transform", points_transform)
    monkeypatch.setattr(point_augmentor, "_points_global", points_global)
