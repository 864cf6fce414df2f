"""Anchor helpers: AnchorGeneratorStride (mmdet/core/anchor/anchor3d_generator.py:3-41,81-103) and the
near-bbox conversion used for anchors_bv (mmdet/core/bbox3d/geometry.py:401-426).  Host-side numpy: anchors are
computed ONCE per dataset (kitti.py:81-88), not per frame."""
import numpy as np


def create_anchors_3d_stride(feature_size, sizes=(1.6, 3.9, 1.56), anchor_strides=(0.4, 0.4, 0.0),
                             anchor_offsets=(0.2, -39.8, -1.78), rotations=(0, np.pi / 2), dtype=np.float32):
    """feature_size [D,H,W] -> anchors [D,H,W,num_sizes,num_rots,7] = (x,y,z,w,l,h,r)."""
    d, h, w = [int(v) for v in feature_size]
    xs = np.arange(w, dtype=dtype) * dtype(anchor_strides[0]) + dtype(anchor_offsets[0])
    ys = np.arange(h, dtype=dtype) * dtype(anchor_strides[1]) + dtype(anchor_offsets[1])
    zs = np.arange(d, dtype=dtype) * dtype(anchor_strides[2]) + dtype(anchor_offsets[2])
    sz = np.asarray(sizes, dtype=dtype).reshape(-1, 3)
    rot = np.asarray(rotations, dtype=dtype)
    out = np.empty((d, h, w, sz.shape[0], rot.shape[0], 7), dtype=dtype)
    out[..., 0] = xs[None, None, :, None, None]
    out[..., 1] = ys[None, :, None, None, None]
    out[..., 2] = zs[:, None, None, None, None]
    out[..., 3:6] = sz[None, None, None, :, None, :]
    out[..., 6] = rot[None, None, None, None, :]
    return out


class AnchorGeneratorStride:
    def __init__(self, sizes=(1.6, 3.9, 1.56), anchor_strides=(0.4, 0.4, 1.0), anchor_offsets=(0.2, -39.8, -1.78),
                 rotations=(0, np.pi / 2), dtype=np.float32):
        self._sizes, self._anchor_strides = sizes, anchor_strides
        self._anchor_offsets, self._rotations, self._dtype = anchor_offsets, rotations, dtype

    @property
    def num_anchors_per_localization(self):
        return len(self._rotations) * np.array(self._sizes).reshape([-1, 3]).shape[0]

    def __call__(self, feature_map_size):
        return create_anchors_3d_stride(feature_map_size, self._sizes, self._anchor_strides, self._anchor_offsets,
                                        self._rotations, self._dtype)


def create_anchors_3d_range(feature_size, anchor_range, sizes=(1.6, 3.9, 1.56), rotations=(0, np.pi / 2),
                            dtype=np.float32):
    """feature_size [D,H,W], anchor_range (x0, y0, z0, x1, y1, z1): centres on np.linspace between the range ends
    -> anchors [D,H,W,num_sizes,num_rots,7] (anchor3d_generator.py:44-79; unused by the shipped configs)."""
    d, h, w = [int(v) for v in feature_size]
    rg = np.array(anchor_range, dtype)
    xs, ys, zs = (np.linspace(rg[a], rg[a + 3], n, dtype=dtype) for a, n in ((0, w), (1, h), (2, d)))
    sz = np.asarray(sizes, dtype=dtype).reshape(-1, 3)
    rot = np.asarray(rotations, dtype=dtype)
    out = np.empty((d, h, w, sz.shape[0], rot.shape[0], 7), dtype=dtype)
    out[..., 0] = xs[None, None, :, None, None]
    out[..., 1] = ys[None, :, None, None, None]
    out[..., 2] = zs[:, None, None, None, None]
    out[..., 3:6] = sz[None, None, None, :, None, :]
    out[..., 6] = rot[None, None, None, None, :]
    return out


class AnchorGeneratorRange:
    def __init__(self, anchor_ranges, sizes=(1.6, 3.9, 1.56), rotations=(0, np.pi / 2), dtype=np.float32):
        self._anchor_ranges, self._sizes, self._rotations, self._dtype = anchor_ranges, sizes, rotations, dtype

    @property
    def num_anchors_per_localization(self):
        return len(self._rotations) * np.array(self._sizes).reshape([-1, 3]).shape[0]

    def __call__(self, feature_map_size):
        return create_anchors_3d_range(feature_map_size, self._anchor_ranges, self._sizes, self._rotations, self._dtype)


def limit_period(val, offset=0.5, period=np.pi):
    return val - np.floor(val / period + offset) * period


def rbbox2d_to_near_bbox(rbboxes):
    """[N,5] (x,y,xdim,ydim,rad) -> nearest axis-aligned [N,4] (xmin,ymin,xmax,ymax)."""
    rots = rbboxes[..., -1]
    swap = (np.abs(limit_period(rots, 0.5, np.pi)) > np.pi / 4)[..., np.newaxis]
    cen = np.where(swap, rbboxes[:, [0, 1, 3, 2]], rbboxes[:, :4])
    # (fancy column indexing above leaves Fortran-ordered arrays behind: hand out a C-contiguous [N,4])
    return np.ascontiguousarray(np.concatenate([cen[:, :2] - cen[:, 2:] / 2, cen[:, :2] + cen[:, 2:] / 2], axis=-1))
