"""Mirror of mmdet/ops/points_op/points_ops.py (reference :104-164) on the HIP voxelizer.

Same name, arguments and return types (numpy in, numpy out) so `VoxelGenerator.generate` and
`KittiLiDAR.prepare_*_img` (kitti.py:212,317) work unchanged.  The arithmetic runs in sassd_voxelize on the
GPU; the device-resident variant (no D2H) is `points_to_voxel_device`.
"""
import numpy as np
import torch

from . import kernels as K


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("sassd.points_ops needs an MI355X (no CPU fallback; the CPU oracle lives in oracle/)")
    return torch.device("cuda", torch.cuda.current_device())


def points_to_voxel_device(points, voxel_size, coors_range, max_points=35, reverse_index=True, max_voxels=20000,
                           **kw):
    """points: torch CUDA [N, ndim] f32.  Returns the dict of kernels.voxelize (capacity-sized device tensors and
    a device int32 `voxel_num`)."""
    r = K.voxelize(points, voxel_size, coors_range, max_points, max_voxels, **kw)
    if not reverse_index:                      # points_ops.py:53-101: the same voxels, coordinates in (x, y, z) order
        c = r["coors"]
        r["coors"] = torch.cat([c[:, :-3], c[:, -3:].flip(1)], 1).contiguous()
    return r


def points_to_voxel(points, voxel_size, coors_range, max_points=35, reverse_index=True, max_voxels=20000):
    """Reference signature (points_ops.py:104): numpy [N, >=3] -> (voxels [M,T,ndim], coors [M,3] int32 -- zyx, or
    xyz with reverse_index=False --, num_points_per_voxel [M] int32).  max_points up to 64 (reference default 35)."""
    pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(_device())
    r = points_to_voxel_device(pts, voxel_size, coors_range, max_points, reverse_index, max_voxels,
                               want_mean=False)
    m = int(r["voxel_num"].item())
    return (r["voxels"][:m].cpu().numpy(), r["coors"][:m].cpu().numpy(), r["num_points"][:m].cpu().numpy())


def pts_in_boxes3d(pts, boxes3d):
    """mmdet/ops/points_op/__init__.py:5-11 (points_op.cpp:107-144) on the device: pts (N,3), boxes3d (M,7) ->
    (pts_in_flag [M,N] int32, reg_target [N,3]).  Inputs may be CPU or device tensors; outputs live where `pts` does."""
    dev = pts.device
    p = pts.contiguous().float().to(_device())
    b = boxes3d.contiguous().float().to(_device())
    flag, reg = K.pts_in_boxes3d(p, b)
    return flag.to(dev), reg.to(dev)
