"""sassd -- MI355X-native (gfx950) implementation of the SA-SSD hot path
(voxelize -> sparse 3-D conv backbone -> BEV dense head -> part-sensitive warp -> rotated IoU/NMS).

Layout:
  csrc/            hand-written HIP kernels + the C ABI (include/sassd.h) -> lib/libsassd.so
  _C.py            ctypes binding of the C ABI (fails loudly when the library is missing)
  points_ops.py    mirror of mmdet.ops.points_op        (points_to_voxel)
  voxel_generator.py  mirror of mmdet.core.point_cloud.voxel_generator (VoxelGenerator)
  spconv.py        mirror of the spconv v1.0 surface the reference uses (SparseConvTensor, SubMConv3d, ...)
  iou3d_utils.py   mirror of mmdet.ops.iou3d.iou3d_utils
  anchors.py       AnchorGeneratorStride / near-bbox helpers (mmdet.core.anchor, core.bbox3d.geometry)
  detector.py      SimpleVoxel / SpMiddleFHD / SSDRotateHead / PSWarpHead / SingleStageDetector + build_detector
  config.py        mmcv-free Config.fromfile / obj_from_dict
  pipeline.py      device-resident whole-frame inference plan (raw points -> detections, no host syncs inside)
  synth.py         synthetic KITTI-range / Waymo-scale clouds (measurement contract)
"""
from . import synth  # noqa: F401

__version__ = "0.1.0"
