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
  autograd.py, train_ops.py, train.py, pointnet2_utils.py   the training step
  kitti_common.py  label / result / calibration formats, kitti_bbox2results, per-frame info records (tools/kitti_common.py)
  kitti_eval.py, eval_ops.py   KITTI AP evaluation: GPU overlap matrices + native host matching (core/evaluation)
  geometry.py      box geometry: numpy for boxes, HIP for points, native host collision test (core/bbox3d/geometry.py)
  point_augmentor.py  GT sampling / per-object noise / global transforms, database resident in HBM (core/point_cloud)
  create_data.py   infos / velodyne_reduced / gt_database preparation in the reference's formats (tools/create_data.py)
  kitti_dataset.py, loader.py, runner.py   KittiLiDAR + get_dataset, samplers + prefetching loader, epoch / test loops
"""
from . import synth  # noqa: F401

__version__ = "0.1.0"
