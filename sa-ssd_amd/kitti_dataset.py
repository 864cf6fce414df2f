"""The dataset side of the hot path: `KittiLiDAR` / `get_dataset` with the constructor arguments of
mmdet/datasets/kitti.py:18-117 and mmdet/datasets/utils.py:80-130, reading the same on-disk tree (velodyne_reduced/,
label_2/, calib/, image_2/, planes/, ImageSets/*.txt), re-designed around one GPU per process:

  reference (per DataLoader worker, CPU)                          here
  ------------------------------------------------------------    -----------------------------------------------------
  mmcv.imread + ImageTransform, only img_shape is ever used       24 bytes of the PNG header (kitti_common.png_shape)
  read_lidar -> numpy                                             read_lidar -> ONE upload of the raw sweep
  PointAugmentor numba loops (train)                              PointAugmentor.augment_frame: HIP kernels, database in HBM
  points_to_voxel numba, 360 MB dense lookup per frame            sassd_voxelize on the device-resident cloud
  sparse_sum_for_anchors_mask + cumsum + fused_get_anchors_area   sassd_anchor_mask
  collate -> pinned copies of voxels / masks to the GPU           nothing to copy: `collate` only lists device tensors

`prepare_train_img` / `prepare_test_img` return the frame as device tensors (points after augmentation, labels, meta);
`collate(samples)` turns a list of them into the keyword arguments of `SingleStageDetector.forward` -- voxels,
coordinates, num_points, anchors, anchors_mask, gt_* and img_meta with the reference's names and per-sample list layout."""
import copy
import os.path as osp
import sys

import numpy as np
import torch

from . import anchors as anchor3d_generator
from . import geometry as G
from . import kernels as K
from . import point_augmentor
from . import voxel_generator
from .config import obj_from_dict
from .kitti_common import Calibration, limit_period, png_shape, project_rect_to_velo, read_lidar


def read_label_boxes(label_path):
    """label_2/%06d.txt -> (boxes [n,7] float32 = camera-frame (x, y, z, w, l, h, ry) as Object3d.box3d, types list),
    DontCare lines dropped (mmdet/datasets/kitti_utils.py:6-36,149-152; kitti.py:147-150)."""
    boxes, types = [], []
    with open(label_path) as f:
        for line in f:
            d = line.rstrip().split(' ')
            if len(d) < 15 or d[0] == 'DontCare':
                continue
            v = [float(x) for x in d[1:15]]
            boxes.append([v[10], v[11], v[12], v[8], v[9], v[7], v[13]])
            types.append(d[0])
    return np.array(boxes, dtype=np.float32).reshape(-1, 7), types


def get_road_plane(plane_file):
    """planes/%06d.txt line 4 -> unit (a, b, c, d) of the ground plane in the rect camera frame, normal pointing up."""
    with open(plane_file, 'r') as f:
        plane = np.asarray([float(v) for v in f.readlines()[3].split()])
    if plane[1] > 0:
        plane = -plane
    return plane / np.linalg.norm(plane[0:3])


class KittiLiDAR:
    def __init__(self, root, ann_file, img_prefix=None, img_norm_cfg=None, img_scale=(1242, 375), size_divisor=32,
                 proposal_file=None, flip_ratio=0.5, with_point=False, with_mask=False, with_label=True, with_plane=False,
                 class_names=('Car', 'Van'), augmentor=None, generator=None, anchor_generator=None,
                 anchor_area_threshold=1, target_encoder=None, out_size_factor=2, test_mode=False, device=None):
        self.root = root
        self.img_scales = img_scale if isinstance(img_scale, list) else [img_scale]
        self.class_names = list(class_names)
        self.test_mode, self.with_label, self.with_point, self.with_plane = test_mode, with_label, with_point, with_plane
        self.img_prefix = osp.join(root, 'image_2')
        self.lidar_prefix = osp.join(root, 'velodyne_reduced')
        self.calib_prefix = osp.join(root, 'calib')
        self.label_prefix = osp.join(root, 'label_2')
        self.plane_prefix = osp.join(root, 'planes')
        with open(ann_file, 'r') as f:
            self.sample_ids = list(map(int, f.read().splitlines()))
        if not self.test_mode:
            self.flag = np.ones(len(self), dtype=np.uint8)
        self.augmentor, self.generator = augmentor, generator
        self.out_size_factor, self.anchor_area_threshold = out_size_factor, anchor_area_threshold
        self.device = None if device is None else torch.device(device)
        self.anchors = self.anchors_bv = None
        self._dev_anchors = None
        if anchor_generator is not None:
            fmap = [*(self.generator.grid_size[:2] // self.out_size_factor), 1][::-1]
            per_class = {k: v(fmap).reshape(-1, 7) for k, v in anchor_generator.items()}
            if self.test_mode:                                   # one concatenated set (kitti.py:82-85)
                self.anchors = np.concatenate(list(per_class.values()), 0)
                self.anchors_bv = anchor3d_generator.rbbox2d_to_near_bbox(self.anchors[..., [0, 1, 3, 4, 6]])
            else:                                                # one set per class (kitti.py:86-88)
                self.anchors = per_class
                self.anchors_bv = {k: anchor3d_generator.rbbox2d_to_near_bbox(v[:, [0, 1, 3, 4, 6]])
                                   for k, v in per_class.items()}

    def __len__(self):
        return len(self.sample_ids)

    # ---- host side: files ------------------------------------------------------------------------------------------------
    def _dev(self):
        if self.device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("sassd.KittiLiDAR prepares frames on an MI355X (no CPU fallback)")
            self.device = torch.device("cuda", torch.cuda.current_device())
        return self.device

    def image_shape(self, sample_id):
        path = osp.join(self.img_prefix, '%06d.png' % sample_id)
        if osp.exists(path):
            return (*png_shape(path), 3)
        w, h = self.img_scales[0]
        return (h, w, 3)

    def load_frame(self, idx, with_label=True):
        """Everything the files hold for sample idx: dict(sample_idx, img_shape, calib, points [N,4] f32 numpy,
        gt_bboxes [G,7] f32 lidar frame, gt_types list, plane)."""
        sample_id = self.sample_ids[idx]
        calib = Calibration(osp.join(self.calib_prefix, '%06d.txt' % sample_id))
        out = dict(sample_idx=sample_id, img_shape=self.image_shape(sample_id), calib=calib, plane=None)
        if with_label:
            boxes, types = read_label_boxes(osp.join(self.label_prefix, '%06d.txt' % sample_id))
            if len(boxes):
                boxes[:, :3] = project_rect_to_velo(boxes[:, :3], calib)      # camera -> lidar (kitti.py:152-154)
            out.update(gt_bboxes=boxes, gt_types=types)
        if self.with_point:
            out['points'] = read_lidar(osp.join(self.lidar_prefix, '%06d.bin' % sample_id))
        if self.with_plane:
            out['plane'] = get_road_plane(osp.join(self.plane_prefix, '%06d.txt' % sample_id))
        return out

    # ---- one frame -> device tensors -------------------------------------------------------------------------------------
    def prepare_train_img(self, idx, frame=None):
        fr = self.load_frame(idx) if frame is None else frame
        dev = self._dev()
        points = torch.from_numpy(fr['points']).to(dev)
        gt_bboxes, gt_types = fr['gt_bboxes'], fr['gt_types']
        if self.augmentor is not None:
            points, gt_bboxes, gt_types, gt_labels = self.augmentor.augment_frame(
                points, gt_bboxes, gt_types, self.class_names, fr['plane'], fr['calib'])
        else:                                                    # (the reference cannot train without an augmentor)
            gt_types = np.array(['Car' if n == 'Van' else n for n in gt_types])
            keep = [i for i in range(len(gt_types)) if gt_types[i] in self.class_names]
            gt_bboxes, gt_types = gt_bboxes[keep, :], gt_types[keep]
            gt_labels = np.array([self.class_names.index(n) + 1 for n in gt_types], dtype=np.int64)
        inside = G.filter_gt_box_outside_range(gt_bboxes, self.generator.point_cloud_range[[0, 1, 3, 4]])
        gt_bboxes, gt_types, gt_labels = gt_bboxes[inside], gt_types[inside], gt_labels[inside]
        if len(gt_bboxes) == 0:
            return None                                          # no object left in range: the caller draws another frame
        gt_bboxes[:, 6] = limit_period(gt_bboxes[:, 6], offset=0.5, period=2 * np.pi)
        return dict(img=None, img_meta=dict(img_shape=fr['img_shape'], sample_idx=fr['sample_idx'], calib=fr['calib']),
                    points=points, gt_bboxes=torch.from_numpy(np.ascontiguousarray(gt_bboxes, dtype=np.float32)).to(dev),
                    gt_labels=torch.from_numpy(gt_labels).to(dev), gt_types=gt_types)

    def prepare_test_img(self, idx, frame=None):
        fr = self.load_frame(idx, with_label=self.with_label) if frame is None else frame
        dev = self._dev()
        out = dict(img=None, img_meta=dict(img_shape=fr['img_shape'], sample_idx=fr['sample_idx'], calib=fr['calib']),
                   points=torch.from_numpy(fr['points']).to(dev), gt_bboxes=None, gt_labels=None, gt_types=None)
        if self.with_label:
            types = np.array(['Car' if n == 'Van' else n for n in fr['gt_types']])
            keep = [i for i in range(len(types)) if types[i] in self.class_names]
            out.update(gt_bboxes=fr['gt_bboxes'][keep, :], gt_types=types[keep],
                       gt_labels=np.array([self.class_names.index(n) + 1 for n in types[keep]], dtype=np.int64))
        return out

    def __getitem__(self, idx):
        if self.test_mode:
            return self.prepare_test_img(idx)
        while True:
            data = self.prepare_train_img(idx)
            if data is not None:
                return data
            idx = np.random.choice(np.where(self.flag == self.flag[idx])[0])

    # ---- frames -> model keyword arguments -------------------------------------------------------------------------------
    def _anchors_on_device(self):
        if self._dev_anchors is None:
            dev = self._dev()
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
            if isinstance(self.anchors, dict):
                self._dev_anchors = ({k: t(v) for k, v in self.anchors.items()}, {k: t(v) for k, v in self.anchors_bv.items()})
            else:
                self._dev_anchors = (t(self.anchors), t(self.anchors_bv))
        return self._dev_anchors

    def collate(self, samples, model=None):
        """list of prepare_*_img results -> kwargs of model(...): voxelization and anchor masks run here, on the device.
        Train mode adds the sparse-conv rulebooks of the batch when `model` is given (sassd.train.device_batch)."""
        from . import train
        gen = self.generator
        an, bv = self._anchors_on_device()
        metas = [s['img_meta'] for s in samples]
        if not self.test_mode:
            kw = train.device_batch([s['points'] for s in samples], [s['gt_bboxes'] for s in samples],
                                    [s['gt_types'] for s in samples], self.class_names, an, bv, gen.voxel_size,
                                    gen.point_cloud_range, gen.max_num_points_per_voxel, gen._max_voxels,
                                    self.anchor_area_threshold, model=model)
            kw['img_meta'] = metas
            kw['gt_labels'] = [s['gt_labels'] for s in samples]
            return kw
        vs, cr = list(gen.voxel_size), list(gen.point_cloud_range)
        w0, h0 = int(gen.grid_size[0]), int(gen.grid_size[1])
        kw = dict(img=None, img_meta=metas, return_loss=False, voxels=[], coordinates=[], num_points=[], anchors=[],
                  anchors_mask=[])
        for s in samples:
            r = K.voxelize(s['points'], vs, cr, gen.max_num_points_per_voxel, gen._max_voxels, batch_idx=0, coors_cols=4,
                           want_mean=False)
            m = int(r["voxel_num"].item())
            kw["voxels"].append(r["voxels"][:m])
            kw["coordinates"].append(r["coors"][:m, 1:])
            kw["num_points"].append(r["num_points"][:m])
            kw["anchors"].append(an)
            zero = torch.zeros(1, dtype=torch.int32, device=s['points'].device)
            mask = K.anchor_mask(r["coors"], zero, r["voxel_num"], h0, w0, bv, vs, cr, self.anchor_area_threshold)
            kw["anchors_mask"].append(mask.bool())
        return kw


def get_dataset(data_cfg, device=None):
    """mmdet/datasets/utils.py:80-130: build generator / augmentor / anchor generators from their config dicts, then the
    dataset named by data_cfg['type'] (one per ann_file)."""
    cfg = copy.deepcopy(dict(data_cfg))
    ann_files = cfg['ann_file'] if isinstance(cfg['ann_file'], (list, tuple)) else [cfg['ann_file']]
    if cfg.get('generator') is not None and isinstance(cfg['generator'], dict):
        cfg['generator'] = obj_from_dict(cfg['generator'], voxel_generator)
    if cfg.get('augmentor') is not None and isinstance(cfg['augmentor'], dict):
        cfg['augmentor'] = obj_from_dict(cfg['augmentor'], point_augmentor, dict(device=device))
    if cfg.get('anchor_generator') is not None:
        cfg['anchor_generator'] = {c: (obj_from_dict(g, anchor3d_generator) if isinstance(g, dict) else g)
                                   for c, g in cfg['anchor_generator'].items()}
    dsets = []
    for ann in ann_files:
        info = dict(cfg, ann_file=ann, device=device)
        dsets.append(obj_from_dict(info, sys.modules[__name__]))
    return dsets[0] if len(dsets) == 1 else dsets
