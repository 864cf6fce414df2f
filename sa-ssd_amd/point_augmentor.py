"""Training-side augmentation -- same class, constructor arguments and method names as
mmdet/core/point_cloud/point_augmentor.py (`PointAugmentor`, `BatchSampler`; used by mmdet/datasets/kitti.py:209-238),
re-designed for one MI355X per process instead of four numba DataLoader workers per GPU:

  * the ground-truth database (tools/create_data.py: kitti_dbinfos_train.pkl + gt_database/*.bin, a few hundred MB for
    KITTI) is read ONCE and kept resident in HBM as one packed [P,4] array; pasting sampled objects is a gather kernel
    (sassd_paste_objects), not `np.fromfile` per object per iteration (point_augmentor.py:232-242);
  * the frame's sweep is uploaded once; point-in-box masks (sassd_points_in_polytopes), the per-object move
    (sassd_points_transform) and flip + global rotation + global scaling fused in one pass
    (sassd_points_global_transform) run one thread per point and hand the cloud straight to the HIP voxelizer;
  * what is sequential over a few dozen boxes -- collision tests between sampled and existing boxes, choosing the first
    non-colliding noise draw per box -- is native host code (sassd_box_collision_test, sassd_noise_per_box).

Random numbers come from numpy's global generator in exactly the reference's order (sampler shuffles, normal centre
noise, uniform yaw noise, flip, global rotation, global scale), so `np.random.seed(s)` reproduces the reference's
choices.  Methods accept numpy arrays (the reference's calling convention: uploaded / downloaded around the kernels) or
GPU tensors (kept on the GPU); there is no CPU fallback for the per-point work."""
import copy
import ctypes
import pathlib
import pickle

import numpy as np
import torch

from . import _C
from . import geometry as G
from .kitti_common import project_rect_to_velo, project_velo_to_rect


def _hp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _on_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("sassd.point_augmentor: per-point work runs on the MI355X only (tensor on %s)" % t.device)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _paste_objects(db, src_start, out_start, n_obj, n_out, shift, lower, out):
    _on_gpu(db, src_start, out_start, shift, lower, out)
    with torch.cuda.device(out.device):
        _C.check(_C.lib().sassd_paste_objects(_ptr(db), _ptr(src_start), _ptr(out_start), n_obj, n_out, _ptr(shift),
                                              _ptr(lower), _ptr(out), _C.stream()), "sassd_paste_objects")


def _points_transform(points, mask8, valid, centers, rot_sin, rot_cos, loc):
    _on_gpu(points, mask8, valid, centers, rot_sin, rot_cos, loc)
    with torch.cuda.device(points.device):
        _C.check(_C.lib().sassd_points_transform(_ptr(points), points.shape[0], points.stride(0), _ptr(mask8),
                                                 mask8.shape[1], _ptr(valid), _ptr(centers), _ptr(rot_sin), _ptr(rot_cos),
                                                 _ptr(loc), _C.stream()), "sassd_points_transform")


def _points_global(points, flip, rot_sin, rot_cos, scale):
    _on_gpu(points)
    with torch.cuda.device(points.device):
        _C.check(_C.lib().sassd_points_global_transform(_ptr(points), points.shape[0], points.stride(0), int(flip),
                                                        float(rot_sin), float(rot_cos), float(scale), _C.stream()),
                 "sassd_points_global_transform")


def noise_per_box(boxes, valid_mask, loc_noises, rot_noises):
    """boxes [N,5] (x, y, w, l, yaw), loc_noises [N,M,3], rot_noises [N,M] -> [N] index of the accepted draw or -1
    (point_augmentor.py:73-105)."""
    b = np.ascontiguousarray(boxes, dtype=np.float32)
    v = np.ascontiguousarray(valid_mask, dtype=np.uint8)
    ln, rn = np.ascontiguousarray(loc_noises, dtype=np.float64), np.ascontiguousarray(rot_noises, dtype=np.float64)
    out = np.full((b.shape[0],), -1, dtype=np.int64)
    _C.check(_C.lib().sassd_noise_per_box(_hp(b), _hp(v), _hp(ln), _hp(rn), b.shape[0], rn.shape[1] if rn.ndim == 2 else 0,
                                          _hp(out)), "sassd_noise_per_box")
    return out


def select_transform(transform, indices):
    """transform [N,M,...], indices [N] -> [N,...] with zeros where the index is -1."""
    result = np.zeros((transform.shape[0], *transform.shape[2:]), dtype=transform.dtype)
    ok = indices != -1
    result[ok] = transform[np.flatnonzero(ok), indices[ok]]
    return result


def box3d_transform_(boxes, loc_transform, rot_transform, valid_mask):
    v = np.asarray(valid_mask, dtype=bool)
    boxes[v, :3] += loc_transform[v]
    boxes[v, 6] += rot_transform[v]


class BatchSampler:
    """Walks a shuffled index list, reshuffling when it runs out (point_augmentor.py:107-139)."""

    def __init__(self, sampled_list, name=None, epoch=None, shuffle=True, drop_reminder=False):
        self._sampled_list = sampled_list
        self._indices = np.arange(len(sampled_list))
        if shuffle:
            np.random.shuffle(self._indices)
        self._idx, self._example_num = 0, len(sampled_list)
        self._name, self._shuffle = name, shuffle

    def sample_indices(self, num):
        if self._idx + num >= self._example_num:
            ret = self._indices[self._idx:].copy()
            if self._shuffle:
                np.random.shuffle(self._indices)
            self._idx = 0
        else:
            ret = self._indices[self._idx:self._idx + num]
            self._idx += num
        return ret

    def sample(self, num):
        return [self._sampled_list[i] for i in self.sample_indices(num)]


class PointAugmentor:
    def __init__(self, root_path, info_path, sample_classes, min_num_points, sample_max_num, removed_difficulties,
                 gt_rot_range=None, global_rot_range=None, center_noise_std=None, scale_range=None, device=None):
        with open(info_path, 'rb') as f:
            db_infos_all = pickle.load(f)
        if isinstance(min_num_points, int):
            min_num_points = [min_num_points] * len(sample_classes)
        self.root_path = root_path
        self._samplers, rows, at = [], [], 0
        for i, cls in enumerate(sample_classes):
            kept = [dict(info) for info in db_infos_all[cls]
                    if info["num_points_in_gt"] >= min_num_points[i] and info["difficulty"] not in removed_difficulties]
            for info in kept:                                    # read every object's points once
                pts = np.fromfile(str(pathlib.Path(root_path) / info["path"]), dtype=np.float32).reshape([-1, 4])
                info["_start"], info["_count"] = at, len(pts)
                rows.append(pts)
                at += len(pts)
            self._samplers.append(BatchSampler(kept, cls))
        self._db_points = np.concatenate(rows, 0) if rows else np.zeros((0, 4), np.float32)
        self._db_dev = None
        self.device = None if device is None else torch.device(device)
        self._sample_classes = sample_classes
        self._sample_max_num = [sample_max_num] * len(sample_classes) if isinstance(sample_max_num, int) else sample_max_num
        self._global_rot_range, self._gt_rot_range = global_rot_range, gt_rot_range
        self._center_noise_std = center_noise_std
        self._min_scale, self._max_scale = scale_range[0], scale_range[1]

    # ---- device plumbing -----------------------------------------------------------------------------------------------
    def _dev(self):
        if self.device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("sassd.PointAugmentor needs an MI355X for per-point work (no CPU fallback)")
            self.device = torch.device("cuda", torch.cuda.current_device())
        return self.device

    def database_on_device(self):
        if self._db_dev is None:
            self._db_dev = torch.from_numpy(self._db_points).to(self._dev())
        return self._db_dev

    def _points_in(self, points):
        """-> (GPU tensor [N,4] f32, the numpy array to write back into or None)"""
        if torch.is_tensor(points):
            return points, None
        return torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(self._dev()), points

    @staticmethod
    def _points_out(dev_points, host):
        if host is None:
            return dev_points
        host[...] = dev_points.cpu().numpy()
        return host

    # ---- ground-truth sampling -------------------------------------------------------------------------------------------
    def sample(self, gt_boxes, num, i):
        """`num` database objects of class i that collide neither with gt_boxes nor with each other (earlier wins)."""
        sampled = copy.deepcopy(self._samplers[i].sample(num))
        if not sampled:                                          # (an empty class list makes the reference's np.stack fail)
            return []
        num_gt = gt_boxes.shape[0]
        sp_boxes = np.stack([s["box3d_lidar"] for s in sampled], axis=0)
        boxes = np.concatenate([gt_boxes, sp_boxes], axis=0).copy()
        bv = G.center_to_corner_box2d(boxes[:, 0:2], boxes[:, 3:5], boxes[:, 6])
        if gt_boxes.dtype != boxes.dtype:                       # the scene's corners are formed in the scene's own dtype
            bv[:num_gt] = G.center_to_corner_box2d(gt_boxes[:, 0:2], gt_boxes[:, 3:5], gt_boxes[:, 6])
        coll = G.box_collision_test(bv, bv)
        np.fill_diagonal(coll, False)
        valid = []
        for k in range(num_gt, num_gt + len(sampled)):
            if coll[k].any():
                coll[k] = False
                coll[:, k] = False
            else:
                valid.append(sampled[k - num_gt])
        return valid

    def select_samples(self, gt_boxes, gt_types, road_planes=None, calib=None):
        """Host part of sample_all: -> (sampled infos, sampled_gt_boxes [S,7] float64 after the road-plane correction,
        mv_height [S] or None)."""
        avoid = gt_boxes
        sampled, boxes = [], []
        for i, cls in enumerate(self._sample_classes):
            want = int(self._sample_max_num[i] - np.sum([n == cls for n in gt_types]))
            got = self.sample(avoid, want, i) if want > 0 else []
            sampled += got
            if got:
                b = np.stack([s["box3d_lidar"] for s in got], axis=0)
                boxes.append(b)
                avoid = np.concatenate([avoid, b], axis=0)
        if not sampled:
            return [], np.empty((0, 7)), None
        boxes = np.concatenate(boxes, axis=0)
        mv_height = None
        if road_planes is not None:                              # put the pasted objects on the frame's road plane
            a, b, c, d = road_planes
            center_cam = project_velo_to_rect(boxes[:, 0:3], calib)
            center_cam[:, 1] = (-d - a * center_cam[:, 0] - c * center_cam[:, 2]) / b
            mv_height = boxes[:, 2] - project_rect_to_velo(center_cam, calib)[:, 2]
            boxes[:, 2] -= mv_height
        return sampled, boxes, mv_height

    def paste(self, sampled, mv_height=None):
        """The sampled objects' points, gathered on the GPU from the resident database: [sum counts, 4] f32."""
        dev = self._dev()
        counts = np.array([s["_count"] for s in sampled], dtype=np.int64)
        out_start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        n_out = int(out_start[-1])
        out = torch.empty((n_out, 4), dtype=torch.float32, device=dev)
        if n_out:
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            src = t(np.array([s["_start"] for s in sampled], dtype=np.int64))
            shift = t(np.stack([s["box3d_lidar"][:3] for s in sampled]).astype(np.float64))
            lower = None if mv_height is None else t(np.asarray(mv_height, dtype=np.float64))
            ostart = t(out_start)
            _paste_objects(self.database_on_device(), src, ostart, len(sampled), n_out, shift, lower, out)
        return out

    def sample_all(self, gt_boxes, gt_types, road_planes=None, calib=None, as_numpy=True):
        """-> (sampled_gt_boxes [S,7] f32, sampled_gt_types, sampled_points [P,4] f32)   (point_augmentor.py:191-247)"""
        sampled, boxes, mv_height = self.select_samples(gt_boxes, gt_types, road_planes, calib)
        if not sampled:
            empty = np.empty((0, 4), dtype=np.float32)
            return np.empty((0, 7), dtype=np.float32), [], empty if as_numpy else torch.from_numpy(empty).to(self._dev())
        pts = self.paste(sampled, mv_height)
        return boxes.astype(np.float32), [s['name'] for s in sampled], pts.cpu().numpy() if as_numpy else pts

    # ---- per-object noise ------------------------------------------------------------------------------------------------
    def draw_object_noise(self, gt_boxes, valid_mask=None, num_try=100):
        """The host decisions of noise_per_object_: -> (loc_transforms [N,3] f64, rot_transforms [N] f64, valid [N])."""
        n = gt_boxes.shape[0]
        valid_mask = np.ones((n,), dtype=np.bool_) if valid_mask is None else valid_mask
        std = np.array(self._center_noise_std, dtype=gt_boxes.dtype)
        loc_noises = np.random.normal(scale=std, size=[n, num_try, 3])
        rot_noises = np.random.uniform(self._global_rot_range[0], self._global_rot_range[1], size=[n, num_try])
        chosen = noise_per_box(gt_boxes[:, [0, 1, 3, 4, 6]], valid_mask, loc_noises, rot_noises)
        return select_transform(loc_noises, chosen), select_transform(rot_noises, chosen), valid_mask

    def move_points(self, points_dev, gt_boxes, loc_t, rot_t, valid):
        """points inside a box follow that box (boxes as they are BEFORE the move)."""
        n, m = points_dev.shape[0], gt_boxes.shape[0]
        if n == 0 or m == 0:
            return
        dev = points_dev.device
        mask = G.points_in_rbbox(points_dev, gt_boxes)                         # [n,m] bool on the GPU
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        mask8 = mask.to(torch.uint8).contiguous()
        centers = t(gt_boxes[:, :3].astype(np.float32))
        rs, rc = t(np.sin(rot_t).astype(np.float32)), t(np.cos(rot_t).astype(np.float32))
        loc, vd = t(loc_t.astype(np.float64)), t(np.asarray(valid, dtype=np.uint8))
        _points_transform(points_dev, mask8, vd, centers, rs, rc, loc)

    def noise_per_object_(self, gt_boxes, points=None, valid_mask=None, num_try=100):
        """Moves every ground-truth box (and the points inside it) by the first of `num_try` random (shift, yaw) draws that
        does not make it collide with another box; in place on gt_boxes and points (point_augmentor.py:306-346)."""
        loc_t, rot_t, valid = self.draw_object_noise(gt_boxes, valid_mask, num_try)
        if points is not None:
            dev_pts, host = self._points_in(points)
            self.move_points(dev_pts, gt_boxes, loc_t, rot_t, valid)
            self._points_out(dev_pts, host)
        box3d_transform_(gt_boxes, loc_t, rot_t, valid)

    # ---- global transforms -----------------------------------------------------------------------------------------------
    def _global(self, points, flip, angle, scale):
        dev_pts, host = self._points_in(points)
        n = dev_pts.shape[0]
        if n:
            _points_global(dev_pts, flip, np.float32(np.sin(angle)), np.float32(np.cos(angle)), np.float32(scale))
        return self._points_out(dev_pts, host)

    def random_flip(self, gt_boxes, points, probability=0.5):
        enable = np.random.choice([False, True], replace=False, p=[1 - probability, probability])
        if enable:
            gt_boxes[:, 1] = -gt_boxes[:, 1]
            gt_boxes[:, 6] = -gt_boxes[:, 6] + np.pi
            points = self._global(points, True, 0.0, 1.0)
        return gt_boxes, points

    def global_rotation(self, gt_boxes, points):
        angle = np.random.uniform(self._global_rot_range[0], self._global_rot_range[1])
        points = self._global(points, False, angle, 1.0)
        gt_boxes[:, :3] = G.rotation_points_single_angle(gt_boxes[:, :3], angle, axis=2)
        gt_boxes[:, 6] += angle
        return gt_boxes, points

    def global_scaling(self, gt_boxes, points):
        scale = np.random.uniform(self._min_scale, self._max_scale)
        points = self._global(points, False, 0.0, scale)
        gt_boxes[:, :6] *= scale
        return gt_boxes, points

    # ---- the whole training-frame recipe, fused (mmdet/datasets/kitti.py:209-238) ---------------------------------------------
    def augment_frame(self, points, gt_boxes, gt_types, class_names, road_planes=None, calib=None, num_try=100):
        """points [N,4] (numpy or GPU tensor), gt_boxes [G,7] f32 lidar, gt_types list -> (points on the GPU, gt_boxes,
        gt_types array, gt_labels): paste sampled objects, drop the scene points they cover, Van -> Car, keep
        `class_names`, per-object noise, flip, global rotation, global scaling -- the last three in ONE pass over the
        points.  Consumes numpy's global random stream in the reference's order."""
        dev_pts, _ = self._points_in(points)
        sampled, s_boxes, mv_height = self.select_samples(gt_boxes, gt_types, road_planes, calib)
        if sampled:
            s_boxes32 = s_boxes.astype(np.float32)
            covered = G.points_in_rbbox(dev_pts, s_boxes32).any(-1)
            dev_pts = torch.cat([self.paste(sampled, mv_height), dev_pts[~covered]], dim=0)
            gt_boxes = np.concatenate([gt_boxes, s_boxes32])
            gt_types = list(gt_types) + [s['name'] for s in sampled]
        else:
            dev_pts = dev_pts.clone()
        gt_types = np.array(['Car' if n == 'Van' else n for n in gt_types])
        keep = [i for i in range(len(gt_types)) if gt_types[i] in class_names]
        gt_boxes, gt_types = gt_boxes[keep, :], gt_types[keep]
        gt_labels = np.array([class_names.index(n) + 1 for n in gt_types], dtype=np.int64)

        loc_t, rot_t, valid = self.draw_object_noise(gt_boxes, None, num_try)
        self.move_points(dev_pts, gt_boxes, loc_t, rot_t, valid)
        box3d_transform_(gt_boxes, loc_t, rot_t, valid)

        flip = np.random.choice([False, True], replace=False, p=[0.5, 0.5])
        angle = np.random.uniform(self._global_rot_range[0], self._global_rot_range[1])
        scale = np.random.uniform(self._min_scale, self._max_scale)
        if flip:
            gt_boxes[:, 1] = -gt_boxes[:, 1]
            gt_boxes[:, 6] = -gt_boxes[:, 6] + np.pi
        gt_boxes[:, :3] = G.rotation_points_single_angle(gt_boxes[:, :3], angle, axis=2)
        gt_boxes[:, 6] += angle
        gt_boxes[:, :6] *= scale
        dev_pts = self._global(dev_pts, flip, angle, scale)
        return dev_pts, gt_boxes, gt_types, gt_labels
