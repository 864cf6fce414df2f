"""KITTI on-disk formats either side of the hot path: label / result files, calibration files, and
the conversion of the detector's lidar-frame boxes into KITTI camera-frame result annotations.

Mirrors, with the same names and dictionary keys:
  * tools/kitti_common.py:560-668   get_label_anno(s), get_start_result_anno, empty_result_anno, anno_to_rbboxes
  * mmdet/datasets/kitti_utils.py:49-126,154-204   Calibration, project_velo_to_rect, project_rect_to_image, read_lidar
  * mmdet/core/bbox3d/geometry.py:289-404   the camera-frame box corners kitti_bbox2results needs
  * mmdet/core/bbox/transforms.py:225-276   kitti_bbox2results
Everything here is host-side numpy on a few dozen boxes per frame; the O(N*K) work of the evaluation is in
sassd.kitti_eval / sassd.eval_ops (HIP)."""
import pathlib
import re

import numpy as np

_ANNO_KEYS = ("name", "truncated", "occluded", "alpha", "bbox", "dimensions", "location", "rotation_y")


def get_image_index_str(img_idx):
    return "{:06d}".format(img_idx)


def get_label_anno(label_path):
    """One KITTI label / result file -> dict of arrays.  A line is
    `type truncated occluded alpha x1 y1 x2 y2 h w l x y z rotation_y [score]`; dimensions are stored l,h,w."""
    with open(label_path, "r") as f:
        rows = [ln.strip().split(" ") for ln in f.readlines()]
    n = len(rows)
    col = lambda a, b: np.array([[float(v) for v in r[a:b]] for r in rows], dtype=np.float64).reshape(n, b - a)
    anno = dict(
        name=np.array([r[0] for r in rows]),
        truncated=np.array([float(r[1]) for r in rows]),
        occluded=np.array([int(float(r[2])) for r in rows]),
        alpha=np.array([float(r[3]) for r in rows]),
        bbox=col(4, 8),
        dimensions=col(8, 11)[:, [2, 0, 1]],
        location=col(11, 14),
        rotation_y=np.array([float(r[14]) for r in rows]).reshape(-1),
    )
    if n and len(rows[0]) == 16:
        anno["score"] = np.array([float(r[15]) for r in rows])
    else:
        anno["score"] = np.zeros((n,))
    cared = sum(1 for r in rows if r[0] != "DontCare")
    anno["index"] = np.array(list(range(cared)) + [-1] * (n - cared), dtype=np.int32)
    anno["group_ids"] = np.arange(n, dtype=np.int32)
    return anno


def get_label_annos(label_folder, image_ids=None):
    folder = pathlib.Path(label_folder)
    if image_ids is None:
        pat = re.compile(r"^\d{6}.txt$")
        image_ids = sorted(int(p.stem) for p in folder.glob("*.txt") if pat.match(p.name))
    if not isinstance(image_ids, list):
        image_ids = list(range(image_ids))
    annos = []
    for idx in image_ids:
        anno = get_label_anno(folder / (get_image_index_str(idx) + ".txt"))
        anno["image_idx"] = np.array([idx] * anno["name"].shape[0], dtype=np.int64)
        annos.append(anno)
    return annos


def get_start_result_anno():
    return {k: [] for k in _ANNO_KEYS + ("score",)}


def empty_result_anno():
    return dict(name=np.array([]), truncated=np.array([]), occluded=np.array([]), alpha=np.array([]),
                bbox=np.zeros([0, 4]), dimensions=np.zeros([0, 3]), location=np.zeros([0, 3]),
                rotation_y=np.array([]), score=np.array([]))


def anno_to_rbboxes(anno):
    return np.concatenate([anno["location"], anno["dimensions"], anno["rotation_y"][..., np.newaxis]], axis=1)


def kitti_result_line(name, alpha, bbox, dimensions, location, rotation_y, score, truncated=0.0, occluded=0):
    """One line of a KITTI result file (dimensions given l,h,w as in the annotation dicts, written h,w,l)."""
    l, h, w = dimensions
    vals = [truncated, occluded, alpha, *bbox, h, w, l, *location, rotation_y, score]
    return "{} {:.2f} {:d} ".format(name, float(vals[0]), int(vals[1])) + " ".join("{:.4f}".format(float(v))
                                                                                   for v in vals[2:])


def write_label_annos(annos, folder):
    """Result annotations -> one `%06d.txt` per frame (image_idx), the format get_label_annos reads back."""
    folder = pathlib.Path(folder)
    folder.mkdir(parents=True, exist_ok=True)
    for anno in annos:
        if "image_idx" not in anno or len(anno["image_idx"]) == 0:
            continue
        lines = [kitti_result_line(anno["name"][i], anno["alpha"][i], anno["bbox"][i], anno["dimensions"][i],
                                   anno["location"][i], anno["rotation_y"][i], anno["score"][i])
                 for i in range(len(anno["name"]))]
        with open(folder / (get_image_index_str(int(anno["image_idx"][0])) + ".txt"), "w") as f:
            f.write("\n".join(lines) + ("\n" if lines else ""))


# ---- calibration -------------------------------------------------------------------------------------------------

class Calibration:
    """calib/%06d.txt: P2 / P3 (rect camera -> image), R0_rect (reference -> rect camera), Tr_velo_to_cam."""

    def __init__(self, calib_filepath=None, matrices=None):
        c = matrices if matrices is not None else self.read_calib_file(calib_filepath)
        self.P2 = np.reshape(c["P2"], [3, 4])
        self.P3 = np.reshape(c["P3"], [3, 4]) if "P3" in c else None
        self.V2C = np.reshape(c["Tr_velo_to_cam"], [3, 4])
        rot = self.V2C[:, :3]
        self.C2V = np.concatenate([rot.T, (-rot.T @ self.V2C[:, 3])[:, None]], axis=1)
        self.R0 = np.reshape(c["R0_rect"], [3, 3])
        self.c_u, self.c_v = self.P2[0, 2], self.P2[1, 2]
        self.f_u, self.f_v = self.P2[0, 0], self.P2[1, 1]
        self.b_x, self.b_y = self.P2[0, 3] / (-self.f_u), self.P2[1, 3] / (-self.f_v)

    @staticmethod
    def read_calib_file(filepath):
        data = {}
        with open(filepath, "r") as f:
            for line in f:
                line = line.rstrip()
                if not line:
                    continue
                key, value = line.split(":", 1)
                try:
                    data[key] = np.array([float(x) for x in value.split()])
                except ValueError:                       # dates and other non-numeric entries
                    pass
        return data


def cart2hom(pts):
    return np.concatenate([pts, np.ones(list(pts.shape[:-1]) + [1])], axis=-1)


def project_velo_to_rect(pts_velo, calib):
    return (cart2hom(pts_velo) @ calib.V2C.T) @ calib.R0.T


def project_rect_to_velo(pts_rect, calib):
    return cart2hom(pts_rect @ np.linalg.inv(calib.R0).T) @ calib.C2V.T


def project_rect_to_image(pts_rect, calib):
    uvw = cart2hom(pts_rect) @ calib.P2.T
    uvw[..., 0] /= uvw[..., 2]
    uvw[..., 1] /= uvw[..., 2]
    return uvw[..., 0:2]


def read_lidar(bin_path):
    return np.fromfile(bin_path, dtype=np.float32).reshape(-1, 4)


# ---- boxes -------------------------------------------------------------------------------------------------------

def limit_period(val, offset=0.5, period=np.pi):
    return val - np.floor(val / period + offset) * period


def _camera_box_corners(boxes_cam):
    """[N,7] (x, y, z, l, h, w, ry), origin at the bottom centre (0.5, 1.0, 0.5), rotation about the camera y axis ->
    [N,8,3] corners in the order x0y0z0, x0y0z1, x0y1z1, x0y1z0, x1y0z0, x1y0z1, x1y1z1, x1y1z0."""
    unit = np.array([[0, 0, 0], [0, 0, 1], [0, 1, 1], [0, 1, 0], [1, 0, 0], [1, 0, 1], [1, 1, 1], [1, 1, 0]],
                    dtype=boxes_cam.dtype) - np.array([0.5, 1.0, 0.5], dtype=boxes_cam.dtype)
    corners = boxes_cam[:, None, 3:6] * unit[None]
    s, c = np.sin(boxes_cam[:, 6]), np.cos(boxes_cam[:, 6])
    o, z = np.ones_like(c), np.zeros_like(c)
    rot_t = np.stack([[c, z, -s], [z, o, z], [s, z, c]])                   # [3,3,N]
    return np.einsum("aij,jka->aik", corners, rot_t) + boxes_cam[:, None, :3]


def kitti_bbox2results(boxes_lidar, scores, labels, meta, class_names=None):
    """Detections of one frame (lidar frame: x, y, z, w, l, h, yaw) -> KITTI result annotation (camera frame), dropping
    boxes whose projection misses the image and clipping the rest to it.  meta: calib, sample_idx, img_shape.
    Like the reference, wraps the yaw of `boxes_lidar` to [-pi, pi) in place."""
    if scores is None or len(scores) == 0 or boxes_lidar is None or len(boxes_lidar) == 0:
        return empty_result_anno()
    calib, sample_id = meta["calib"], meta["sample_idx"]
    img_h, img_w = meta["img_shape"][:2]
    boxes_lidar[:, -1] = limit_period(boxes_lidar[:, -1], offset=0.5, period=np.pi * 2)
    boxes_cam = np.zeros_like(boxes_lidar)
    boxes_cam[:, :3] = project_velo_to_rect(boxes_lidar[:, :3], calib)
    boxes_cam[:, 3:] = boxes_lidar[:, [4, 5, 3, 6]]
    uv = project_rect_to_image(_camera_box_corners(boxes_cam), calib)
    box2d = np.concatenate([uv.min(axis=1), uv.max(axis=1)], axis=1)
    alphas = -np.arctan2(-boxes_lidar[:, 1], boxes_lidar[:, 0]) + boxes_lidar[:, 6]
    keep = ~((box2d[:, 0] > img_w) | (box2d[:, 1] > img_h) | (box2d[:, 2] < 0) | (box2d[:, 3] < 0))
    if not keep.any():
        return empty_result_anno()
    box2d = box2d[keep]
    box2d[:, 2:] = np.minimum(box2d[:, 2:], np.array([img_w, img_h], dtype=box2d.dtype))
    box2d[:, :2] = np.maximum(box2d[:, :2], 0)
    cam = boxes_cam[keep]
    k = int(keep.sum())
    return dict(name=np.array([class_names[int(lb)] for lb in np.asarray(labels)[keep]]),
                truncated=np.zeros(k), occluded=np.zeros(k, dtype=np.int64), alpha=alphas[keep], bbox=box2d,
                dimensions=cam[:, 3:6], location=cam[:, :3], rotation_y=cam[:, 6],
                score=np.asarray(scores)[keep], image_idx=np.full(k, int(sample_id), dtype=np.int64))


# ---- per-frame info records (tools/kitti_common.py:77-213,476-518; consumed by sassd.create_data) ----------------------

def png_shape(path):
    """(height, width) from a PNG file's IHDR chunk -- all the image the lidar path ever needs (the reference decodes the
    whole picture with imageio / mmcv just to read its shape, kitti_common.py:152-153, kitti.py:140-142)."""
    with open(path, "rb") as f:
        head = f.read(24)
    if len(head) < 24 or head[:8] != b"\x89PNG\r\n\x1a\n" or head[12:16] != b"IHDR":
        raise ValueError("not a PNG file: %s" % path)
    return int.from_bytes(head[20:24], "big"), int.from_bytes(head[16:20], "big")


def get_kitti_info_path(idx, prefix, info_type='image_2', file_tail='.png', training=True, relative_path=True,
                        exist_check=True):
    rel = pathlib.Path('training' if training else 'testing') / info_type / (get_image_index_str(idx) + file_tail)
    if exist_check and not (pathlib.Path(prefix) / rel).exists():
        raise ValueError("file not exist: {}".format(rel))
    return str(rel) if relative_path else str(pathlib.Path(prefix) / rel)


def get_image_path(idx, prefix, training=True, relative_path=True, exist_check=True):
    return get_kitti_info_path(idx, prefix, 'image_2', '.png', training, relative_path, exist_check)


def get_label_path(idx, prefix, training=True, relative_path=True, exist_check=True):
    return get_kitti_info_path(idx, prefix, 'label_2', '.txt', training, relative_path, exist_check)


def get_velodyne_path(idx, prefix, training=True, relative_path=True, exist_check=True):
    return get_kitti_info_path(idx, prefix, 'velodyne', '.bin', training, relative_path, exist_check)


def get_calib_path(idx, prefix, training=True, relative_path=True, exist_check=True):
    return get_kitti_info_path(idx, prefix, 'calib', '.txt', training, relative_path, exist_check)


def add_difficulty_to_annos(info):
    """annos['difficulty'] [n] int32: 0 easy / 1 moderate / 2 hard / -1 none, from image-box height, occlusion, truncation."""
    annos = info['annos']
    height = annos['bbox'][:, 3] - annos['bbox'][:, 1]
    occ, trunc = np.asarray(annos['occluded']), np.asarray(annos['truncated'])
    ok = [~((occ > o) | (height <= h) | (trunc > t)) for h, o, t in zip((40, 25, 25), (0, 1, 2), (0.15, 0.3, 0.5))]
    diff = np.full(len(height), -1, dtype=np.int32)
    diff[ok[2] ^ ok[1]] = 2
    diff[ok[0] ^ ok[1]] = 1
    diff[ok[0]] = 0
    annos["difficulty"] = diff
    return diff.tolist()


def _calib_rows(path):
    with open(path, 'r') as f:
        lines = f.readlines()
    return lambda i, n: np.array([float(v) for v in lines[i].split(' ')[1:n + 1]])


def get_kitti_image_info(path, training=True, label_info=True, velodyne=False, calib=False, image_ids=7481,
                         extend_matrix=True, num_worker=8, relative_path=True, with_imageshape=True):
    """One dict per frame: image_idx, pointcloud_num_features, velodyne_path, img_path, img_shape, calib/P0..P3,
    calib/R0_rect, calib/Tr_velo_to_cam, calib/Tr_imu_to_velo (4x4 when extend_matrix), annos (+ difficulty)."""
    root = pathlib.Path(path)
    if not isinstance(image_ids, list):
        image_ids = list(range(image_ids))
    infos = []
    for idx in image_ids:
        info = {'image_idx': idx, 'pointcloud_num_features': 4}
        if velodyne:
            info['velodyne_path'] = get_velodyne_path(idx, path, training, relative_path)
        info['img_path'] = get_image_path(idx, path, training, relative_path)
        if with_imageshape:
            img = info['img_path']
            info['img_shape'] = np.array(png_shape(str(root / img) if relative_path else img), dtype=np.int32)
        annotations = None
        if label_info:
            label = get_label_path(idx, path, training, relative_path)
            annotations = get_label_anno(str(root / label) if relative_path else label)
        if calib:
            row = _calib_rows(get_calib_path(idx, path, training, relative_path=False))
            pad = lambda m: np.concatenate([m, np.array([[0., 0., 0., 1.]])], axis=0) if extend_matrix else m
            for k in range(4):
                info['calib/P%d' % k] = pad(row(k, 12).reshape([3, 4]))
            r0 = row(4, 9).reshape([3, 3])
            if extend_matrix:
                r4 = np.zeros([4, 4], dtype=r0.dtype)
                r4[3, 3], r4[:3, :3] = 1., r0
                r0 = r4
            info['calib/R0_rect'] = r0
            info['calib/Tr_velo_to_cam'] = pad(row(5, 12).reshape([3, 4]))
            info['calib/Tr_imu_to_velo'] = pad(row(6, 12).reshape([3, 4]))
        if annotations is not None:
            info['annos'] = annotations
            add_difficulty_to_annos(info)
        infos.append(info)
    return infos
