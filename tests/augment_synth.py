"""Seeded synthetic inputs for the augmentation / data-preparation tests: lidar scenes, ground-truth
boxes, a small ground-truth database in the on-disk layout of tools/create_data.py (kitti_dbinfos_train.pkl +
gt_database/*.bin), calibration matrices.  numpy only; shared by tests/golden/make_golden_augment.py and the tests."""
import os
import pickle

import numpy as np

PLANE = np.array([0.012, -0.99988, 0.0096, 1.66])
PLANE = PLANE / np.linalg.norm(PLANE[:3])
SIZES = {"Car": (1.6, 3.9, 1.56), "Van": (1.9, 5.0, 2.2), "Pedestrian": (0.6, 0.8, 1.73), "Cyclist": (0.6, 1.76, 1.73),
         "Truck": (2.6, 10.0, 3.2)}                                     # w, l, h (lidar box order)
AUGMENTOR_CONFIGS = {
    "car": dict(sample_classes=["Car"], min_num_points=[5], sample_max_num=[15], removed_difficulties=[-1],
                global_rot_range=[-0.78539816, 0.78539816], gt_rot_range=[-0.78539816, 0.78539816],
                center_noise_std=[1., 1., .5], scale_range=[0.95, 1.05]),
    "multi": dict(sample_classes=["Car", "Pedestrian", "Cyclist"], min_num_points=[5, 5, 5], sample_max_num=[15, 10, 10],
                  removed_difficulties=[-1], global_rot_range=[-0.78539816, 0.78539816],
                  gt_rot_range=[-0.78539816, 0.78539816], center_noise_std=[1., 1., .5], scale_range=[0.95, 1.05]),
}


def calib_matrices():
    return {
        "P2": np.array([721.5377, 0, 609.5593, 44.85728, 0, 721.5377, 172.854, 0.2163791, 0, 0, 1, 0.002745884]),
        "R0_rect": np.array([0.9999239, 0.00983776, -0.007445048, -0.009869795, 0.9999421, -0.004278459, 0.007402527,
                             0.004351614, 0.9999631]),
        "Tr_velo_to_cam": np.array([0.007533745, -0.9999714, -0.000616602, -0.004069766, 0.01480249, 0.0007280733,
                                    -0.9998902, -0.07631618, 0.9998621, 0.00752379, 0.01480755, -0.2717806]),
    }


def extend(m):
    """3x3 / 3x4 calibration matrix -> 4x4 homogeneous (tools/kitti_common.py _extend_matrix, R0 padded likewise)."""
    m = np.asarray(m, dtype=np.float64)
    out = np.eye(4)
    if m.size == 9:
        out[:3, :3] = m.reshape(3, 3)
    else:
        out[:3, :4] = m.reshape(3, 4)
    return out


def bev_boxes(r, n, spread=12.0):
    """[n,5] (x, y, w, l, angle) float64."""
    return np.stack([r.uniform(0, spread, n), r.uniform(0, spread, n), r.uniform(1.2, 2.2, n), r.uniform(3.0, 5.0, n),
                     r.uniform(-3.3, 3.3, n)], 1)


def lidar_boxes(r, n, spread=None, names=None):
    """[n,7] (x, y, z, w, l, h, yaw) float64, z at the box bottom, inside the KITTI range."""
    b = np.zeros((n, 7))
    b[:, 0] = r.uniform(4, 66, n) if spread is None else r.uniform(10, 10 + spread, n)
    b[:, 1] = r.uniform(-34, 34, n) if spread is None else r.uniform(-spread / 2, spread / 2, n)
    b[:, 2] = r.uniform(-1.9, -1.3, n)
    for i in range(n):
        b[i, 3:6] = np.array(SIZES[names[i] if names is not None else "Car"]) * r.uniform(0.9, 1.1, 3)
    b[:, 6] = r.uniform(-np.pi, np.pi, n)
    return b


def _cluster(r, box, k):
    """k points inside a lidar box (a little inside its faces)."""
    u = r.uniform(-0.46, 0.46, (k, 3)) * box[3:6]
    c, s = np.cos(box[6]), np.sin(box[6])
    x = c * u[:, 0] + s * u[:, 1]                                   # same sense as the reference's box corners
    y = -s * u[:, 0] + c * u[:, 1]
    return np.stack([x + box[0], y + box[1], u[:, 2] + box[2] + box[5] / 2, r.uniform(0, 1, k)], 1)


def scene_points(seed, boxes=None, n_ground=3500):
    r = np.random.default_rng(100 + seed)
    ground = np.stack([r.uniform(0, 70, n_ground), r.uniform(-40, 40, n_ground), r.normal(-1.7, 0.04, n_ground),
                       r.uniform(0, 1, n_ground)], 1)
    walls = np.stack([r.uniform(0, 70, 600), r.choice([-22.0, 19.0], 600) + r.normal(0, 0.1, 600), r.uniform(-1.7, 1.0, 600),
                      r.uniform(0, 1, 600)], 1)
    parts = [ground, walls]
    if boxes is None:
        boxes = lidar_boxes(np.random.default_rng(200 + seed), 10)
    for b in boxes:
        parts.append(_cluster(r, b, int(r.integers(10, 60))))
    pts = np.concatenate(parts).astype(np.float32)
    return pts[r.permutation(len(pts))]


def full_sweep(seed, n=6000):
    """points all around the sensor (before the camera-frustum reduction)."""
    r = np.random.default_rng(300 + seed)
    ang, rad = r.uniform(-np.pi, np.pi, n), r.uniform(2, 70, n)
    return np.stack([rad * np.cos(ang), rad * np.sin(ang), r.uniform(-2.5, 1.5, n), r.uniform(0, 1, n)], 1).astype(np.float32)


def frame(f):
    """-> (points [N,4] f32, gt_boxes [G,7] f32 lidar, gt_types list) of training frame f."""
    r = np.random.default_rng(400 + f)
    names = [["Car", "Car", "Van", "Pedestrian", "Car", "Cyclist", "Truck", "Car"],
             ["Car", "Pedestrian", "Pedestrian", "Car"],
             ["Cyclist", "Car", "Car", "Car", "Van", "Car", "Car", "Pedestrian", "Car", "Car", "Car", "Car"]][f % 3]
    boxes = lidar_boxes(r, len(names), names=names)
    return scene_points(f, boxes), boxes.astype(np.float32), list(names)


# ---- ground-truth database ----------------------------------------------------------------------------------------------

def make_database(seed=2, counts=(("Car", 40), ("Pedestrian", 25), ("Cyclist", 25))):
    """dict class -> list of db infos (tools/create_data.py:232-250) with the object's points (relative to the box
    centre, as stored in gt_database/*.bin) under the extra key 'points'."""
    r = np.random.default_rng(seed)
    db, group = {}, 0
    for name, n in counts:
        boxes = lidar_boxes(r, n, names=[name] * n)
        infos = []
        for i in range(n):
            k = int(r.integers(2, 70))
            pts = _cluster(r, boxes[i], k).astype(np.float32)
            pts[:, :3] -= boxes[i, :3].astype(np.float32)
            infos.append(dict(name=name, path="gt_database/%d_%s_%d.bin" % (1000 + i, name, i % 5), image_idx=1000 + i,
                              gt_idx=i % 5, box3d_lidar=boxes[i].copy(), num_points_in_gt=k,
                              difficulty=int(r.choice([0, 1, 2, -1], p=[0.4, 0.3, 0.2, 0.1])), group_id=group,
                              points=pts))
            group += 1
        db[name] = infos
    return db


def write_database(db, root):
    os.makedirs(os.path.join(root, "gt_database"), exist_ok=True)
    plain = {}
    for name, infos in db.items():
        plain[name] = []
        for info in infos:
            info["points"].tofile(os.path.join(root, info["path"]))
            plain[name].append({k: v for k, v in info.items() if k != "points"})
    with open(os.path.join(root, "kitti_dbinfos_train.pkl"), "wb") as f:
        pickle.dump(plain, f)


def pack_database(db):
    out = {}
    for name, infos in db.items():
        out["db_%s_boxes" % name] = np.stack([i["box3d_lidar"] for i in infos])
        out["db_%s_meta" % name] = np.array([[i["image_idx"], i["gt_idx"], i["num_points_in_gt"], i["difficulty"],
                                              i["group_id"]] for i in infos], dtype=np.int64)
        out["db_%s_points" % name] = np.concatenate([i["points"] for i in infos])
    return out


def unpack_database(npz, names=("Car", "Pedestrian", "Cyclist")):
    db = {}
    for name in names:
        boxes, meta, pts = npz["db_%s_boxes" % name], npz["db_%s_meta" % name], npz["db_%s_points" % name]
        infos, at = [], 0
        for b, m in zip(boxes, meta):
            k = int(m[2])
            infos.append(dict(name=name, path="gt_database/%d_%s_%d.bin" % (m[0], name, m[1]), image_idx=int(m[0]),
                              gt_idx=int(m[1]), box3d_lidar=b.copy(), num_points_in_gt=k, difficulty=int(m[3]),
                              group_id=int(m[4]), points=pts[at:at + k].copy()))
            at += k
        db[name] = infos
    return db


# ---- a tiny raw KITTI tree (tools/create_data.py input layout) -----------------------------------------------------------

CALIB_TXT = """P0: 7.215377e+02 0.000000e+00 6.095593e+02 0.000000e+00 0.000000e+00 7.215377e+02 1.728540e+02 0.000000e+00 0.000000e+00 0.000000e+00 1.000000e+00 0.000000e+00
P1: 7.215377e+02 0.000000e+00 6.095593e+02 -3.875744e+02 0.000000e+00 7.215377e+02 1.728540e+02 0.000000e+00 0.000000e+00 0.000000e+00 1.000000e+00 0.000000e+00
P2: 7.215377e+02 0.000000e+00 6.095593e+02 4.485728e+01 0.000000e+00 7.215377e+02 1.728540e+02 2.163791e-01 0.000000e+00 0.000000e+00 1.000000e+00 2.745884e-03
P3: 7.215377e+02 0.000000e+00 6.095593e+02 -3.395242e+02 0.000000e+00 7.215377e+02 1.728540e+02 2.199936e+00 0.000000e+00 0.000000e+00 1.000000e+00 2.729905e-03
R0_rect: 9.999239e-01 9.837760e-03 -7.445048e-03 -9.869795e-03 9.999421e-01 -4.278459e-03 7.402527e-03 4.351614e-03 9.999631e-01
Tr_velo_to_cam: 7.533745e-03 -9.999714e-01 -6.166020e-04 -4.069766e-03 1.480249e-02 7.280733e-04 -9.998902e-01 -7.631618e-02 9.998621e-01 7.523790e-03 1.480755e-02 -2.717806e-01
Tr_imu_to_velo: 9.999976e-01 7.553071e-04 -2.035826e-03 -8.086759e-01 -7.854027e-04 9.998898e-01 -1.482298e-02 3.195559e-01 2.024406e-03 1.482454e-02 9.998881e-01 -7.997231e-01
"""
TREE_SETS = {"train": [0, 1, 3], "val": [2, 5], "trainval": [0, 1, 3, 2, 5], "test": [0, 1]}
TREE_IMG_HW = {0: (375, 1242), 1: (370, 1224), 2: (374, 1238), 3: (376, 1241), 5: (375, 1242)}


def write_png(path, h, w):
    """a valid all-black 8-bit grayscale PNG of the given size (the data path only ever reads its header)."""
    import struct
    import zlib

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xffffffff)
    raw = b"".join(b"\x00" + b"\x00" * w for _ in range(h))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 1)) + chunk(b"IEND", b""))


def _camera_objects(r, n):
    """n labelled objects in the camera frame: (name, truncated, occluded, dims l h w, location, ry)."""
    names = ["Car", "Car", "Pedestrian", "Cyclist", "Van", "Truck", "Car", "Person_sitting"]
    out = []
    for _ in range(n):
        name = names[int(r.integers(0, len(names)))]
        w, l, h = np.array(SIZES.get(name, SIZES["Pedestrian"])) * r.uniform(0.9, 1.1, 3)
        out.append((name, float(r.choice([0.0, 0.0, 0.1, 0.4])), int(r.choice([0, 0, 1, 2, 3])), (l, h, w),
                    np.array([r.uniform(-12, 12), r.uniform(1.5, 1.8), r.uniform(7, 48)]), r.uniform(-np.pi, np.pi)))
    return out


def write_kitti_tree(root):
    """training/{image_2,label_2,calib,velodyne} for frames 0,1,2,3,5, testing/{image_2,calib,velodyne} for 0,1, ImageSets."""
    c = calib_matrices()
    rect4, v2c4, p2 = extend(c["R0_rect"]), extend(c["Tr_velo_to_cam"]), c["P2"].reshape(3, 4)
    to_lidar = np.linalg.inv((rect4 @ v2c4).T)
    os.makedirs(os.path.join(root, "ImageSets"), exist_ok=True)
    for k, ids in TREE_SETS.items():
        with open(os.path.join(root, "ImageSets", k + ".txt"), "w") as f:
            f.write("\n".join("%06d" % i for i in ids) + "\n")
    for split, ids in (("training", sorted(TREE_IMG_HW)), ("testing", TREE_SETS["test"])):
        for sub in ("image_2", "calib", "velodyne") + (("label_2",) if split == "training" else ()):
            os.makedirs(os.path.join(root, split, sub), exist_ok=True)
        for idx in ids:
            r = np.random.default_rng(700 + idx + (50 if split == "testing" else 0))
            h, w = TREE_IMG_HW[idx]
            write_png(os.path.join(root, split, "image_2", "%06d.png" % idx), h, w)
            with open(os.path.join(root, split, "calib", "%06d.txt" % idx), "w") as f:
                f.write(CALIB_TXT)
            objs = _camera_objects(r, int(r.integers(5, 10)) if idx != 3 else 0)
            lines, boxes = [], []
            for name, trunc, occ, (l, hh, ww), loc, ry in objs:
                lidar_c = (np.append(loc, 1.0) @ to_lidar)[:3]
                box = np.array([lidar_c[0], lidar_c[1], lidar_c[2], ww, l, hh, ry])   # same yaw number as box_camera_to_lidar
                boxes.append(box)
                corners = np.array([[sx * l / 2, -sy * hh, sz * ww / 2] for sx in (-1, 1) for sy in (0, 1) for sz in (-1, 1)])
                cs, sn = np.cos(ry), np.sin(ry)
                cam = np.stack([cs * corners[:, 0] + sn * corners[:, 2], corners[:, 1],
                                -sn * corners[:, 0] + cs * corners[:, 2]], 1) + loc
                uvw = np.concatenate([cam, np.ones((8, 1))], 1) @ p2.T
                uv = uvw[:, :2] / uvw[:, 2:3]
                bb = [max(uv[:, 0].min(), 0), max(uv[:, 1].min(), 0), min(uv[:, 0].max(), w - 1), min(uv[:, 1].max(), h - 1)]
                alpha = ry - np.arctan2(loc[0], loc[2])
                lines.append("%s %.2f %d %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f" % (
                    name, trunc, occ, alpha, bb[0], bb[1], bb[2], bb[3], hh, ww, l, loc[0], loc[1], loc[2], ry))
            for _ in range(int(r.integers(0, 3))):
                x1, y1 = r.uniform(0, 1000), r.uniform(100, 250)
                lines.append("DontCare -1 -1 -10 %.2f %.2f %.2f %.2f -1 -1 -1 -1000 -1000 -1000 -10" % (
                    x1, y1, x1 + r.uniform(40, 200), y1 + r.uniform(30, 90)))
            if split == "training":
                with open(os.path.join(root, split, "label_2", "%06d.txt" % idx), "w") as f:
                    f.write("\n".join(lines) + ("\n" if lines else ""))
            sweep = full_sweep(idx, n=3500)
            parts = [sweep]
            for b in boxes:                                   # the labelled objects return some points
                parts.append(_cluster(r, b, int(r.integers(3, 50))).astype(np.float32))
            np.concatenate(parts).astype(np.float32).tofile(os.path.join(root, split, "velodyne", "%06d.bin" % idx))
