"""Generate tests/golden/create_data_ref.npz by running the reference's OWN data preparation (tools/create_data.py:
create_kitti_info_file, create_reduced_point_cloud, create_groundtruth_database, with tools/kitti_common.py and
mmdet/core/bbox3d/geometry.py) on the tiny synthetic raw-KITTI tree of tests/augment_synth.py::write_kitti_tree
(build container only; nothing is copied).  Stubs: numba -> identity decorators; tqdm -> identity; imageio.imread ->
an object carrying the PNG header's shape; `np.bool` (removed from NumPy) -> bool; and the same `is True` -> `== True`
substitution as make_golden_augment.py (not exercised on this path).

    python tests/golden/make_golden_create_data.py
"""
import hashlib
import os
import pickle
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import augment_synth as S                                            # noqa: E402
from make_golden_augment import _load, _stub                         # noqa: E402


def digest(path):
    with open(path, "rb") as f:
        return hashlib.sha1(f.read()).hexdigest()


def flatten_infos(infos, tag, out):
    out[tag + "count"] = np.array(len(infos))
    for i, info in enumerate(infos):
        for k, v in info.items():
            if k == "annos":
                for ak, av in v.items():
                    out["%s%d_annos_%s" % (tag, i, ak)] = np.array("\n".join(av)) if ak == "name" else np.asarray(av)
            else:
                out["%s%d_%s" % (tag, i, k.replace("/", "_"))] = np.asarray(v)


def main():
    class _Img:
        def __init__(self, path):
            with open(path, "rb") as f:
                head = f.read(24)
            self.shape = (int.from_bytes(head[20:24], "big"), int.from_bytes(head[16:20], "big"), 3)

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    _stub("numba", jit=jit, njit=jit, prange=range)
    _stub("cv2"), _stub("imageio", imread=_Img), _stub("tqdm", tqdm=lambda x: x)
    for pkg in ("mmdet", "mmdet.core", "mmdet.core.bbox3d", "mmdet.datasets", "tools"):
        _stub(pkg)
    np.bool = bool                                                   # noqa: the reference predates its removal
    _load("mmdet.core.bbox3d.geometry", "mmdet/core/bbox3d/geometry.py",
          lambda s: s.replace(" is True", " == True").replace(" is False", " == False"))
    kc = _load("tools.kitti_common", "tools/kitti_common.py")
    sys.modules["tools"].kitti_common = kc
    cd = _load("tools.create_data", "tools/create_data.py", lambda s: s.split("if __name__ == '__main__':")[0])

    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        S.write_kitti_tree(tmp)
        os.makedirs(os.path.join(tmp, "training", "velodyne_reduced"))
        os.makedirs(os.path.join(tmp, "testing", "velodyne_reduced"))
        cd.create_kitti_info_file(tmp)
        cd.create_reduced_point_cloud(tmp)
        cd.create_groundtruth_database(tmp)
        ku = _load("mmdet.datasets.kitti_utils", "mmdet/datasets/kitti_utils.py")
        for idx in sorted(S.TREE_IMG_HW):                        # the label -> lidar-frame boxes step of kitti.py:147-154
            objects = ku.read_label(os.path.join(tmp, "training", "label_2", "%06d.txt" % idx))
            boxes = np.array([o.box3d for o in objects if o.type not in ["DontCare"]], dtype=np.float32)
            calib = ku.Calibration(os.path.join(tmp, "training", "calib", "%06d.txt" % idx))
            if len(boxes) != 0:
                boxes[:, :3] = ku.project_rect_to_velo(boxes[:, :3], calib)
            out["frame%d_gt_bboxes" % idx] = boxes.reshape(-1, 7)
            out["frame%d_gt_types" % idx] = np.array("\n".join(o.type for o in objects if o.type != "DontCare"))
        for name in ("train", "val", "trainval", "test"):
            with open(os.path.join(tmp, "kitti_infos_%s.pkl" % name), "rb") as f:
                flatten_infos(pickle.load(f), "infos_%s_" % name, out)
        files = {}
        for sub in ("training/velodyne_reduced", "testing/velodyne_reduced", "gt_database"):
            for fn in sorted(os.listdir(os.path.join(tmp, sub))):
                p = os.path.join(tmp, sub, fn)
                files[sub + "/" + fn] = "%s %d" % (digest(p), os.path.getsize(p))
        out["files"] = np.array("\n".join("%s %s" % kv for kv in sorted(files.items())))
        with open(os.path.join(tmp, "kitti_dbinfos_train.pkl"), "rb") as f:
            db = pickle.load(f)
        out["db_classes"] = np.array("\n".join(db.keys()))
        for cls, infos in db.items():
            out["db_%s_count" % cls] = np.array(len(infos))
            if infos:
                out["db_%s_path" % cls] = np.array("\n".join(i["path"] for i in infos))
                out["db_%s_box" % cls] = np.stack([i["box3d_lidar"] for i in infos])
                out["db_%s_meta" % cls] = np.array([[i["image_idx"], i["gt_idx"], i["num_points_in_gt"], i["difficulty"],
                                                     i["group_id"]] for i in infos], dtype=np.int64)
                out["db_%s_score" % cls] = np.array([i["score"] for i in infos])
    np.savez_compressed(os.path.join(HERE, "create_data_ref.npz"), **out)
    print(str(out["files"]))
    print("create_data_ref.npz: %d arrays; db:" % len(out), {c: int(out["db_%s_count" % c]) for c in str(out["db_classes"]).split("\n")})


if __name__ == "__main__":
    main()
