"""Offline data preparation against the artefacts the reference's own tools/create_data.py produced for
the same tiny raw-KITTI tree (tests/golden/make_golden_create_data.py -> create_data_ref.npz): info records field by
field, velodyne_reduced and gt_database files byte for byte (SHA-1), database infos.  GPU-less: the per-point kernel is
replaced by the CPU harness over the product's own __host__ __device__ code (tests/harness.py); test_gpu_train.py repeats
the file comparison with the real kernels."""
import hashlib
import os
import pickle

import numpy as np
import pytest
import torch

import sassd  # noqa: F401
from sassd import create_data as CD
from sassd import kitti_common as kc

import augment_synth as S
import harness

HERE = os.path.dirname(os.path.abspath(__file__))


def digest(path):
    with open(path, "rb") as f:
        return hashlib.sha1(f.read()).hexdigest()


def check_tree(root, R):
    """compare everything under `root` with the reference's outputs"""
    for name in ("train", "val", "trainval", "test"):
        with open(os.path.join(root, "kitti_infos_%s.pkl" % name), "rb") as f:
            infos = pickle.load(f)
        tag = "infos_%s_" % name
        assert len(infos) == int(R[tag + "count"])
        for i, info in enumerate(infos):
            want_keys = sorted(k[len("%s%d_" % (tag, i)):] for k in R.files
                               if k.startswith("%s%d_" % (tag, i)) and "_annos_" not in k)
            assert sorted(k.replace("/", "_") for k in info if k != "annos") == want_keys, (name, i)
            for k, v in info.items():
                if k == "annos":
                    akeys = sorted(x[len("%s%d_annos_" % (tag, i)):] for x in R.files
                                   if x.startswith("%s%d_annos_" % (tag, i)))
                    assert sorted(v.keys()) == akeys, (name, i)
                    for ak, av in v.items():
                        want = R["%s%d_annos_%s" % (tag, i, ak)]
                        if ak == "name":
                            assert "\n".join(av) == str(want)
                        else:
                            assert np.array_equal(np.asarray(av), want) and np.asarray(av).dtype == want.dtype, (name, i, ak)
                else:
                    want = R["%s%d_%s" % (tag, i, k.replace("/", "_"))]
                    assert np.array_equal(np.asarray(v), want) and np.asarray(v).dtype == want.dtype, (name, i, k)
    files = {}
    for sub in ("training/velodyne_reduced", "testing/velodyne_reduced", "gt_database"):
        for fn in sorted(os.listdir(os.path.join(root, sub))):
            p = os.path.join(root, sub, fn)
            files[sub + "/" + fn] = "%s %d" % (digest(p), os.path.getsize(p))
    assert "\n".join("%s %s" % kv for kv in sorted(files.items())) == str(R["files"])
    with open(os.path.join(root, "kitti_dbinfos_train.pkl"), "rb") as f:
        db = pickle.load(f)
    assert "\n".join(db.keys()) == str(R["db_classes"])
    total = 0
    for cls, infos in db.items():
        assert len(infos) == int(R["db_%s_count" % cls])
        if infos:
            assert "\n".join(i["path"] for i in infos) == str(R["db_%s_path" % cls])
            assert np.array_equal(np.stack([i["box3d_lidar"] for i in infos]), R["db_%s_box" % cls])
            meta = np.array([[i["image_idx"], i["gt_idx"], i["num_points_in_gt"], i["difficulty"], i["group_id"]]
                             for i in infos], dtype=np.int64)
            assert np.array_equal(meta, R["db_%s_meta" % cls])
            assert np.array_equal([i["score"] for i in infos], R["db_%s_score" % cls])
            assert sorted(infos[0].keys()) == sorted(["name", "path", "image_idx", "gt_idx", "box3d_lidar",
                                                      "num_points_in_gt", "difficulty", "group_id", "score"])
            total += len(infos)
    assert total >= 10


def run_preparation(root, device):
    S.write_kitti_tree(root)
    CD.create_kitti_info_file(root, device=device)
    CD.create_reduced_point_cloud(root, device=device)
    CD.create_groundtruth_database(root, device=device)


def test_png_shape_and_difficulty(tmp_path):
    S.write_png(str(tmp_path / "a.png"), 37, 1242)
    assert kc.png_shape(str(tmp_path / "a.png")) == (37, 1242)
    (tmp_path / "b.png").write_bytes(b"not a png at all, but long enough to read")
    with pytest.raises(ValueError):
        kc.png_shape(str(tmp_path / "b.png"))


def test_create_data_matches_reference(tmp_path, monkeypatch):
    harness.patch(monkeypatch)
    R = np.load(os.path.join(HERE, "golden", "create_data_ref.npz"))
    run_preparation(str(tmp_path), torch.device("cpu"))
    check_tree(str(tmp_path), R)
    # the prepared tree feeds the augmentor
    from sassd.point_augmentor import PointAugmentor
    np.random.seed(0)
    aug = PointAugmentor(str(tmp_path), str(tmp_path / "kitti_dbinfos_train.pkl"), **dict(
        S.AUGMENTOR_CONFIGS["car"], sample_classes=["Van"], min_num_points=[2], sample_max_num=[3]), device="cpu")
    assert len(aug._samplers[0]._sampled_list) >= 1 and len(aug._db_points) > 0


def test_dataset_frames(tmp_path, monkeypatch):
    """KittiLiDAR over the prepared tree: labels -> lidar-frame boxes as the reference's read_label + Calibration give
    them; a training frame through the augmentor (device arithmetic via the CPU harness)."""
    harness.patch(monkeypatch)
    from sassd.kitti_dataset import get_dataset
    R = np.load(os.path.join(HERE, "golden", "create_data_ref.npz"))
    root = str(tmp_path)
    run_preparation(root, torch.device("cpu"))
    data_cfg = dict(type='KittiLiDAR', root=root + '/training/', ann_file=root + '/ImageSets/train.txt', img_prefix=None,
                    img_scale=(1242, 375), img_norm_cfg=dict(mean=[0, 0, 0], std=[1, 1, 1], to_rgb=True), size_divisor=32,
                    flip_ratio=0.5, with_mask=False, with_label=True, with_point=True, class_names=['Car', 'Pedestrian'],
                    augmentor=dict(type='PointAugmentor', root_path=root + '/', info_path=root + '/kitti_dbinfos_train.pkl',
                                   sample_classes=['Van', 'Pedestrian'], min_num_points=[2, 2], sample_max_num=[4, 3],
                                   removed_difficulties=[-1], global_rot_range=[-0.78539816, 0.78539816],
                                   gt_rot_range=[-0.78539816, 0.78539816], center_noise_std=[1., 1., .5],
                                   scale_range=[0.95, 1.05]),
                    generator=dict(type='VoxelGenerator', voxel_size=[0.05, 0.05, 0.1],
                                   point_cloud_range=[0, -40., -3., 70.4, 40., 1.], max_num_points=5, max_voxels=20000),
                    anchor_generator=dict(Car=dict(type='AnchorGeneratorStride', sizes=[1.6, 3.9, 1.56],
                                                   anchor_strides=[0.4, 0.4, 1.0], anchor_offsets=[0.2, -39.8, -1.78],
                                                   rotations=[0, 1.57])),
                    anchor_area_threshold=1, out_size_factor=8, test_mode=False)
    np.random.seed(5)
    ds = get_dataset(data_cfg, device="cpu")
    assert len(ds) == 3 and ds.sample_ids == [0, 1, 3]
    assert ds.anchors['Car'].shape == (200 * 176 * 2, 7) and ds.anchors_bv['Car'].shape == (70400, 4)
    for i, idx in enumerate(ds.sample_ids):
        fr = ds.load_frame(i)
        assert np.array_equal(fr['gt_bboxes'], R["frame%d_gt_bboxes" % idx]) and fr['gt_bboxes'].dtype == np.float32
        assert "\n".join(fr['gt_types']) == str(R["frame%d_gt_types" % idx])
        assert fr['img_shape'] == (*S.TREE_IMG_HW[idx], 3)
        assert fr['points'].shape[1] == 4 and len(fr['points']) * 16 == os.path.getsize(
            os.path.join(root, 'training', 'velodyne_reduced', '%06d.bin' % idx))
    s = ds[0]
    assert s['points'].dtype == torch.float32 and s['points'].shape[1] == 4
    assert set(s['gt_types']) <= {'Car', 'Pedestrian'} and len(s['gt_types']) == len(s['gt_bboxes']) == len(s['gt_labels'])
    assert np.array_equal(s['gt_labels'].numpy(), [['Car', 'Pedestrian'].index(t) + 1 for t in s['gt_types']])
    yaw = s['gt_bboxes'][:, 6].numpy()
    assert (yaw >= -np.pi - 1e-6).all() and (yaw < np.pi + 1e-6).all()
    assert s['img_meta']['sample_idx'] == 0 and s['img_meta']['calib'].P2.shape == (3, 4)
    # frame 3 has no labels: with nothing pasted in range it yields None and __getitem__ draws another frame
    ds.augmentor = None
    assert ds.prepare_train_img(2) is None and ds[2] is not None
    test_cfg = dict(data_cfg, ann_file=root + '/ImageSets/val.txt', with_label=False, augmentor=None, test_mode=True)
    dv = get_dataset(test_cfg, device="cpu")
    t = dv[1]
    assert t['gt_bboxes'] is None and t['img_meta']['sample_idx'] == 5 and dv.anchors.shape == (70400, 7)
