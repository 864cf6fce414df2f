"""Training-side augmentation against vectors produced by the reference's own PointAugmentor /
geometry code (tests/golden/make_golden_augment.py -> augment_ref.npz).  GPU-less: the host parts of the product (box
geometry, native collision / noise selection, sampler, random-number order) are tested directly; the per-point device
arithmetic is tested through the CPU harness that loops the product's __host__ __device__ functions
(tests/harness.py).  tests/test_gpu_train.py runs the same vectors through the real kernels."""
import os

import numpy as np
import pytest
import torch

import sassd  # noqa: F401
from sassd import geometry as G
from sassd import kitti_common as kc
from sassd import point_augmentor as PA

import augment_synth as S
import harness

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def R():
    return np.load(os.path.join(HERE, "golden", "augment_ref.npz"))


def _unpack(bits, n, m):
    return np.unpackbits(bits)[:n * m].reshape(n, m).astype(bool)


def test_box_collision(R):
    for tag in ("f64", "f32"):
        a, b = R["coll_a_" + tag], R["coll_b_" + tag]
        assert np.array_equal(G.box_collision_test(a, b), R["coll_ab_" + tag]), tag
        assert np.array_equal(G.box_collision_test(a, a), R["coll_aa_" + tag]), tag
        # corners come out of the same numpy arithmetic as the reference's
        box = R["coll_boxes_a"].astype(a.dtype)
        assert np.array_equal(G.center_to_corner_box2d(box[:, :2], box[:, 2:4], box[:, 4]), a)
    got = G.box_collision_test(R["coll_a_f64"], R["coll_b_f64"])
    assert got[1, 1] and got[2, 2] and got[4, 4]                    # b inside a, a inside b, crossed
    assert not got[0, 0]      # reference quirk: exactly coincident boxes neither cross (collinear edges) nor strictly contain
    assert G.box_collision_test(np.zeros((0, 4, 2)), R["coll_b_f64"]).shape == (0, 20)


def test_host_geometry(R):
    c = S.calib_matrices()
    rect, trv2c, p2 = (S.extend(c[k]) for k in ("R0_rect", "Tr_velo_to_cam", "P2"))
    assert np.array_equal(G.box_camera_to_lidar(R["cam_boxes"], rect, trv2c), R["cam_boxes_lidar"])
    assert np.array_equal(G.frustum_in_lidar(rect, trv2c, p2, (375, 1242)), R["fov_frustum"])
    assert np.array_equal(G.filter_gt_box_outside_range(R["range_boxes"], np.array([0, -40.0, 70.4, 40.0])), R["range_mask"])
    assert 0 < R["range_mask"].sum() < len(R["range_mask"])


def test_points_in_boxes_device_arithmetic(R, monkeypatch):
    harness.patch(monkeypatch)
    pts = R["pts"]
    for tag in ("32", "64"):
        boxes = R["pib_boxes" + tag]
        got = G.points_in_rbbox(pts, boxes)
        assert got.dtype == np.bool_ and np.array_equal(got, _unpack(R["pib_mask" + tag], len(pts), len(boxes))), tag
    assert np.array_equal(G.points_in_rbbox(pts, R["pib_boxes32"]).sum(0), R["pib_count32"])
    c = S.calib_matrices()
    rect, trv2c, p2 = (S.extend(c[k]) for k in ("R0_rect", "Tr_velo_to_cam", "P2"))
    full = R["fov_points"]
    kept = G.remove_outside_points(full, rect, trv2c, p2, (375, 1242))
    assert np.array_equal(kept, full[_unpack(R["fov_mask"], len(full), 1)[:, 0]]) and len(kept) == int(R["fov_count"])
    assert 200 < len(kept) < len(full) // 2
    assert G.points_in_rbbox(pts[:0], R["pib_boxes32"]).shape == (0, 14)
    assert G.points_in_rbbox(pts, R["pib_boxes32"][:0]).shape == (len(pts), 0)


def test_noise_per_box_and_point_move(R, monkeypatch):
    harness.patch(monkeypatch)
    boxes, valid = R["npb_boxes"], R["npb_valid"]
    sel = PA.noise_per_box(boxes[:, [0, 1, 3, 4, 6]], valid, R["npb_loc"], R["npb_rot"])
    assert np.array_equal(sel, R["npb_sel"]) and sel[5] == -1
    loc_t, rot_t = PA.select_transform(R["npb_loc"], sel), PA.select_transform(R["npb_rot"], sel)
    aug = PA.PointAugmentor.__new__(PA.PointAugmentor)
    aug.device = torch.device("cpu")
    pts = torch.from_numpy(R["pts"].copy())
    aug.move_points(pts, boxes, loc_t, rot_t, valid)
    want = R["npb_points_out"]
    assert np.abs(pts.numpy() - want).max() < 4e-6            # (the reference rotates with a BLAS float32 product)
    moved = (pts.numpy() != R["pts"]).any(1)
    assert moved.sum() == int(R["npb_moved_count"]) and np.array_equal(moved, (want != R["pts"]).any(1))
    b2 = boxes.copy()
    PA.box3d_transform_(b2, loc_t, rot_t, valid)
    assert np.array_equal(b2, R["npb_boxes_out"])


@pytest.mark.parametrize("cfg_name", ["car", "multi"])
def test_augment_frames(R, cfg_name, tmp_path, monkeypatch):
    """Three consecutive training frames through PointAugmentor with the reference's seed: the same database objects are
    pasted, the same noise draws accepted, the same flip / rotation / scale drawn; boxes agree to float32 rounding, the
    point clouds row for row."""
    harness.patch(monkeypatch)
    S.write_database(S.unpack_database(R), str(tmp_path))
    cfg = S.AUGMENTOR_CONFIGS[cfg_name]
    calib = kc.Calibration(matrices=S.calib_matrices())
    np.random.seed(1234)
    aug = PA.PointAugmentor(root_path=str(tmp_path), info_path=str(tmp_path / "kitti_dbinfos_train.pkl"), device="cpu",
                            **cfg)
    np.random.seed(1234)
    step = PA.PointAugmentor(root_path=str(tmp_path), info_path=str(tmp_path / "kitti_dbinfos_train.pkl"), device="cpu",
                             **cfg)
    state_after_init = np.random.get_state()
    for f in range(3):
        points, gt_boxes, gt_types = S.frame(f)
        plane = S.PLANE if cfg_name == "multi" else None
        tag = "aug_%s_%d_" % (cfg_name, f)

        # (a) the reference's method-by-method calling convention, numpy in / numpy out
        np.random.set_state(state_after_init)
        s_boxes, s_types, s_points = step.sample_all(gt_boxes, gt_types, plane, calib)
        assert np.array_equal(s_boxes, R[tag + "s_boxes"]) and s_boxes.dtype == np.float32
        assert "\n".join(s_types) == str(R[tag + "s_types"])
        assert np.array_equal(s_points, R[tag + "s_points"])
        boxes, pts = R[tag + "pre_boxes"].copy(), R[tag + "pre_points"].copy()
        step.noise_per_object_(boxes, pts, num_try=100)
        assert np.abs(boxes - R[tag + "noise_boxes"]).max() < 1e-5
        assert np.abs(pts - R[tag + "noise_points"]).max() < 4e-6
        boxes, pts = step.random_flip(boxes, pts)
        assert np.abs(boxes - R[tag + "flip_boxes"]).max() < 1e-5
        boxes, pts = step.global_rotation(boxes, pts)
        boxes, pts = step.global_scaling(boxes, pts)
        assert np.abs(boxes - R[tag + "out_boxes"]).max() < 2e-5
        assert np.abs(pts - R[tag + "out_points"]).max() < 2e-5
        state_after_frame = np.random.get_state()

        # (b) the fused frame recipe consumes the random stream identically and gives the same frame
        np.random.set_state(state_after_init)
        out_pts, out_boxes, out_types, out_labels = aug.augment_frame(torch.from_numpy(points), gt_boxes.copy(), gt_types,
                                                                      cfg["sample_classes"], plane, calib)
        assert "\n".join(out_types) == str(R[tag + "out_types"])
        assert np.array_equal(out_labels, [cfg["sample_classes"].index(t) + 1 for t in out_types])
        assert np.abs(out_boxes - R[tag + "out_boxes"]).max() < 2e-5
        assert out_pts.shape == R[tag + "out_points"].shape
        assert np.abs(out_pts.numpy() - R[tag + "out_points"]).max() < 2e-5
        s1 = np.random.get_state()
        assert s1[2] == state_after_frame[2] and np.array_equal(s1[1], state_after_frame[1])
        state_after_init = state_after_frame


def test_device_only(R):
    """without the harness patch the per-point entry points refuse CPU tensors / a GPU-less box"""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        G.points_in_rbbox(R["pts"], R["pib_boxes32"])
    with pytest.raises(RuntimeError):
        PA._points_global(torch.zeros(4, 4), 0, 0.0, 1.0, 1.0)
