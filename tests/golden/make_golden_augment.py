"""Generate tests/golden/augment_ref.npz by running the reference's OWN training-side augmentation code
 on a seeded synthetic scene and ground-truth database (build container only; nothing is copied):

  * mmdet/core/point_cloud/point_augmentor.py (PointAugmentor, BatchSampler, noise_per_box, points_transform_, ...),
    mmdet/core/bbox3d/geometry.py (box_collision_test, points_in_rbbox, remove_outside_points, ...) and
    mmdet/datasets/kitti_utils.py are imported with `numba` replaced by an identity-decorator stub, so the jitted loops
    run as plain Python.
  * One textual substitution is applied to geometry.py IN MEMORY before it is executed: `is True` / `is False` ->
    `== True` / `== False` inside box_collision_test.  Under numba (the reference's runtime) `x is True` on a boolean is
    compiled as an equality test, so the "one box completely inside the other" branch runs; under plain Python the
    same expression on a numpy bool_ is always False and that branch would be dead.  The substitution restores the
    numba behaviour.

    python tests/golden/make_golden_augment.py
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import augment_synth                                                 # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _load(name, rel, patch=None):
    path = os.path.join(REF, rel)
    src = open(path).read()
    if patch:
        src = patch(src)
    m = types.ModuleType(name)
    m.__file__ = path
    sys.modules[name] = m
    exec(compile(src, path, "exec"), m.__dict__)
    return m


def load_reference():
    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    _stub("numba", jit=jit, njit=jit, prange=range)
    _stub("cv2")
    for pkg in ("mmdet", "mmdet.core", "mmdet.core.bbox3d", "mmdet.core.point_cloud", "mmdet.datasets"):
        _stub(pkg)

    def numba_bool_identity(src):
        assert src.count(" is True") == 1 and src.count(" is False") == 4, "geometry.py changed"
        return src.replace(" is True", " == True").replace(" is False", " == False")
    geo = _load("mmdet.core.bbox3d.geometry", "mmdet/core/bbox3d/geometry.py", numba_bool_identity)
    ku = _load("mmdet.datasets.kitti_utils", "mmdet/datasets/kitti_utils.py")
    pa = _load("mmdet.core.point_cloud.point_augmentor", "mmdet/core/point_cloud/point_augmentor.py")
    return geo, ku, pa


def main():
    geo, ku, pa = load_reference()
    out = {}
    r = np.random.default_rng(11)

    # ---- box_collision_test: random boxes + contained / identical / touching cases, float64 and float32 corners ----
    a = augment_synth.bev_boxes(r, 24, spread=14.0)
    b = augment_synth.bev_boxes(r, 20, spread=14.0)
    b[0] = a[0]                                                    # identical
    b[1] = a[1]; b[1, 2:4] *= 0.4                                  # b inside a, no edge crossing
    b[2] = a[2]; b[2, 2:4] *= 2.5                                  # a inside b
    b[3] = a[3]; b[3, 0] += a[3, 2]                                # shifted by exactly one width (touching when axis aligned)
    a[3, 4] = b[3, 4] = 0.0
    b[4] = a[4]; b[4, 4] += np.pi / 2                              # crossed
    for dt in (np.float64, np.float32):
        ca = geo.center_to_corner_box2d(a[:, :2].astype(dt), a[:, 2:4].astype(dt), a[:, 4].astype(dt))
        cb = geo.center_to_corner_box2d(b[:, :2].astype(dt), b[:, 2:4].astype(dt), b[:, 4].astype(dt))
        tag = "f64" if dt is np.float64 else "f32"
        out["coll_a_" + tag], out["coll_b_" + tag] = ca, cb
        out["coll_ab_" + tag] = geo.box_collision_test(ca, cb)
        out["coll_aa_" + tag] = geo.box_collision_test(ca, ca)
    out["coll_boxes_a"], out["coll_boxes_b"] = a, b
    assert out["coll_ab_f64"][1, 1] and out["coll_ab_f64"][2, 2], "containment must count as a collision"

    # ---- points in rotated boxes / in the camera frustum ---------------------------------------------------------------
    boxes32 = augment_synth.lidar_boxes(r, 14).astype(np.float32)
    boxes64 = augment_synth.lidar_boxes(r, 9)
    pts = augment_synth.scene_points(5, boxes=np.concatenate([boxes32.astype(np.float64), boxes64]))
    out["pts"] = pts
    out["pib_boxes32"], out["pib_boxes64"] = boxes32, boxes64
    out["pib_mask32"] = np.packbits(geo.points_in_rbbox(pts, boxes32))
    out["pib_mask64"] = np.packbits(geo.points_in_rbbox(pts, boxes64))
    out["pib_count32"] = geo.points_in_rbbox(pts, boxes32).sum(0)
    assert out["pib_count32"].sum() > 500
    calib = augment_synth.calib_matrices()
    rect4, trv2c4, p24 = (augment_synth.extend(calib[k]) for k in ("R0_rect", "Tr_velo_to_cam", "P2"))
    full = augment_synth.full_sweep(6)
    kept = geo.remove_outside_points(full, rect4, trv2c4, p24, (375, 1242))
    out["fov_points"] = full
    c, rr, t = geo.projection_matrix_to_CRT_kitti(p24)
    fr = geo.get_frustum([0, 0, 1242, 375], c)
    fr -= t
    fr = np.linalg.inv(rr) @ fr.T
    fr = geo.camera_to_lidar(fr.T, rect4, trv2c4)
    surf = geo.corner_to_surfaces_3d_jit(fr[np.newaxis, ...])
    out["fov_frustum"] = fr
    out["fov_mask"] = np.packbits(geo.points_in_convex_polygon_3d_jit(full[:, :3], surf).reshape(-1))
    out["fov_count"] = np.array(len(kept))
    cam_boxes = np.concatenate([r.uniform(-10, 10, (6, 1)), r.uniform(1, 2, (6, 1)), r.uniform(5, 40, (6, 1)),
                                r.uniform(1, 4, (6, 3)), r.uniform(-3, 3, (6, 1))], 1)
    out["cam_boxes"], out["cam_boxes_lidar"] = cam_boxes, geo.box_camera_to_lidar(cam_boxes, rect4, trv2c4)
    gtb = augment_synth.lidar_boxes(r, 30).astype(np.float32)
    gtb[:6, 0] += 60.0                                             # some outside the range
    out["range_boxes"] = gtb
    out["range_mask"] = geo.filter_gt_box_outside_range(gtb, np.array([0, -40.0, 70.4, 40.0]))

    # ---- noise_per_box / points_transform_ / box3d_transform_ with given noise -------------------------------------------
    nb = augment_synth.lidar_boxes(r, 16, spread=18.0).astype(np.float32)
    valid = np.ones(16, dtype=np.bool_)
    valid[5] = False
    loc_n = r.normal(scale=[1.0, 1.0, 0.5], size=[16, 100, 3])
    rot_n = r.uniform(-0.78539816, 0.78539816, size=[16, 100])
    sel = pa.noise_per_box(nb[:, [0, 1, 3, 4, 6]], valid, loc_n, rot_n)
    out["npb_boxes"], out["npb_valid"], out["npb_loc"], out["npb_rot"], out["npb_sel"] = nb, valid, loc_n, rot_n, sel
    assert (sel > 0).any() and (sel == 0).any(), "want both first-try and later-try acceptances"
    loc_t, rot_t = pa.select_transform(loc_n, sel), pa.select_transform(rot_n, sel)
    corners = geo.center_to_corner_box3d(nb, origin=[0.5, 0.5, 0], axis=2)
    pm = geo.points_in_convex_polygon_3d_jit(pts[:, :3], geo.corner_to_surfaces_3d_jit(corners))
    moved = pts.copy()
    pa.points_transform_(moved, nb[:, :3], pm, loc_t, rot_t, valid)
    nb2 = nb.copy()
    pa.box3d_transform_(nb2, loc_t, rot_t, valid)
    out["npb_point_mask"], out["npb_points_out"], out["npb_boxes_out"] = np.packbits(pm), moved, nb2
    out["npb_moved_count"] = np.array(int((moved != pts).any(1).sum()))

    # ---- the whole augmentor on three consecutive frames (sampler state carries over) -------------------------------------
    db = augment_synth.make_database(seed=2)
    out.update(augment_synth.pack_database(db))
    with tempfile.TemporaryDirectory() as tmp:
        augment_synth.write_database(db, tmp)
        for cfg_name, cfg in augment_synth.AUGMENTOR_CONFIGS.items():
            np.random.seed(1234)
            aug = pa.PointAugmentor(root_path=tmp, info_path=os.path.join(tmp, "kitti_dbinfos_train.pkl"), **cfg)
            ku_calib = ku.Calibration.__new__(ku.Calibration)
            ku_calib.V2C, ku_calib.R0 = calib["Tr_velo_to_cam"].reshape(3, 4), calib["R0_rect"].reshape(3, 3)
            ku_calib.C2V = np.zeros_like(ku_calib.V2C)
            ku_calib.C2V[:, :3] = ku_calib.V2C[:, :3].T
            ku_calib.C2V[:, 3] = -ku_calib.V2C[:, :3].T @ ku_calib.V2C[:, 3]
            for f in range(3):
                points, gt_boxes, gt_types = augment_synth.frame(f)
                plane = augment_synth.PLANE if cfg_name == "multi" else None
                tag = "aug_%s_%d_" % (cfg_name, f)
                s_boxes, s_types, s_points = aug.sample_all(gt_boxes, gt_types, plane, ku_calib)
                out[tag + "s_boxes"], out[tag + "s_types"], out[tag + "s_points"] = \
                    s_boxes, np.array("\n".join(s_types)), s_points
                boxes = np.concatenate([gt_boxes, s_boxes])
                types_ = gt_types + s_types
                masks = geo.points_in_rbbox(points, s_boxes)
                points = points[np.logical_not(masks.any(-1))]
                points = np.concatenate([s_points, points], axis=0)
                types_ = np.array(['Car' if n == 'Van' else n for n in types_])
                sel_cls = [i for i in range(len(types_)) if types_[i] in cfg["sample_classes"]]
                boxes, types_ = boxes[sel_cls, :], types_[sel_cls]
                out[tag + "pre_boxes"], out[tag + "pre_points"] = boxes.copy(), points.copy()
                aug.noise_per_object_(boxes, points, num_try=100)
                out[tag + "noise_boxes"], out[tag + "noise_points"] = boxes.copy(), points.copy()
                boxes, points = aug.random_flip(boxes, points)
                out[tag + "flip_boxes"] = boxes.copy()
                boxes, points = aug.global_rotation(boxes, points)
                boxes, points = aug.global_scaling(boxes, points)
                out[tag + "out_boxes"], out[tag + "out_points"], out[tag + "out_types"] = \
                    boxes, points, np.array("\n".join(types_))
                print(cfg_name, f, "sampled", len(s_types), "points", len(points), "boxes", len(boxes))

    np.savez_compressed(os.path.join(HERE, "augment_ref.npz"), **out)
    print("augment_ref.npz: %d arrays, %.1f MB" % (len(out), os.path.getsize(os.path.join(HERE, "augment_ref.npz")) / 1e6))


if __name__ == "__main__":
    main()
