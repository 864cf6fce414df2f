"""Generate tests/golden/kitti_eval_ref.npz by running the reference's OWN evaluation code on seeded synthetic
annotations (build container only; nothing is copied into the repo):

  * mmdet/core/evaluation/kitti_eval.py is imported unchanged with `numba` replaced by an identity-decorator stub (so
    compute_statistics_jit / fused_compute_statistics / image_box_overlap / d3_box_overlap_kernel run as plain Python)
    and with its one device dependency, `rotate_iou_gpu_eval` (numba.cuda), served by the CPU oracle's restatement
    (oracle/sassd_oracle.c orc_rotate_iou_eval, itself pinned on the reference's device functions by
    make_golden_eval.py) under the wrapper's dtype rules (float32 compute, result cast to the input dtype);
  * mmdet/core/bbox/transforms.py kitti_bbox2results, mmdet/datasets/kitti_utils.py Calibration and
    tools/kitti_common.py get_label_anno are imported unchanged (mmcv / cv2 / imageio stubbed: unused on these paths).

    python tests/golden/make_golden_kitti_eval.py          (a few minutes: the matching runs as interpreted Python)
"""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"

import kitti_synth                                                   # noqa: E402
from oracle import clib                                              # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def oracle_rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    return clib.rotate_iou_eval(boxes.astype(np.float32), query_boxes.astype(np.float32),
                                criterion).astype(boxes.dtype)


def load_reference():
    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    _stub("numba", jit=jit, njit=jit, prange=range)
    _stub("mmcv"), _stub("cv2"), _stub("imageio", imread=None)
    for pkg in ("mmdet", "mmdet.core", "mmdet.core.post_processing", "mmdet.core.bbox3d", "mmdet.core.bbox",
                "mmdet.core.evaluation", "mmdet.datasets", "tools"):
        _stub(pkg)
    _stub("mmdet.core.post_processing.rotate_nms_gpu", rotate_iou_gpu_eval=oracle_rotate_iou_gpu_eval)
    _load("mmdet.core.bbox3d.geometry", "mmdet/core/bbox3d/geometry.py")
    ku = _load("mmdet.datasets.kitti_utils", "mmdet/datasets/kitti_utils.py")
    kc = _load("tools.kitti_common", "tools/kitti_common.py")
    sys.modules["tools"].kitti_common = kc
    tr = _load("mmdet.core.bbox.transforms", "mmdet/core/bbox/transforms.py")
    ev = _load("mmdet.core.evaluation.kitti_eval", "mmdet/core/evaluation/kitti_eval.py")
    return ev, tr, ku, kc


CALIB_TXT = """P0: 7.215377e+02 0.000000e+00 6.095593e+02 0.000000e+00 0.000000e+00 7.215377e+02 1.728540e+02 0.000000e+00 0.000000e+00 0.000000e+00 1.000000e+00 0.000000e+00
P1: 7.215377e+02 0.000000e+00 6.095593e+02 -3.875744e+02 0.000000e+00 7.215377e+02 1.728540e+02 0.000000e+00 0.000000e+00 0.000000e+00 1.000000e+00 0.000000e+00
P2: 7.215377e+02 0.000000e+00 6.095593e+02 4.485728e+01 0.000000e+00 7.215377e+02 1.728540e+02 2.163791e-01 0.000000e+00 0.000000e+00 1.000000e+00 2.745884e-03
P3: 7.215377e+02 0.000000e+00 6.095593e+02 -3.395242e+02 0.000000e+00 7.215377e+02 1.728540e+02 2.199936e+00 0.000000e+00 0.000000e+00 1.000000e+00 2.729905e-03
R0_rect: 9.999239e-01 9.837760e-03 -7.445048e-03 -9.869795e-03 9.999421e-01 -4.278459e-03 7.402527e-03 4.351614e-03 9.999631e-01
Tr_velo_to_cam: 7.533745e-03 -9.999714e-01 -6.166020e-04 -4.069766e-03 1.480249e-02 7.280733e-04 -9.998902e-01 -7.631618e-02 9.998621e-01 7.523790e-03 1.480755e-02 -2.717806e-01
Tr_imu_to_velo: 9.999976e-01 7.553071e-04 -2.035826e-03 -8.086759e-01 -7.854027e-04 9.998898e-01 -1.482298e-02 3.195559e-01 2.024406e-03 1.482454e-02 9.998881e-01 -7.997231e-01
"""

LABEL_TXT = """Car 0.00 0 -1.58 587.01 173.33 614.12 200.12 1.65 1.67 3.64 -0.65 1.71 46.70 -1.59
Cyclist 0.00 0 -2.46 665.45 160.00 717.93 217.99 1.72 0.47 1.65 2.45 1.35 22.10 -2.35
Pedestrian 0.21 2 0.21 423.17 173.67 433.17 224.03 1.60 0.38 0.30 -5.87 1.63 23.11 -0.03
DontCare -1 -1 -10 503.89 169.71 590.61 190.13 -1 -1 -1 -1000 -1000 -1000 -10
DontCare -1 -1 -10 511.35 174.96 527.81 187.45 -1 -1 -1 -1000 -1000 -1000 -10
"""

RESULT_TXT = """Car 0.00 0 -1.5800 587.0100 173.3300 614.1200 200.1200 1.6500 1.6700 3.6400 -0.6500 1.7100 46.7000 -1.5900 0.8731
Pedestrian 0.00 0 0.2100 423.1700 173.6700 433.1700 224.0300 1.6000 0.3800 0.3000 -5.8700 1.6300 23.1100 -0.0300 0.1250
"""


def main():
    ev, tr, ku, kc = load_reference()
    out = {}
    gts, dts = kitti_synth.make_annos(56, seed=0)
    out.update(kitti_synth.pack(gts, "gt_"))
    out.update(kitti_synth.pack(dts, "dt_"))

    # --- pieces ---------------------------------------------------------------------------------------------------
    r = np.random.default_rng(3)
    for k, (n, num_gt) in enumerate([(57, 80), (5, 5), (1, 3), (200, 150)]):
        s = r.uniform(0, 1, n)
        out["thr%d_scores" % k], out["thr%d_num_gt" % k] = s.copy(), np.array(num_gt)
        out["thr%d_out" % k] = np.array(ev.get_thresholds(s.copy(), num_gt))
    for metric in (0, 1, 2):
        _, parted, n_a, n_b = ev.calculate_iou_partly(dts, gts, metric, 50)
        out["ov%d_last_part" % metric] = parted[-1]
        out["ov%d_part3" % metric] = parted[3]
    for crit in (-1, 0, 1, 2):
        a, b = np.concatenate([g["bbox"] for g in gts[50:]]), np.concatenate([d["bbox"] for d in dts[50:]])
        out["imgov_%d" % crit] = ev.image_box_overlap(a, b, crit)
        a, b = (np.concatenate([kc.anno_to_rbboxes(x) for x in xs[50:]]) for xs in (gts, dts))
        out["d3ov_%d" % crit] = ev.d3_box_overlap(a, b, crit)
    clean = []
    for cls in (0, 1, 2):
        for diff in (0, 1, 2):
            nv, ig, idt, dc = [], [], [], []
            for g, d in zip(gts, dts):
                a, b, c, e = ev.clean_data(g, d, cls, diff)
                nv.append(a), ig.extend(b), idt.extend(c), dc.extend(e)
            out["clean_%d_%d_nvalid" % (cls, diff)] = np.array(nv)
            out["clean_%d_%d_ign_gt" % (cls, diff)] = np.array(ig, dtype=np.int64)
            out["clean_%d_%d_ign_dt" % (cls, diff)] = np.array(idt, dtype=np.int64)
            out["clean_%d_%d_dc" % (cls, diff)] = np.array(dc, dtype=np.float64).reshape(-1, 4)

    # single-image matching, both passes, all metrics (car, moderate)
    stats = []
    prep = ev._prepare_data(gts, dts, 0, 1)
    for metric, mo in ((0, 0.7), (1, 0.7), (2, 0.5)):
        overlaps = ev.calculate_iou_partly(dts, gts, metric, 50)[0]
        for i in range(len(gts)):
            args = (overlaps[i], prep[0][i], prep[1][i], prep[2][i], prep[3][i], prep[4][i], metric)
            tp, fp, fn, sim, th = ev.compute_statistics_jit(*args, min_overlap=mo, thresh=0.0, compute_fp=False)
            row = [metric, i, tp, fp, fn, sim, float(np.sum(th))]
            for thresh in (0.0, 0.3, 0.6):
                tp, fp, fn, sim, _ = ev.compute_statistics_jit(*args, min_overlap=mo, thresh=thresh, compute_fp=True,
                                                               compute_aos=True)
                row += [tp, fp, fn, sim]
            stats.append(row)
    out["image_stats"] = np.array(stats, dtype=np.float64)

    # --- whole evaluation ----------------------------------------------------------------------------------------
    seen = []
    real_v3 = ev.eval_class_v3

    def spy(*a, **k):
        ret = real_v3(*a, **k)
        seen.append(ret)
        return ret
    ev.eval_class_v3 = spy
    out["official_text"] = np.array(ev.get_official_eval_result(gts, dts, ["Car", "Pedestrian", "Cyclist"]))
    for metric, ret in enumerate(seen[:3]):
        for key in ("precision", "recall", "orientation"):
            out["official_m%d_%s" % (metric, key)] = ret[key]
    del seen[:]
    real_linspace = np.linspace                 # the reference passes num as a float (accepted by the NumPy of its day)
    np.linspace = lambda a, b, num: real_linspace(a, b, int(num))
    try:
        out["coco_text"] = np.array(ev.get_coco_eval_result(gts, dts, ["Car"]))
    finally:
        np.linspace = real_linspace
    out["coco_m2_precision"] = seen[2]["precision"]
    # no orientation (alpha == -10) and a single class given as an int
    dts_na = [dict(d, alpha=np.full_like(d["alpha"], -10.0)) for d in dts]
    out["official_text_noaos"] = np.array(ev.get_official_eval_result(gts, dts_na, 0))

    # --- formats --------------------------------------------------------------------------------------------------
    with tempfile.TemporaryDirectory() as tmp:
        open(os.path.join(tmp, "000007.txt"), "w").write(LABEL_TXT)
        open(os.path.join(tmp, "000008.txt"), "w").write(RESULT_TXT)
        open(os.path.join(tmp, "000009.txt"), "w").write("")
        open(os.path.join(tmp, "calib.txt"), "w").write(CALIB_TXT)
        annos = kc.get_label_annos(tmp, [7, 8, 9])
        for i, a in enumerate(annos):
            for k, v in a.items():
                out["label%d_%s" % (i, k)] = np.array("\n".join(v)) if k == "name" else np.asarray(v)
        calib = ku.Calibration(os.path.join(tmp, "calib.txt"))
    for k in ("P2", "V2C", "C2V", "R0"):
        out["calib_" + k] = getattr(calib, k)
    pts = r.uniform(-20, 40, (12, 3))
    out["proj_pts"], out["proj_rect"] = pts, ku.project_velo_to_rect(pts, calib)
    out["proj_img"] = ku.project_rect_to_image(out["proj_rect"], calib)
    out["proj_back"] = ku.project_rect_to_velo(out["proj_rect"], calib)
    boxes = np.zeros((30, 7), np.float32)
    boxes[:, 0], boxes[:, 1], boxes[:, 2] = r.uniform(-5, 70, 30), r.uniform(-40, 40, 30), r.uniform(-2.2, -0.6, 30)
    boxes[:, 3], boxes[:, 4], boxes[:, 5] = r.uniform(1.4, 1.9, 30), r.uniform(3.2, 4.6, 30), r.uniform(1.4, 1.8, 30)
    boxes[:, 6] = r.uniform(-7, 7, 30)
    scores, labels = r.uniform(0.3, 1, 30).astype(np.float32), r.integers(0, 3, 30)
    out["b2r_boxes"], out["b2r_scores"], out["b2r_labels"] = boxes.copy(), scores, labels
    meta = dict(calib=calib, sample_idx=123, img_shape=(375, 1242, 3))
    res = tr.kitti_bbox2results(boxes.copy(), scores, labels, meta, class_names=["Car", "Pedestrian", "Cyclist"])
    for k, v in res.items():
        out["b2r_out_" + k] = np.array("\n".join(v)) if k == "name" else np.asarray(v)
    behind = boxes[:3].copy()
    behind[:, 0] = -30.0                                         # behind the camera / outside the image
    res = tr.kitti_bbox2results(behind, scores[:3], labels[:3], meta, class_names=["Car", "Pedestrian", "Cyclist"])
    out["b2r_behind_count"] = np.array(len(res["name"]))

    np.savez_compressed(os.path.join(HERE, "kitti_eval_ref.npz"), **out)
    print(str(out["official_text"]))
    print(str(out["coco_text"]))
    print("kitti_eval_ref.npz: %d arrays, b2r kept %d of 30, behind kept %d" %
          (len(out), len(out["b2r_out_score"]), int(out["b2r_behind_count"])))


if __name__ == "__main__":
    main()
