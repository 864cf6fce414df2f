"""KITTI evaluation + on-disk formats against vectors produced by the reference's own code
(tests/golden/make_golden_kitti_eval.py -> kitti_eval_ref.npz).  Runs without a GPU: the host parts of the product
(sassd.kitti_common, sassd.kitti_eval's numpy overlaps, the native sassd_kitti_eval_statistics) are the code under
test; the one device call of the evaluation, rotate_iou_gpu_eval, is served here by the CPU oracle -- the same
substitution the fixture was generated with -- and checked on the GPU in test_gpu_kernels.py::test_rotate_iou_eval and
test_gpu_train.py::test_kitti_eval_on_gpu."""
import os

import numpy as np
import pytest

import sassd  # noqa: F401
from sassd import kitti_common as kc
from sassd import kitti_eval as ke
from oracle import clib

import kitti_synth

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(HERE, "golden", "kitti_eval_ref.npz"))


@pytest.fixture()
def oracle_iou(monkeypatch):
    def fn(boxes, query_boxes, criterion=-1, device_id=0):
        return clib.rotate_iou_eval(boxes.astype(np.float32), query_boxes.astype(np.float32),
                                    criterion).astype(boxes.dtype)
    monkeypatch.setattr(ke, "rotate_iou_gpu_eval", fn)


@pytest.fixture(scope="module")
def annos(G):
    gts, dts = kitti_synth.unpack(G, "gt_"), kitti_synth.unpack(G, "dt_")
    again = kitti_synth.make_annos(56, seed=0)                       # the generator is part of the fixture's definition
    for a, b in zip(gts + dts, again[0] + again[1]):
        assert list(a["name"]) == list(b["name"]) and np.array_equal(a["bbox"], b["bbox"])
    return gts, dts


def test_thresholds(G):
    for k in range(4):
        got = ke.get_thresholds(G["thr%d_scores" % k], int(G["thr%d_num_gt" % k]))
        assert np.array_equal(np.array(got), G["thr%d_out" % k]), k
    assert ke.get_thresholds(np.zeros(0), 5) == []


def test_overlap_matrices(G, annos, oracle_iou):
    gts, dts = annos
    for metric in (0, 1, 2):
        overlaps, parted, n_dt, n_gt = ke.calculate_iou_partly(dts, gts, metric, 50)
        assert len(parted) == 51 and len(overlaps) == 56
        assert np.array_equal(parted[-1], G["ov%d_last_part" % metric]), metric
        assert np.array_equal(parted[3], G["ov%d_part3" % metric]), metric
        assert overlaps[55].shape == (len(dts[55]["name"]), len(gts[55]["name"]))
    a, b = np.concatenate([g["bbox"] for g in gts[50:]]), np.concatenate([d["bbox"] for d in dts[50:]])
    a3, b3 = (np.concatenate([kc.anno_to_rbboxes(x) for x in xs[50:]]) for xs in (gts, dts))
    for crit in (-1, 0, 1, 2):
        assert np.array_equal(ke.image_box_overlap(a, b, crit), G["imgov_%d" % crit]), crit
        assert np.array_equal(ke.d3_box_overlap(a3, b3, crit), G["d3ov_%d" % crit]), crit
    # fewer images than parts: the reference cannot run this (empty concatenation); here the empty parts are dropped
    assert len(ke.calculate_iou_partly(dts[:7], gts[:7], 0, 50)[1]) == 1


def test_clean_data(G, annos):
    gts, dts = annos
    for cls in (0, 1, 2):
        for diff in (0, 1, 2):
            rows = [ke.clean_data(g, d, cls, diff) for g, d in zip(gts, dts)]
            tag = "clean_%d_%d_" % (cls, diff)
            assert np.array_equal([r[0] for r in rows], G[tag + "nvalid"])
            assert np.array_equal(np.concatenate([r[1] for r in rows]), G[tag + "ign_gt"])
            assert np.array_equal(np.concatenate([r[2] for r in rows]), G[tag + "ign_dt"])
            assert np.array_equal(np.concatenate([r[3] for r in rows]), G[tag + "dc"])


def test_single_image_matching(G, annos, oracle_iou):
    """sassd_kitti_eval_statistics on one image at a time == compute_statistics_jit, both passes, three thresholds."""
    gts, dts = annos
    ref = G["image_stats"]
    prep = ke._prepare_data(gts, dts, 0, 1)
    at = 0
    for metric, mo in ((0, 0.7), (1, 0.7), (2, 0.5)):
        overlaps = ke.calculate_iou_partly(dts, gts, metric, 50)[0]
        for i in range(len(gts)):
            args = (overlaps[i], prep[0][i], prep[1][i], prep[2][i], prep[3][i], prep[4][i], metric)
            row = ref[at]
            at += 1
            assert (row[0], row[1]) == (metric, i)
            tp, fp, fn, sim, th = ke.compute_statistics_jit(*args, min_overlap=mo, thresh=0.0, compute_fp=False)
            assert (tp, fp, fn, sim) == tuple(row[2:6]) and len(th) == tp
            assert abs(float(np.sum(th)) - row[6]) < 1e-12
            for k, thresh in enumerate((0.0, 0.3, 0.6)):
                tp, fp, fn, sim, _ = ke.compute_statistics_jit(*args, min_overlap=mo, thresh=thresh, compute_fp=True,
                                                               compute_aos=True)
                want = row[7 + 4 * k:11 + 4 * k]
                assert (tp, fp, fn) == tuple(want[:3]), (metric, i, thresh)
                assert abs(sim - want[3]) < 1e-12, (metric, i, thresh)
    assert at == len(ref)
    assert ref[:, 8].sum() > 20 and ref[:, 2].sum() > 50             # the fixture has false positives and matches


def test_official_result(G, annos, oracle_iou):
    gts, dts = annos
    seen = []
    real = ke.eval_class_v3
    ke.eval_class_v3 = lambda *a, **k: seen.append(real(*a, **k)) or seen[-1]
    try:
        text = ke.get_official_eval_result(gts, dts, ["Car", "Pedestrian", "Cyclist"])
    finally:
        ke.eval_class_v3 = real
    assert text == str(G["official_text"])
    for metric, ret in enumerate(seen):
        for key in ("precision", "recall", "orientation"):
            assert np.allclose(ret[key], G["official_m%d_%s" % (metric, key)], rtol=0, atol=1e-12), (metric, key)
    dts_na = [dict(d, alpha=np.full_like(d["alpha"], -10.0)) for d in dts]
    assert ke.get_official_eval_result(gts, dts_na, 0) == str(G["official_text_noaos"])
    assert "aos" not in str(G["official_text_noaos"])


def test_coco_result(G, annos, oracle_iou):
    gts, dts = annos
    assert ke.get_coco_eval_result(gts, dts, ["Car"]) == str(G["coco_text"])


def test_num_parts_does_not_change_the_result(annos, oracle_iou):
    gts, dts = annos
    mo = np.array([[[0.7], [0.7], [0.7]]])
    a = ke.eval_class_v3(gts, dts, [0], [0, 1, 2], 2, mo, num_parts=50)
    b = ke.eval_class_v3(gts, dts, [0], [0, 1, 2], 2, mo, num_parts=3)
    c = ke.eval_class_v3(gts, dts, [0], [0, 1, 2], 2, mo, num_parts=200)
    for key in a:
        assert np.array_equal(a[key], b[key]) and np.array_equal(a[key], c[key])


def test_perfect_detections(annos, oracle_iou):
    """Detections identical to the labels: precision 1 at every score cut-off and full recall on the image-box metric
    (the AP itself stays under 100 on a set this small: 41 recall samples need 41 counted objects)."""
    gts, dts = [], []
    for g in annos[0]:
        inside = (g["bbox"][:, 2] - g["bbox"][:, 0] > 1) & (g["bbox"][:, 3] - g["bbox"][:, 1] > 1)
        g = {k: v[inside] for k, v in g.items()}          # (objects projected outside the image have empty boxes)
        gts.append(g)
        keep = g["name"] != "DontCare"
        d = {k: v[keep].copy() for k, v in g.items()}
        d["score"] = np.linspace(0.9, 0.5, int(keep.sum()))
        dts.append(d)
    ret = ke.eval_class_v3(gts, dts, [0, 1, 2], [0, 1, 2], 0, np.full((1, 3, 3), 0.7), compute_aos=True)
    for key in ("precision", "orientation"):
        vals = ret[key][ret["recall"] > 0]
        assert len(vals) > 30 and np.all(vals == 1.0), key
    assert np.all(ret["recall"].max(-1) == 1.0)


def test_statistics_abi_errors():
    from sassd import _C
    import ctypes
    f = _C.lib().sassd_kitti_eval_statistics
    one = np.zeros(8)
    n1 = np.ones(1, dtype=np.int64)
    cnt = ctypes.c_int64(0)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    ok = (p(one), 1, 1, p(n1), p(n1), p(n1), p(one), p(one), p(one), p(n1), p(n1), 1, 0.5, None, 0, 0, None, p(one),
          ctypes.addressof(cnt))
    assert f(*ok) == _C.OK
    bad_metric = list(ok); bad_metric[11] = 3
    assert f(*bad_metric) == _C.EINVAL
    small_ld = list(ok); small_ld[1] = 0
    assert f(*small_ld) == _C.EINVAL
    no_out = list(ok); no_out[17] = None
    assert f(*no_out) == _C.EINVAL
    no_pr = list(ok); no_pr[13], no_pr[14] = p(one), 1
    assert f(*no_pr) == _C.EINVAL


# ---- formats ---------------------------------------------------------------------------------------------------------

def test_label_files(G, tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(HERE, "golden", "make_golden_kitti_eval.py"))
    src = open(spec.origin).read()
    texts = {}
    for name in ("LABEL_TXT", "RESULT_TXT", "CALIB_TXT"):
        texts[name] = src.split(name + ' = """')[1].split('"""')[0]
    (tmp_path / "000007.txt").write_text(texts["LABEL_TXT"])
    (tmp_path / "000008.txt").write_text(texts["RESULT_TXT"])
    (tmp_path / "000009.txt").write_text("")
    (tmp_path / "calib.txt").write_text(texts["CALIB_TXT"])
    annos = kc.get_label_annos(tmp_path, [7, 8, 9])
    for i, a in enumerate(annos):
        keys = [k[len("label%d_" % i):] for k in G.files if k.startswith("label%d_" % i)]
        assert sorted(keys) == sorted(a.keys())
        for k in keys:
            want = G["label%d_%s" % (i, k)]
            if k == "name":
                assert "\n".join(a[k]) == str(want)
            else:
                assert np.array_equal(np.asarray(a[k]), want) and np.asarray(a[k]).shape == want.shape, (i, k)
    assert [int(a["image_idx"][0]) for a in kc.get_label_annos(tmp_path)[:2]] == [7, 8]     # folder scan
    calib = kc.Calibration(tmp_path / "calib.txt")
    for k in ("P2", "V2C", "C2V", "R0"):
        assert np.array_equal(getattr(calib, k), G["calib_" + k]), k
    assert np.array_equal(kc.project_velo_to_rect(G["proj_pts"], calib), G["proj_rect"])
    assert np.array_equal(kc.project_rect_to_image(G["proj_rect"], calib), G["proj_img"])
    assert np.allclose(kc.project_rect_to_velo(G["proj_rect"], calib), G["proj_back"], rtol=0, atol=1e-12)
    assert np.allclose(G["proj_back"], G["proj_pts"], atol=1e-4)     # (the files carry 7 significant digits)

    # detector output -> result annotation -> result file -> annotation
    meta = dict(calib=calib, sample_idx=123, img_shape=(375, 1242, 3))
    res = kc.kitti_bbox2results(G["b2r_boxes"].copy(), G["b2r_scores"], G["b2r_labels"], meta,
                                class_names=["Car", "Pedestrian", "Cyclist"])
    for k in [f[len("b2r_out_"):] for f in G.files if f.startswith("b2r_out_")]:
        want = G["b2r_out_" + k]
        if k == "name":
            assert "\n".join(res[k]) == str(want)
        else:
            assert np.array_equal(res[k], want) and res[k].dtype == want.dtype, (k, res[k].dtype, want.dtype)
    assert len(res["score"]) == 21
    empty = kc.kitti_bbox2results(np.zeros((0, 7), np.float32), np.zeros(0), np.zeros(0, int), meta, ["Car"])
    assert empty["bbox"].shape == (0, 4) and len(empty["name"]) == 0
    kc.write_label_annos([res], tmp_path / "out")
    back = kc.get_label_annos(tmp_path / "out", [123])[0]
    assert list(back["name"]) == list(res["name"])
    for k in ("bbox", "dimensions", "location", "rotation_y", "score", "alpha"):
        assert np.allclose(back[k], res[k], atol=6e-5), k
