"""KITTI average-precision evaluation -- same entry points and result text as
mmdet/core/evaluation/kitti_eval.py (`get_official_eval_result` is what tools/test.py:156-157 calls).

Split of the work, MI355X-first:
  * the rotated BEV / 3-D overlap matrices, O(boxes^2) per part of the image list, run on the GPU through
    sassd_rotate_iou_eval (eval_ops.rotate_iou_gpu_eval; kitti_eval.py:125-162 used a numba.cuda kernel);
  * the axis-aligned image-box overlaps and the height term of the 3-D overlap are a few vectorised float64 numpy
    expressions (kitti_eval.py:96-122,131-154, numba CPU loops in the reference);
  * the greedy matching over images x score thresholds is the native host function sassd_kitti_eval_statistics
    (kitti_eval.py:165-343, numba CPU in the reference).
Annotations are the dictionaries of sassd.kitti_common (camera frame)."""
import ctypes

import numpy as np

from . import _C
from .eval_ops import rotate_iou_gpu_eval

N_SAMPLE_PTS = 41
_CLASS_NAMES = ['car', 'pedestrian', 'cyclist', 'van', 'person_sitting', 'car', 'tractor', 'trailer']
_CLASS_TO_NAME = {0: 'Car', 1: 'Pedestrian', 2: 'Cyclist', 3: 'Van', 4: 'Person_sitting', 5: 'car', 6: 'tractor',
                  7: 'trailer'}
_MIN_HEIGHT = (40, 25, 25)
_MAX_OCCLUSION = (0, 1, 2)
_MAX_TRUNCATION = (0.15, 0.3, 0.5)


# ---- overlap matrices --------------------------------------------------------------------------------------------

def image_box_overlap(boxes, query_boxes, criterion=-1):
    """[N,4] x [K,4] axis-aligned (x1, y1, x2, y2) -> [N,K]; criterion -1 IoU, 0 / 1 intersection over the box's / the
    query's area, otherwise the intersection area (kitti_eval.py:96-122)."""
    b, q = boxes[:, None, :], query_boxes[None, :, :]
    iw = np.minimum(b[..., 2], q[..., 2]) - np.maximum(b[..., 0], q[..., 0])
    ih = np.minimum(b[..., 3], q[..., 3]) - np.maximum(b[..., 1], q[..., 1])
    hit = (iw > 0) & (ih > 0)
    area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    area_q = (q[..., 2] - q[..., 0]) * (q[..., 3] - q[..., 1])
    inter = iw * ih
    if criterion == -1:
        ua = area_b + area_q - inter
    elif criterion == 0:
        ua = np.broadcast_to(area_b, inter.shape)
    elif criterion == 1:
        ua = np.broadcast_to(area_q, inter.shape)
    else:
        ua = np.ones_like(inter)
    out = np.zeros(inter.shape, dtype=boxes.dtype)
    np.divide(inter, ua, out=out, where=hit)
    return out


def bev_box_overlap(boxes, qboxes, criterion=-1):
    return rotate_iou_gpu_eval(boxes, qboxes, criterion)


def d3_box_overlap(boxes, qboxes, criterion=-1):
    """Camera-frame boxes (x, y, z, l, h, w, ry), y pointing down with the box bottom at y: BEV intersection (GPU) times
    the overlap of the vertical extents, over the requested union (kitti_eval.py:131-162)."""
    rinc = rotate_iou_gpu_eval(boxes[:, [0, 2, 3, 5, 6]], qboxes[:, [0, 2, 3, 5, 6]], 2)
    b, q = boxes[:, None, :], qboxes[None, :, :]
    iw = np.minimum(b[..., 1], q[..., 1]) - np.maximum(b[..., 1] - b[..., 4], q[..., 1] - q[..., 4])
    vol_b = b[..., 3] * b[..., 4] * b[..., 5]
    vol_q = q[..., 3] * q[..., 4] * q[..., 5]
    inc = iw * rinc
    if criterion == -1:
        ua = vol_b + vol_q - inc
    elif criterion == 0:
        ua = np.broadcast_to(vol_b, inc.shape)
    elif criterion == 1:
        ua = np.broadcast_to(vol_q, inc.shape)
    else:
        ua = np.ones_like(inc)
    out = rinc.copy()                                  # entries with no BEV intersection stay as they are
    out[(rinc > 0) & ~(iw > 0)] = 0.0
    np.divide(inc, ua, out=out, where=(rinc > 0) & (iw > 0))
    return out


def get_split_parts(num, num_part):
    """Sizes of the consecutive image groups one overlap matrix is computed for (empty groups are dropped)."""
    same, rest = divmod(num, num_part)
    return [p for p in [same] * num_part + [rest] if p > 0]


def _boxes_for(annos, metric):
    if metric == 0:
        return np.concatenate([a["bbox"] for a in annos], 0)
    cols = [0, 2] if metric == 1 else [0, 1, 2]
    loc = np.concatenate([a["location"][:, cols] for a in annos], 0)
    dims = np.concatenate([a["dimensions"][:, cols] for a in annos], 0)
    rots = np.concatenate([a["rotation_y"] for a in annos], 0)
    return np.concatenate([loc, dims, rots[..., np.newaxis]], axis=1)


def calculate_iou_partly(gt_annos, dt_annos, metric, num_parts=50):
    """-> (per-image [n_first, n_second] views, per-part matrices, boxes per image of the first / second list).
    metric 0 image bbox, 1 BEV, 2 3-D; camera frame (kitti_eval.py:345-419)."""
    assert len(gt_annos) == len(dt_annos)
    if metric not in (0, 1, 2):
        raise ValueError("unknown metric")
    n_first = np.array([len(a["name"]) for a in gt_annos], dtype=np.int64)
    n_second = np.array([len(a["name"]) for a in dt_annos], dtype=np.int64)
    overlaps, parted, at = [], [], 0
    for size in get_split_parts(len(gt_annos), num_parts):
        first, second = _boxes_for(gt_annos[at:at + size], metric), _boxes_for(dt_annos[at:at + size], metric)
        if metric == 0:
            part = image_box_overlap(first, second)
        elif metric == 1:
            part = bev_box_overlap(first, second).astype(np.float64)
        else:
            part = d3_box_overlap(first, second).astype(np.float64)
        parted.append(part)
        r = c = 0
        for i in range(at, at + size):
            overlaps.append(part[r:r + n_first[i], c:c + n_second[i]])
            r, c = r + n_first[i], c + n_second[i]
        at += size
    return overlaps, parted, n_first, n_second


# ---- which boxes count -------------------------------------------------------------------------------------------

def clean_data(gt_anno, dt_anno, current_class, difficulty):
    """-> (number of counted ground truths, ignored_gt [G], ignored_dt [D], DontCare boxes [C,4]); flags are 0 counted,
    1 neutral (neighbouring class or too hard for this difficulty), -1 other class (kitti_eval.py:39-93)."""
    cls = _CLASS_NAMES[current_class].lower()
    names = np.char.lower(np.asarray(gt_anno["name"], dtype=str)) if len(gt_anno["name"]) else np.zeros(0, dtype=str)
    same = names == cls
    neighbour = ((names == "person_sitting") & (cls == "pedestrian")) | ((names == "van") & (cls == "car"))
    neighbour &= ~same
    bbox = np.asarray(gt_anno["bbox"]).reshape(-1, 4)
    too_hard = ((np.asarray(gt_anno["occluded"]) > _MAX_OCCLUSION[difficulty])
                | (np.asarray(gt_anno["truncated"]) > _MAX_TRUNCATION[difficulty])
                | ((bbox[:, 3] - bbox[:, 1]) <= _MIN_HEIGHT[difficulty]))
    ignored_gt = np.full(len(names), -1, dtype=np.int64)
    ignored_gt[neighbour | (same & too_hard)] = 1
    ignored_gt[same & ~too_hard] = 0
    dc = bbox[np.asarray(gt_anno["name"], dtype=str) == "DontCare"] if len(names) else np.zeros((0, 4))

    dnames = np.char.lower(np.asarray(dt_anno["name"], dtype=str)) if len(dt_anno["name"]) else np.zeros(0, dtype=str)
    dbox = np.asarray(dt_anno["bbox"]).reshape(-1, 4)
    ignored_dt = np.where(dnames == cls, 0, -1).astype(np.int64)
    ignored_dt[np.abs(dbox[:, 3] - dbox[:, 1]) < _MIN_HEIGHT[difficulty]] = 1
    return int((ignored_gt == 0).sum()), ignored_gt, ignored_dt, dc.astype(np.float64).reshape(-1, 4)


def _prepare_data(gt_annos, dt_annos, current_class, difficulty):
    gt_datas, dt_datas, ign_gts, ign_dts, dcs = [], [], [], [], []
    n_valid = 0
    for g, d in zip(gt_annos, dt_annos):
        nv, ig, idt, dc = clean_data(g, d, current_class, difficulty)
        n_valid += nv
        ign_gts.append(ig)
        ign_dts.append(idt)
        dcs.append(dc)
        gt_datas.append(np.concatenate([np.asarray(g["bbox"], np.float64).reshape(-1, 4),
                                        np.asarray(g["alpha"], np.float64).reshape(-1, 1)], 1))
        dt_datas.append(np.concatenate([np.asarray(d["bbox"], np.float64).reshape(-1, 4),
                                        np.asarray(d["alpha"], np.float64).reshape(-1, 1),
                                        np.asarray(d["score"], np.float64).reshape(-1, 1)], 1))
    dc_nums = np.array([len(x) for x in dcs], dtype=np.int64)
    return gt_datas, dt_datas, ign_gts, ign_dts, dcs, dc_nums, n_valid


def get_thresholds(scores, num_gt, num_sample_pts=41):
    """Score cut-offs at which recall crosses k/(num_sample_pts-1), k = 0, 1, ... (kitti_eval.py:18-36)."""
    ranked = np.sort(np.asarray(scores, dtype=np.float64))[::-1]
    step = 1 / (num_sample_pts - 1.0)
    picked, want = [], 0
    for i, s in enumerate(ranked):
        here = (i + 1) / num_gt
        last = i == len(ranked) - 1
        ahead = here if last else (i + 2) / num_gt
        if not last and (ahead - want) < (want - here):
            continue                                   # the next detection gets closer to the wanted recall
        picked.append(s)
        want += step
    return picked


# ---- matching (native) -------------------------------------------------------------------------------------------

def _dp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class _Part:
    """The images of one overlap matrix, concatenated the way sassd_kitti_eval_statistics reads them."""

    def __init__(self, overlap, gt_nums, dt_nums, dc_nums, gt_datas, dt_datas, dcs, ign_gts, ign_dts):
        cat = lambda xs, w, t: np.ascontiguousarray(np.concatenate(xs, 0) if len(xs) else np.zeros((0, w)), dtype=t)
        self.overlap = np.ascontiguousarray(overlap, dtype=np.float64)
        self.gt_nums, self.dt_nums, self.dc_nums = (np.ascontiguousarray(x, dtype=np.int64)
                                                    for x in (gt_nums, dt_nums, dc_nums))
        self.gt, self.dt, self.dc = cat(gt_datas, 5, np.float64), cat(dt_datas, 6, np.float64), cat(dcs, 4, np.float64)
        self.ign_gt = np.ascontiguousarray(np.concatenate(ign_gts, 0), dtype=np.int64)
        self.ign_dt = np.ascontiguousarray(np.concatenate(ign_dts, 0), dtype=np.int64)
        assert self.overlap.shape == (int(self.dt_nums.sum()), int(self.gt_nums.sum()))

    def run(self, metric, min_overlap, thresholds=None, compute_aos=False, pr=None):
        """thresholds None: first pass -> scores of the matched detections (pr [4], if given, += tp, 0, fn, 0);
        otherwise pr [len(thresholds),4] += (tp, fp, fn, similarity) per threshold."""
        n_thr = 0 if thresholds is None else len(thresholds)
        if thresholds is not None and n_thr == 0:
            return None
        scores = np.zeros(max(int(self.gt_nums.sum()), 1), dtype=np.float64)
        count = ctypes.c_int64(0)
        thr = np.ascontiguousarray(thresholds, dtype=np.float64) if n_thr else None
        _C.check(_C.lib().sassd_kitti_eval_statistics(
            _dp(self.overlap), self.overlap.shape[1], len(self.gt_nums), _dp(self.gt_nums), _dp(self.dt_nums),
            _dp(self.dc_nums), _dp(self.gt), _dp(self.dt), _dp(self.dc), _dp(self.ign_gt), _dp(self.ign_dt), int(metric),
            float(min_overlap), _dp(thr) if n_thr else None, n_thr, int(bool(compute_aos)),
            None if pr is None else _dp(pr), None if n_thr else _dp(scores),
            None if n_thr else ctypes.addressof(count)), "sassd_kitti_eval_statistics")
        return None if n_thr else scores[:count.value]


def compute_statistics_jit(overlaps, gt_datas, dt_datas, ignored_gt, ignored_det, dc_bboxes, metric, min_overlap,
                           thresh=0, compute_fp=False, compute_aos=False):
    """One image -> (tp, fp, fn, similarity, scores of the true positives); the contract of kitti_eval.py:165-283
    (overlaps [n_dt, n_gt]; the score list is only produced by the compute_fp=False pass)."""
    part = _Part(overlaps, [len(gt_datas)], [len(dt_datas)], [len(dc_bboxes)], [gt_datas], [dt_datas],
                 [np.asarray(dc_bboxes, np.float64).reshape(-1, 4)], [ignored_gt], [ignored_det])
    pr = np.zeros((1, 4))
    if not compute_fp:
        scores = part.run(metric, min_overlap, pr=pr)
        return int(pr[0, 0]), 0, int(pr[0, 2]), 0, scores
    part.run(metric, min_overlap, [thresh], compute_aos, pr)
    tp, fp, fn = int(pr[0, 0]), int(pr[0, 1]), int(pr[0, 2])
    similarity = pr[0, 3] if (not compute_aos or tp > 0 or fp > 0) else -1
    return tp, fp, fn, similarity, np.zeros(0)


def eval_class_v3(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos=False,
                  num_parts=50):
    """-> dict(recall, precision, orientation), each [class, difficulty, min_overlap, 41] (kitti_eval.py:549-656).
    min_overlaps: [num_minoverlap, metric, class]."""
    assert len(gt_annos) == len(dt_annos)
    sizes = get_split_parts(len(gt_annos), num_parts)
    _, parted, dt_nums, gt_nums = calculate_iou_partly(dt_annos, gt_annos, metric, num_parts)
    shape = [len(current_classes), len(difficultys), len(min_overlaps), N_SAMPLE_PTS]
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for m, current_class in enumerate(current_classes):
        for l, difficulty in enumerate(difficultys):
            gt_datas, dt_datas, ign_gts, ign_dts, dcs, dc_nums, n_valid = _prepare_data(gt_annos, dt_annos,
                                                                                       current_class, difficulty)
            parts, at = [], 0
            for j, size in enumerate(sizes):
                s = slice(at, at + size)
                parts.append(_Part(parted[j], gt_nums[s], dt_nums[s], dc_nums[s], gt_datas[s], dt_datas[s], dcs[s],
                                   ign_gts[s], ign_dts[s]))
                at += size
            for k, min_overlap in enumerate(min_overlaps[:, metric, m]):
                tp_scores = [p.run(metric, min_overlap) for p in parts]
                tp_scores = np.concatenate(tp_scores) if tp_scores else np.zeros(0)
                thresholds = np.array(get_thresholds(tp_scores, n_valid))
                n = len(thresholds)
                pr = np.zeros([n, 4])
                if n:
                    for p in parts:
                        p.run(metric, min_overlap, thresholds, compute_aos, pr)
                with np.errstate(divide="ignore", invalid="ignore"):
                    recall[m, l, k, :n] = pr[:, 0] / (pr[:, 0] + pr[:, 2])
                    precision[m, l, k, :n] = pr[:, 0] / (pr[:, 0] + pr[:, 1])
                    if compute_aos:
                        aos[m, l, k, :n] = pr[:, 3] / (pr[:, 0] + pr[:, 1])
                for arr in (precision, recall) + ((aos,) if compute_aos else ()):
                    row = arr[m, l, k]
                    for i in range(n):                          # value at i := best value at or after i
                        row[i] = np.max(row[i:])
    return {"recall": recall, "precision": precision, "orientation": aos}


def get_mAP_v2(prec):
    return prec[..., ::4].sum(-1) / 11 * 100


def do_eval_v2(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos=False, difficultys=(0, 1, 2)):
    """-> (mAP_bbox, mAP_bev, mAP_3d, mAP_aos), each [class, difficulty, min_overlap]."""
    ret = eval_class_v3(gt_annos, dt_annos, current_classes, difficultys, 0, min_overlaps, compute_aos)
    mAP_bbox = get_mAP_v2(ret["precision"])
    mAP_aos = get_mAP_v2(ret["orientation"]) if compute_aos else None
    mAP_bev = get_mAP_v2(eval_class_v3(gt_annos, dt_annos, current_classes, difficultys, 1, min_overlaps)["precision"])
    mAP_3d = get_mAP_v2(eval_class_v3(gt_annos, dt_annos, current_classes, difficultys, 2, min_overlaps)["precision"])
    return mAP_bbox, mAP_bev, mAP_3d, mAP_aos


def do_coco_style_eval(gt_annos, dt_annos, current_classes, overlap_ranges, compute_aos):
    """overlap_ranges [3 = (start, stop, num), metric, class] -> mAPs averaged over the 10 overlap levels."""
    min_overlaps = np.zeros([10, *overlap_ranges.shape[1:]])
    for i in range(overlap_ranges.shape[1]):
        for j in range(overlap_ranges.shape[2]):
            lo, hi, num = overlap_ranges[:, i, j]
            min_overlaps[:, i, j] = np.linspace(lo, hi, int(num))
    maps = do_eval_v2(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos)
    return tuple(None if x is None else x.mean(-1) for x in maps)


def _class_ids(current_classes):
    name_to_class = {v: n for n, v in _CLASS_TO_NAME.items()}
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    return [name_to_class[c] if isinstance(c, str) else c for c in current_classes]


def _has_alpha(dt_annos):
    for anno in dt_annos:                                     # the first non-empty frame decides, as in the reference
        if anno['alpha'].shape[0] != 0:
            return bool(anno['alpha'][0] != -10)
    return False


def _ap_lines(bbox, bev, d3, aos):
    fmt = lambda tag, v: "{}AP:{:.2f}, {:.2f}, {:.2f}\n".format(tag, v[0], v[1], v[2])
    out = fmt("bbox ", bbox) + fmt("bev  ", bev) + fmt("3d   ", d3)
    return out + (fmt("aos  ", aos) if aos is not None else "")


def get_official_eval_result(gt_annos, dt_annos, current_classes, difficultys=(0, 1, 2)):
    """The KITTI report tools/test.py prints: per class, AP at the official overlaps (0.7/0.5/0.5 ...) and at the
    relaxed ones, for the image bbox, BEV, 3-D (and orientation) metrics x (easy, moderate, hard)."""
    strict = [0.7, 0.5, 0.5, 0.7, 0.5, 0.7, 0.7, 0.7]
    relaxed = [0.5, 0.25, 0.25, 0.5, 0.25, 0.5, 0.5, 0.5]
    min_overlaps = np.array([[strict, strict, strict], [strict, relaxed, relaxed]])      # [2, metric, class]
    classes = _class_ids(current_classes)
    min_overlaps = min_overlaps[:, :, classes]
    compute_aos = _has_alpha(dt_annos)
    bbox, bev, d3, aos = do_eval_v2(gt_annos, dt_annos, classes, min_overlaps, compute_aos, difficultys)
    result = ''
    for j, c in enumerate(classes):
        for i in range(min_overlaps.shape[0]):
            result += "{} AP@{:.2f}, {:.2f}, {:.2f}:\n".format(_CLASS_TO_NAME[c], *min_overlaps[i, :, j])
            result += _ap_lines(bbox[j, :, i], bev[j, :, i], d3[j, :, i], aos[j, :, i] if compute_aos else None)
    return result


def get_coco_eval_result(gt_annos, dt_annos, current_classes):
    """COCO-style report: AP averaged over 10 overlap levels per class (0.5:0.95 vehicles, 0.25:0.7 people)."""
    ranges = {c: ([0.5, 0.95, 10] if c in (0, 3, 5, 6, 7) else [0.25, 0.7, 10]) for c in range(8)}
    classes = _class_ids(current_classes)
    overlap_ranges = np.zeros([3, 3, len(classes)])
    for i, c in enumerate(classes):
        overlap_ranges[:, :, i] = np.array(ranges[c])[:, np.newaxis]
    compute_aos = _has_alpha(dt_annos)
    bbox, bev, d3, aos = do_coco_style_eval(gt_annos, dt_annos, classes, overlap_ranges, compute_aos)
    result = ''
    for j, c in enumerate(classes):
        lo, hi, num = ranges[c]
        result += "{} coco AP@{:.2f}:{:.2f}:{:.2f}:\n".format(_CLASS_TO_NAME[c], lo, (hi - lo) / (num - 1), hi)
        result += _ap_lines(bbox[j], bev[j], d3[j], aos[j] if compute_aos else None)
    return result
