"""Seeded synthetic KITTI annotations (camera frame) for the evaluation tests, and their (de)serialisation into flat
arrays for the golden fixtures.  numpy only; used by tests/golden/make_golden_kitti_eval.py and tests/test_kitti_eval_cpu.py."""
import numpy as np

FIELDS = ("truncated", "occluded", "alpha", "bbox", "dimensions", "location", "rotation_y", "score")
WIDTH = dict(truncated=0, occluded=0, alpha=0, bbox=4, dimensions=3, location=3, rotation_y=0, score=0)
P2 = np.array([[721.5, 0.0, 609.6, 44.9], [0.0, 721.5, 172.9, 0.2], [0.0, 0.0, 1.0, 0.003]])
IMG_HW = (375, 1242)
SIZES = {"Car": (3.9, 1.56, 1.6), "Van": (5.0, 2.2, 1.9), "Pedestrian": (0.8, 1.73, 0.6), "Cyclist": (1.76, 1.73, 0.6),
         "Person_sitting": (0.8, 1.3, 0.6), "Truck": (10.0, 3.2, 2.6)}        # l, h, w


def _project(loc, dims, ry):
    """image box (x1, y1, x2, y2) of a camera-frame box with its bottom centre at loc; clipped to the image."""
    l, h, w = dims
    x = np.array([l, l, -l, -l, l, l, -l, -l]) / 2
    y = np.array([0, 0, 0, 0, -h, -h, -h, -h], dtype=np.float64)
    z = np.array([w, -w, -w, w, w, -w, -w, w]) / 2
    c, s = np.cos(ry), np.sin(ry)
    pts = np.stack([c * x + s * z + loc[0], y + loc[1], -s * x + c * z + loc[2], np.ones(8)], 1) @ P2.T
    uv = pts[:, :2] / pts[:, 2:3]
    lo, hi = uv.min(0), uv.max(0)
    return np.array([max(lo[0], 0), max(lo[1], 0), min(hi[0], IMG_HW[1] - 1), min(hi[1], IMG_HW[0] - 1)])


def _stack(rows):
    out = {"name": np.array([r["name"] for r in rows])}
    for f in FIELDS:
        vals = [r[f] for r in rows]
        out[f] = (np.array(vals, dtype=np.float64).reshape(len(rows), WIDTH[f]) if WIDTH[f]
                  else np.array(vals, dtype=np.int64 if f == "occluded" else np.float64))
    return out


def make_annos(num_images=56, seed=0):
    """-> (gt_annos, dt_annos): lists of annotation dicts like sassd.kitti_common.get_label_annos returns."""
    r = np.random.default_rng(seed)
    names = list(SIZES)
    gts, dts = [], []
    for img in range(num_images):
        g_rows, d_rows = [], []
        for _ in range(int(r.integers(0, 9)) if img % 11 else 0):
            name = names[int(r.choice(len(names), p=[0.45, 0.1, 0.2, 0.12, 0.05, 0.08]))]
            dims = np.array(SIZES[name]) * r.uniform(0.9, 1.1, 3)
            loc = np.array([r.uniform(-18, 18), r.uniform(1.4, 1.9), r.uniform(6, 55)])
            ry = r.uniform(-np.pi, np.pi)
            g = dict(name=name, truncated=float(r.choice([0.0, 0.0, 0.1, 0.2, 0.4, 0.7])),
                     occluded=int(r.choice([0, 0, 1, 2, 3])), alpha=ry - np.arctan2(loc[0], loc[2]),
                     bbox=_project(loc, dims, ry), dimensions=dims, location=loc, rotation_y=ry, score=0.0)
            g_rows.append(g)
            if r.random() < 0.8:                          # a detection near this object
                d = dict(g)
                d["name"] = name if r.random() < 0.9 else names[int(r.integers(0, 4))]
                d["dimensions"] = dims * r.uniform(0.95, 1.05, 3)
                d["location"] = loc + r.normal(0, [0.15, 0.05, 0.2])
                d["rotation_y"] = ry + r.normal(0, 0.06) + (np.pi if r.random() < 0.1 else 0.0)
                d["alpha"] = d["rotation_y"] - np.arctan2(d["location"][0], d["location"][2])
                d["bbox"] = g["bbox"] + r.normal(0, 2.0, 4)
                d["truncated"], d["occluded"], d["score"] = 0.0, 0, float(r.uniform(0.05, 1.0))
                d_rows.append(d)
        for _ in range(int(r.integers(0, 3))):            # DontCare regions
            x1, y1 = r.uniform(0, 1000), r.uniform(100, 250)
            g_rows.append(dict(name="DontCare", truncated=-1.0, occluded=-1, alpha=-10.0,
                               bbox=np.array([x1, y1, x1 + r.uniform(40, 200), y1 + r.uniform(30, 90)]),
                               dimensions=np.array([-1.0, -1.0, -1.0]), location=np.array([-1000.0, -1000.0, -1000.0]),
                               rotation_y=-10.0, score=0.0))
        dcs = [g for g in g_rows if g["name"] == "DontCare"]
        for _ in range(int(r.integers(0, 4)) if img % 7 else 0):      # false positives, some inside DontCare regions
            name = names[int(r.integers(0, 4))]
            dims = np.array(SIZES[name]) * r.uniform(0.9, 1.1, 3)
            loc = np.array([r.uniform(-18, 18), r.uniform(1.4, 1.9), r.uniform(6, 55)])
            ry = r.uniform(-np.pi, np.pi)
            bbox = _project(loc, dims, ry)
            if dcs and r.random() < 0.5:
                b = dcs[int(r.integers(0, len(dcs)))]["bbox"]
                bbox = np.array([b[0] + 3, b[1] + 2, b[2] - 5, b[3] + 6])
            d_rows.append(dict(name=name, truncated=0.0, occluded=0, alpha=ry - np.arctan2(loc[0], loc[2]), bbox=bbox,
                               dimensions=dims, location=loc, rotation_y=ry, score=float(r.uniform(0.05, 0.9))))
        order = r.permutation(len(d_rows))
        gts.append(_stack(g_rows))
        dts.append(_stack([d_rows[i] for i in order]))
    return gts, dts


def pack(annos, prefix):
    out = {prefix + "count": np.array([len(a["name"]) for a in annos], dtype=np.int64),
           prefix + "name": np.array("\n".join(str(n) for a in annos for n in a["name"]))}
    for f in FIELDS:
        out[prefix + f] = np.concatenate([a[f].reshape(len(a["name"]), max(WIDTH[f], 1)) for a in annos], 0)
    return out


def unpack(npz, prefix):
    counts = npz[prefix + "count"]
    names = str(npz[prefix + "name"]).split("\n") if counts.sum() else []
    annos, at = [], 0
    for n in counts:
        a = {"name": np.array(names[at:at + n])}
        for f in FIELDS:
            v = npz[prefix + f][at:at + n]
            a[f] = v.reshape(n, WIDTH[f]) if WIDTH[f] else v.reshape(n)
            if f == "occluded":
                a[f] = a[f].astype(np.int64)
        annos.append(a)
        at += n
    return annos
