"""Throughput of the data side of a training step (SURVEY 8f ranks 1 + 4): files -> KittiLiDAR.load_frame -> PointAugmentor
(GT sampling, per-object noise, flip / rotation / scaling; database resident in HBM) -> collate (HIP voxelizer, anchor
masks, rulebooks when a model is given).  Prints one JSON line: frames/s, ms per frame split into read / augment /
collate, and -- with --cpu-port -- the same augmentation through the CPU harness (the product's __host__ __device__
per-point code looped on one core) as a reported baseline.

    python tools/bench_data.py --frames 64 --points 20000 --objects 600          (on an MI355X)
    python tools/bench_data.py --frames 4 --points 3000 --objects 60 --cpu-only   (logic check without a GPU)

The tree is synthetic (tests/augment_synth.py), written to a temporary directory and prepared with sassd.create_data."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import sassd  # noqa: E402,F401
import augment_synth as S  # noqa: E402


def write_tree(root, frames, points, objects):
    """`frames` training frames with ~`points` points and 6-10 labelled objects each, and a database of `objects` cars."""
    r = np.random.default_rng(0)
    os.makedirs(os.path.join(root, "ImageSets"), exist_ok=True)
    with open(os.path.join(root, "ImageSets", "train.txt"), "w") as f:
        f.write("\n".join("%06d" % i for i in range(frames)) + "\n")
    for sub in ("velodyne_reduced", "label_2", "calib", "image_2"):
        os.makedirs(os.path.join(root, "training", sub), exist_ok=True)
    c = S.calib_matrices()
    from sassd import kitti_common as kc
    calib = kc.Calibration(matrices=c)
    for i in range(frames):
        names = ["Car"] * int(r.integers(4, 9)) + ["Pedestrian", "Cyclist"]
        boxes = S.lidar_boxes(r, len(names), names=names)
        S.scene_points(1000 + i, boxes, n_ground=points).tofile(os.path.join(root, "training", "velodyne_reduced", "%06d.bin" % i))
        cam = kc.project_velo_to_rect(boxes[:, :3], calib)
        with open(os.path.join(root, "training", "label_2", "%06d.txt" % i), "w") as f:
            for n, b, cc in zip(names, boxes, cam):
                f.write("%s 0.00 0 0.00 100.00 100.00 200.00 200.00 %.2f %.2f %.2f %.2f %.2f %.2f %.2f\n" % (
                    n, b[5], b[3], b[4], cc[0], cc[1], cc[2], b[6]))
        with open(os.path.join(root, "training", "calib", "%06d.txt" % i), "w") as f:
            f.write(S.CALIB_TXT)
        S.write_png(os.path.join(root, "training", "image_2", "%06d.png" % i), 375, 1242)
    S.write_database(S.make_database(seed=1, counts=(("Car", objects),)), root)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--objects", type=int, default=600)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--cpu-port", action="store_true", help="also time the augmentation through the CPU harness")
    ap.add_argument("--cpu-only", action="store_true", help="no GPU: harness only, collate skipped (script logic check)")
    a = ap.parse_args()
    from sassd.config import Config
    from sassd.kitti_dataset import get_dataset
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "car_cfg.py"))
    out = dict(metric="data-side frames/s", unit="frames/s", data="synthetic", config=dict(
        workload="KittiLiDAR train frames: %d pts, car_cfg augmentor, db %d objects" % (a.points, a.objects), batch=a.batch))
    with tempfile.TemporaryDirectory() as root:
        write_tree(root, a.frames, a.points, a.objects)
        tr = dict(cfg.data.train, root=root + "/training/", ann_file=root + "/ImageSets/train.txt")
        tr["augmentor"] = dict(tr["augmentor"], root_path=root + "/", info_path=root + "/kitti_dbinfos_train.pkl")

        def run(device, harness_patch):
            undo = []
            if harness_patch:
                import harness
                from sassd import geometry, point_augmentor
                for mod, name, fn in ((geometry, "points_in_polytopes", harness.points_in_polytopes),
                                      (point_augmentor, "_paste_objects", harness.paste_objects),
                                      (point_augmentor, "_points_transform", harness.points_transform),
                                      (point_augmentor, "_points_global", harness.points_global)):
                    undo.append((mod, name, getattr(mod, name)))
                    setattr(mod, name, fn)
            try:
                np.random.seed(0)
                ds = get_dataset(tr, device=device)
                t = dict(read=0.0, augment=0.0, collate=0.0)
                sync = (lambda: torch.cuda.synchronize()) if str(device).startswith("cuda") else (lambda: None)
                n = 0
                for ep in range(a.epochs + 1):                     # first pass = warm-up
                    if ep == 1:
                        t = dict(read=0.0, augment=0.0, collate=0.0)
                        n = 0
                    batch = []
                    for i in range(len(ds)):
                        t0 = time.perf_counter()
                        fr = ds.load_frame(i)
                        t1 = time.perf_counter()
                        s = ds.prepare_train_img(i, frame=fr)
                        sync()
                        t2 = time.perf_counter()
                        t["read"] += t1 - t0
                        t["augment"] += t2 - t1
                        if s is not None:
                            batch.append(s)
                        if len(batch) == a.batch and not harness_patch:
                            ds.collate(batch)
                            sync()
                            t["collate"] += time.perf_counter() - t2
                            batch = []
                        elif len(batch) == a.batch:
                            batch = []
                        n += 1
                return n, t
            finally:
                for mod, name, fn in undo:
                    setattr(mod, name, fn)

        if not a.cpu_only:
            n, t = run(torch.device("cuda", 0), False)
            total = sum(t.values())
            out.update(value=n / total, ms_per_frame={k: 1e3 * v / n for k, v in t.items()}, frames=n)
        if a.cpu_port or a.cpu_only:
            n, t = run(torch.device("cpu"), True)
            out["cpu_port"] = dict(value=n / (t["read"] + t["augment"]), unit="frames/s (read + augment only)", cores=1,
                                   ms_per_frame={k: 1e3 * v / n for k, v in t.items() if k != "collate"})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
