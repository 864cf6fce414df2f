#!/usr/bin/env python
"""f-4 measurement: the fused GT-sampling augmentation (sassd.point_augmentor.PointAugmentor.augment_frame) on synthetic
frames with the object database resident in HBM -- the target of `rocprofv3 --kernel-trace --stats`, and wall-clock per
frame.  usage: python tools/run_augment.py [--frames 50]"""
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
import augment_synth as S  # noqa: E402
import sassd  # noqa: E402,F401
from sassd import kitti_common as kc, point_augmentor as PA  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=50)
a = ap.parse_args()
dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp()
S.write_database(S.make_database(), tmp)
calib = kc.Calibration(matrices=S.calib_matrices())
cfg = S.AUGMENTOR_CONFIGS["multi"]
aug = PA.PointAugmentor(root_path=tmp, info_path=os.path.join(tmp, "kitti_dbinfos_train.pkl"), device=dev, **cfg)
frames = [S.frame(f % 3) for f in range(3)]
sweeps = [torch.from_numpy(np.concatenate([p, S.full_sweep(7 + i)[:15000]], 0).astype(np.float32)).to(dev)
          for i, (p, _, _) in enumerate(frames)]
np.random.seed(0)
for i in range(3):
    aug.augment_frame(sweeps[i], frames[i][1].copy(), frames[i][2], cfg["sample_classes"], S.PLANE, calib)
torch.cuda.synchronize()
t0 = time.perf_counter()
n_out = 0
for i in range(a.frames):
    k = i % 3
    out = aug.augment_frame(sweeps[k], frames[k][1].copy(), frames[k][2], cfg["sample_classes"], S.PLANE, calib)
    n_out += int(out[0].shape[0])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps(dict(frames=a.frames, ms_per_frame=round(dt / a.frames * 1e3, 3), points_in=int(sweeps[0].shape[0]),
                      points_out_mean=n_out // a.frames, classes=cfg["sample_classes"])))
