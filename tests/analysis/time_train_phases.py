"""Steady-state phase timing of one training step (wall clock with device syncs between phases, and the same loop
without syncs) -- tells host-bound from GPU-bound.  usage: python tests/analysis/time_train_phases.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import helpers as H  # noqa: E402
import sassd  # noqa: E402,F401
from sassd import synth, anchors as A, train, autograd as AG  # noqa: E402
from sassd.config import Config  # noqa: E402
from sassd.detector import build_detector  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
AG.set_bev_precision(os.environ.get('PRECISION', 'bf16'))
dev = torch.device("cuda", 0)
cfg = Config.fromfile(os.path.join(ROOT, "configs", "car_cfg.py"))
model = H.randomize_detector(build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg), 0, cls_bias=-3.0).to(dev)
an = A.AnchorGeneratorStride(sizes=[1.6, 3.9, 1.56], anchor_strides=[.4, .4, 1.], anchor_offsets=[.2, -39.8, -1.78],
                             rotations=[0, 1.57])([1, 200, 176]).reshape(-1, 7)
bv = A.rbbox2d_to_near_bbox(an[:, [0, 1, 3, 4, 6]]).astype(np.float32)
anchors, anchors_bv = dict(Car=torch.from_numpy(an).to(dev)), dict(Car=torch.from_numpy(bv).to(dev))
opt = train.build_optimizer(model, cfg.optimizer, 1)
sched = train.build_scheduler(opt, 1000, 1, cfg.optimizer, cfg.lr_config)
sync = train.GradSync(opt.flat)
clouds = [torch.from_numpy(synth.k21(i)).to(dev) for i in range(4)]
gts = [torch.from_numpy(bench.synth_gt_on_points(synth.k21(i), i)).to(dev) for i in range(4)]
types = [np.array(["Car"] * 8)] * 4
S = torch.cuda.synchronize
acc = dict(data=0., fwd=0., bwd=0., opt=0.)
import cProfile, pstats
pr = cProfile.Profile()
for it in range(steps + 6):
    ids = [(2 * it) % 4, (2 * it + 1) % 4]
    S(); t0 = time.perf_counter()
    batch = train.device_batch([clouds[k] for k in ids], [gts[k] for k in ids], [types[k] for k in ids], ["Car"],
                               anchors, anchors_bv, synth.KITTI_VOXEL, synth.KITTI_RANGE)
    S(); t1 = time.perf_counter()
    sched.step(it); model.train(); opt.zero_grad()
    if it >= 6 and os.environ.get('PROF'):
        pr.enable()
    loss, terms = train.parse_losses(model(**batch))
    pr.disable()
    t1h = time.perf_counter(); S(); t2 = time.perf_counter()
    loss.backward()
    t2h = time.perf_counter(); S(); t3 = time.perf_counter()
    sync.all_reduce_grads(); opt.step()
    S(); t4 = time.perf_counter()
    if it >= 6:
        acc["data"] += t1 - t0; acc["fwd"] += t2 - t1; acc["bwd"] += t3 - t2; acc["opt"] += t4 - t3
        acc["fwd_host"] = acc.get("fwd_host", 0.) + t1h - t1
        acc["bwd_host"] = acc.get("bwd_host", 0.) + t2h - t2
print({k: round(v / steps * 1e3, 2) for k, v in acc.items()}, "ms per step")
if os.environ.get('PROF'):
    pstats.Stats(pr).sort_stats('tottime').print_stats(45)
    pstats.Stats(pr).sort_stats('cumulative').print_stats('sa-ssd_amd|sassd', 70)
