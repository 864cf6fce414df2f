#!/usr/bin/env python
"""bf16 vs fp32 training TRAJECTORY: N steps of BASELINE configs[2] (car_cfg, batch 2, the bench's
training workload: K21 frames + 8 synthetic car boxes per frame on occupied voxels, adam_onecycle, grad clip 10) from the same
seeds -- same initial weights, same frames in the same order, same schedule -- once with fp32 BEV convolutions, once more with
fp32 (the run-to-run spread: the auxiliary head scatters with float atomics, so two fp32 runs are not bit-equal and 200
chaotic steps amplify that), once with bf16 BEV convolutions.  The six loss terms per step of every run go to a JSON record
(profiles/rNN_train_trajectory.json); tests/test_gpu_train.py::test_bf16_and_fp32_training_trajectories_agree asserts on it.

    python tests/analysis/train_trajectory.py [--steps 200] [--out profiles/r06_train_trajectory.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sassd  # noqa: E402,F401
from sassd import synth, train, autograd as AG  # noqa: E402
import bench  # noqa: E402


def run(precision, steps, dev, seed=0, batch=2, frames=16):
    """-> {term: [steps values]} + 'loss'.  Everything seeded: model init, frames, boxes, batch order."""
    prev = AG.bev_precision()
    AG.set_bev_precision(precision)
    try:
        torch.manual_seed(seed)
        w = synth.workload("car")
        model, cfg = synth.build_detector_for(w, seed, train=True, cls_bias=-3.0)
        model = model.to(dev)
        anchors = dict(Car=torch.from_numpy(w["anchors"]).to(dev))
        anchors_bv = dict(Car=torch.from_numpy(w["anchors_bv"]).to(dev))
        opt = train.build_optimizer(model, cfg.optimizer, 1)
        sched = train.build_scheduler(opt, steps, 1, cfg.optimizer, cfg.lr_config)
        sync = train.GradSync(opt.flat)
        host = [w["frame"](i) for i in range(frames)]
        clouds = [torch.from_numpy(p).to(dev) for p in host]
        gts = [torch.from_numpy(bench.synth_gt_on_points(p, i, 8, "car")).to(dev) for i, p in enumerate(host)]
        types = [np.array(["Car"] * 8) for _ in range(frames)]
        cal = w["cal"]
        curves = {}
        for it in range(steps):
            ids = [(it * batch + j) % frames for j in range(batch)]
            b = train.device_batch([clouds[k] for k in ids], [gts[k] for k in ids], [types[k] for k in ids], ["Car"],
                                   anchors, anchors_bv, cal["voxel_size"], cal["pc_range"], max_points=cal["max_points"],
                                   max_voxels=cal["max_voxels"], model=model)
            loss, terms = train.train_one_iter(model, opt, sched, sync, b, it)
            curves.setdefault("loss", []).append(float(loss))
            for k, v in terms.items():
                curves.setdefault(k, []).append(float(v))
        torch.cuda.synchronize()
        return curves
    finally:
        AG.set_bev_precision(prev)


def summary(curves, tail=20):
    return {k: float(np.mean(v[-tail:])) for k, v in curves.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r6", "train_trajectory.json"))
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    runs = {"fp32": run("fp32", a.steps, dev), "fp32_again": run("fp32", a.steps, dev), "bf16": run("bf16", a.steps, dev)}
    rec = {"workload": "BASELINE configs[2]: car_cfg, batch 2, K21 frames + 8 synthetic car boxes / frame, adam_onecycle, "
                       "%d steps, seed 0" % a.steps,
           "tail_mean_last_20_steps": {k: summary(v) for k, v in runs.items()}, "curves": runs}
    from sassd import _C
    rec["csrc_hash"] = _C.csrc_hash()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rec, open(a.out, "w"))
    for k, v in rec["tail_mean_last_20_steps"].items():
        print(k, {t: round(x, 4) for t, x in v.items()})
    f, g, h = (rec["tail_mean_last_20_steps"][k] for k in ("fp32", "fp32_again", "bf16"))
    for t in f:
        print("%-24s fp32 run-to-run %.3g   bf16 - fp32 %.3g   (fp32 %.4g)" % (t, abs(f[t] - g[t]), abs(h[t] - f[t]), f[t]))
    first = {k: {t: v[t][0] for t in v} for k, v in runs.items()}
    print("step 0:", {k: round(v["loss"], 5) for k, v in first.items()})


if __name__ == "__main__":
    main()
