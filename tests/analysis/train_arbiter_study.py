#!/usr/bin/env python
"""How far apart are the CPU oracle's own variants of ONE training step?  (analysis script; its summary is what
tests/test_train_arbiter_cpu.py asserts and what the GPU bars in tests/test_gpu_train.py are derived from)

    python tests/analysis/train_arbiter_study.py [car|multi] [half|full]

Runs oracle.train_ref.train_step on the two-cloud workload of test_training_step_vs_oracle as
  f32   the fp32 oracle                                f64   the same step in float64 (the arbiter)
  b32   dense-conv operands rounded to bf16, fp32      b64   rounded operands, float64 everywhere else
and prints, per pair, the relative L2 distance of the whole-model gradient and the worst tensors.  d(f32, f64) is the fp32
floor of the step; d(b32, b64) is the floor of the ROUNDED step: two runs that round the same way but whose activations
differ in the last fp32 bits round a few operands to different bf16 neighbours (a 2^-8 jump each), and those jumps feed the
next layer's roundings."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def dist(ga, gb, names):
    num = sum(float((ga[n].double() - gb[n].double()).pow(2).sum()) for n in names)
    den = sum(float(gb[n].double().pow(2).sum()) for n in names)
    per = {n: float((ga[n].double() - gb[n].double()).norm() / max(float(gb[n].double().norm()), 1e-30)) for n in names}
    return (num / den) ** 0.5, sorted(per.items(), key=lambda kv: -kv[1])[:5]


def main():
    import test_gpu_train as T
    from oracle import train_ref
    which = sys.argv[1] if len(sys.argv) > 1 else "car"
    grid = T.FULL if (len(sys.argv) > 2 and sys.argv[2] == "full") else T.HALF
    cfg, names = ("configs/car_cfg.py", ["Car"]) if which == "car" else ("configs/multi_cfg.py", ["Car", "Pedestrian", "Cyclist"])
    case = T.oracle_case(cfg, names, grid)
    out = {}
    sel = None
    for tag, kw in (("f32", {}), ("f64", dict(dtype=torch.float64)), ("b32", dict(bf16=("bev",))),
                    ("b64", dict(bf16=("bev",), dtype=torch.float64))):
        t0 = time.time()
        l, g, ex = train_ref.train_step(*case["ref_args"], guided_sel=sel, **kw)
        if sel is None:
            sel = ex["guided_sel"]            # every variant differentiates the same candidate set
        out[tag] = (l, {k: v for k, v in g.items() if v is not None})
        print(tag, "%.1f s" % (time.time() - t0), {k: round(v, 6) for k, v in l.items()}, flush=True)
    names_ = [k for k in out["f64"][1] if float(out["f64"][1][k].norm()) > 1e-7]
    for a, b in (("f32", "f64"), ("b32", "b64"), ("b32", "f32"), ("b64", "f64")):
        whole, worst = dist(out[a][1], out[b][1], names_)
        print("d(%s, %s): whole-model %.2e; worst %s" % (a, b, whole, [(k, "%.1e" % v) for k, v in worst]))
        print("   losses rel:", {k: "%.1e" % (abs(out[a][0][k] - out[b][0][k]) / max(abs(out[b][0][k]), 1e-12)) for k in out[a][0]})


if __name__ == "__main__":
    main()
