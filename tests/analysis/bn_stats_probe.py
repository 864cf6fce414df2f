#!/usr/bin/env python
"""How far are the BatchNorm batch statistics of the fused kernels from a float64 evaluation ON THE LIVE ACTIVATIONS of a training
step?  Runs the forward pass of tests/test_gpu_train.py's seeded multi_cfg full-grid case with K.bn2d_relu_fwd / K.bn_relu_fwd
wrapped: every call's saved mean / invstd is compared with torch float64 statistics of the same input.  (Round 6: used to tell
which BatchNorm variant moves the step away from the float64 arbiter.)   python tests/analysis/bn_stats_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sassd  # noqa: E402,F401
from sassd import kernels as K  # noqa: E402
import test_gpu_train as T  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    names = ["Car", "Pedestrian", "Cyclist"]
    case = T.oracle_case("configs/multi_cfg.py", names, T.FULL)
    model = case["model"].to(dev).train()
    npi, gts, types = case["np_inputs"], case["gts"], case["types"]
    kw = dict(voxels=[torch.from_numpy(v).to(dev) for v in npi["voxels"]],
              coordinates=[torch.from_numpy(v).to(dev) for v in npi["coordinates"]],
              num_points=[torch.from_numpy(v).to(dev) for v in npi["num_points"]],
              anchors={n: [torch.from_numpy(a).to(dev) for a in npi["anchors"][n]] for n in names},
              anchors_mask={n: [torch.from_numpy(a).to(dev) for a in npi["anchors_mask"][n]] for n in names},
              gt_bboxes=[torch.from_numpy(g).to(dev) for g in gts],
              gt_labels=[torch.tensor(l, dtype=torch.int64, device=dev) for l in npi["gt_labels"]], gt_types=types)
    rows = []
    f2, f1 = K.bn2d_relu_fwd, K.bn_relu_fwd

    def rec(kind, x, mean, invstd, eps, dims):
        x64 = x.double()
        m64 = x64.mean(dims)
        v64 = x64.var(dims, unbiased=False)
        is64 = 1.0 / torch.sqrt(v64 + eps)
        em = ((mean.double() - m64).abs() / (v64 + eps).sqrt()).max().item()          # mean error in units of the std
        ei = ((invstd.double() - is64).abs() / is64).max().item()
        c = int(((invstd.double() - is64).abs() / is64).argmax())
        rows.append((kind, tuple(x.shape), em, ei, float(m64[c]), float(v64[c])))

    def w2(x, gamma, beta, rm, rv, momentum, eps):
        y, mean, invstd = f2(x, gamma, beta, rm, rv, momentum, eps)
        rec("nchw", x, mean, invstd, eps, (0, 2, 3))
        return y, mean, invstd

    def w1(x, gamma, beta, rm, rv, momentum, eps):
        y, mean, invstd = f1(x, gamma, beta, rm, rv, momentum, eps)
        rec("sparse", x, mean, invstd, eps, (0,))
        return y, mean, invstd
    K.bn2d_relu_fwd, K.bn_relu_fwd = w2, w1
    try:
        model(None, [dict(sample_idx=0), dict(sample_idx=1)], return_loss=True, **kw)
    finally:
        K.bn2d_relu_fwd, K.bn_relu_fwd = f2, f1
    torch.cuda.synchronize()
    for r in rows:
        print("%-6s %-22s mean err / std %.2e   invstd rel err %.2e   (worst channel: mean %.4g var %.3g)" % r)


if __name__ == "__main__":
    main()
