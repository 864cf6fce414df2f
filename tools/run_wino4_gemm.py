#!/usr/bin/env python
"""Time the Winograd GEMM launch alone (36 GEMMs 256 x 256 x tiles, B=1 @200x176) on a random transformed input, with the
A/B and ablation switches of the Winograd cfg word:   python tools/run_wino4_gemm.py [--reps 100]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sassd  # noqa: E402,F401
from sassd import kernels as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=100)
ap.add_argument("--batch", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0")
B, C, H, W = a.batch, 256, 200, 176
w = torch.randn(256, C, 3, 3, device=dev) * (2.0 / (C * 9)) ** 0.5
w4 = K.conv2d_wino4_pack_weight(w)
ws = K.conv2d_wino4_chain_workspace(B, 256, H, W, dev)
ws.view(torch.float32).normal_()                                     # V (and M) regions: dense random operands
sc, sh = torch.ones(256, device=dev), torch.zeros(256, device=dev)


def timed(flags):
    call = lambda: K.conv2d_wino4_chain(None, (sc, sh, True), w4, 256, 256, 256, B, H, W, sc, sh, True, None, ws,  # noqa: E731
                                        cfg=K.wino4_cfg(0, flags))
    for _ in range(max(10, a.reps // 2)):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps


GEMM_ONLY = 128 | 16                             # skip the fused transform and the input transform
out = {}
for name, flags in (("gemm_ms", 0), ("gemm_no_dma_ms", 1), ("gemm_no_mfma_ms", 2), ("gemm_skeleton_ms", 3),
                    ("gemm_again_ms", 0)):
    out[name] = timed(GEMM_ONLY | flags)
flops = 2.0 * 36 * 256 * 256 * B * (H // 4) * (W // 4)
out["gemm_tflops"] = flops / min(out["gemm_ms"], out["gemm_again_ms"]) / 1e9
print(json.dumps(out, indent=1))
