#!/usr/bin/env python
"""Winograd conv time vs number of 16-channel chunks (Cin) at fixed Cout=256, 200x176, B=1: fixed per-workgroup cost vs
per-chunk cost."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sassd  # noqa: E402,F401
from sassd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for cin in (32, 64, 128, 256, 512):
    x = torch.randn(1, cin, 200, 176, generator=g).to(dev)
    w = (torch.randn(256, cin, 3, 3, generator=g) * 0.02).to(dev)
    ww = K.conv2d_wino_pack_weight(w)
    y = torch.empty(1, 256, 200, 176, device=dev)
    for _ in range(3):
        K.conv2d_wino_fwd(x, ww, 256, None, None, True, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        K.conv2d_wino_fwd(x, ww, 256, None, None, True, y)
    e1.record()
    torch.cuda.synchronize()
    print("Cin=%3d (%2d chunks of 32): %.4f ms" % (cin, cin // 32, e0.elapsed_time(e1) / 20))
