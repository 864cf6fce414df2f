#!/usr/bin/env python
"""Fixed vs per-chunk cost of the BEV conv kernel: time vs Cin at Cout=256, 200x176."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sassd
from sassd import kernels as K
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for cin in (8, 16, 64, 128, 256, 320):
    x = torch.randn(1, cin, 200, 176, generator=g).to(dev)
    w = (torch.randn(256, cin, 3, 3, generator=g) * 0.02).to(dev)
    wp = K.conv2d_pack_weight(w)
    y = torch.empty(1, 256, 200, 176, device=dev)
    for _ in range(5):
        K.conv2d_fwd(x, wp, 256, 3, None, None, True, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        K.conv2d_fwd(x, wp, 256, 3, None, None, True, y)
    e1.record(); torch.cuda.synchronize()
    print("Cin=%3d chunks=%2d  %.1f us" % (cin, cin // 8, e0.elapsed_time(e1) / 20 * 1e3))
