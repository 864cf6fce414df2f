#!/usr/bin/env python
"""Run the BEV 3x3 256->256 conv (the dominant kernel) a few times -- target for rocprofv3 --pmc passes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sassd  # noqa: E402
from sassd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn(1, 256, 200, 176, generator=g).to(dev)
CO = int(os.environ.get("CONV_COUT", "256"))          # 28 = the part-sensitive head's narrow 3x3 layer
w = (torch.randn(CO, 256, 3, 3, generator=g) * 0.02).to(dev)
wp = K.conv2d_pack_weight(w)
sc = torch.ones(CO, device=dev)
sh = torch.zeros(CO, device=dev)
y = torch.empty(1, CO, 200, 176, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for _ in range(n):
    K.conv2d_fwd(x, wp, CO, 3, sc, sh, True, y)
torch.cuda.synchronize()
flags_list = [0] if len(sys.argv) < 3 else [int(v) for v in sys.argv[2].split(",")]
for flags in flags_list:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        K.conv2d_fwd(x, wp, CO, 3, sc, sh, True, y, cfg=flags)           # per-call ablation word (sassd_conv2d_fwd_cfg)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("flags=%d conv 256->%d 3x3 @200x176: %.4f ms  %.1f TFLOP/s" % (flags, CO, ms, 2 * CO * 256 * 9 * 200 * 176 / ms / 1e9))
