#!/usr/bin/env python
"""Run the Winograd BEV conv (256->256 3x3 @200x176, B=1) a few times -- target for rocprofv3 --pmc passes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sassd  # noqa: E402,F401
from sassd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
b = int(sys.argv[2]) if len(sys.argv) > 2 else 1
x = torch.randn(b, 256, 200, 176, generator=g).to(dev)
w = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).to(dev)
ww = K.conv2d_wino_pack_weight(w)
sc, sh = torch.ones(256, device=dev), torch.zeros(256, device=dev)
y = torch.empty(b, 256, 200, 176, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    K.conv2d_wino_fwd(x, ww, 256, sc, sh, True, y)
torch.cuda.synchronize()
