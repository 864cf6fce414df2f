#!/usr/bin/env python
"""Time the Winograd and the direct BEV 3x3 conv (256->256 and 320->256 @200x176) back to back."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sassd  # noqa: E402,F401
from sassd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for b, cin in ((1, 256), (1, 320), (2, 256)):
    x = torch.randn(b, cin, 200, 176, generator=g).to(dev)
    w = (torch.randn(256, cin, 3, 3, generator=g) * 0.02).to(dev)
    sc, sh = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    wd, ww = K.conv2d_pack_weight(w), K.conv2d_wino_pack_weight(w)
    y = torch.empty(b, 256, 200, 176, device=dev)
    for name, fn in (("direct", lambda: K.conv2d_fwd(x, wd, 256, 3, sc, sh, True, y)),
                     ("winograd", lambda: K.conv2d_wino_fwd(x, ww, 256, sc, sh, True, y))):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print("B=%d %d->256 3x3 %-8s %.4f ms  (%.1f direct-equivalent TFLOP/s)" %
              (b, cin, name, ms, 2 * 256 * cin * 9 * 200 * 176 * b / ms / 1e9))



# conv0-like input: 320 channels, 92 % of the pixels empty (BEV occupancy of a densified sparse tensor)
x = torch.randn(1, 320, 200, 176, generator=g)
occ = (torch.rand(1, 5, 200, 176, generator=g) < 0.03).float().repeat_interleave(64, 1)
x = (x * occ).to(dev)
w = (torch.randn(256, 320, 3, 3, generator=g) * 0.02).to(dev)
ww, wd = K.conv2d_wino_pack_weight(w), K.conv2d_pack_weight(w)
y = torch.empty(1, 256, 200, 176, device=dev)
ref = K.conv2d_fwd(x, wd, 256, 3)
got = K.conv2d_wino_fwd(x, ww, 256)
print("sparse conv0 max |wino - direct| = %.3e" % (got - ref).abs().max().item())
for name, fn in (("direct", lambda: K.conv2d_fwd(x, wd, 256, 3, None, None, True, y)),
                 ("winograd", lambda: K.conv2d_wino_fwd(x, ww, 256, None, None, True, y))):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("sparse 320->256 %-8s %.4f ms" % (name, e0.elapsed_time(e1) / n))
