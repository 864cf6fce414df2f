#!/usr/bin/env python
"""Time the bf16 1x1 convolution (sassd_conv1x1_bf16_fwd) on the training shapes next to the fp32-MFMA direct kernel it replaces
under set_bev_precision("bf16") -- also the target of rocprofv3 passes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sassd  # noqa: E402,F401
from sassd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for b, cin, cout in ((2, 256, 256), (2, 20, 256), (2, 256, 20), (1, 256, 256), (4, 256, 256)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(b, cin, 200, 176, generator=g).to(dev)
    w = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(dev)
    pk = K.conv1x1_bf16_pack_weight(w)
    pf = K.conv2d_pack_weight(w)
    y = torch.empty(b, cout, 200, 176, device=dev)
    t_b = timeit(lambda: K.conv1x1_bf16_fwd(x, pk, cout, None, y))
    t_f = timeit(lambda: K.conv2d_fwd(x, pf, cout, 1, None, None, False, y))
    mb = (x.numel() + y.numel()) * 4 / 1e6
    print("1x1 %d -> %d, batch %d: bf16 %.1f us (%.2f TB/s over %.0f MB), fp32 direct %.1f us" % (cin, cout, b, t_b, mb / t_b, mb, t_f))
