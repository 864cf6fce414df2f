#!/usr/bin/env python
"""Ablation timing of the 64->64 submanifold sparse conv on a K21 frame (debug switches in spconv.hip):
bit0 no gather loads, bit1 no LDS scatter-accumulate, bit2 no MFMA, bit3 no weight-fragment loads."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sassd  # noqa: E402
from sassd import _C, kernels as K, synth  # noqa: E402
import bench  # noqa: E402
from sassd.pipeline import InferencePlan  # noqa: E402

dev = torch.device("cuda:0")
model, an, bv, cal = bench.build_model(0)
plan = InferencePlan(model.state_dict(), batch_size=1, anchors=an, anchors_bv=bv, device=dev)
plan.run_from_points([torch.from_numpy(synth.k21(0)).to(dev)])
torch.cuda.synchronize()
lib = _C.lib()
lib._handle  # noqa
setdbg = ctypes.CDLL(_C.LIB_PATH).sassd_debug_set_spconv
kind, cin, cout, key, wp, scale, shift = plan.sp[12]      # conv3.2 (subm3, 64->64)
x = plan.feat[0].clone()
y = torch.empty_like(x)
for lvl, key in ((3, "subm3"), (2, "subm2")):
    for flags in (0,):
        setdbg(flags)
        for _ in range(3):
            K.spconv_fwd(x, plan.nbr[key], plan.n[lvl], plan.caps[lvl], wp, 27, 64, 64, scale, shift, True, y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            K.spconv_fwd(x, plan.nbr[key], plan.n[lvl], plan.caps[lvl], wp, 27, 64, 64, scale, shift, True, y)
        e1.record()
        torch.cuda.synchronize()
        print("%s flags=%2d  %.1f us" % (key, flags, e0.elapsed_time(e1) / 20 * 1e3))
setdbg(0)
