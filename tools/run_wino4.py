#!/usr/bin/env python
"""Time the three kernels of the Winograd F(4x4,3x3) BEV conv (256->256 @200x176, B=1 by default) next to the fused
F(2x2) kernel -- also the target for rocprofv3 --kernel-trace / --pmc passes."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sassd  # noqa: E402
from sassd import kernels as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--cin", type=int, default=256)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--cfgs", default="", help="comma-separated geometry numbers of the Winograd cfg word (include/sassd.h) (default: all)")
ap.add_argument("--profile", action="store_true", help="default geometry only, `reps` launches: the target of rocprofv3 "
                "--kernel-trace / --pmc passes")
args = ap.parse_args()
dev = torch.device("cuda:0")
B, C, H, W = args.batch, args.cin, 200, 176
x = torch.randn(B, C, H, W, device=dev).clamp(min=0)
w = torch.randn(256, C, 3, 3, device=dev) * (2.0 / (C * 9)) ** 0.5
sc, sh = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1
y = torch.empty(B, 256, H, W, device=dev)
w4, w2 = K.conv2d_wino4_pack_weight(w), K.conv2d_wino_pack_weight(w)
ws = K.conv2d_wino4_workspace(B, C, 256, H, W, dev)


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps * 1e3


if args.profile:
    for _ in range(args.reps):
        K.conv2d_wino4_fwd(x, w4, 256, sc, sh, True, y, ws)
    torch.cuda.synchronize()
    sys.exit(0)
from sassd import _C  # noqa: E402
flops = 2.0 * 256 * C * 9 * H * W * B
ref = K.conv2d_wino_fwd(x, w2, 256, sc, sh, True).clone()
# fp64 reference: im2col + dgemm on the device
with torch.no_grad():
    cols = torch.nn.functional.unfold(x.double(), 3, padding=1)                       # [B, C*9, H*W]
    ref64 = torch.matmul(w.double().reshape(256, -1), cols).reshape(B, 256, H, W)
    ref64 = (ref64 * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]).clamp(min=0)
    del cols
print("F(2x2) fused vs fp64: max abs %.2e   (max |y| %.2f)" % ((ref.double() - ref64).abs().max().item(), ref64.abs().max().item()))
names = {0: "split 128x128 (default)", 1: "fp32 auto", 2: "fp32 128x64", 3: "fp32 128x96", 4: "fp32 128x128", 5: "fp32 128x160",
         6: "fp32 128x192", 11: "split 128x64", 12: "split 128x128", 13: "split 128x192", 14: "split 128x128 kc16"}
cfgs = tuple(int(c) for c in args.cfgs.split(",")) if args.cfgs else (0, 1, 2, 3, 4, 5, 6, 11, 12, 13, 14)
for cfg in cfgs:
    line = "cfg %2d %-22s" % (cfg, names[cfg])
    split = cfg >= 11 or cfg == 0
    modes = ((0, "full"), (1, "no-dma"), (4, "no-split-valu")) if split else ((0, "full"), (1, "no-dma"), (2, "no-mfma"), (3, "neither"))
    for dbg, nm in modes:
        word = K.wino4_cfg(cfg, dbg)
        t4 = timeit(lambda: K.conv2d_wino4_fwd(x, w4, 256, sc, sh, True, y, ws, cfg=word))
        line += "  %s %6.1f us" % (nm, t4)
        if dbg == 0:
            line += " (%.0f TF eq, err vs F(2x2) %.1e, vs fp64 %.2e rms %.2e)" % (
                flops / t4 / 1e6, (y - ref).abs().max().item(), (y.double() - ref64).abs().max().item(),
                (y.double() - ref64).pow(2).mean().sqrt().item())
    print(line, flush=True)
t2 = timeit(lambda: K.conv2d_wino_fwd(x, w2, 256, sc, sh, True, y))
print("F(2x2) fused: %.1f us (%.1f TF direct-equivalent)" % (t2, flops / t2 / 1e6))
