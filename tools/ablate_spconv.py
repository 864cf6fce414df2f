#!/usr/bin/env python
"""Per-layer timing + ablation of the sparse path on a synthetic frame batch (debug switches of spconv.hip):
bit1 no slab accumulate, bit2 no MFMA, bit4 ticket (dynamic) offset assignment (round-3 kernel only),
bit8 legacy register-stationary kernel, bits 16+ kernel / workgroup geometry.  Also times the rulebook build (fused pyramid vs the per-op chain).

  python tools/ablate_spconv.py [--config car|multi|waymo] [--batch B] [--ablate]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sassd  # noqa: E402
from sassd import kernels as K, synth  # noqa: E402
from sassd.pipeline import InferencePlan  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="car")
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--ablate", action="store_true")
args = ap.parse_args()

dev = torch.device("cuda:0")
w = synth.workload(args.config)
B = args.batch or w["batch"]
model, _ = synth.build_detector_for(w, 0)
sd = model.state_dict()
clouds = [torch.from_numpy(w["frame"](i)).to(dev) for i in range(B)]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


plans = {f: InferencePlan(sd, batch_size=B, anchors=w["anchors"], anchors_bv=w["anchors_bv"], device=dev,
                          fused_rulebooks=f, overlap=False, **w["plan"]) for f in (True, False)}
for f, plan in plans.items():
    plan.run_from_points(clouds)
    torch.cuda.synchronize()
    assert (int(plan.status.item()) & ~4) == 0            # bit 2 = candidate-box overflow of the uncalibrated head
    with K.ws_scope(plan._wsid):
        print("rulebooks %-9s %8.1f us   (rows %s)" % ("fused" if f else "per-op", timeit(plan.rulebooks),
                                                       [int(t.item()) for t in plan.n]))
plan = plans[True]
work = plan.sparse_work()
print("pairs", work["pairs"], "bytes_gs %.1f MB" % (work["bytes_gs"] / 1e6))

# round 4: the balanced kernel (spconv_gq.h) next to the round-3 kernel; the gather / weight-load ablation bits are gone
# (a load under a branch makes the compiler drain vmcnt(0) in front of every tile -- the very thing being measured)
MODES = [("r3", 10 << 16), ("gs4w", 1 << 16), ("gs8w", 5 << 16), ("gq4x4w", 8 << 16), ("gq16x4w", 9 << 16), ("default", 0)]
layers = []
lvl = 0
for kind, cin, cout, key, wp, scale, shift in plan.sp:
    if kind == "down":
        lvl += 1
    layers.append((kind, cin, cout, key, wp, scale, shift, lvl))
seen = set()
tot = {m: 0.0 for m, _ in MODES}
for kind, cin, cout, key, wp, scale, shift, lvl in layers:
    nin = plan.caps[lvl - 1] if kind == "down" else plan.caps[lvl]
    x = torch.randn(nin, cin, device=dev)
    y = torch.empty(plan.caps[lvl], cout, device=dev)
    nbr = plan.nbr[key] if key else None
    k = 27 if key else 1
    line = "%-5s %-6s %2d->%2d rows %7d:" % (kind, key or "-", cin, cout, work["n"][lvl])
    for mname, flags in MODES:
        us = timeit(lambda: K.spconv_fwd(x, nbr, plan.n[lvl], plan.caps[lvl], wp, k, cin, cout, scale, shift, True, y, cfg=flags))
        tot[mname] += us
        line += "  %s %7.1f us" % (mname, us)
    print(line)
    if args.ablate and (kind, key, cin) not in seen and cin >= 16 and key:
        seen.add((kind, key, cin))
        for mname, base in [MODES[0], MODES[2], MODES[3]]:
            s = "        ablation %-8s" % mname
            bits = ((2, "no-slab"), (4, "no-mfma"), (6, "neither")) if mname == "r3" else \
                   ((4, "no-mfma"), (32, "hot-gather"), (64, "hot-w"), (96, "hot-both"), (100, "skeleton"))
            for bit, nm in bits:
                s += "  %s %6.1f" % (nm, timeit(lambda: K.spconv_fwd(x, nbr, plan.n[lvl], plan.caps[lvl], wp, k, cin,
                                                                      cout, scale, shift, True, y, cfg=base | bit)))
            print(s)
print("sum of 14 layers:", {m: round(v, 1) for m, v in tot.items()}, "us;  bytes_gs GB/s:",
      {m: round(work["bytes_gs"] / v / 1e3, 1) for m, v in tot.items()})
