#!/usr/bin/env python
"""Run the sparse segment (7 rulebooks + 14 sparse convs) of a workload a few times -- the target of rocprofv3
--kernel-trace / --pmc passes for roofline_sparse.   python tools/run_sparse_only.py [--config multi] [--reps 5]"""
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
ap.add_argument("--config", default="multi")
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--graph", action="store_true", help="replay the segment as its own hipGraph (what roofline_sparse times) "
                "instead of issuing it eagerly on two streams")
ap.add_argument("--no-overlap", action="store_true", help="rulebooks on the main stream (no side stream)")
ap.add_argument("--spconv-cfg", type=int, default=0, help="kernel / workgroup geometry switch of the sparse conv (spconv.hip)")
args = ap.parse_args()
dev = torch.device("cuda:0")
K.DEFAULT_CFG["spconv"] = K.spconv_cfg(args.spconv_cfg)      # host-side default of the binding: passed per call
w = synth.workload(args.config)
B = args.batch or w["batch"]
model, _ = synth.build_detector_for(w, 0)
plan = InferencePlan(model.state_dict(), batch_size=B, anchors=w["anchors"], anchors_bv=w["anchors_bv"], device=dev,
                     **w["plan"])
clouds = [torch.from_numpy(w["frame"](i)).to(dev) for i in range(B)]
if args.no_overlap:
    plan.overlap = False
if args.graph:
    with K.ws_scope(plan._wsid):
        plan.voxelize(clouds)
    plan.capture(w["points_cap"], stages=("sparse",))
    for _ in range(args.reps):
        plan.graph.launch()
        torch.cuda.synchronize()
else:
    with K.ws_scope(plan._wsid):
        plan.voxelize(clouds)
        for _ in range(args.reps):
            plan.backbone(densify=False, masks=False)
            if plan.overlap:
                torch.cuda.current_stream().wait_event(plan.mask_ev)
            torch.cuda.synchronize()
torch.cuda.synchronize()
work = plan.sparse_work()
print({k: work[k] for k in ("bytes_gs", "bytes_min", "rulebook_bytes", "flops", "n")})
