"""The DDP exchange of the training loop on ONE GPU (launched by tests/test_gpu_multi.py): a one-rank `nccl` (RCCL)
communicator, GradSync forced on, and a real forward_train / backward of the car_cfg detector -- buckets launched from the
post-accumulate hooks during backward as asynchronous collectives on the HIP stream, `comm_ms()` measured, gradients
equal to the step without any exchange (a sum over one rank), `reset()` after an aborted step.  Reference:
tools/train.py:78 (MMDistributedDataParallel), mmdet/core/utils/dist_utils.py:9-41."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sassd  # noqa: E402,F401
import bench  # noqa: E402
from sassd import dist as D, synth, train  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    rank, _, world = D.init("nccl", force_single=True)
    assert world == 1 and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
    w = synth.workload("car")
    model, cfg = synth.build_detector_for(w, 0, train=True, cls_bias=-3.0)
    model = model.to(dev).train()
    anchors = dict(Car=torch.from_numpy(w["anchors"]).to(dev))
    anchors_bv = dict(Car=torch.from_numpy(w["anchors_bv"]).to(dev))
    opt = train.build_optimizer(model, cfg.optimizer, world)
    host = [w["frame"](i)[::2] for i in range(2)]
    clouds = [torch.from_numpy(np.ascontiguousarray(p)).to(dev) for p in host]
    gts = [torch.from_numpy(bench.synth_gt_on_points(p, i, 6, "car")).to(dev) for i, p in enumerate(host)]
    types = [np.array(["Car"] * 6) for _ in host]
    cal = w["cal"]

    def batch():
        return train.device_batch(clouds, gts, types, ["Car"], anchors, anchors_bv, cal["voxel_size"], cal["pc_range"],
                                  max_points=cal["max_points"], max_voxels=cal["max_voxels"], model=model)

    def fwd_bwd():
        opt.zero_grad()
        loss, _ = train.parse_losses(model(**batch()))
        loss.backward()
        return float(loss)

    # 1. the step without any exchange
    off = train.GradSync(opt.flat)
    assert not off.on, "a one-rank group must not switch the exchange on by itself"
    l0 = fwd_bwd()
    opt.flat.collect(0, len(opt.flat.params))
    g0 = opt.flat.grad.clone()
    assert torch.isfinite(g0).all() and float(g0.abs().sum()) > 0

    # 2. the same step with the exchange forced through RCCL
    sync = train.GradSync(opt.flat, buckets=4, time_comm=True, force=True)
    assert sync.on and sync.overlap and len(sync.ranges) == 4
    sync.reset()
    l1 = fwd_bwd()
    launched_in_backward = sync._next
    assert launched_in_backward >= 1, "no bucket left from a gradient hook during backward"
    sync.all_reduce_grads()
    ms = sync.comm_ms()
    assert ms is not None and ms > 0.0, ms
    torch.cuda.synchronize()
    g1 = opt.flat.grad.clone()
    # atomics in the auxiliary-head scatter / the sampling backward make two runs differ in the last bits
    rel = float((g1 - g0).norm() / g0.norm())
    assert abs(l1 - l0) <= 1e-4 * abs(l0) and rel < 1e-4, (l0, l1, rel)

    # 3. an aborted step (backward ran, nobody waited) followed by reset() and a clean step
    fwd_bwd()
    assert sync.works, "the aborted step left no collective in flight"
    sync.reset()
    assert not sync.works and sync._next == 0
    fwd_bwd()
    sync.all_reduce_grads()
    torch.cuda.synchronize()
    rel = float((opt.flat.grad - g0).norm() / g0.norm())
    assert rel < 1e-4, rel
    opt.step()
    torch.cuda.synchronize()
    print("RCCL_SINGLE_OK buckets_from_hooks=%d allreduce_ms=%.3f" % (launched_in_backward, ms))
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
