"""One rank of the multi-GPU RCCL check (launched by tests/test_gpu_multi.py through torch.distributed.run): the
training path's parameter broadcast + bucketed, overlapped gradient all-reduce over the flat buffers on real GPUs, and
one fused optimizer step that must leave every rank with identical weights."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sassd  # noqa: E402,F401
from sassd import dist as D, train  # noqa: E402


def main():
    rank, local_rank, world = D.init("nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.manual_seed(100 + rank)
    m = torch.nn.Sequential(torch.nn.Linear(64, 512), torch.nn.ReLU(), torch.nn.Linear(512, 512), torch.nn.ReLU(),
                            torch.nn.Linear(512, 8)).to(dev)
    opt = train.AdamOneCycle(m, lr=1e-3, weight_decay=0.01, grad_clip=dict(max_norm=10, norm_type=2), world_size=world)
    sync = train.GradSync(opt.flat, buckets=3)
    ref = opt.flat.data.clone()
    torch.distributed.broadcast(ref, src=0)
    assert torch.equal(ref, opt.flat.data), "parameter broadcast"
    x = torch.randn(32, 64, device=dev) * (rank + 1)
    # local gradient WITHOUT the exchange: autograd.grad does not run the post-accumulate hooks that launch the buckets
    gl = torch.autograd.grad(m(x).pow(2).mean(), opt.flat.params)
    local = torch.zeros_like(opt.flat.data)
    for g, o in zip(gl, opt.flat.offsets):
        local[o:o + g.numel()] = g.reshape(-1)
    opt.zero_grad()
    m(x).pow(2).mean().backward()                 # buckets go out from the gradient hooks during this backward
    sync.all_reduce_grads()
    total = local.clone()
    torch.distributed.all_reduce(total)
    assert torch.allclose(opt.flat.grad, total, rtol=1e-5, atol=1e-6), "bucketed all-reduce != one all-reduce"
    opt.step()
    w = opt.flat.data.clone()
    torch.distributed.broadcast(w, src=0)
    assert torch.equal(w, opt.flat.data), "weights diverged across ranks after the update"
    # weight images after the broadcast: per-rank seeds differ, PackPlan ran BEFORE the broadcast inside
    # build_optimizer -- every rank's packed images must be packs of the broadcast (rank 0) weights
    from sassd import kernels as K, spconv
    from sassd.detector import _HipConv2d

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.sp = spconv.SubMConv3d(16, 32, 3, bias=False, indice_key="k")
            self.cv = _HipConv2d(32, 14, 1)

    torch.manual_seed(200 + rank)
    net = Net().to(dev)
    opt2 = train.build_optimizer(net, dict(type="adam_onecycle", lr=1e-3, weight_decay=0.01), world)
    assert opt2.pack_plan is not None
    stale = net.sp.packed_weight().clone()
    train.GradSync(opt2.flat, buckets=2)
    fresh = K.spconv_pack_weight(net.sp.weight.detach().reshape(27, 16, 32).contiguous())
    assert torch.equal(net.sp.packed_weight(), fresh), "sparse pack is not a pack of the broadcast weights"
    assert torch.equal(net.cv.packed_weight(), K.conv2d_pack_weight(net.cv.weight.detach().contiguous()))
    if rank != 0:
        assert not torch.equal(stale, fresh), "per-rank seeds should differ"
    D.barrier()
    if rank == 0:
        print("RCCL_OK world=%d" % world)


if __name__ == "__main__":
    main()
