"""-m gpu: the fused clip + weight-decay + Adam kernels (sassd_grad_sumsq / sassd_adam_step) driven by
sassd.train.AdamOneCycle / OneCycle against the parameter trajectory of the reference's own optimizer stack
(tests/golden/make_golden_optim.py -> optim_ref.npz: 20 iterations, every third one clipped)."""
import os

import numpy as np
import pytest
import torch

import sassd  # noqa: F401
from sassd import kernels as K
from sassd import train

pytestmark = pytest.mark.gpu
O = np.load(os.path.join(os.path.dirname(__file__), "golden", "optim_ref.npz"))


def test_grad_sumsq(dev):
    for n in (1, 3, 4, 1023, 5_340_001):
        g = torch.randn(n + 4, device=dev)[:n] if n % 4 else torch.randn(n, device=dev)
        g = g.contiguous()
        got = K.grad_sumsq(g).item()
        ref = float((g.double() ** 2).sum())
        assert abs(got - ref) <= 1e-4 * max(ref, 1.0), (n, got, ref)


def test_adam_onecycle_trajectory(dev):
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.ReLU(),
                                torch.nn.Linear(5, 3, bias=False))
    names = [str(n) for n in O["names"]]
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(O["init/" + n]))
    model = model.to(dev)
    opt = train.build_optimizer(model, dict(type="adam_onecycle", lr=0.003, weight_decay=0.01,
                                            grad_clip=dict(max_norm=10, norm_type=2)))
    sched = train.build_scheduler(opt, 5, 4, dict(lr=0.003), dict(policy="onecycle", moms=[0.95, 0.85],
                                                                   div_factor=10, pct_start=0.4))
    params = dict(model.named_parameters())
    assert sorted(params) == sorted(names)
    for it in range(20):
        sched.step(it)
        opt.zero_grad()
        for n in names:          # zero_grad() leaves .grad = None; the optimizer gathers whatever autograd (or we) set
            params[n].grad = torch.from_numpy(O["grad%d/%s" % (it, n)]).to(dev)
        opt.step()
        for n in names:
            ref = O["step%d/%s" % (it, n)]
            err = np.abs(params[n].detach().cpu().numpy() - ref).max()
            assert err <= 2e-6 * max(1.0, np.abs(ref).max()), (it, n, err)
