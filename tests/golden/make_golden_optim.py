"""Generate tests/golden/optim_ref.npz by running the reference's own optimizer stack on the CPU (build container
only): tools/train_utils/optimization/{__init__,fastai_optim,learning_schedules_fastai}.py -- build_optimizer(
'adam_onecycle') + build_scheduler('onecycle') + clip_grad_norm_ exactly as train_one_epoch calls them
(tools/train_utils/__init__.py:39-61) -- on a tiny Linear/BatchNorm model with seeded synthetic gradients.

    python tests/golden/make_golden_optim.py
"""
import collections
import collections.abc
import importlib.util
import os

import numpy as np
import torch
from torch import nn
from torch.nn.utils import clip_grad_norm_

collections.Iterable = collections.abc.Iterable          # the reference predates python 3.10
HERE = os.path.dirname(os.path.abspath(__file__))
PKG = "/root/reference/tools/train_utils/optimization"


class AttrDict(dict):
    __getattr__ = dict.__getitem__


def tiny_model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(6, 5), nn.BatchNorm1d(5), nn.ReLU(), nn.Linear(5, 3, bias=False))


def main():
    spec = importlib.util.spec_from_file_location("ref_optimization", os.path.join(PKG, "__init__.py"),
                                                  submodule_search_locations=[PKG])
    ro = importlib.util.module_from_spec(spec)
    import sys
    sys.modules["ref_optimization"] = ro
    spec.loader.exec_module(ro)
    model = tiny_model()
    ocfg = AttrDict(type="adam_onecycle", lr=0.003, weight_decay=0.01, grad_clip=AttrDict(max_norm=10, norm_type=2))
    lcfg = AttrDict(policy="onecycle", moms=[0.95, 0.85], div_factor=10, pct_start=0.4)
    opt = ro.build_optimizer(model, ocfg)
    sched, _ = ro.build_scheduler(opt, total_iters_each_epoch=5, total_epochs=4, last_epoch=-1, optim_cfg=ocfg,
                                  lr_cfg=lcfg)
    names = [n for n, _ in model.named_parameters()]
    out = {"init/" + n: p.detach().clone().numpy() for n, p in model.named_parameters()}
    g = torch.Generator().manual_seed(1)
    lrs, moms = [], []
    for it in range(20):
        sched.step(it)
        lrs.append(float(opt.lr)); moms.append(float(opt.mom))
        opt.zero_grad()
        scale = 40.0 if it % 3 == 0 else 0.5                          # every third step exceeds max_norm -> clipped
        for n, p in model.named_parameters():
            p.grad = torch.randn(p.shape, generator=g) * scale
            out["grad%d/%s" % (it, n)] = p.grad.clone().numpy()
        clip_grad_norm_(model.parameters(), **ocfg.grad_clip)
        opt.step()
        for n, p in model.named_parameters():
            out["step%d/%s" % (it, n)] = p.detach().clone().numpy()
    out["lr"], out["mom"], out["names"] = np.array(lrs), np.array(moms), np.array(names)
    np.savez_compressed(os.path.join(HERE, "optim_ref.npz"), **out)
    print("optim_ref.npz", len(out), lrs[:3], moms[:3])


if __name__ == "__main__":
    main()
