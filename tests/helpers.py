"""Shared test inputs: seeded synthetic frames and random network parameters (same on CPU oracle and GPU)."""
import numpy as np
import torch

import sassd
from sassd import synth
from oracle import nets as onets


def frame(name="k21", seed=0):
    if name == "k21":
        return synth.k21(seed)
    if name == "k17":
        return synth.k17(seed)
    if name == "small":
        return synth.lidar64(seed)[:3000]
    raise KeyError(name)


def random_bn(c, g):
    return dict(weight=torch.rand(c, generator=g) * 0.5 + 0.75, bias=torch.randn(c, generator=g) * 0.1,
                running_mean=torch.randn(c, generator=g) * 0.1, running_var=torch.rand(c, generator=g) + 0.5)


def vxnet_params(seed=0):
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, kind, cin, cout, key in onets.VXNET_LAYERS:
        k = 1 if kind == "1x1" else 27
        w = torch.randn(k, cin, cout, generator=g) * (2.0 / (cin * min(k, 8))) ** 0.5
        p[name] = dict(weight=w, bn=random_bn(cout, g))
    return p


def bev_params(seed=1, cin=320, c=256):
    g = torch.Generator().manual_seed(seed)
    p = {}
    for i in range(8):
        ci = cin if i == 0 else c
        ks = 1 if i == 7 else 3
        w = torch.randn(c, ci, ks, ks, generator=g) * (2.0 / (ci * ks * ks)) ** 0.5
        p["conv%d" % i] = dict(weight=w, bn=random_bn(c, g))
    return p


def head_params(seed=2, c=256, num_class=1, a=2):
    g = torch.Generator().manual_seed(seed)
    na = a * num_class
    p = {
        "conv_box": dict(weight=torch.randn(na * 7, c, 1, 1, generator=g) * 0.02, bias=torch.randn(na * 7, generator=g) * 0.05),
        "conv_cls": dict(weight=torch.randn(na * num_class, c, 1, 1, generator=g) * 0.02,
                         bias=torch.full((na * num_class,), -2.0) + torch.randn(na * num_class, generator=g) * 0.1),
        "conv_dir_cls": dict(weight=torch.randn(na * 2, c, 1, 1, generator=g) * 0.02, bias=torch.randn(na * 2, generator=g) * 0.05),
    }
    return p


def pswarp_params(seed=3, c=256, parts=28):
    g = torch.Generator().manual_seed(seed)
    return {
        "conv0": dict(weight=torch.randn(parts, c, 3, 3, generator=g) * (2.0 / (c * 9)) ** 0.5, bn=random_bn(parts, g)),
        "conv1": dict(weight=torch.randn(parts, parts, 1, 1, generator=g) * (1.0 / parts) ** 0.5),
    }


def fold_bn(bn, eps=1e-3):
    scale = bn["weight"] / torch.sqrt(bn["running_var"] + eps)
    shift = bn["bias"] - bn["running_mean"] * scale
    return scale.contiguous(), shift.contiguous()


def rand_bev_boxes(rng, k, spread=12.0):
    x = rng.uniform(0, spread, k); y = rng.uniform(0, spread, k)
    w = rng.uniform(1.2, 2.2, k); l = rng.uniform(3, 5, k); a = rng.uniform(-3.3, 3.3, k)
    return np.stack([x - w / 2, y - l / 2, x + w / 2, y + l / 2, a], 1).astype(np.float32)
