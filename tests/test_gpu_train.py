"""-m gpu parity of the TRAINING path: autograd wrappers over the HIP forward / backward kernels
against torch-CPU references, and one whole training step (forward_train: losses + every parameter gradient)
against the CPU oracle oracle/train_ref.py on the same seeded inputs and weights.  Floating point: losses within
1e-3 relative, gradients within 2e-3 relative L2 per tensor (fp32 sums in a different order through ~30 layers with
batch-statistics BatchNorm); index / label work exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import sassd
from sassd import synth, anchors as A
from sassd import kernels as K
from sassd.autograd import Conv2dFn, PSWarpFn
from sassd.config import Config
from sassd.detector import build_detector
import helpers as H

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / max(b.norm().item(), 1e-12))


@pytest.mark.parametrize("cin,cout,ks,hw", [(32, 64, 3, (40, 48)), (256, 14, 1, (24, 32)), (28, 28, 1, (20, 24)),
                                            (256, 28, 3, (16, 24)), (320, 256, 3, (50, 88)), (256, 256, 3, (13, 180)),
                                            (72, 136, 1, (9, 44))])
def test_conv2d_autograd(dev, cin, cout, ks, hw):
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(2, cin, *hw, generator=g)
    w = torch.randn(cout, cin, ks, ks, generator=g) * (1.0 / (cin * ks * ks)) ** 0.5
    b = torch.randn(cout, generator=g)
    dy = torch.randn(2, cout, *hw, generator=g)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    F.conv2d(xr, wr, br, 1, ks // 2).backward(dy)
    xd, wd, bd = (t.to(dev).requires_grad_() for t in (x, w, b))
    y = Conv2dFn.apply(xd, wd, bd, None, None)
    assert _rel(y, F.conv2d(x, w, b, 1, ks // 2)) < 1e-5
    y.backward(dy.to(dev))
    assert _rel(xd.grad, xr.grad) < 1e-5
    assert _rel(wd.grad, wr.grad) < 1e-5          # sassd_conv2d_bwd_weight (split-K MFMA)
    assert _rel(bd.grad, br.grad) < 1e-5


def test_pswarp_backward_vs_grid_sample(dev):
    """d(logits)/d(feature map) and d(logits)/d(box) of the HIP part-sensitive warp against torch grid_sample
    autograd on the CPU (the reference's own formulation, ssd_rotate_head.py:374-447)."""
    g = torch.Generator().manual_seed(5)
    feat = torch.randn(1, 28, 200, 176, generator=g)
    k = 300
    boxes = torch.zeros(k, 7)
    boxes[:, 0] = torch.rand(k, generator=g) * 74 - 2          # a few partly outside the map
    boxes[:, 1] = torch.rand(k, generator=g) * 84 - 42
    boxes[:, 2] = -1.0
    boxes[:, 3] = 1.6 + torch.rand(k, generator=g)
    boxes[:, 4] = 3.9 + torch.rand(k, generator=g)
    boxes[:, 5] = 1.5
    boxes[:, 6] = (torch.rand(k, generator=g) - 0.5) * 6.3
    dl = torch.randn(k, generator=g)
    fr, br = feat.clone().requires_grad_(), boxes.clone().requires_grad_()
    n = k
    ct, st = torch.cos(br[:, 6]).view(n, 1, 1), torch.sin(br[:, 6]).view(n, 1, 1)
    xx = torch.linspace(-.5, .5, 4).view(1, 4, 1) * br[:, 3].view(n, 1, 1)
    yy = torch.linspace(-.5, .5, 7).view(1, 1, 7) * br[:, 4].view(n, 1, 1)
    sx = (xx * ct + yy * st + br[:, 0].view(n, 1, 1) + 0.0) * 2.5
    sy = (yy * ct - xx * st + br[:, 1].view(n, 1, 1) + 40.0) * 2.5
    grid = torch.stack([sx.reshape(n, 28).t() / 175, sy.reshape(n, 28).t() / 199], -1).view(28, n, 1, 2) * 2 - 1
    ref = F.grid_sample(fr[0].unsqueeze(1), grid, align_corners=True).mean(0).view(-1)
    ref.backward(dl)
    fd, bd = feat.to(dev).requires_grad_(), boxes.to(dev).requires_grad_()
    out = PSWarpFn.apply(fd, bd, (0.0, 40.0), 2.5)
    assert (out.cpu() - ref.detach()).abs().max() < 1e-5
    out.backward(dl.to(dev))
    assert _rel(fd.grad, fr.grad) < 1e-5
    gb, rb_ = bd.grad.cpu(), br.grad
    assert (gb[:, [2, 5]] == 0).all()
    assert (gb - rb_).abs().max() < 1e-4 * max(1.0, rb_.abs().max().item()), (gb - rb_).abs().max()


HALF = dict(voxel_size=synth.KITTI_VOXEL, pc_range=[0, -40., -3., 35.2, 40., 1.], max_points=5, max_voxels=20000,
            sparse_shape=(40, 1600, 704), grid_xyz=(704, 1600, 40))


def _half_anchors():
    an = A.AnchorGeneratorStride(sizes=[1.6, 3.9, 1.56], anchor_strides=[.4, .4, 1.], anchor_offsets=[.2, -39.8, -1.78],
                                 rotations=[0, 1.57])([1, 200, 88]).reshape(-1, 7)
    return an, A.rbbox2d_to_near_bbox(an[:, [0, 1, 3, 4, 6]]).astype(np.float32)


def _gt(seed, n):
    r = np.random.default_rng(seed)
    b = np.zeros((n, 7), np.float32)
    b[:, 0] = r.uniform(4, 32, n); b[:, 1] = r.uniform(-30, 30, n); b[:, 2] = r.uniform(-1.9, -1.5, n)
    b[:, 3] = r.uniform(1.5, 1.8, n); b[:, 4] = r.uniform(3.5, 4.4, n); b[:, 5] = r.uniform(1.4, 1.7, n)
    b[:, 6] = r.uniform(-3.1, 3.1, n)
    return b


def test_training_step_vs_oracle(dev):
    """forward_train on the GPU (HIP kernels under autograd) vs oracle/train_ref.train_step on the CPU: the six loss
    terms and the gradient of their sum with respect to every parameter."""
    from oracle import clib, nets as onets, train_ref
    c = Config.fromfile("configs/car_cfg.py")
    mcfg = dict(c.model)
    mcfg["neck"] = dict(mcfg["neck"], output_shape=list(HALF["sparse_shape"]))
    model = H.randomize_detector(build_detector(mcfg, c.train_cfg, c.test_cfg), seed=7, cls_bias=-3.0)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    an, bv = _half_anchors()
    clouds = [H.frame("small", 31), H.frame("k17", 32)[::3]]
    clouds = [p[p[:, 0] < 35.2] for p in clouds]
    gts = [_gt(1, 5), _gt(2, 7)]
    types = [np.array(["Car"] * 5), np.array(["Car"] * 6 + ["Van"])]
    kw = dict(voxels=[], coordinates=[], num_points=[], anchors=dict(Car=[]), anchors_mask=dict(Car=[]),
              gt_bboxes=[], gt_labels=[], gt_types=types)
    feats, coors, masks = [], [], []
    for b, p in enumerate(clouds):
        v, co, n = clib.points_to_voxel(p, HALF["voxel_size"], HALF["pc_range"], 5, True, 20000)
        m = onets.anchors_mask(co, bv, HALF["voxel_size"], HALF["pc_range"], HALF["grid_xyz"], 1)
        feats.append(clib.voxel_mean(v, n))
        coors.append(np.concatenate([np.full((len(co), 1), b, np.int32), co], 1))
        masks.append(m)
        kw["voxels"].append(torch.from_numpy(v).to(dev)); kw["coordinates"].append(torch.from_numpy(co).to(dev))
        kw["num_points"].append(torch.from_numpy(n).to(dev))
        kw["anchors"]["Car"].append(torch.from_numpy(an).to(dev))
        kw["anchors_mask"]["Car"].append(torch.from_numpy(m).to(dev))
        kw["gt_bboxes"].append(torch.from_numpy(gts[b]).to(dev))
        kw["gt_labels"].append(torch.ones(len(gts[b]), dtype=torch.int64, device=dev))
    losses = model(None, [dict(sample_idx=0), dict(sample_idx=1)], return_loss=True, **kw)
    total = sum(v.sum() for v in losses.values())
    total.backward()
    torch.cuda.synchronize()
    ref_l, ref_g, ex = train_ref.train_step(
        sd, np.concatenate(feats), np.concatenate(coors), 2, HALF["sparse_shape"], gts, types, ["Car"],
        dict(Car=np.stack([an, an])), dict(Car=np.stack(masks)), dict(Car=(0.6, 0.45)))
    assert set(losses) == set(ref_l) == {"aux_loss_cls", "aux_loss_reg", "rpn_loc_loss", "rpn_cls_loss",
                                        "rpn_dir_loss", "loss_cls"}
    assert int((ex["labels"] > 0).sum()) > 10 and int((ex["ext_labels"] > 0).sum()) >= 12
    for k, v in ref_l.items():
        got = float(losses[k].sum())
        assert np.isfinite(got) and abs(got - v) <= 1e-3 * max(1.0, abs(v)), (k, got, v)
    worst = {}
    checked = 0
    for name, p in model.named_parameters():
        rg = ref_g.get(name)
        if rg is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0, name
            continue
        assert p.grad is not None, name
        if float(rg.norm()) < 1e-7:
            continue
        worst[name] = _rel(p.grad, rg)
        checked += 1
    bad = {k: v for k, v in worst.items() if not v < 2e-3}
    assert checked >= 60 and not bad, (checked, bad)
