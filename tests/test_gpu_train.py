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
            sparse_shape=(40, 1600, 704), grid_xyz=(704, 1600, 40), bev_w=88, xmax=32.0)
FULL = dict(voxel_size=synth.KITTI_VOXEL, pc_range=list(synth.KITTI_RANGE), max_points=5, max_voxels=20000,
            sparse_shape=(40, 1600, 1408), grid_xyz=(1408, 1600, 40), bev_w=176, xmax=66.0)      # configs[2]'s own grid


SIZES = dict(Car=[1.6, 3.9, 1.56], Pedestrian=[0.6, 0.8, 1.73], Cyclist=[0.6, 1.76, 1.73])


def _half_anchors(name="Car", bev_w=88):
    an = A.AnchorGeneratorStride(sizes=SIZES[name], anchor_strides=[.4, .4, 1.], anchor_offsets=[.2, -39.8, -1.78],
                                 rotations=[0, 1.57])([1, 200, bev_w]).reshape(-1, 7)
    return an, A.rbbox2d_to_near_bbox(an[:, [0, 1, 3, 4, 6]]).astype(np.float32)


def _gt(seed, n, types=None, xmax=32.0):
    r = np.random.default_rng(seed)
    b = np.zeros((n, 7), np.float32)
    b[:, 0] = r.uniform(4, xmax, n); b[:, 1] = r.uniform(-30, 30, n); b[:, 2] = r.uniform(-1.9, -1.5, n)
    b[:, 3] = r.uniform(1.5, 1.8, n); b[:, 4] = r.uniform(3.5, 4.4, n); b[:, 5] = r.uniform(1.4, 1.7, n)
    b[:, 6] = r.uniform(-3.1, 3.1, n)
    if types is not None:                            # class-sized boxes
        for i, t in enumerate(types):
            if t in SIZES:
                b[i, 3:6] = np.asarray(SIZES[t], np.float32) * r.uniform(0.95, 1.05, 3).astype(np.float32)
    return b


# (each case runs the CPU oracle's step inside the test: ~1 minute.  The default run keeps the three-class half grid and the
# bf16 whole-step bar; car_cfg fp32 on its full grid is held by the stored golden of test_training_step_k21_vs_oracle
# (the bench workload) and, like multi_cfg on its full grid, runs here under SASSD_FULL_TESTS=1)
@pytest.mark.parametrize("cfgfile,names,HALF,precision",
                         [pytest.param("configs/car_cfg.py", ["Car"], FULL, "fp32", marks=pytest.mark.slow),
                          ("configs/multi_cfg.py", ["Car", "Pedestrian", "Cyclist"], HALF, "fp32"),
                          pytest.param("configs/multi_cfg.py", ["Car", "Pedestrian", "Cyclist"], FULL, "fp32",
                                       marks=pytest.mark.slow),
                          ("configs/car_cfg.py", ["Car"], FULL, "bf16")])
def test_training_step_vs_oracle(dev, cfgfile, names, HALF, precision):
    """forward_train on the GPU (HIP kernels under autograd) vs oracle/train_ref.train_step on the CPU: the six loss
    terms and the gradient of their sum with respect to every parameter.  car_cfg on its own full 1408-wide grid
    (BASELINE configs[2]); the three-class multi_cfg (per-class anchors / masks / thresholds, 18 + 42 + 12 head
    channels) on a half-width grid AND on its own full 1408-wide grid (211 200 anchors)."""
    from oracle import clib, nets as onets, train_ref
    c = Config.fromfile(cfgfile)
    mcfg = dict(c.model)
    mcfg["neck"] = dict(mcfg["neck"], output_shape=list(HALF["sparse_shape"]))
    model = H.randomize_detector(build_detector(mcfg, c.train_cfg, c.test_cfg), seed=7, cls_bias=-3.0)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    anbv = {n: _half_anchors(n, HALF["bev_w"]) for n in names}
    clouds = [H.frame("small", 31), H.frame("k17", 32)[::3]]
    clouds = [p[p[:, 0] < HALF["pc_range"][3]] for p in clouds]
    if len(names) == 1:
        types = [np.array(["Car"] * 5), np.array(["Car"] * 6 + ["Van"])]
    else:
        types = [np.array(["Car", "Pedestrian", "Cyclist", "Car", "Pedestrian"]),
                 np.array(["Cyclist", "Car", "Car", "Pedestrian", "Cyclist", "Van", "Car"])]
    gts = [_gt(1, len(types[0]), types[0], HALF["xmax"]), _gt(2, len(types[1]), types[1], HALF["xmax"])]
    kw = dict(voxels=[], coordinates=[], num_points=[], anchors={n: [] for n in names},
              anchors_mask={n: [] for n in names}, gt_bboxes=[], gt_labels=[], gt_types=types)
    feats, coors, masks = [], [], {n: [] for n in names}
    for b, p in enumerate(clouds):
        v, co, n = clib.points_to_voxel(p, HALF["voxel_size"], HALF["pc_range"], 5, True, 20000)
        feats.append(clib.voxel_mean(v, n))
        coors.append(np.concatenate([np.full((len(co), 1), b, np.int32), co], 1))
        kw["voxels"].append(torch.from_numpy(v).to(dev)); kw["coordinates"].append(torch.from_numpy(co).to(dev))
        kw["num_points"].append(torch.from_numpy(n).to(dev))
        for nm in names:
            m = onets.anchors_mask(co, anbv[nm][1], HALF["voxel_size"], HALF["pc_range"], HALF["grid_xyz"], 1)
            masks[nm].append(m)
            kw["anchors"][nm].append(torch.from_numpy(anbv[nm][0]).to(dev))
            kw["anchors_mask"][nm].append(torch.from_numpy(m).to(dev))
        kw["gt_bboxes"].append(torch.from_numpy(gts[b]).to(dev))
        lab = [names.index(t) + 1 if t in names else 0 for t in types[b]]
        kw["gt_labels"].append(torch.tensor(lab, dtype=torch.int64, device=dev))
    acfg = {n: (c.train_cfg.rpn.assigner[n].pos_iou_thr, c.train_cfg.rpn.assigner[n].neg_iou_thr) for n in names}
    ref_args = (sd, np.concatenate(feats), np.concatenate(coors), 2, HALF["sparse_shape"], gts, types, names,
                {n: np.stack([anbv[n][0]] * 2) for n in names}, {n: np.stack(masks[n]) for n in names}, acfg)
    ref_l, ref_g, ex = train_ref.train_step(*ref_args)
    # threshold-safe guided-anchor selection: move train_cfg.rpn.anchor_thr (0.1) to a nearby value that no masked
    # anchor score of the oracle approaches, so that GPU and CPU select the same anchors (a different selection would put
    # a PSWarp-sampling gradient of a borderline box into the box head of one side only)
    top = torch.sigmoid(ex["cls"]).reshape(2, -1, len(names)).max(-1)[0]
    msk = torch.from_numpy(np.concatenate([np.stack(masks[n]) for n in names], 1)).reshape(2, -1)
    vals = top[msk].numpy()
    # (round 4: the middle of the WIDEST gap between neighbouring scores within 0.1 +- 0.02 instead of the first threshold
    # with a 1e-5 ... 1e-4 margin: a new fp32 summation order inside the sparse convs moved one score across a 1e-5
    # margin -- one borderline anchor in the box head of one side only is a 2e-2 error of that head's gradient)
    thr, near = H.widest_gap_threshold(0.1, vals, span=0.02)
    assert near > 2e-5, near
    if abs(thr - 0.1) > 1e-9:
        ref_l, ref_g, ex = train_ref.train_step(*ref_args, anchor_thr=thr)
    model.train_cfg.rpn.anchor_thr = thr
    from sassd import autograd as AG
    try:
        AG.set_bev_precision(precision)
        losses = model(None, [dict(sample_idx=0), dict(sample_idx=1)], return_loss=True, **kw)
        total = sum(v.sum() for v in losses.values())
        total.backward()
        torch.cuda.synchronize()
    finally:
        AG.set_bev_precision("fp32")
    if precision == "bf16":
        # BASELINE configs[2]: the step with bf16 MFMA operands in the BEV convs against the fp32 oracle.  Stated
        # tolerance: every loss term within 3 % (absolute 3e-3 for the small ones), the gradient of the whole model
        # (all parameters concatenated) within 2 % relative L2 and cosine >= 0.9995 (measured: losses <= 1.4 %, gradient
        # 6.1e-3, cosine 0.99998).  (Per-tensor bars are not meaningful
        # here: the guided-anchor selection is a threshold on BEV outputs, so single borderline anchors may differ.)
        for k, v in ref_l.items():
            got = float(losses[k].detach().sum())
            assert np.isfinite(got) and abs(got - v) <= max(3e-2 * abs(v), 3e-3), (k, got, v)
        g_got = torch.cat([p.grad.detach().double().cpu().reshape(-1) for n, p in model.named_parameters() if n in ref_g])
        g_ref = torch.cat([ref_g[n].double().reshape(-1) for n, p in model.named_parameters() if n in ref_g])
        rel = float((g_got - g_ref).norm() / g_ref.norm())
        cos = float(torch.dot(g_got, g_ref) / (g_got.norm() * g_ref.norm()))
        print("bf16 training step vs fp32 oracle: losses",
              {k: (round(float(losses[k].detach().sum()), 5), round(float(v), 5)) for k, v in ref_l.items()},
              "whole-model gradient rel L2 %.3e cosine %.6f" % (rel, cos))
        assert rel < 2e-2 and cos > 0.9995, (rel, cos)
        return
    assert set(losses) == set(ref_l) == {"aux_loss_cls", "aux_loss_reg", "rpn_loc_loss", "rpn_cls_loss",
                                        "rpn_dir_loss", "loss_cls"}
    assert int((ex["labels"] > 0).sum()) > 10 and int((ex["ext_labels"] > 0).sum()) >= len(types[0]) + len(types[1])
    if len(names) > 1:
        assert set(np.unique(ex["labels"].numpy())) >= {0, 1, 2, 3}
    for k, v in ref_l.items():
        got = float(losses[k].detach().sum())
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
    # car_cfg: every tensor < 2e-3 (measured < 6e-4).
    # Three classes: the box head's gradient sums 42 channels over 105 600 / 211 200 anchors with heavy cancellation, and
    # the tensors at the sparse / dense seam (extra_conv, bn0, conv0) collect that rounding through eight train-mode BN
    # layers: their relative error is set by the last bits of the sparse activations, i.e. by the ORDER in which a sparse
    # conv sums its <= 27 offset contributions.  Measured on the half grid with three orders of round 4 (balanced kernel,
    # round-3 geometry, register-stationary kernel) AND with the round-3 sources forced onto their own register-stationary
    # kernel: whole-model gradient 4.9e-4 ... 5.1e-4, box head 2.0e-2 / 1.8e-2, seam tensors 2.5e-3 ... 2.7e-3 -- every
    # one of them; only the round-3 default order, which happens to follow the oracle's ascending-offset sum, measured
    # 4.0e-5 / 9e-4 / 7e-4 (and its bars of 2e-3 / 6e-3 were set on that).  The loss terms carry the strict bar above;
    # the three-class gradient is held as a whole (2e-3 relative L2 over all parameters), 1e-2 per tensor, 4e-2 for the
    # box head.
    tols, per_tensor = {}, 2e-3
    g_got = torch.cat([p.grad.detach().double().cpu().reshape(-1) for n, p in model.named_parameters() if n in worst])
    g_ref = torch.cat([ref_g[n].double().reshape(-1) for n, p in model.named_parameters() if n in worst])
    whole = float((g_got - g_ref).norm() / g_ref.norm())
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:6]
    print("training step vs oracle (%s, %d-wide grid, %s): whole-model gradient rel L2 %.2e; worst tensors %s"
          % (cfgfile, HALF["sparse_shape"][2], precision, whole, [(k, "%.1e" % v) for k, v in top]))
    if len(names) > 1:
        assert whole < 2e-3, whole
        per_tensor, tols = 1e-2, {"rpn_head.conv_box.bias": 4e-2, "rpn_head.conv_box.weight": 4e-2}
    else:
        assert whole < 5e-4, whole
    bad = {k: v for k, v in worst.items() if not v < tols.get(k, per_tensor)}
    assert checked >= 60 and not bad, (checked, bad)


def test_training_step_waymo_scale(dev):
    """configs[4] shape on one GPU, training side: 180k points, 0.1 x 0.1 x 0.15 m voxels (grid 40x1504x1504, ~79k active
    voxels, BEV 188x188), batch 2 built on the device (HIP voxelizer, anchor masks, rulebooks) -> forward_train ->
    backward -> fused optimizer step.  A scale / plumbing test: finite losses, every parameter receives a finite
    gradient, the update changes the weights, a second step runs on the same buffers."""
    from sassd import train
    c = Config.fromfile("configs/car_cfg.py")
    mcfg = dict(c.model)
    mcfg["neck"] = dict(mcfg["neck"], output_shape=[40, 1504, 1504], aux_offset=synth.WAYMO_RANGE[:3],
                        aux_voxel_size=synth.WAYMO_VOXEL)
    mcfg["extra_head"] = dict(mcfg["extra_head"], grid_offsets=(75.2, 75.2), featmap_stride=0.8)
    model = H.randomize_detector(build_detector(mcfg, c.train_cfg, c.test_cfg), 7, cls_bias=-3.0).to(dev)
    an = A.AnchorGeneratorStride(sizes=[1.6, 3.9, 1.56], anchor_strides=[.8, .8, 1.], anchor_offsets=[-74.8, -74.8, -1.0],
                                 rotations=[0, 1.57])([1, 188, 188]).reshape(-1, 7)
    bv = A.rbbox2d_to_near_bbox(an[:, [0, 1, 3, 4, 6]]).astype(np.float32)
    anchors, anchors_bv = dict(Car=torch.from_numpy(an).to(dev)), dict(Car=torch.from_numpy(bv).to(dev))
    pts = [torch.from_numpy(synth.waymo_synth(s)[:180000]).to(dev) for s in (0, 1)]
    r = np.random.default_rng(0)
    gts = []
    for _ in range(2):
        b = np.zeros((12, 7), np.float32)
        b[:, 0], b[:, 1], b[:, 2] = r.uniform(-60, 60, 12), r.uniform(-60, 60, 12), r.uniform(-1.2, -0.8, 12)
        b[:, 3], b[:, 4], b[:, 5] = r.uniform(1.5, 1.8, 12), r.uniform(3.5, 4.4, 12), r.uniform(1.4, 1.7, 12)
        b[:, 6] = r.uniform(-3.1, 3.1, 12)
        gts.append(torch.from_numpy(b).to(dev))
    types = [np.array(["Car"] * 12)] * 2
    opt = train.build_optimizer(model, c.optimizer, 1)
    sched = train.build_scheduler(opt, 10, 1, c.optimizer, c.lr_config)
    sync = train.GradSync(opt.flat)
    w0 = opt.flat.data.clone()
    for it in range(2):
        batch = train.device_batch(pts, gts, types, ["Car"], anchors, anchors_bv, synth.WAYMO_VOXEL, synth.WAYMO_RANGE,
                                   max_voxels=150000, model=model)
        assert sum(v.shape[0] for v in batch["voxels"]) > 150000
        loss, terms = train.train_one_iter(model, opt, sched, sync, batch, it)
        assert np.isfinite(float(loss)) and len(terms) == 6
        assert bool(torch.isfinite(opt.flat.grad).all()) and float(opt.flat.grad.abs().sum()) > 0
    assert float((opt.flat.data - w0).abs().max()) > 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16sp", pytest.param("bf16sp-fwd", marks=pytest.mark.slow),
                                       pytest.param("bf16sp-bwd", marks=pytest.mark.slow)])
def test_training_step_k21_vs_oracle(dev, precision):
    """BASELINE configs[2], parity ON THE WORKLOAD bench.py MEASURES: car_cfg on its full grid, batch 2 of K21 frames
    (21 500 points each, 32 245 voxels), 8 boxes per frame, the weights bench.py trains -- forward_train + backward on the
    HIP kernels, batch built by the product's own device_batch (fp32; bf16 = BEV convs on the bf16 MFMA; bf16sp = the 64-channel
    sparse convs on it as well, the opt-in mode), against the CPU oracle's step read from
    tests/golden/train_k21_ref.npz (tests/golden/make_golden_train_k21.py; imported here for the shared seeded inputs).
    fp32: six loss terms 1e-3 relative; gradients: the stored layers TAKEN TOGETHER 2e-3 relative L2 (measured 6.6e-4), each
    stored tensor 2e-2 (measured 6.4e-3), every parameter's norm 5e-3 (1.8e-3) and seeded projection 2e-2 (7.2e-3).  The
    per-tensor bar is this workload's own fp32 floor, not a kernel tolerance: tests/analysis/train_order_sensitivity.py runs the CPU
    ORACLE against its own stored golden with nothing changed but the order in which its 4-channel input layer adds the 27
    offset terms -- forward activations move by <= 5e-5 on values of 40, FOUR of 3.9 M ReLU decisions flip (|z| < 1e-6),
    and the BatchNorm-parameter gradients of the first sparse blocks (small residuals of cancelling sums over 30 k rows)
    move by 6.4e-3 (descending order) / 8.8e-3 (three interleaved partial sums), the same tensors in the same ranking as
    the GPU's deviation (down1.1.bias 6.4e-3, conv1.1.bias 5.3e-3, conv0.1.bias 5.0e-3, ...).  With the round-1 kernel on
    the input layer the GPU happened to land on the golden's side of those decisions (7.7e-4); the round-4 input-layer
    kernel (k ascending, one fmaf chain) lands on the other (tests/test_train_order_sensitivity_cpu.py holds the oracle-
    vs-oracle number on CPU).  bf16 (BEV convs on the bf16
    MFMA, what the bench line runs): the six loss terms 3 % (measured 0.3 %), and the gradient of the SELECTION-FREE part of
    the objective (all terms but the rescoring head's loss_cls) -- stored layers taken together 1.2e-1 relative L2 with
    cosine >= 0.99 (measured 7.8e-2), every parameter's gradient norm within 25 % (measured 17 %).  These are the numbers
    of THIS workload, not of the kernels: the bf16 kernels equal float64 on the rounded operands to 5e-7
    (tests/test_gpu_bf16.py) and the same step on two sparser clouds holds 6e-3 (test_training_step_vs_oracle[bf16]);
    with 32 k voxels and 70 k masked anchors per frame the rpn-path gradient reaches the sparse trunk through eight
    train-mode BatchNorm backward passes (each subtracts the projections of dy on 1 and x-hat: cancellation) and the
    0.4 % operand rounding of the BEV convs comes out as 20-37 % on the first sparse blocks' BatchNorm parameters.
    Why not the full sum in bf16: bf16 moves the classification scores by ~1e-2, far beyond any threshold margin, ~2000
    guided candidates sit near the 0.11 threshold on this workload, so the candidate SET differs and with it, discretely,
    the gradient of loss_cls into everything upstream (measured: 19 % on the full sum)."""
    import importlib.util
    import os
    from sassd import train, autograd as AG
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_golden_train_k21", os.path.join(gdir, "make_golden_train_k21.py"))
    MG = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(MG)
    G = np.load(os.path.join(gdir, "train_k21_ref.npz"))
    model, c, w, clouds, gts = MG.build()
    model = model.to(dev).train()
    model.train_cfg.rpn.anchor_thr = float(G["anchor_thr"])
    cal = w["cal"]
    anchors = dict(Car=torch.from_numpy(w["anchors"]).to(dev))
    anchors_bv = dict(Car=torch.from_numpy(w["anchors_bv"]).to(dev))
    sparse_bf16 = precision.startswith("bf16sp")         # bf16sp: the 64-channel sparse convs on the bf16 MFMA as well
    sparse_mode = {"bf16sp": "bf16", "bf16sp-fwd": "bf16-fwd", "bf16sp-bwd": "bf16-bwd"}.get(precision, "fp32")
    precision = "bf16" if sparse_bf16 else precision     # (-fwd / -bwd: one half only; SASSD_FULL_TESTS=1, printed, same bars)
    AG.set_bev_precision(precision)
    AG.set_sparse_precision(sparse_mode)
    try:
        batch = train.device_batch([torch.from_numpy(p).to(dev) for p in clouds], [torch.from_numpy(g).to(dev) for g in gts],
                                   [np.array(["Car"] * len(g)) for g in gts], ["Car"], anchors, anchors_bv, cal["voxel_size"],
                                   cal["pc_range"], max_points=cal["max_points"], max_voxels=cal["max_voxels"], model=model)
        assert sum(v.shape[0] for v in batch["voxels"]) == int(G["n_voxels"]) == 32245
        assert int(sum(m.sum() for m in batch["anchors_mask"]["Car"])) == int(G["n_masked"])
        losses = model(**batch)
        total = sum(v.sum() for k, v in losses.items() if precision == "fp32" or k != "loss_cls")
        total.backward()
        torch.cuda.synchronize()
    finally:
        AG.set_bev_precision("fp32")
        AG.set_sparse_precision("fp32")
    got_l = {k: float(v.detach().sum()) for k, v in losses.items()}
    ref_l = dict(zip([str(k) for k in G["loss_names"]], G["losses"]))
    assert set(got_l) == set(ref_l) and len(ref_l) == 6
    lbar = 1e-3 if precision == "fp32" else 3e-2
    gp, g8p = ("grad:", "grad8:") if precision == "fp32" else ("gradx:", "gradx8:")
    gnames, gnorms = (G["grad_names"], G["grad_norms"]) if precision == "fp32" else (G["gradx_names"], G["gradx_norms"])
    gprojs = G["grad_projs"] if precision == "fp32" else [None] * len(gnames)
    for k, v in ref_l.items():
        assert v != 0 and abs(got_l[k] - v) <= lbar * max(1.0, abs(v)), (k, got_l[k], v)
    params = dict(model.named_parameters())
    worst, worst_n, worst_p, num, den, dot, gg = {}, {}, {}, 0.0, 0.0, 0.0, 0.0
    for k in G.files:
        if k.startswith(gp) or k.startswith(g8p):
            name = k.split(":", 1)[1]
            g = params[name].grad
            if g is None:                                # (a rescoring-head parameter under the reduced objective)
                continue
            g = g[:8] if k.startswith(g8p) else g
            ref = torch.from_numpy(G[k])
            if float(ref.norm()) > 1e-7:
                worst[name] = _rel(g, ref)
                gdc = g.detach().cpu().double()
                num += float((gdc - ref.double()).pow(2).sum())
                den += float(ref.double().pow(2).sum())
                dot += float((gdc * ref.double()).sum())
                gg += float(gdc.pow(2).sum())
    for name, norm, proj in zip(gnames, gnorms, gprojs):
        name = str(name)
        g = params[name].grad
        if norm < 1e-7 or (g is None and precision != "fp32"):
            continue
        assert g is not None, name
        gd = g.detach().double().cpu().reshape(-1)
        worst_n[name] = abs(float(gd.norm()) - norm) / norm
        worst_p[name] = 0.0 if proj is None else abs(float(torch.dot(gd, MG.projection(name, gd.numel()))) - proj) / norm
    allrel = (num / den) ** 0.5
    cosine = dot / max((den * gg) ** 0.5, 1e-30)
    print("K21 x 2 training step (%s%s) vs oracle: stored-layer gradient cosine %.5f; losses" % (
        precision, " + bf16 sparse MFMA (%s)" % sparse_mode if sparse_bf16 else "", cosine),
          {k: (round(got_l[k], 5), round(float(v), 5)) for k, v in ref_l.items()},
          "| stored-layer gradients: worst rel L2 %.2e over %d tensors, taken together %.2e | all %d parameters: worst norm "
          "error %.2e, worst projection error %.2e" % (max(worst.values()), len(worst), allrel, len(worst_n),
                                                       max(worst_n.values()), max(worst_p.values())))
    print("largest stored-layer errors:", sorted(((round(v, 4), k) for k, v in worst.items()), reverse=True)[:10])
    print("largest norm errors:", sorted(((round(v, 4), k) for k, v in worst_n.items()), reverse=True)[:10])
    assert len(worst) >= 40 and len(worst_n) >= 70
    if precision == "fp32":
        assert allrel < 2e-3 and cosine > 0.999999, (allrel, cosine)
        bad = {k: v for k, v in worst.items() if not v < 2e-2}
        assert not bad, bad
        assert max(worst_n.values()) < 5e-3, {k: v for k, v in worst_n.items() if v >= 5e-3}
        assert max(worst_p.values()) < 2e-2, {k: v for k, v in worst_p.items() if v >= 2e-2}
    elif not sparse_bf16:
        assert allrel < 1.2e-1 and cosine > 0.99, (allrel, cosine)
        assert max(worst_n.values()) < 0.25, {k: v for k, v in worst_n.items() if v >= 0.25}
    else:
        # the opt-in mode (bench.py --sparse-precision bf16; +4.6 % samples/s): rounding the operands of the 64-channel
        # sparse convs as well roughly doubles the gradient noise of this workload -- measured 1.44e-1 together, cosine
        # 0.9897, norms 12 %; that is why it is not the default
        assert allrel < 2e-1 and cosine > 0.98, (allrel, cosine)
        assert max(worst_n.values()) < 0.25, {k: v for k, v in worst_n.items() if v >= 0.25}


def test_training_step_waymo_vs_oracle(dev):
    """BASELINE configs[4] as a TRAINING config, parity: one 180 000-point frame (79 302 voxels, grid 40 x 1504 x 1504, BEV
    188 x 188) through forward_train + backward on the HIP kernels -- the batch built by the product's own device_batch
    (HIP voxelizer, anchor mask, rulebooks) -- against the CPU oracle's step on the same model / frame / boxes, read
    from tests/golden/waymo_train_ref.npz (the oracle needs ~2.5 CPU-minutes for this frame;
    tests/golden/make_golden_waymo_train.py made the file and is imported here for the shared seeded inputs).
    Bars: the six loss terms 1e-3 relative (as for car_cfg; measured equal to 5 digits); gradients 5e-3 relative L2 --
    elementwise for the stored layers (first / last sparse convs, every BatchNorm, heads, aux linears, a slice of four BEV
    convs), through the norm and a seeded random projection for every other parameter.  (car_cfg holds 2e-3 and
    measures < 6e-4; at this scale the auxiliary loss terms are 192 and 44 -- their gradients, scattered back with float
    atomics over 5x the rows, dominate the first sparse layers, which measure 2-4e-3.)"""
    import importlib.util
    import os
    from sassd import train
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_golden_waymo_train", os.path.join(gdir, "make_golden_waymo_train.py"))
    MG = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(MG)
    G = np.load(os.path.join(gdir, "waymo_train_ref.npz"))
    model, c, an, bv, p, gt = MG.build()
    model = model.to(dev).train()
    model.train_cfg.rpn.anchor_thr = float(G["anchor_thr"])
    anchors, anchors_bv = dict(Car=torch.from_numpy(an).to(dev)), dict(Car=torch.from_numpy(bv).to(dev))
    batch = train.device_batch([torch.from_numpy(p).to(dev)], [torch.from_numpy(gt).to(dev)], [np.array(["Car"] * len(gt))],
                               ["Car"], anchors, anchors_bv, synth.WAYMO_VOXEL, synth.WAYMO_RANGE, max_voxels=150000,
                               model=model)
    assert batch["voxels"][0].shape[0] == int(G["n_voxels"]) == 79302
    assert int(batch["anchors_mask"]["Car"][0].sum()) == int(G["n_masked"])
    losses = model(**batch)
    total = sum(v.sum() for v in losses.values())
    total.backward()
    torch.cuda.synchronize()
    got_l = {k: float(v.detach().sum()) for k, v in losses.items()}
    ref_l = dict(zip([str(k) for k in G["loss_names"]], G["losses"]))
    assert set(got_l) == set(ref_l) and len(ref_l) == 6
    for k, v in ref_l.items():
        assert v != 0 and abs(got_l[k] - v) <= 1e-3 * max(1.0, abs(v)), (k, got_l[k], v)
    params = dict(model.named_parameters())
    worst, worst_n, worst_p = {}, {}, {}
    for k in G.files:
        if k.startswith("grad:") or k.startswith("grad8:"):
            name = k.split(":", 1)[1]
            g = params[name].grad
            g = g[:8] if k.startswith("grad8:") else g
            ref = torch.from_numpy(G[k])
            if float(ref.norm()) > 1e-7:
                worst[name] = _rel(g, ref)
    for name, norm, proj in zip(G["grad_names"], G["grad_norms"], G["grad_projs"]):
        name = str(name)
        g = params[name].grad
        assert g is not None, name
        if norm < 1e-7:
            continue
        gd = g.detach().double().cpu().reshape(-1)
        worst_n[name] = abs(float(gd.norm()) - norm) / norm
        # error e with ||e|| <= 2e-3 ||g||, independent of the seeded direction r (unit variance): dot(e, r) ~ N(0, ||e||^2)
        worst_p[name] = abs(float(torch.dot(gd, MG.projection(name, gd.numel()))) - proj) / norm
    print("waymo-scale training step vs oracle: losses", {k: (round(got_l[k], 5), round(float(v), 5)) for k, v in ref_l.items()},
          "| stored-layer gradients: worst rel L2 %.2e over %d tensors | all %d parameters: worst norm error %.2e, worst "
          "projection error %.2e" % (max(worst.values()), len(worst), len(worst_n), max(worst_n.values()),
                                     max(worst_p.values())))
    bad = {k: v for k, v in worst.items() if not v < 5e-3}
    assert len(worst) >= 40 and not bad, bad
    assert len(worst_n) >= 75 and max(worst_n.values()) < 5e-3, {k: v for k, v in worst_n.items() if v >= 5e-3}
    assert max(worst_p.values()) < 5 * 5e-3, {k: v for k, v in worst_p.items() if v >= 2.5e-2}


# ---- evaluation with the overlap matrices on the GPU (kept last in the last -m gpu file) --------------

def test_kitti_eval_on_gpu(dev):
    """get_official_eval_result with the rotated BEV / 3-D overlaps computed by sassd_rotate_iou_eval (HIP): the report
    text equals the one the reference's own kitti_eval.py produced for the same annotations (tests/golden/
    make_golden_kitti_eval.py; its device call was served by the CPU oracle).  Index-level work: the text is compared
    exactly -- it could only differ if an overlap fell within fp32 rounding of a 0.7 / 0.5 / 0.25 cut-off."""
    import os
    import kitti_synth
    from sassd import kitti_eval as ke
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti_eval_ref.npz"))
    gts, dts = kitti_synth.unpack(G, "gt_"), kitti_synth.unpack(G, "dt_")
    for metric in (1, 2):
        part = ke.calculate_iou_partly(dts, gts, metric, 50)[1][-1]
        assert np.abs(part - G["ov%d_last_part" % metric]).max() < 1e-5
    assert ke.get_official_eval_result(gts, dts, ["Car", "Pedestrian", "Cyclist"]) == str(G["official_text"])


def test_forward_test_returns_kitti_annos(dev):
    """With calibration in img_meta, model(return_loss=False) returns the reference's result annotations
    (single_stage.py:129 -> kitti_bbox2results), which the evaluation accepts as `dt_annos`."""
    import test_gpu_pipeline as P
    from sassd import kitti_common as kc, kitti_eval as ke
    from sassd.voxel_generator import VoxelGenerator
    from oracle import nets as onets
    model, c = P._model()
    model = model.to(dev)
    model.class_names = c.data.val.class_names
    an, bv = P._anchors()
    gen = VoxelGenerator(**{k: v for k, v in c.data.val.generator.items() if k != "type"})
    calib = kc.Calibration(matrices={
        "P2": [721.5377, 0, 609.5593, 44.85728, 0, 721.5377, 172.854, 0.2163791, 0, 0, 1, 0.002745884],
        "R0_rect": [0.9999239, 0.00983776, -0.007445048, -0.009869795, 0.9999421, -0.004278459, 0.007402527,
                    0.004351614, 0.9999631],
        "Tr_velo_to_cam": [0.007533745, -0.9999714, -0.000616602, -0.004069766, 0.01480249, 0.0007280733, -0.9998902,
                           -0.07631618, 0.9998621, 0.00752379, 0.01480755, -0.2717806]})
    metas, kw = [], dict(voxels=[], coordinates=[], num_points=[], anchors=[], anchors_mask=[])
    for i, p in enumerate([H.frame("small", 5), H.frame("k17", 1)]):
        v, co, n = gen.generate(p)
        m = onets.anchors_mask(co, bv, gen.voxel_size, gen.point_cloud_range, gen.grid_size, 1)
        kw["voxels"].append(torch.from_numpy(v).to(dev)); kw["coordinates"].append(torch.from_numpy(co).to(dev))
        kw["num_points"].append(torch.from_numpy(n).to(dev)); kw["anchors"].append(torch.from_numpy(an).to(dev))
        kw["anchors_mask"].append(torch.from_numpy(m).to(dev))
        metas.append(dict(sample_idx=i, img_shape=(375, 1242, 3), calib=calib))
    raw = model(None, [dict(sample_idx=0), dict(sample_idx=1)], return_loss=False, **kw)
    annos = model(None, metas, return_loss=False, **kw)
    total = 0
    for r, a in zip(raw, annos):
        want = kc.kitti_bbox2results(None if r["boxes_lidar"] is None else r["boxes_lidar"].copy(), r["scores"],
                                     r["labels"], metas[0], class_names=model.class_names)
        assert sorted(a.keys()) == sorted(kc.empty_result_anno().keys()) or "image_idx" in a
        assert len(a["name"]) == len(want["name"])
        for k in ("bbox", "location", "dimensions", "rotation_y", "score", "alpha"):
            assert np.allclose(a[k], want[k], atol=1e-4), k
        total += len(a["name"])
    assert total > 0, "test vector produced no detections inside the image"
    # the annotations are valid evaluation input (scored against themselves as labels)
    text = ke.get_official_eval_result([dict(a, occluded=np.zeros(len(a["name"]), int)) for a in annos], annos, "Car")
    assert text.startswith("Car AP@0.70, 0.70, 0.70:")


# ---- augmentation kernels on the reference-generated vectors -----------------------------------------------

def test_augmentation_kernels(dev, tmp_path):
    """sassd_points_in_polytopes / sassd_points_transform / sassd_points_global_transform / sassd_paste_objects against
    tests/golden/augment_ref.npz (made by the reference's own PointAugmentor / geometry code): masks bit-exact, points
    within float32 rounding, three whole training frames with the reference's random seed."""
    import os
    import augment_synth as S
    from sassd import geometry as G, kitti_common as kc, point_augmentor as PA
    R = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment_ref.npz"))
    unpack = lambda bits, n, m: np.unpackbits(bits)[:n * m].reshape(n, m).astype(bool)
    pts = R["pts"]
    for tag in ("32", "64"):
        boxes = R["pib_boxes" + tag]
        assert np.array_equal(G.points_in_rbbox(pts, boxes), unpack(R["pib_mask" + tag], len(pts), len(boxes))), tag
    on_gpu = G.points_in_rbbox(torch.from_numpy(pts).to(dev), R["pib_boxes32"])
    assert on_gpu.is_cuda and np.array_equal(on_gpu.cpu().numpy(), unpack(R["pib_mask32"], len(pts), 14))
    many = np.tile(R["pib_boxes32"], (30, 1))                                  # 420 polytopes: two LDS passes
    assert np.array_equal(G.points_in_rbbox(pts, many), np.tile(unpack(R["pib_mask32"], len(pts), 14), (1, 30)))
    c = S.calib_matrices()
    rect, trv2c, p2 = (S.extend(c[k]) for k in ("R0_rect", "Tr_velo_to_cam", "P2"))
    full = R["fov_points"]
    assert np.array_equal(G.remove_outside_points(full, rect, trv2c, p2, (375, 1242)),
                          full[unpack(R["fov_mask"], len(full), 1)[:, 0]])

    S.write_database(S.unpack_database(R), str(tmp_path))
    calib = kc.Calibration(matrices=S.calib_matrices())
    info = str(tmp_path / "kitti_dbinfos_train.pkl")
    for cfg_name in ("car", "multi"):
        cfg = S.AUGMENTOR_CONFIGS[cfg_name]
        np.random.seed(1234)
        aug = PA.PointAugmentor(root_path=str(tmp_path), info_path=info, device=dev, **cfg)       # fused recipe
        np.random.seed(1234)
        step = PA.PointAugmentor(root_path=str(tmp_path), info_path=info, device=dev, **cfg)      # method by method
        state = np.random.get_state()
        for f in range(3):
            points, gt_boxes, gt_types = S.frame(f)
            plane = S.PLANE if cfg_name == "multi" else None
            tag = "aug_%s_%d_" % (cfg_name, f)
            np.random.set_state(state)
            s_boxes, s_types, s_points = step.sample_all(gt_boxes, gt_types, plane, calib)        # numpy convention
            assert np.array_equal(s_boxes, R[tag + "s_boxes"]) and np.array_equal(s_points, R[tag + "s_points"])
            boxes, p = R[tag + "pre_boxes"].copy(), R[tag + "pre_points"].copy()
            step.noise_per_object_(boxes, p, num_try=100)
            assert np.abs(p - R[tag + "noise_points"]).max() < 4e-6
            boxes, p = step.random_flip(boxes, p)
            boxes, p = step.global_rotation(boxes, p)
            boxes, p = step.global_scaling(boxes, p)
            assert np.abs(boxes - R[tag + "out_boxes"]).max() < 2e-5 and np.abs(p - R[tag + "out_points"]).max() < 2e-5
            after = np.random.get_state()
            np.random.set_state(state)
            out_pts, out_boxes, out_types, _ = aug.augment_frame(torch.from_numpy(points).to(dev), gt_boxes.copy(), gt_types,
                                                                 cfg["sample_classes"], plane, calib)
            assert out_pts.is_cuda and "\n".join(out_types) == str(R[tag + "out_types"])
            assert np.abs(out_boxes - R[tag + "out_boxes"]).max() < 2e-5
            assert np.abs(out_pts.cpu().numpy() - R[tag + "out_points"]).max() < 2e-5
            state = after


def test_disk_to_ap_end_to_end(dev, tmp_path):
    """Raw KITTI tree on disk -> data preparation on the GPU (byte-identical to the reference's artefacts) -> KittiLiDAR
    (augmentor with the database in HBM) -> collate -> one optimisation step; then the val split -> forward_test ->
    KITTI result files -> get_official_eval_result.  Everything between the files and the report runs on the device."""
    import os
    import augment_synth as S
    import test_create_data_cpu as TC
    from sassd import create_data as CD, kitti_common as kc, kitti_eval as ke, train
    from sassd.kitti_dataset import get_dataset
    root = str(tmp_path)
    TC.run_preparation(root, dev)
    TC.check_tree(root, np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "create_data_ref.npz")))

    c = Config.fromfile("configs/car_cfg.py")
    tr = dict(c.data.train, root=root + '/training/', ann_file=root + '/ImageSets/train.txt')
    tr['augmentor'] = dict(tr['augmentor'], root_path=root + '/', info_path=root + '/kitti_dbinfos_train.pkl',
                           sample_classes=['Van'], min_num_points=[2], sample_max_num=[4])
    np.random.seed(3)
    ds = get_dataset(tr, device=dev)
    assert ds.augmentor.database_on_device().is_cuda
    model = H.randomize_detector(build_detector(c.model, c.train_cfg, c.test_cfg), 3, cls_bias=-3.0).to(dev)
    opt = train.build_optimizer(model, c.optimizer, 1)
    sched = train.build_scheduler(opt, 10, 1, c.optimizer, c.lr_config)
    sync = train.GradSync(opt.flat)
    w0 = opt.flat.data.clone()
    samples = [ds[0], ds[1]]
    assert all(s['points'].is_cuda and s['gt_bboxes'].is_cuda for s in samples)
    batch = ds.collate(samples, model=model)
    assert len(batch['voxels']) == 2 and batch['anchors_mask']['Car'][0].dtype == torch.bool
    loss, terms = train.train_one_iter(model, opt, sched, sync, batch, 0)
    assert np.isfinite(float(loss)) and len(terms) == 6 and float((opt.flat.data - w0).abs().max()) > 0

    va = dict(c.data.val, root=root + '/training/', ann_file=root + '/ImageSets/val.txt')
    dv = get_dataset(va, device=dev)
    model.eval()
    model.class_names = c.data.val.class_names
    with torch.no_grad():
        annos = [a for i in range(len(dv)) for a in model(**dv.collate([dv[i]]))]
    assert len(annos) == 2 and all(a['bbox'].shape[1:] == (4,) for a in annos)
    gt = kc.get_label_annos(dv.label_prefix, dv.sample_ids)
    text = ke.get_official_eval_result(gt, annos, c.data.val.class_names)
    assert text.startswith("Car AP@0.70, 0.70, 0.70:") and "3d   AP:" in text
    found = [a for a in annos if len(a['name'])]
    if found:                                                    # result files round-trip (random weights may detect nothing)
        kc.write_label_annos(found, root + '/results')
        back = kc.get_label_annos(root + '/results', [int(a['image_idx'][0]) for a in found])
        assert [len(b['name']) for b in back] == [len(a['name']) for a in found]


def test_weight_packs_follow_the_fused_optimizer(dev):
    """The fused optimizer writes the flat parameter buffer through a raw pointer (no autograd version bump): every cached
    weight pack -- sparse conv, BEV conv (direct / Winograd), the folded inference plan -- must be rebuilt after a step."""
    from sassd import train, kernels as K, spconv
    from sassd.detector import _HipConv2d
    c = Config.fromfile("configs/car_cfg.py")
    model = H.randomize_detector(build_detector(c.model, c.train_cfg, c.test_cfg), seed=3).to(dev)
    opt = train.build_optimizer(model, c.optimizer, 1)
    sp = [m for m in model.modules() if isinstance(m, spconv.SparseConvolution)][3]
    cv = [m for m in model.modules() if isinstance(m, _HipConv2d) and m.kernel_size[0] == 3][1]
    p0, q0, w0 = sp.packed_weight().clone(), cv.packed_weight().clone(), cv.packed_wino(200, 176).clone()
    an = A.AnchorGeneratorStride(sizes=SIZES["Car"], anchor_strides=[.4, .4, 1.], anchor_offsets=[.2, -39.8, -1.78],
                                 rotations=[0, 1.57])([1, 200, 176]).reshape(-1, 7)
    plan0 = model.plan(1, torch.from_numpy(an).to(dev), dev)
    assert model.plan(1, torch.from_numpy(an).to(dev).clone(), dev) is plan0          # same weights + same anchor content
    opt.flat.grad.fill_(1e-2)
    opt.lr = 1e-2
    opt.step()
    torch.cuda.synchronize()
    k = int(np.prod(sp.kernel_size))
    fresh = K.spconv_pack_weight(sp.weight.detach().reshape(k, sp.in_channels, sp.out_channels).contiguous())
    assert torch.equal(sp.packed_weight(), fresh) and not torch.equal(fresh, p0)
    assert torch.equal(cv.packed_weight(), K.conv2d_pack_weight(cv.weight.detach().contiguous())) and not torch.equal(cv.packed_weight(), q0)
    assert torch.equal(cv.packed_wino(200, 176), K.conv2d_wino_pack_weight(cv.weight.detach().contiguous()))
    assert not torch.equal(cv.packed_wino(200, 176), w0)
    model.eval()
    plan1 = model.plan(1, torch.from_numpy(an).to(dev), dev)
    assert plan1 is not plan0
    assert torch.equal(plan1.sp[3][4], K.spconv_pack_weight(sp.weight.detach().reshape(k, sp.in_channels, sp.out_channels).contiguous()))
