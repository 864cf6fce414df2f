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


def oracle_case(cfgfile, names, HALF):
    """The seeded two-cloud workload of test_training_step_vs_oracle, built on the CPU (shared with the CPU-only arbiter tests,
    tests/test_train_arbiter_cpu.py): -> dict(c, model (CPU), sd, types, gts, masks, np_inputs (per-sample voxel arrays / anchors
    / masks as numpy), ref_args (positional arguments of oracle.train_ref.train_step))."""
    from oracle import clib, nets as onets
    c = Config.fromfile(cfgfile)
    mcfg = dict(c.model)
    mcfg["neck"] = dict(mcfg["neck"], output_shape=list(HALF["sparse_shape"]))
    model = H.randomize_detector(build_detector(mcfg, c.train_cfg, c.test_cfg), seed=7, cls_bias=-3.0)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    anbv = {n: _half_anchors(n, HALF["bev_w"]) for n in names}
    clouds = [H.frame("small", 31), H.frame("k17", 32)[::3]]
    clouds = [p[p[:, 0] < HALF["pc_range"][3]] for p in clouds]
    if len(names) == 1:
        types = [np.array(["Car"] * 5), np.array(["Car"] * 6 + ["Van"])]
    else:
        types = [np.array(["Car", "Pedestrian", "Cyclist", "Car", "Pedestrian"]),
                 np.array(["Cyclist", "Car", "Car", "Pedestrian", "Cyclist", "Van", "Car"])]
    gts = [_gt(1, len(types[0]), types[0], HALF["xmax"]), _gt(2, len(types[1]), types[1], HALF["xmax"])]
    npi = dict(voxels=[], coordinates=[], num_points=[], anchors={n: [] for n in names}, anchors_mask={n: [] for n in names},
               gt_bboxes=gts, gt_labels=[[names.index(t) + 1 if t in names else 0 for t in ty] for ty in types])
    feats, coors, masks = [], [], {n: [] for n in names}
    for b, p in enumerate(clouds):
        v, co, n = clib.points_to_voxel(p, HALF["voxel_size"], HALF["pc_range"], 5, True, 20000)
        feats.append(clib.voxel_mean(v, n))
        coors.append(np.concatenate([np.full((len(co), 1), b, np.int32), co], 1))
        npi["voxels"].append(v); npi["coordinates"].append(co); npi["num_points"].append(n)
        for nm in names:
            m = onets.anchors_mask(co, anbv[nm][1], HALF["voxel_size"], HALF["pc_range"], HALF["grid_xyz"], 1)
            masks[nm].append(m)
            npi["anchors"][nm].append(anbv[nm][0]); npi["anchors_mask"][nm].append(m)
    acfg = {n: (c.train_cfg.rpn.assigner[n].pos_iou_thr, c.train_cfg.rpn.assigner[n].neg_iou_thr) for n in names}
    ref_args = (sd, np.concatenate(feats), np.concatenate(coors), 2, HALF["sparse_shape"], gts, types, names,
                {n: np.stack([anbv[n][0]] * 2) for n in names}, {n: np.stack(masks[n]) for n in names}, acfg)
    return dict(c=c, model=model, sd=sd, types=types, gts=gts, masks=masks, np_inputs=npi, ref_args=ref_args)


# (each case runs the CPU oracle's step inside the test, twice -- float64 arbiter + fp32 floor: ~1-2 minutes.  The default run
# holds car_cfg on its own full grid in fp32 and bf16 and the three-class half grid; multi_cfg on its full grid runs under
# SASSD_FULL_TESTS=1)
@pytest.mark.parametrize("cfgfile,names,HALF,precision",
                         [("configs/car_cfg.py", ["Car"], FULL, "fp32"),
                          ("configs/multi_cfg.py", ["Car", "Pedestrian", "Cyclist"], HALF, "fp32"),
                          pytest.param("configs/multi_cfg.py", ["Car", "Pedestrian", "Cyclist"], FULL, "fp32",
                                       marks=pytest.mark.slow),
                          ("configs/car_cfg.py", ["Car"], FULL, "bf16")])
def test_training_step_vs_oracle(dev, cfgfile, names, HALF, precision):
    """forward_train on the GPU (HIP kernels under autograd) vs oracle/train_ref.train_step on the CPU: the six loss
    terms and the gradient of their sum with respect to every parameter.  car_cfg on its own full 1408-wide grid
    (BASELINE configs[2]) in fp32 and with the bf16 dense convs; the three-class multi_cfg (per-class anchors / masks /
    thresholds, 18 + 42 + 12 head channels) on a half-width grid AND on its own full 1408-wide grid (211 200 anchors).
    Round 5: the reference value is the oracle's step in FLOAT64 (bf16: with the dense-conv operands rounded like the HIP
    kernels round them), evaluated on the candidate set the GPU selected at the configured threshold; the GPU gradient must
    be no farther from it than a small multiple of the distance the CPU oracle's own fp32 evaluation has."""
    from oracle import train_ref
    case = oracle_case(cfgfile, names, HALF)
    c, sd, types, gts, masks, ref_args = (case[k] for k in ("c", "sd", "types", "gts", "masks", "ref_args"))
    model = case["model"].to(dev).train()
    npi = case["np_inputs"]
    kw = dict(voxels=[torch.from_numpy(v).to(dev) for v in npi["voxels"]],
              coordinates=[torch.from_numpy(v).to(dev) for v in npi["coordinates"]],
              num_points=[torch.from_numpy(v).to(dev) for v in npi["num_points"]],
              anchors={n: [torch.from_numpy(a).to(dev) for a in npi["anchors"][n]] for n in names},
              anchors_mask={n: [torch.from_numpy(a).to(dev) for a in npi["anchors_mask"][n]] for n in names},
              gt_bboxes=[torch.from_numpy(g).to(dev) for g in gts],
              gt_labels=[torch.tensor(l, dtype=torch.int64, device=dev) for l in npi["gt_labels"]], gt_types=types)
    from sassd import autograd as AG
    # 1. the GPU step, at the CONFIGURED guided-anchor threshold (train_cfg.rpn.anchor_thr = 0.1); the selection its kernel
    #    makes is recorded
    try:
        AG.set_bev_precision(precision)
        with H.GuidedSelectionTap() as tap:
            losses = model(None, [dict(sample_idx=0), dict(sample_idx=1)], return_loss=True, **kw)
        total = sum(v.sum() for v in losses.values())
        total.backward()
        torch.cuda.synchronize()
    finally:
        AG.set_bev_precision("fp32")
    # 2. the CPU oracle on the same candidate set (guided_sel): the ARBITER -- the step in float64 -- and the oracle's own
    #    fp32 evaluation, whose distance to the arbiter is the floor of this workload.  bf16: both with the dense-conv
    #    operands rounded where the HIP kernels round them.
    msk = np.concatenate([np.stack(masks[n]) for n in names], 1).reshape(2, -1)
    gsel = [H.selection_as_mask_ranks(tap.selected[b], msk[b]) for b in range(2)]
    okw = dict(bf16=("bev",)) if precision == "bf16" else {}
    arb_l, arb_g, ex = train_ref.train_step(*ref_args, guided_sel=gsel, dtype=torch.float64, **okw)
    ref_l, ref_g, _ = train_ref.train_step(*ref_args, guided_sel=gsel, **okw)
    # 3. the selection itself: the GPU may differ from the arbiter's own selection only in candidates whose arbiter score lies
    #    within the score tolerance of the threshold (fp32: 1e-4, the north_star's box / score tolerance; bf16: 1e-2 -- measured
    #    2.2e-3: rounding the BEV operands moves the scores by that much, on the GPU and between the two CPU evaluations alike)
    stol = 1e-4 if precision == "fp32" else 1e-2
    nsel = ndiff = 0
    for b in range(2):
        diff = np.setxor1d(gsel[b], ex["guided_free"][b])
        nsel, ndiff = nsel + len(gsel[b]), ndiff + len(diff)
        far = [int(i) for i in diff if abs(ex["masked_top"][b][i] - 0.1) > stol]
        assert not far, (b, far[:5], [float(ex["masked_top"][b][i]) for i in far[:5]])
    assert set(losses) == set(arb_l) == {"aux_loss_cls", "aux_loss_reg", "rpn_loc_loss", "rpn_cls_loss", "rpn_dir_loss",
                                         "loss_cls"}
    assert int((ex["labels"] > 0).sum()) > 10 and int((ex["ext_labels"] > 0).sum()) >= len(types[0]) + len(types[1])
    if len(names) > 1:
        assert set(np.unique(ex["labels"].numpy())) >= {0, 1, 2, 3}
    # 4. losses: fp32 1e-4 relative to the arbiter (round 4: 1e-3 against the fp32 oracle); bf16: 3 x the distance of the two
    #    rounded CPU evaluations, at least 2e-3 (round 4: 3e-2 against the UNROUNDED oracle)
    for k, v in arb_l.items():
        got = float(losses[k].detach().sum())
        bar = 1e-4 * max(1.0, abs(v)) if precision == "fp32" else max(3 * abs(ref_l[k] - v), 2e-3 * max(1.0, abs(v)))
        assert np.isfinite(got) and abs(got - v) <= bar, (k, got, v, ref_l[k], bar)
    # 5. gradients, every parameter: ||g_gpu - g_arbiter|| <= max(3 x ||g_cpu32 - g_arbiter||, 5e-3 ||g_arbiter||), the whole
    #    model 3 x (floor 5e-4 / bf16 1e-3).  The 5e-3 floor is the GPU's OWN fp32 noise where the CPU oracle happens to be
    #    exact: its fp32 BEV stack runs Winograd F(4x4,3x3) (forward and data gradient), whose transforms cost ~2e-4 of the
    #    feature maximum per layer where the CPU's direct convolution costs 1e-6 -- measured 1.0-1.5e-3 on the dense stack's
    #    tensors and 2.7e-3 on the box head's bias (a cancelling sum over 140 800 anchors) at K21 scale, 50-250 x the CPU
    #    oracle's distance, 2-5 x under the floor.  Round 4 held 2e-2 / 4e-2 per tensor against the fp32 oracle, which said
    #    "two fp32 evaluations disagree", not which one is right; no fp32 bar here is looser than 5e-3 unless the CPU oracle
    #    itself is more than 1.7e-3 from float64 on that tensor.
    got_g = {n: p.grad for n, p in model.named_parameters()}
    for name, p in model.named_parameters():
        if arb_g.get(name) is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0, name
        else:
            assert p.grad is not None, name
    fl = dict(rel_floor=5e-3, whole_factor=3.0, whole_floor=5e-4) if precision == "fp32" else dict(rel_floor=5e-3, whole_factor=3.0, whole_floor=1e-3)
    bad, whole, whole_ok, rows = H.arbiter_report(got_g, {k: v for k, v in arb_g.items() if v is not None}, ref_g, **fl)
    H.dump_rows("step_vs_oracle %s %d %s" % (cfgfile, HALF["sparse_shape"][2], precision), rows)
    print("training step vs float64 arbiter (%s, %d-wide grid, %s): %d candidates selected, %d differ from the arbiter's own "
          "selection (all within %.0e of the threshold); whole-model gradient GPU %.2e / CPU oracle %.2e; worst tensors "
          "(GPU, CPU oracle) %s" % (cfgfile, HALF["sparse_shape"][2], precision, nsel, ndiff, stol, whole[0], whole[1],
                                    [(k, "%.1e" % a, "%.1e" % b_) for a, b_, k in rows[:6]]))
    print("   losses (GPU, arbiter, CPU oracle):", {k: (round(float(losses[k].detach().sum()), 6), round(arb_l[k], 6),
                                                        round(ref_l[k], 6)) for k in arb_l})
    assert len(rows) >= 60 and not bad, bad
    assert whole_ok, whole


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


def _k21_golden():
    import importlib.util
    import os
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_golden_train_k21", os.path.join(gdir, "make_golden_train_k21.py"))
    MG = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(MG)
    return MG, np.load(os.path.join(gdir, "train_k21_ref.npz"))


def _k21_step(dev, precision, force_sel=None, hook=None):
    """forward_train + backward of the bench's training workload on the HIP kernels -> (model, losses, golden module, golden)"""
    from sassd import train, autograd as AG
    MG, G = _k21_golden()
    model, c, w, clouds, gts = MG.build()
    model = model.to(dev).train()
    model.train_cfg.rpn.anchor_thr = float(G["anchor_thr"])
    cal = w["cal"]
    anchors = dict(Car=torch.from_numpy(w["anchors"]).to(dev))
    anchors_bv = dict(Car=torch.from_numpy(w["anchors_bv"]).to(dev))
    AG.set_bev_precision(precision)
    try:
        batch = train.device_batch([torch.from_numpy(p).to(dev) for p in clouds], [torch.from_numpy(g).to(dev) for g in gts],
                                   [np.array(["Car"] * len(g)) for g in gts], ["Car"], anchors, anchors_bv, cal["voxel_size"],
                                   cal["pc_range"], max_points=cal["max_points"], max_voxels=cal["max_voxels"], model=model)
        assert sum(v.shape[0] for v in batch["voxels"]) == int(G["n_voxels"]) == 32245
        assert int(sum(m.sum() for m in batch["anchors_mask"]["Car"])) == int(G["n_masked"])
        with H.GuidedSelectionTap(force=force_sel) as tap:
            losses = model(**batch)
        total = sum(v.sum() for v in losses.values())
        if hook is not None:
            hook("backward")
        total.backward()
        torch.cuda.synchronize()
    finally:
        AG.set_bev_precision("fp32")
    return model, losses, tap, MG, G


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_training_step_k21_vs_oracle(dev, precision):
    """BASELINE configs[2], parity ON THE WORKLOAD bench.py MEASURES: car_cfg on its full grid, batch 2 of K21 frames
    (21 500 points each, 32 245 voxels), 8 boxes per frame, the weights bench.py trains -- forward_train + backward on the
    HIP kernels, batch built by the product's own device_batch -- against the CPU oracle's step stored in
    tests/golden/train_k21_ref.npz (tests/golden/make_golden_train_k21.py; imported here for the shared seeded inputs).

    Round 5 -- what the GPU is compared WITH.  The golden holds four evaluations of the step on one candidate set: the
    fp32 oracle, its FLOAT64 arbiter, and the rounded-operand step (dense-conv operands rounded to bf16 exactly where the
    HIP kernels round them) in fp32 and float64.
      fp32: the GPU runs at the golden's threshold-safe guided-anchor threshold and must select the golden's candidates
        itself; six loss terms 1e-4 relative to the arbiter; every stored tensor ||g_gpu - g64|| <= max(3 x ||g_cpu32 - g64||,
        5e-3 ||g64||), the stored layers together 3 x (measured: GPU 1.39e-3, CPU oracle 6.5e-4; worst GPU tensor 3.5e-3 where
        the CPU oracle has 4.1e-3; the dense stack 1.0-1.5e-3 where the CPU's direct convolution has 2e-5 -- the Winograd
        F(4x4) transforms of the GPU's fp32 BEV path); every parameter's norm and seeded projection within the same multiple of
        its stored distance ||g_cpu32 - g64||.  (Round 4 held 2e-2 per tensor against the fp32 oracle: two
        fp32 summation orders disagree by 6.4e-3 on this workload -- tests/test_train_order_sensitivity_cpu.py -- and a
        bar against one of them says nothing about which is right.)
      bf16 (BEV convs on the bf16 MFMA, what the bench line runs): the FULL objective, loss_cls included -- the golden's
        candidate set is forced onto the GPU step (bf16 moves ~2000 scores near the threshold by more than any margin; the
        selection kernel itself is held by the fp32 case) -- against the rounded-operand float64 step, same multiples of
        the distance between the two rounded CPU evaluations (measured: stored layers together GPU 1.58e-1, the two CPU
        evaluations 1.58e-1 apart).  That distance is NOT small: rounding
        is discontinuous, activations that differ in the last fp32 bits round a few operands to different bf16 neighbours,
        and those 2^-8 jumps feed the next layer's roundings -- the two CPU evaluations of the rounded step are about as
        far apart as the rounded step is from the unrounded one (tests/test_train_arbiter_cpu.py asserts it from this
        fixture; tests/analysis/train_arbiter_study.py reproduces it on a small workload).  A whole-step comparison of a
        bf16 step can therefore not be tight for ANY implementation; what pins the bf16 step tightly is
        test_bf16_step_launches_vs_rounded_reference below (every bf16 launch of this very step, on its live operands,
        1e-5 against the rounded-operand product on the CPU)."""
    tag, ftag = ("f64/", "") if precision == "fp32" else ("b64/", "b32/")
    MG, G = _k21_golden()
    force = [G["sel0"], G["sel1"]] if precision == "bf16" else None
    model, losses, tap, MG, G = _k21_step(dev, precision, force)
    if precision == "fp32":
        for b in range(2):
            assert np.array_equal(tap.own[b], G["sel%d" % b]), "sample %d: the GPU selected other guided anchors" % b
    else:
        moved = [len(np.setxor1d(tap.own[b], G["sel%d" % b])) for b in range(2)]
        print("bf16: the kernel's own selection differs from the golden's in %s of %s candidates (golden set forced)"
              % (moved, [len(G["sel0"]), len(G["sel1"])]))
    got_l = {k: float(v.detach().sum()) for k, v in losses.items()}
    names = [str(k) for k in G["loss_names"]]
    arb_l, flo_l = dict(zip(names, G[tag + "losses"])), dict(zip(names, G[ftag + "losses"]))
    assert set(got_l) == set(arb_l) and len(arb_l) == 6
    for k, v in arb_l.items():
        bar = 1e-4 * max(1.0, abs(v)) if precision == "fp32" else max(3 * abs(flo_l[k] - v), 2e-3 * max(1.0, abs(v)))
        assert v != 0 and abs(got_l[k] - v) <= bar, (k, got_l[k], v, flo_l[k], bar)
    params = dict(model.named_parameters())
    # stored layers, elementwise
    got, arb, flo = {}, {}, {}
    for k in G.files:
        if k.startswith(tag + "grad:") or k.startswith(tag + "grad8:"):
            kind, name = k[len(tag):].split(":", 1)
            g = params[name].grad
            assert g is not None, name
            got[name] = g[:8] if kind == "grad8" else g
            arb[name] = torch.from_numpy(G[k])
            flo[name] = torch.from_numpy(G[ftag + kind + ":" + name])
    fl = dict(rel_floor=5e-3, whole_factor=3.0, whole_floor=5e-4) if precision == "fp32" else dict(rel_floor=5e-3, whole_factor=3.0, whole_floor=1e-3)
    bad, whole, whole_ok, rows = H.arbiter_report(got, arb, flo, **fl)
    H.dump_rows("k21 golden %s" % precision, rows)
    # every parameter: norm and seeded projection against the arbiter's, in multiples of the stored distance
    dist = dict(zip([str(n) for n in G[ftag + "grad_names"]], G[ftag + "grad_dist"]))
    worst_n, worst_p, bad_np = {}, {}, {}
    for name, norm, proj in zip(G[tag + "grad_names"], G[tag + "grad_norms"], G[tag + "grad_projs"]):
        name = str(name)
        if norm < 1e-7:
            continue
        g = params[name].grad
        assert g is not None, name
        gd = g.detach().double().cpu().reshape(-1)
        slack = max(3 * dist[name], fl["rel_floor"] * norm)
        en = abs(float(gd.norm()) - norm)
        # an error e with ||e|| <= slack, seen through a seeded unit-variance direction r: dot(e, r) ~ N(0, ||e||^2) -> 4 sigma
        ep = abs(float(torch.dot(gd, MG.projection(name, gd.numel()))) - proj)
        worst_n[name], worst_p[name] = en / norm, ep / norm
        if en > slack or ep > 4 * slack:
            bad_np[name] = (en / norm, ep / norm, slack / norm)
    print("K21 x 2 training step (%s) vs the %s float64 arbiter: losses (GPU, arbiter, CPU fp32)" % (
        precision, "rounded-operand" if precision == "bf16" else "fp32 oracle's"),
        {k: (round(got_l[k], 6), round(float(v), 6), round(float(flo_l[k]), 6)) for k, v in arb_l.items()},
        "| stored layers together: GPU %.2e / CPU oracle %.2e | worst stored tensors (GPU, CPU oracle): %s | all %d parameters: "
        "worst norm error %.2e, worst projection error %.2e" % (whole[0], whole[1], [(k, "%.1e" % a, "%.1e" % b_) for a, b_, k in rows[:8]],
                                                                len(worst_n), max(worst_n.values()), max(worst_p.values())))
    assert len(rows) >= 40 and len(worst_n) >= 70
    assert not bad, bad
    assert whole_ok, whole
    assert not bad_np, bad_np


def test_bf16_step_launches_vs_rounded_reference(dev):
    """The TIGHT pin of the bf16 training step (BASELINE configs[2]; the configuration the bench's `train` record measures):
    every dense-convolution launch of the K21 x 2 step under set_bev_precision("bf16") -- forward, data gradient and weight
    gradient of the eight BEVNet layers, the fused RPN head and the two rescoring-head layers -- is replayed ON ITS LIVE
    OPERANDS (the tensors the step itself produced on the GPU) by torch-CPU convolutions over operands rounded to bf16 with
    torch.bfloat16 where oracle.train_ref.bf16_conv_rule says the HIP kernel rounds, and must agree to 1e-5 relative L2 (weight
    gradients 2e-5; fp32 CPU sums; the kernels measure 5e-7 against float64 in tests/test_gpu_bf16.py).  Everything between those launches is the
    code of the fp32 step, which test_training_step_k21_vs_oracle[fp32] holds against the float64 arbiter: together the two
    tests pin the bf16 step launch by launch, which no whole-step comparison can (see there)."""
    from oracle import train_ref
    from sassd import autograd as AG
    log = []
    fwd0, bwd0 = AG.Conv2dFn.forward, AG.Conv2dFn.backward

    def fwd(ctx, x, weight, bias, packed, wino, wino4=None):
        y = fwd0(ctx, x, weight, bias, packed, wino, wino4)
        log.append(("fwd", x.detach(), weight.detach(), None if bias is None else bias.detach(), y.detach()))
        return y

    def bwd(ctx, dy):
        out = bwd0(ctx, dy)
        x, weight = ctx.saved_tensors
        log.append(("bwd", x.detach(), weight.detach(), dy.detach(), out[0], out[1]))
        return out

    # (the replay hooks Conv2dFn: BatchNorm folded into the next layer's loaders -- round 6, bit-identical to this layer-by-layer
    # form, tests/test_gpu_bf16.py::test_bevnet_with_fused_bn_is_bit_identical -- is switched off for it)
    from sassd.detector import BEVNet
    AG.Conv2dFn.forward, AG.Conv2dFn.backward = staticmethod(fwd), staticmethod(bwd)
    fuse0, BEVNet.fuse_bn_into_conv = BEVNet.fuse_bn_into_conv, False
    try:
        _k21_step(dev, "bf16")
    finally:
        AG.Conv2dFn.forward, AG.Conv2dFn.backward = staticmethod(fwd0), staticmethod(bwd0)
        BEVNet.fuse_bn_into_conv = fuse0
    nf = sum(1 for e in log if e[0] == "fwd")
    assert nf >= 11 and len(log) == 2 * nf, (nf, len(log))
    rb = train_ref.round_bf16
    worst, rounded = {}, [0, 0, 0]
    for e in log:
        x, w = e[1].cpu(), e[2].cpu()
        cout, cin, ks = w.shape[0], w.shape[1], w.shape[2]
        rf, rd, rw = train_ref.bf16_conv_rule(cin, cout, ks, x.shape[3], x.shape[2])
        tagk = "%dx%d %d->%d" % (ks, ks, cin, cout)
        if e[0] == "fwd":
            ref = F.conv2d(rb(x) if rf else x, rb(w) if rf else w, None if e[3] is None else e[3].cpu(), 1, ks // 2)
            worst[("forward", tagk)] = max(worst.get(("forward", tagk), 0.0), _rel(e[4], ref))
            rounded[0] += rf
        else:
            dy = e[3].cpu()
            if e[4] is not None:
                ref = torch.nn.grad.conv2d_input(x.shape, rb(w) if rd else w, rb(dy) if rd else dy, padding=ks // 2)
                worst[("data gradient", tagk)] = max(worst.get(("data gradient", tagk), 0.0), _rel(e[4], ref))
                rounded[1] += rd
            ref = torch.nn.grad.conv2d_weight(rb(x) if rw else x, w.shape, rb(dy) if rw else dy, padding=ks // 2)
            worst[("weight gradient", tagk)] = max(worst.get(("weight gradient", tagk), 0.0), _rel(e[5], ref))
            rounded[2] += rw
    print("bf16 step, every dense-conv launch on its live operands vs the rounded-operand CPU product (%d forward / %d data-"
          "gradient / %d weight-gradient launches on the bf16 MFMA of %d layers): %s"
          % (rounded[0], rounded[1], rounded[2], nf, {"%s %s" % k: "%.1e" % v for k, v in sorted(worst.items())}))
    assert rounded[0] >= 7 and rounded[1] >= 7 and rounded[2] >= 11, rounded
    # forward / data gradient 1e-5 (measured <= 3.5e-7).  Weight gradient 2e-5: its CPU reference sums 70 400 products per
    # element in fp32 (oneDNN's blocking depends on the host's core count) -- the measured 5.8e-6 on the 256 -> 256 layers is that
    # reference's own rounding, the kernel is 5e-7 from float64 on such operands (tests/test_gpu_bf16.py)
    bad = {k: v for k, v in worst.items() if not v < (2e-5 if k[0] == "weight gradient" else 1e-5)}
    assert not bad, bad


def test_training_step_waymo_vs_oracle(dev):
    """BASELINE configs[4] as a TRAINING config, parity: one 180 000-point frame (79 302 voxels, grid 40 x 1504 x 1504, BEV
    188 x 188) through forward_train + backward on the HIP kernels -- the batch built by the product's own device_batch
    (HIP voxelizer, anchor mask, rulebooks) -- against the CPU oracle's step on the same model / frame / boxes, read from
    tests/golden/waymo_train_ref.npz (the oracle needs minutes for this frame; tests/golden/make_golden_waymo_train.py made
    the file and is imported here for the shared seeded inputs).
    Round 5: the golden holds the fp32 oracle AND its float64 arbiter; the bars are those of the K21 workload -- six loss
    terms 1e-4 relative to the arbiter; every stored tensor ||g_gpu - g64|| <= max(3 x ||g_cpu32 - g64||, 5e-3 ||g64||), the
    stored layers together 3 x; every parameter's norm and seeded projection within the same multiple of its stored distance.
    (Rounds 3-4: fixed 5e-3 bars against the fp32 oracle and 2.5e-2 on the projections.)"""
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
    names = [str(k) for k in G["loss_names"]]
    arb_l, flo_l = dict(zip(names, G["f64/losses"])), dict(zip(names, G["losses"]))
    assert set(got_l) == set(arb_l) and len(arb_l) == 6
    for k, v in arb_l.items():
        assert v != 0 and abs(got_l[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, got_l[k], v, flo_l[k])
    params = dict(model.named_parameters())
    got, arb, flo = {}, {}, {}
    for k in G.files:
        if k.startswith("f64/grad:") or k.startswith("f64/grad8:"):
            kind, name = k[4:].split(":", 1)
            g = params[name].grad
            assert g is not None, name
            got[name] = g[:8] if kind == "grad8" else g
            arb[name] = torch.from_numpy(G[k])
            flo[name] = torch.from_numpy(G[kind + ":" + name])
    # (measured: stored layers together GPU 4.0e-5 / CPU oracle 1.4e-5; worst tensor extra_conv.1.bias 2.2e-3 / 8.1e-4)
    fl = dict(rel_floor=5e-3, whole_factor=3.0, whole_floor=5e-4)
    bad, whole, whole_ok, rows = H.arbiter_report(got, arb, flo, **fl)
    H.dump_rows("waymo golden fp32", rows)
    dist = dict(zip([str(n) for n in G["grad_names"]], G["grad_dist"]))
    worst_n, worst_p, bad_np = {}, {}, {}
    for name, norm, proj in zip(G["f64/grad_names"], G["f64/grad_norms"], G["f64/grad_projs"]):
        name = str(name)
        if norm < 1e-7:
            continue
        g = params[name].grad
        assert g is not None, name
        gd = g.detach().double().cpu().reshape(-1)
        slack = max(3 * dist[name], fl["rel_floor"] * norm)
        en = abs(float(gd.norm()) - norm)
        ep = abs(float(torch.dot(gd, MG.projection(name, gd.numel()))) - proj)      # dot(e, r) ~ N(0, ||e||^2): 4 sigma
        worst_n[name], worst_p[name] = en / norm, ep / norm
        if en > slack or ep > 4 * slack:
            bad_np[name] = (en / norm, ep / norm, slack / norm)
    print("waymo-scale training step vs the float64 arbiter: losses (GPU, arbiter, CPU fp32)",
          {k: (round(got_l[k], 5), round(float(v), 5), round(float(flo_l[k]), 5)) for k, v in arb_l.items()},
          "| stored layers together: GPU %.2e / CPU oracle %.2e | worst stored tensors (GPU, CPU oracle): %s | all %d parameters: "
          "worst norm error %.2e, worst projection error %.2e" % (whole[0], whole[1], [(k, "%.1e" % a, "%.1e" % b_) for a, b_, k in rows[:6]],
                                                                len(worst_n), max(worst_n.values()), max(worst_p.values())))
    assert len(rows) >= 40 and len(worst_n) >= 75
    assert not bad, bad
    assert whole_ok, whole
    assert not bad_np, bad_np

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


def test_bf16_and_fp32_training_trajectories_agree(dev):
    """The parity statement the launch-by-launch and whole-step tests cannot make: do bf16 and fp32
    training CONVERGE alike?  200 steps of BASELINE configs[2] (the bench's training workload: car_cfg, batch 2, K21 frames + 8
    synthetic boxes per frame, adam_onecycle) from the same seeds, fp32 / fp32 again / bf16 BEV convolutions
    (tests/analysis/train_trajectory.py; its record of the round is profiles/r06_train_trajectory.json).  Two fp32 runs are
    not bit-equal -- the auxiliary head scatters with float atomics -- and 200 steps amplify that, so the yardstick is the
    fp32 run-to-run spread.  Measured on an MI355X (mean of the last 20 steps): total loss 1.522 / 1.588 / 1.466 -- fp32
    run-to-run 4.3 %, bf16 vs fp32 3.7 %; worst single term bf16 vs fp32: aux_loss_reg 15 % (fp32 run-to-run: rpn_dir_loss
    12 %, rpn_loc_loss 11 %); step 0: 105.5128 / 105.5128 / 105.5209.  Later boxes of the round gave fp32 pairs 1.422 / 1.629
    (14.6 % apart) and bf16 1.545 / 1.466 (the 1x1 layers and the 256 -> 28 layer on the bf16 MFMA as well): over all runs seen,
    fp32 ends at 1.42 .. 1.63 (mean 1.56, sigma 6 %), bf16 at 1.47 .. 1.55 -- inside the fp32 spread.  Two single draws from
    distributions 6 % wide differ by more than 10 % one time in five, so the bands are 3 sigma of that difference: total within
    25 % of the fp32 run (or 1.5 x the fp32 pair's own distance, if larger), every one of the six terms within 45 %, step-0 losses
    within 1e-3, and both runs must have LEARNED (loss under 5 % of its start; measured 1.4-1.5 %)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "analysis"))
    import train_trajectory as TT
    steps = 200
    runs = {"fp32": TT.run("fp32", steps, dev), "fp32_again": TT.run("fp32", steps, dev), "bf16": TT.run("bf16", steps, dev)}
    tail = {k: TT.summary(v) for k, v in runs.items()}
    f, g, h = tail["fp32"], tail["fp32_again"], tail["bf16"]
    terms = [t for t in f if t != "loss"]
    assert len(terms) == 6, terms
    line = []
    for t in ["loss"] + terms:
        rr, bf = abs(f[t] - g[t]) / abs(f[t]), abs(h[t] - f[t]) / abs(f[t])
        line.append("%s %.4g/%.4g/%.4g (fp32 run-to-run %.1f%%, bf16 %.1f%%)" % (t, f[t], g[t], h[t], 100 * rr, 100 * bf))
        assert np.isfinite(h[t]) and np.isfinite(f[t])
        assert bf <= max(0.25 if t == "loss" else 0.45, 1.5 * rr), (t, f[t], g[t], h[t], bf)
    print("training trajectories after %d steps, mean of the last 20 (fp32 / fp32 again / bf16): %s" % (steps, "; ".join(line)))
    l0 = [runs[k]["loss"][0] for k in ("fp32", "fp32_again", "bf16")]
    assert abs(l0[0] - l0[1]) <= 1e-3 * l0[0] and abs(l0[2] - l0[0]) <= 1e-3 * l0[0], l0
    for k in ("fp32", "bf16"):
        assert tail[k]["loss"] < 0.05 * runs[k]["loss"][0], (k, tail[k]["loss"], runs[k]["loss"][0])
