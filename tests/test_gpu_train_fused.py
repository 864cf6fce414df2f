"""-m gpu: the fused training kernels (sassd_assign_targets, sassd_rpn_loss) against the elementwise torch formulation
they replace (sassd.train_ops.create_target_torch / SSDRotateHead.loss with fused_loss=False), which is itself held
to the CPU oracle by test_gpu_train.py.  Labels / argmax choices exact, targets and best overlaps bit-equal (same fp32
operation sequence), loss sums and gradients within 2e-6 relative (different summation order only)."""
import numpy as np
import pytest
import torch

from sassd import kernels as K
from sassd import train_ops as T
from sassd import synth
from sassd.config import Config, _wrap
from sassd.detector import SSDRotateHead

pytestmark = pytest.mark.gpu


def _scene(seed, n_gt, dev, classes=("Car",)):
    w = synth.workload("car")
    anchors = torch.from_numpy(w["anchors"]).to(dev)                    # [A,7]
    r = np.random.RandomState(seed)
    gt = np.zeros((n_gt, 7), np.float32)
    gt[:, 0], gt[:, 1] = r.uniform(2, 68, n_gt), r.uniform(-38, 38, n_gt)
    gt[:, 2] = r.uniform(-1.9, -1.5, n_gt)
    gt[:, 3], gt[:, 4], gt[:, 5] = r.uniform(1.4, 1.9, n_gt), r.uniform(3.2, 4.6, n_gt), r.uniform(1.4, 1.7, n_gt)
    gt[:, 6] = r.uniform(-3.3, 3.3, n_gt)
    if n_gt > 2:
        gt[1] = gt[0]                                                   # duplicated box: ties between two ground truths
        gt[2, :2] = w["anchors"][12345, :2]                             # one box exactly on an anchor centre
    mask = torch.from_numpy(r.rand(anchors.shape[0]) < 0.3).to(dev)
    return anchors, torch.from_numpy(gt).to(dev), mask


@pytest.mark.parametrize("n_gt", [0, 1, 8, 37, 150])
def test_assign_targets_matches_create_target_torch(dev, n_gt):
    B = 2
    anchors, gt0, mask0 = _scene(1 + n_gt, n_gt, dev)
    _, gt1, mask1 = _scene(77 + n_gt, max(n_gt - 1, 0), dev)
    A = anchors.shape[0]
    g = torch.Generator().manual_seed(n_gt)
    gts, masks = [gt0, gt1], [mask0, mask1]
    cls = [torch.randint(1, 4, (t.shape[0],), generator=g).to(dev) for t in gts]
    oks = [(torch.rand(t.shape[0], generator=g) < 0.8).to(dev) for t in gts]
    ref = [T.create_target_torch(anchors, masks[b], gts[b], cls[b], oks[b], T.NearestIouSimilarity(),
                                 T.second_box_encode, 0.6, 0.45) for b in range(B)]
    labels = torch.empty(B, A, dtype=torch.int64, device=dev)
    targets = torch.empty(B, A, 7, device=dev)
    best = torch.empty(B, A, device=dev)
    npos = torch.empty(B, dtype=torch.int32, device=dev)
    counts = [t.shape[0] for t in gts]
    tot = sum(counts)
    K.assign_targets(anchors, torch.stack(masks).view(torch.uint8), torch.cat(gts) if tot else None,
                     torch.cat(cls) if tot else None, torch.cat(oks).view(torch.uint8) if tot else None,
                     K.gt_offsets(counts, dev), 0.6, 0.45, labels, targets, npos, best=best)
    for b in range(B):
        assert torch.equal(labels[b], ref[b][0]), (b, int((labels[b] != ref[b][0]).sum()))
        assert torch.equal(targets[b], ref[b][1])
        assert torch.equal(best[b], ref[b][2])
        assert int(npos[b]) == int((ref[b][0] > 0).sum())
    if n_gt >= 8:
        assert int(npos.sum()) > 0


def test_assign_targets_strided_outputs_and_accumulating_positives(dev):
    """Two 'classes' writing into one [B, 2, A] tensor (the multi-class layout of SSDRotateHead.loss)."""
    anchors, gt, mask = _scene(5, 12, dev)
    A = anchors.shape[0]
    labels = torch.full((2, 2, A), -7, dtype=torch.int64, device=dev)
    targets = torch.full((2, 2, A, 7), -7.0, device=dev)
    npos = torch.empty(2, dtype=torch.int32, device=dev)
    off = K.gt_offsets([12, 12], dev)
    gts = torch.cat([gt, gt])
    for c, (hi, lo) in enumerate(((0.6, 0.45), (0.35, 0.2))):
        K.assign_targets(anchors, None, gts, None, None, off, hi, lo, labels[:, c], targets[:, c], npos,
                         zero_num_pos=c == 0, out_stride=2 * A)
    for c, (hi, lo) in enumerate(((0.6, 0.45), (0.35, 0.2))):
        ref = T.create_target_torch(anchors, None, gt, None, None, T.NearestIouSimilarity(), T.second_box_encode, hi, lo)
        for b in range(2):
            assert torch.equal(labels[b, c], ref[0]) and torch.equal(targets[b, c], ref[1])
    assert int(npos[0]) == int((labels[0] > 0).sum())


@pytest.mark.parametrize("cfg_path,names", [("configs/car_cfg.py", ["Car"]),
                                            ("configs/multi_cfg.py", ["Car", "Pedestrian", "Cyclist"])])
def test_fused_rpn_loss_matches_elementwise_path(dev, cfg_path, names):
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, cfg_path))
    head_cfg = dict(cfg.model.bbox_head)
    head_cfg.pop("type")
    head = SSDRotateHead(**head_cfg).to(dev)
    ncls, B, H, W = len(names), 2, 40, 48
    g = torch.Generator().manual_seed(3)
    a_c = H * W * 2
    # anchors on a small grid (real sizes / rotations), gts near anchors so that every label value occurs
    from sassd import anchors as AN
    r = np.random.RandomState(0)
    anchors, masks, gtb, gtl, gtt = {}, {}, [], [], []
    sizes = {"Car": [1.6, 3.9, 1.56], "Pedestrian": [0.6, 0.8, 1.73], "Cyclist": [0.6, 1.76, 1.73]}
    for n in names:
        an = AN.AnchorGeneratorStride(sizes=sizes[n], anchor_strides=[.4, .4, 1.], anchor_offsets=[.2, -9.4, -1.78],
                                      rotations=[0, 1.57])([1, H, W]).reshape(-1, 7).astype(np.float32)
        anchors[n] = torch.from_numpy(np.stack([an] * B)).to(dev)
        masks[n] = torch.from_numpy(r.rand(B, a_c) < 0.6).to(dev)
    for b in range(B):
        k = 9
        t = [names[i % ncls] for i in range(k)]
        box = np.zeros((k, 7), np.float32)
        for i, n in enumerate(t):
            base = anchors[n][b, r.randint(a_c)].cpu().numpy()
            box[i] = base + np.array([0.1, -0.07, 0.05, 0.03, 0.1, 0.02, r.uniform(-0.3, 0.3)], np.float32)
        gtb.append(torch.from_numpy(box).to(dev))
        gtl.append(torch.tensor([names.index(n) + 1 for n in t], dtype=torch.int64, device=dev))
        gtt.append(np.array(t))
    nloc = head._num_anchor_per_loc
    outs = []
    for ch in (nloc * 7, nloc * head._num_class, nloc * 2):
        y = (torch.randn(B, ch, H, W, generator=g) * 0.5).to(dev)
        outs.append(y.view(B, ncls, -1, H, W).permute(0, 1, 3, 4, 2).contiguous().requires_grad_())
    res = {}
    for fused in (False, True):
        c = _wrap(dict(cfg.train_cfg.rpn))
        c["fused_loss"] = fused
        for o in outs:
            o.grad = None
        losses = head.loss(*outs, gtb, gtl, gtt, anchors, masks, c)
        sum(v.sum() for v in losses.values()).backward()
        res[fused] = ({k: v.detach().clone() for k, v in losses.items()}, [o.grad.clone() for o in outs])
    for k in res[False][0]:
        a, b = float(res[True][0][k].sum()), float(res[False][0][k].sum())
        assert abs(a - b) <= 2e-6 * max(1.0, abs(b)), (k, a, b)
        assert res[True][0][k].shape == res[False][0][k].shape
    assert float(res[False][0]["rpn_loc_loss"]) > 0 and float(res[False][0]["rpn_dir_loss"]) > 0
    for ga, gb in zip(res[True][1], res[False][1]):
        err = float((ga - gb).abs().max()) / max(float(gb.abs().max()), 1e-12)
        assert err < 2e-5, err


# ---- the same kernels against the vectors generated from the REFERENCE's own Python (tests/golden/train_fns.npz) -----
import os as _os

_G = np.load(_os.path.join(_os.path.dirname(__file__), "golden", "train_fns.npz"))


def _g(name, dev):
    return torch.from_numpy(_G[name]).to(dev)


@pytest.mark.parametrize("case", ["masked", "nomask", "nogt"])
def test_assign_targets_vs_reference_golden(dev, case):
    """create_target_np of the reference (target_ops.py:139-277) on its own inputs: labels exact, targets / overlaps 1e-6."""
    anchors, gt = _g("anchors", dev), _g("gt", dev)
    gmask = torch.tensor([1, 1, 0, 1, 1, 1, 0, 1, 1], dtype=torch.uint8, device=dev)
    am, gm, gb = dict(masked=(_g("anchor_mask", dev), gmask, gt), nomask=(None, None, gt),
                      nogt=(_g("anchor_mask", dev), None, gt[:0]))[case]
    A = anchors.shape[0]
    labels = torch.empty(1, A, dtype=torch.int64, device=dev)
    targets = torch.empty(1, A, 7, device=dev)
    best = torch.empty(1, A, device=dev)
    npos = torch.empty(1, dtype=torch.int32, device=dev)
    n = gb.shape[0]
    K.assign_targets(anchors.contiguous(), am.view(1, A).view(torch.uint8).contiguous() if am is not None else None,
                     gb.contiguous() if n else None, None, gm, K.gt_offsets([n], dev), 0.6, 0.45, labels, targets, npos,
                     best=best)
    assert np.array_equal(labels[0].cpu().numpy(), _G["ct_%s_labels" % case])
    assert np.abs(targets[0].cpu().numpy() - _G["ct_%s_targets" % case]).max() <= 1e-6
    mx = best[0].cpu()
    ref_mx = _G["ct_%s_max" % case]
    got = mx[am.cpu().bool()].numpy() if am is not None else mx.numpy()
    assert np.abs(got - ref_mx).max() <= 1e-6
    assert int(npos[0]) == int((_G["ct_%s_labels" % case] > 0).sum())


def test_fused_rpn_loss_vs_reference_golden(dev):
    """SSDRotateHead.loss of the reference on its own inputs: the three losses 2e-6, the three gradients 1e-6."""
    from sassd.config import ConfigDict
    head = SSDRotateHead(num_class=1, num_output_filters=8, num_anchor_per_loc=2, box_code_size=7).to(dev)
    anc = dict(Car=torch.stack([_g("anchors", dev), _g("a2", dev)]))
    msk = dict(Car=torch.stack([_g("anchor_mask", dev), _g("m2", dev)]))
    gtb = [_g("gt", dev), _g("gt2", dev)]
    gtl = [torch.ones(9, dtype=torch.int64, device=dev), torch.ones(6, dtype=torch.int64, device=dev)]
    gtt = [np.array(["Car"] * 9), np.array(["Car"] * 5 + ["Van"])]
    box, cls, dr = (_g(k, dev).clone().requires_grad_() for k in ("rpn_box", "rpn_cls", "rpn_dir"))
    cfg = ConfigDict(assigner=ConfigDict(Car=ConfigDict(pos_iou_thr=0.6, neg_iou_thr=0.45, min_pos_iou=0.45),
                                         ignore_iof_thr=-1, similarity_fn="NearestIouSimilarity"), anchor_thr=0.1)
    ls = head.loss(box, cls, dr, gtb, gtl, gtt, anc, msk, cfg)                  # fused path (CUDA tensors)

    def close(a, b, tol):
        a, b = np.asarray(a.detach().cpu(), np.float64), np.asarray(b, np.float64)
        assert a.shape == b.shape and np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), np.abs(a - b).max()
    for k in ("rpn_loc_loss", "rpn_cls_loss", "rpn_dir_loss"):
        close(ls[k], _G[k], 2e-6)
    tot = ls["rpn_loc_loss"] + ls["rpn_cls_loss"] + ls["rpn_dir_loss"]
    gb, gc, gd = torch.autograd.grad(tot.sum(), [box, cls, dr])
    close(gb, _G["rpn_gbox"], 1e-6)
    close(gc, _G["rpn_gcls"], 1e-6)
    close(gd, _G["rpn_gdir"], 1e-6)


def test_pack_plan_matches_individual_packs(dev):
    """sassd.train.PackPlan: after an optimizer step every cached weight image (sparse fwd / transposed, direct conv fwd
    / data-gradient, bf16 fwd / data-gradient) equals what the individual pack routine produces from the new weights."""
    import os
    from sassd import autograd as AG, spconv as SP, train
    from sassd.detector import _HipConv2d, build_detector
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "car_cfg.py"))
    try:
        AG.set_bev_precision("bf16")
        model = synth.randomize_detector(build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg), 0).to(dev)
        opt = train.build_optimizer(model, cfg.optimizer, 1)
        assert opt.pack_plan is not None
        opt.flat.grad.normal_(0, 1e-2)
        opt.lr = 1e-2
        opt.step()
        torch.cuda.synchronize()
        checked = dict(sp=0, spt=0, direct=0, dgrad=0, bf=0, bft=0, c1=0, c1t=0)
        for m in model.modules():
            if isinstance(m, SP.SparseConvolution):
                k = int(np.prod(m.kernel_size))
                w = m.weight.detach().reshape(k, m.in_channels, m.out_channels).contiguous()
                assert m._packed_version == K.weight_key(m.weight)
                assert torch.equal(m._packed, K.spconv_pack_weight(w)); checked["sp"] += 1
                if m.in_channels >= 16:
                    # submanifold layers differentiate on the forward rulebook: offset-reversed transposed image
                    rev = isinstance(m, SP.SubMConv3d) and k == 27 and AG.SparseConvFn.subm_on_forward_table
                    gen, pk, src = AG._sp_t_packs[(m.weight.data_ptr(), (k, m.in_channels, m.out_channels)) + ((True,) if rev else ())]
                    ref = K.spconv_pack_weight_t(w.flip(0).contiguous() if rev else w)
                    assert gen == K.weight_key(m.weight) and torch.equal(pk, ref); checked["spt"] += 1
                    assert src is m.weight                   # (the entry pins the tensor it was packed from)
            elif isinstance(m, _HipConv2d):
                w = m.weight.detach()
                key = (w.data_ptr(), tuple(w.shape))
                if m.kernel_size[0] == 1:
                    assert m._pkv == K.weight_key(m.weight) and torch.equal(m._pk, K.conv2d_pack_weight(w.contiguous()))
                    gen, d = AG._dgrad_direct[key]
                    wt = w.transpose(0, 1).flip(2, 3).contiguous()
                    assert gen == K.weight_key(m.weight) and torch.equal(d["packed"], K.conv2d_pack_weight(wt))
                    assert tuple(d["wt"].shape) == tuple(wt.shape)
                    checked["direct"] += 1; checked["dgrad"] += 1
                    for tr, name in ((False, "c1"), (True, "c1t")):                 # (round 6) bf16 MFMA fragments of the 1x1 convs
                        co_g, ci_g = (m.in_channels, m.out_channels) if tr else (m.out_channels, m.in_channels)
                        if K.conv1x1_bf16_supported(ci_g, co_g, 4):
                            gen, pk, src = AG._bf16_1x1_packs[key + (tr,)]
                            assert src is m.weight and gen == K.weight_key(m.weight)
                            assert torch.equal(pk, K.conv1x1_bf16_pack_weight(w, tr)); checked[name] += 1
                else:
                    gen, pk, src = AG._bf16_packs[key + (False,)]
                    assert src is m.weight
                    wp = w                                              # (round 6) Cout zero-padded to a multiple of 32: 28 -> 32
                    if m.out_channels % 32:
                        wp = torch.cat([w, w.new_zeros((AG.bf16_cout_pad(m.out_channels) - m.out_channels,) + tuple(w.shape[1:]))], 0)
                    assert gen == K.weight_key(m.weight) and torch.equal(pk, K.conv2d_bf16_pack_weight(wp.contiguous()))
                    checked["bf"] += 1
                    if m.in_channels % 32 == 0:
                        gen, pk, src = AG._bf16_packs[key + (True,)]
                        assert src is m.weight
                        wt = w.transpose(0, 1).flip(2, 3).contiguous()
                        assert gen == K.weight_key(m.weight) and torch.equal(pk, K.conv2d_bf16_pack_weight(wt))
                        checked["bft"] += 1
        assert checked["sp"] >= 14 and checked["spt"] >= 13 and checked["bf"] == 8 and checked["bft"] == 8, checked
        assert checked["direct"] >= 5 and checked["dgrad"] >= 5 and checked["c1"] >= 5 and checked["c1t"] >= 5, checked
    finally:
        AG.set_bev_precision("fp32")


def test_padded_guided_path_equals_list_path(dev):
    """forward_train with the sync-free padded guided anchors (sassd_guided_select + batched PSWarp + one assignment
    call) against the per-sample list formulation it replaces: every loss term and every parameter gradient."""
    import bench
    from sassd import train
    w = synth.workload("car")
    model, cfg = synth.build_detector_for(w, 0, train=True, cls_bias=-2.0)     # enough anchors above anchor_thr
    model = model.to(dev).train()
    anchors = dict(Car=torch.from_numpy(w["anchors"]).to(dev))
    anchors_bv = dict(Car=torch.from_numpy(w["anchors_bv"]).to(dev))
    clouds = [torch.from_numpy(synth.k21(i)).to(dev) for i in range(2)]
    gts = [torch.from_numpy(bench.synth_gt_on_points(synth.k21(i), i)[:8 - 3 * i]).to(dev) for i in range(2)]  # 8 and 5
    types = [np.array(["Car"] * int(g.shape[0])) for g in gts]
    batch = train.device_batch(clouds, gts, types, ["Car"], anchors, anchors_bv, synth.KITTI_VOXEL, synth.KITTI_RANGE,
                               model=model)
    res = {}
    for padded in (False, True):
        model.train_cfg.rpn["padded_guided"] = padded
        model.zero_grad(set_to_none=True)
        losses = model(**batch)
        sum(v.sum() for v in losses.values()).backward()
        res[padded] = ({k: float(v.sum()) for k, v in losses.items()},
                       {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    model.rpn_head.check_guided_capacity()              # default capacity = all anchors: cannot overflow
    model.train_cfg.rpn["guided_cap"] = 64               # a capacity that IS too small must be reported, not ignored
    model(**batch)
    with pytest.raises(RuntimeError, match="guided_cap"):
        model.rpn_head.check_guided_capacity()
    del model.train_cfg.rpn["guided_cap"]
    assert res[True][0].keys() == res[False][0].keys()
    for k, v in res[False][0].items():
        assert abs(res[True][0][k] - v) <= 2e-6 * max(1.0, abs(v)), (k, res[True][0][k], v)
    assert float(res[False][0]["loss_cls"]) > 0
    assert res[True][1].keys() == res[False][1].keys()
    worst = 0.0
    for n, g in res[False][1].items():
        d = float((res[True][1][n] - g).norm()) / max(float(g.norm()), 1e-12)
        worst = max(worst, d)
        assert d < 1e-4, (n, d)
    print("padded vs list guided path: worst relative gradient difference %.2e" % worst)


@pytest.mark.parametrize("n,c", [(16111, 64), (18355, 32), (777, 16), (3, 64), (40000, 4)])
def test_bn_relu_fused_matches_torch(dev, n, c):
    """sassd_bn_relu_fwd / _bwd against torch BatchNorm1d(training) + ReLU: output, the three gradients and the running
    statistics (fp32; sums in a different order: 2e-5 relative to each tensor's scale)."""
    from sassd.autograd import BnReluFn
    g = torch.Generator().manual_seed(n + c)
    x = (torch.randn(n, c, generator=g) * 1.7 + 0.4).to(dev)
    dy = torch.randn(n, c, generator=g).to(dev)
    bn = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev).train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g).to(dev) + 0.5)
        bn.bias.copy_((torch.randn(c, generator=g) * 0.3).to(dev))
        bn.running_mean.normal_(0, 0.1)
        bn.running_var.uniform_(0.5, 1.5)
    rm, rv = bn.running_mean.clone(), bn.running_var.clone()
    xr = x.clone().requires_grad_()
    torch.relu(bn(xr)).backward(dy)
    xf = x.clone().requires_grad_()
    gam, bet = bn.weight.detach().clone().requires_grad_(), bn.bias.detach().clone().requires_grad_()
    y = BnReluFn.apply(xf, gam, bet, rm, rv, 0.01, 1e-3)
    y.backward(dy)
    with torch.no_grad():
        y_ref = torch.relu(torch.nn.functional.batch_norm(x, None, None, bn.weight, bn.bias, True, 0.0, 1e-3))

    def close(a, b, tol=2e-5):
        scale = max(float(b.abs().max()), 1e-6)
        assert float((a - b).abs().max()) <= tol * scale, (float((a - b).abs().max()), scale)
    close(y, y_ref)
    close(xf.grad, xr.grad, 1e-4)
    close(gam.grad, bn.weight.grad, 1e-4)
    close(bet.grad, bn.bias.grad, 1e-4)
    close(rm, bn.running_mean)
    close(rv, bn.running_var)
    y2 = BnReluFn.apply(x, gam.detach(), bet.detach(), None, None, 0.01, 1e-3)       # bit-reproducible, stats optional
    assert torch.equal(y2, y.detach())


@pytest.mark.parametrize("shape", [(2, 256, 200, 176), (1, 256, 188, 188), (2, 28, 200, 176), (3, 7, 6, 10)])
def test_bn2d_relu_fused_matches_torch(dev, shape):
    """sassd_bn2d_relu_fwd / _bwd (NCHW) against torch BatchNorm2d(training) + ReLU: output, the three gradients and the
    running statistics (fp32; sums in a different order: 2e-5 / 1e-4 relative to each tensor's scale, as for the sparse
    pair).  Shapes: the BEV stack at batch 2 / the Waymo-scale map, the 28-channel part-sensitive head, a tiny map."""
    from sassd.autograd import BnRelu2dFn
    b, c, h, w = shape
    g = torch.Generator().manual_seed(c + h)
    x = (torch.randn(*shape, generator=g) * 1.7 + 0.4).to(dev)
    dy = torch.randn(*shape, generator=g).to(dev)
    bn = torch.nn.BatchNorm2d(c, eps=1e-3, momentum=0.01).to(dev).train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g).to(dev) + 0.5)
        bn.bias.copy_((torch.randn(c, generator=g) * 0.3).to(dev))
        bn.running_mean.normal_(0, 0.1)
        bn.running_var.uniform_(0.5, 1.5)
    rm, rv = bn.running_mean.clone(), bn.running_var.clone()
    xf = x.clone().requires_grad_()
    gam, bet = bn.weight.detach().clone().requires_grad_(), bn.bias.detach().clone().requires_grad_()
    y = BnRelu2dFn.apply(xf, gam, bet, rm, rv, 0.01, 1e-3)
    y.backward(dy)
    # reference gradients with the kernel's OWN ReLU mask: among 9 M pre-activations one or two lie within fp32 rounding of
    # zero, where torch's z and the kernel's z may take different signs -- a legitimate O(|dy|) difference in that
    # element (and in the channel's dgamma / dbeta) that says nothing about the arithmetic
    xr = x.clone().requires_grad_()
    (bn(xr) * (y.detach() > 0)).backward(dy)
    with torch.no_grad():
        y_ref = torch.relu(torch.nn.functional.batch_norm(x, None, None, bn.weight, bn.bias, True, 0.0, 1e-3))

    def close(a, b, tol=2e-5):
        scale = max(float(b.abs().max()), 1e-6)
        assert float((a - b).abs().max()) <= tol * scale, (float((a - b).abs().max()), scale)
    close(y, y_ref)
    close(xf.grad, xr.grad, 1e-4)
    close(gam.grad, bn.weight.grad, 1e-4)
    close(bet.grad, bn.bias.grad, 1e-4)
    close(rm, bn.running_mean)
    close(rv, bn.running_var)
    y2 = BnRelu2dFn.apply(x, gam.detach(), bet.detach(), None, None, 0.01, 1e-3)       # bit-reproducible, stats optional
    assert torch.equal(y2, y.detach())


@pytest.mark.parametrize("kind", ["sparse", "nchw"])
def test_bn_statistics_of_a_large_mean_small_variance_channel(dev, kind):
    """the statistics kernels add <= 16 (64) elements in fp32 before they widen to double; with raw x^2 sums a channel
    with |mean| >> std (mean 40, variance 1e-2) lost ~1 % of its invstd to the cancellation E[x^2] - mean^2.  Every thread now sums
    the RESIDUALS to the first value it loads in fp32 and re-bases them to zero in double (a first version shifted the whole channel
    by its first element: worse than no shift on near-constant maps with an outlier corner, caught by the full-grid multi_cfg
    step), so the saved mean / invstd must match a FLOAT64 evaluation: invstd to 1e-4
    relative (the fp32 input itself carries 40 * 2^-24 = 2.4e-6 of noise per element against a std of 0.1), the output to 1e-3
    of its scale.  One ordinary channel rides along."""
    from sassd import kernels as K
    g = torch.Generator().manual_seed(7)
    if kind == "sparse":
        x = torch.randn(20011, 16, generator=g, dtype=torch.float64)
        x[:, 3] = 40.0 + 0.1 * torch.randn(20011, generator=g, dtype=torch.float64)
        x[:, 7] = -1e3 + 0.5 * torch.randn(20011, generator=g, dtype=torch.float64)
        x[:, 9] = 0.2 + 1e-3 * torch.randn(20011, generator=g, dtype=torch.float64)        # near-constant, outlier first row
        x[0, 9] = 3.0
        xf = x.float().to(dev)
        gam, bet = torch.ones(16, device=dev), torch.zeros(16, device=dev)
        y, mean, invstd = K.bn_relu_fwd(xf, gam, bet, None, None, 0.01, 1e-3)
        x64 = xf.double().cpu()                               # what the kernel was given, in float64
        m64, v64 = x64.mean(0), x64.var(0, unbiased=False)
    else:
        x = torch.randn(2, 8, 40, 44, generator=g, dtype=torch.float64)
        x[:, 3] = 40.0 + 0.1 * torch.randn(2, 40, 44, generator=g, dtype=torch.float64)
        x[:, 5] = -1e3 + 0.5 * torch.randn(2, 40, 44, generator=g, dtype=torch.float64)
        x[:, 6] = 0.2 + 1e-3 * torch.randn(2, 40, 44, generator=g, dtype=torch.float64)    # near-constant map, outlier corner pixel
        x[0, 6, 0, 0] = 3.0
        xf = x.float().to(dev)
        gam, bet = torch.ones(8, device=dev), torch.zeros(8, device=dev)
        y, mean, invstd = K.bn2d_relu_fwd(xf, gam, bet, None, None, 0.01, 1e-3)
        x64 = xf.double().cpu()
        m64, v64 = x64.mean((0, 2, 3)), x64.var((0, 2, 3), unbiased=False)
    is64 = 1.0 / torch.sqrt(v64 + 1e-3)
    rel = ((invstd.double().cpu() - is64).abs() / is64).max().item()
    dm = ((mean.double().cpu() - m64).abs() / m64.abs().clamp(min=1.0)).max().item()
    print("BN statistics (%s) with mean 40 / -1000 channels: invstd rel err %.2e, mean rel err %.2e" % (kind, rel, dm))
    assert rel < 1e-4 and dm < 1e-6, (rel, dm)
    shape = (1, -1) if kind == "sparse" else (1, -1, 1, 1)
    ref = torch.relu((x64 - m64.view(shape)) * is64.view(shape))
    assert (y.double().cpu() - ref).abs().max().item() < 1e-3 * max(1.0, ref.abs().max().item())


def test_fused_aux_head_matches_module_path(dev):
    """The fused auxiliary head (sassd_aux_prepare / sassd_aux_head_fwd / _bwd: voxel centres, point-in-box labels,
    interpolation weights, three interpolations, three Linear layers, focal + smooth-L1 and the whole backward) against the
    module-by-module formulation it replaces (tensor2points, nearest_neighbor_interpolate, nn.Linear, build_aux_target,
    train_ops losses -- itself held to the CPU oracle by test_gpu_train.py): the intermediate tensors first, then both
    aux loss terms and EVERY parameter gradient of a whole forward_train."""
    import bench
    from sassd import train
    from sassd.detector import SpMiddleFHD
    w = synth.workload("car")
    model, cfg = synth.build_detector_for(w, 0, train=True, cls_bias=-3.0)
    model = model.to(dev).train()
    anchors = dict(Car=torch.from_numpy(w["anchors"]).to(dev))
    anchors_bv = dict(Car=torch.from_numpy(w["anchors_bv"]).to(dev))
    clouds = [torch.from_numpy(synth.k21(i)).to(dev) for i in range(2)]
    gts = [torch.from_numpy(bench.synth_gt_on_points(synth.k21(i), i)[:8 - 3 * i]).to(dev) for i in range(2)]
    gts[1][1, :3] = gts[1][0, :3] + 0.3                       # two overlapping boxes: the LAST containing box wins
    types = [np.array(["Car"] * int(g.shape[0])) for g in gts]
    batch = train.device_batch(clouds, gts, types, ["Car"], anchors, anchors_bv, synth.KITTI_VOXEL, synth.KITTI_RANGE,
                               model=model)
    # ---- pieces: prepare kernel against the torch expressions -----------------------------------------------------------
    ret = model.merge_second_batch({k: v for k, v in batch.items() if k not in ("img", "img_meta", "return_loss")})
    vx = model.backbone(ret["voxels"], ret["num_points"])
    neck = model.neck
    with torch.no_grad():
        SpMiddleFHD.fused_aux = False
        try:
            _, _, (pm, pc, pr) = neck(vx, ret["coordinates"], 2, is_test=False, indice_dict=ret.get("sassd_rulebooks"))
        finally:
            SpMiddleFHD.fused_aux = True
        lab_ref, off_ref = neck.build_aux_target(pm, ret["gt_bboxes"])
    counts = [int(g.shape[0]) for g in gts]
    points, known, label, target, npos = K.aux_prepare(vx.contiguous(), ret["coordinates"].int().contiguous(),
                                                       [torch.zeros(1, 4, dtype=torch.int32, device=dev)] * 3,
                                                       neck.aux_voxel_size, neck.aux_offset, torch.cat(gts).contiguous(),
                                                       K.gt_offsets(counts, dev), 2)
    assert torch.equal(points, pm) and torch.equal(label, lab_ref) and int(npos) == int((lab_ref > 0).sum()) > 0
    assert torch.equal(target, off_ref)
    # ---- whole step, fused vs module path -------------------------------------------------------------------------------
    res = {}
    for fused in (False, True):
        SpMiddleFHD.fused_aux = fused
        try:
            model.zero_grad(set_to_none=True)
            losses = model(**batch)
            sum(v.sum() for v in losses.values()).backward()
        finally:
            SpMiddleFHD.fused_aux = True
        res[fused] = ({k: float(v.sum()) for k, v in losses.items()},
                      {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    for k, v in res[False][0].items():
        assert abs(res[True][0][k] - v) <= 1e-5 * max(1.0, abs(v)), (k, res[True][0][k], v)
    assert res[False][0]["aux_loss_cls"] > 0 and res[False][0]["aux_loss_reg"] > 0
    assert res[True][1].keys() == res[False][1].keys()
    worst = {}
    for n, g in res[False][1].items():
        worst[n] = float((res[True][1][n] - g).norm()) / max(float(g.norm()), 1e-12)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print("fused vs module aux head: losses", {k: (res[True][0][k], res[False][0][k]) for k in ("aux_loss_cls", "aux_loss_reg")},
          "worst gradient differences", [(k, "%.1e" % v) for k, v in top])
    assert max(worst.values()) < 1e-4, top


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_fused_head_data_gradient_follows_in_place_weight_edits(dev, precision):
    """the fused RPN head hands Conv2dFn a `torch.cat` of its three conv weights -- a derived tensor whose
    `_version` is always 0 and whose storage address the caching allocator repeats.  Its packed data-gradient image must
    not be cached: after an in-place edit of the source parameters (what load_state_dict / any torch.optim step does,
    without sassd's generation bump) the gradient into the shared BEV features has to use the NEW weights."""
    from sassd import autograd as AG
    torch.manual_seed(3)
    head = SSDRotateHead(num_class=1, num_output_filters=256).to(dev).train()
    x = torch.randn(1, 256, 24, 32, device=dev, requires_grad=True)

    def dx_of():
        x.grad = None
        ys = head(x)
        sum((y * (i + 1.0)).sum() for i, y in enumerate(ys)).backward()
        return x.grad.clone()

    def ref_of():
        # (bf16, round 6: the 1x1 data gradient runs on the bf16 MFMA -- operands rounded; the upstream gradients 1, 2, 3 are exact)
        rnd = (lambda t: t.bfloat16().float()) if precision == "bf16" else (lambda t: t)
        xr = x.detach().clone().requires_grad_(True)
        ys = [torch.nn.functional.conv2d(xr, rnd(c.weight.detach()), c.bias.detach()) for c in
              (head.conv_box, head.conv_cls, head.conv_dir_cls)]
        sum((y * (i + 1.0)).sum() for i, y in enumerate(ys)).backward()
        return xr.grad

    AG.set_bev_precision(precision)
    try:
        for rep in range(3):
            got, ref = dx_of(), ref_of()
            assert (got - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item()), (rep, precision)
            with torch.no_grad():                            # no K.bump_weights_generation(): a foreign optimizer's step
                for c in (head.conv_box, head.conv_cls, head.conv_dir_cls):
                    c.weight.mul_(-1.7).add_(0.01)
    finally:
        AG.set_bev_precision("fp32")
