"""-m gpu end-to-end parity: the fused HIP pipeline (through the C ABI) against the whole-path CPU oracle on the
same seeded inputs and weights; stage-by-stage tolerances, final boxes / scores within 1e-4 (north_star)."""
import numpy as np
import pytest
import torch

import sassd
from sassd import synth, anchors as A
from sassd.config import Config
from sassd.detector import build_detector
from sassd.pipeline import InferencePlan
import helpers as H

pytestmark = pytest.mark.gpu

CFG = dict(voxel_size=synth.KITTI_VOXEL, pc_range=synth.KITTI_RANGE, max_points=5, max_voxels=20000,
           sparse_shape=(40, 1600, 1408), grid_xyz=(1408, 1600, 40))


def _anchors(names=("Car",)):
    sizes = dict(Car=[1.6, 3.9, 1.56], Pedestrian=[0.6, 0.8, 1.73], Cyclist=[0.6, 1.76, 1.73])
    an = np.concatenate([A.AnchorGeneratorStride(sizes=sizes[n], anchor_strides=[.4, .4, 1.], anchor_offsets=[.2, -39.8, -1.78],
                                                 rotations=[0, 1.57])([1, 200, 176]).reshape(-1, 7) for n in names], 0)
    return an, A.rbbox2d_to_near_bbox(an[:, [0, 1, 3, 4, 6]]).astype(np.float32)


def _model(cfgfile="configs/car_cfg.py", seed=0):
    c = Config.fromfile(cfgfile)
    m = H.randomize_detector(build_detector(c.model, c.train_cfg, c.test_cfg).eval(), seed)
    names = c.data.val.class_names
    H.calibrate_cls_head(m, H.frame("small", 11), _anchors(names)[1], CFG)
    return m, c


def _abs_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max()) if a.size else 0.0


def _box_ok(got, want):
    """1e-4 ABSOLUTE on every box field (north_star), every workload.  (Rounds 1-5 carried an fp32-relative escape for the
    Waymo-scale random-weight model; its boxes measure 4.8e-7 absolute since the synthetic regression head was scaled: deleted in
    round 6)"""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return bool(np.all(np.abs(got - want) <= 1e-4))


def _check_sample(tag, plan, res, ref, b, errs):
    """One sample against the oracle: guided anchors (same order), PSWarp logits, final boxes / scores / labels.
    ABSOLUTE 1e-4 on every box field and score (north_star), no skipped cases: the thresholds handed to the plan were
    chosen away from every oracle score (helpers.oracle_forward_safe)."""
    gb, gl, gs = ref["guided"][b]
    k = int(plan.df["counts"][b].item())
    assert k == len(gb), (tag, b, k, len(gb))
    got = plan.df["guided"][b, :k].cpu().numpy()
    if k:
        fe = np.abs(got.astype(np.float64) - gb.numpy()).max(0)
        errs["guided_by_field(x,y,z,w,l,h,r)"] = np.maximum(errs.get("guided_by_field(x,y,z,w,l,h,r)", 0.0), fe)
    e = _abs_err(got, gb.numpy())
    errs["guided_boxes"] = max(errs.get("guided_boxes", 0.0), e)
    assert _box_ok(got, gb.numpy()), (tag, "guided anchors", e)
    assert np.array_equal(plan.df["labels"][b, :k].cpu().numpy(), gl.numpy())
    e = _abs_err(plan.logits[b, :k].cpu().numpy(), ref["logits"][b].numpy())
    errs["pswarp_logits"] = max(errs.get("pswarp_logits", 0.0), e)
    # part-sensitive logits average 28 samples of a feature map whose values reach |f| ~ 10 with random weights:
    # 1e-4 absolute for O(1) features, fp32-relative to the sampled map otherwise
    assert e <= max(1e-4, 2e-5 * float(ref["psfeat"].abs().max())), (tag, "logits", e)
    d = ref["dets"][b]
    if d is None:
        assert res[b][0] is None, (tag, b)
        return 0
    assert res[b][0] is not None and len(res[b][0]) == len(d[0]), (tag, b, None if res[b][0] is None else len(res[b][0]), len(d[0]))
    eb, es = _abs_err(res[b][0], d[0]), _abs_err(res[b][1], d[1])
    errs["det_boxes"] = max(errs.get("det_boxes", 0.0), eb)
    errs["det_scores"] = max(errs.get("det_scores", 0.0), es)
    assert _box_ok(res[b][0], d[0]) and es <= 1e-4, (tag, "detections", eb, es)
    assert np.array_equal(res[b][2], d[2])
    return len(d[0])


@pytest.mark.parametrize("frames,seed,score_thr", [(("k21",), 0, 0.3), (("small", "k17"), 1, 0.6)])
def test_pipeline_vs_oracle(dev, frames, seed, score_thr):
    model, c = _model()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    an, bv = _anchors()
    clouds = [H.frame(f, seed + i) for i, f in enumerate(frames)]
    ref, rpn_thr, score_thr = H.oracle_forward_safe(sd, clouds, an, bv, CFG, score_thr=score_thr)
    B = len(clouds)
    plan = InferencePlan(sd, batch_size=B, anchors=an, anchors_bv=bv, device=dev, rpn_thr=rpn_thr, score_thr=score_thr)
    plan.run_from_points([torch.from_numpy(p).to(dev) for p in clouds])
    torch.cuda.synchronize()
    assert int(plan.status.item()) == 0
    # -- voxel features / coords bit exact
    n0 = int(plan.n[0].item())
    assert n0 == len(ref["coors"])
    assert np.array_equal(plan.idx[0][:n0].cpu().numpy(), ref["coors"])
    assert np.array_equal(plan.mean[:n0].cpu().numpy(), ref["feats"])
    # -- level-3 indices bit exact, sparse features close
    n3 = int(plan.n[3].item())
    assert n3 == len(ref["idx3"]) and np.array_equal(plan.idx[3][:n3].cpu().numpy(), ref["idx3"])
    sp = plan.sp_out[:n3].cpu()
    # (features: bars RELATIVE to the map's maximum, ~85 for this model.  Round 5 tightened them 10 x -- sparse 1e-5, BEV 2e-5 of
    # the maximum; measured 1.1e-6 / 2.1e-6 -- and prints the relative figure: rounds 1-4 printed only the absolute error, whose
    # 1.9e-4 read as if it sat on the old 2e-4 bar)
    e = (sp - ref["x3"]).abs().max().item()
    m3 = max(1.0, ref["x3"].abs().max().item())
    assert e < 1e-5 * m3, (e, m3)
    errs = {"sparse_features": e, "sparse_features/max": e / m3}
    # -- dense BEV stack
    for name, got in (("conv6", plan.conv6), ("x", plan.x)):
        r = ref[name]
        e = (got.cpu() - r).abs().max().item()
        errs["bev_" + name] = e
        errs["bev_" + name + "/max"] = e / max(1.0, r.abs().max().item())
        assert e < 2e-5 * max(1.0, r.abs().max().item()), (name, e, r.abs().max().item())
    # -- anchors mask exact
    assert np.array_equal(plan.mask.cpu().numpy().astype(bool), ref["masks"])
    res = plan.results()
    ndet = sum(_check_sample("car", plan, res, ref, b, errs) for b in range(B))
    print("max abs errors vs the CPU oracle (%s): %s" % ("+".join(frames), {k: (["%.1e" % x for x in v] if isinstance(v, np.ndarray) else "%.2e" % v) for k, v in errs.items()}))
    assert ndet >= 1, "test vector produced no detections at all"


def _det_state(plan):
    """what a frame leaves behind: detections and guided anchors up to their counts (rows past a count are stale by contract),
    part-sensitive logits of the candidates, anchor masks, the two BEV maps the heads read"""
    k = int(plan.det["counts"][0].item())
    c = int(plan.df["counts"][0].item())
    return [t.clone() for t in (plan.det["counts"], plan.det["boxes"][0, :k], plan.det["scores"][0, :k], plan.det["labels"][0, :k],
                                plan.df["counts"], plan.df["guided"][0, :c], plan.logits[0, :c], plan.mask, plan.x, plan.conv6)]


@pytest.mark.parametrize("overlap", [True, False])
def test_frame_graph_replays_equal_the_eager_frame(dev, overlap):
    """The production launch path: the whole frame captured once (plan.capture) and replayed per frame (plan.run_graph), as the
    TWO-BRANCH graph (coordinate side stream, `overlap=True`) and as the ONE-BRANCH graph bench.py keeps in flight
    (`overlap=False`: rulebooks / anchor masks in front of the feature path on the frame's own stream; round 6).  Every replay
    must leave exactly what the eager frame leaves -- detections, guided-anchor counts, part-sensitive logits, anchor masks, the BEV
    maps, bit for bit -- for frames of different sizes replayed in turn, for THREE plans in flight on three streams, and AFTER THE
    HOST HAS RECYCLED DEVICE MEMORY (allocate + fill + free blocks of 2 MB .. 1 GB between replays): a captured hipMemsetAsync --
    a memset NODE, the one non-kernel node the frame had -- went wrong in exactly that situation (the first plan's dense map full
    of foreign data, SASSD_ST_BOX_OVERFLOW in 5 of 6 bench processes at Waymo scale); the frame clears its maps with fill kernels
    since (sassd_densify, sassd_anchor_mask*)."""
    model, c = _model()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    an, bv = _anchors()
    clouds = [torch.from_numpy(H.frame(f, i)).to(dev) for i, f in enumerate(("k21", "small", "k17", "k21"))]
    cap = max(int(p.shape[0]) for p in clouds) + 64
    eager = InferencePlan(sd, batch_size=1, anchors=an, anchors_bv=bv, device=dev)            # two-branch, launched from the host
    want = []
    for p in clouds:
        eager.run_from_points([p])
        torch.cuda.synchronize()
        assert int(eager.status.item()) == 0
        want.append(_det_state(eager))
    assert sum(int(w_[0].sum().item()) for w_ in want) >= 1, "no detections at all"
    plans = [InferencePlan(sd, batch_size=1, anchors=an, anchors_bv=bv, device=dev, overlap=overlap) for _ in range(3)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    for pl, st in zip(plans, streams):
        with torch.cuda.stream(st):
            pl.capture(cap)
    torch.cuda.synchronize()
    # one plan, frames in turn (a smaller frame after a larger one: nothing of the previous replay may leak)
    with torch.cuda.stream(streams[0]):
        for i in (0, 1, 2, 3, 1, 0):
            plans[0].run_graph([clouds[i]])
            torch.cuda.synchronize()
            for j, (got, ref) in enumerate(zip(_det_state(plans[0]), want[i])):
                assert got.shape == ref.shape and torch.equal(got, ref), ("sequential replay", overlap, i, j)
    # the host recycles device memory between replays (what any application around the plan does)
    for nbytes in (2 << 20, 8 << 20, 64 << 20, 1 << 30):
        junk = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        junk.fill_(255)
        torch.cuda.synchronize()
        del junk
        for pl, st in zip(plans, streams):
            with torch.cuda.stream(st):
                pl.run_graph([clouds[0]])
        torch.cuda.synchronize()
        for pl in plans:
            assert int(pl.status.item()) == 0, ("after the host recycled memory", overlap, nbytes)
            for j, (got, ref) in enumerate(zip(_det_state(pl), want[0])):
                assert got.shape == ref.shape and torch.equal(got, ref), ("after the host recycled memory", overlap, nbytes, j)
    # three plans in flight, each its own stream, no synchronisation between the launches
    for rnd in range(4):
        order = [(rnd + j) % 4 for j in range(3)]
        for pl, st, i in zip(plans, streams, order):
            with torch.cuda.stream(st):
                pl.run_graph([clouds[i]])
        torch.cuda.synchronize()
        for pl, i in zip(plans, order):
            assert int(pl.status.item()) == 0
            for j, (got, ref) in enumerate(zip(_det_state(pl), want[i])):
                assert got.shape == ref.shape and torch.equal(got, ref), ("in flight", overlap, rnd, i, j)


def test_reference_style_forward_test_api(dev):
    """model(img, img_meta, return_loss=False, voxels=[..], coordinates=[..], num_points=[..], anchors=[..],
    anchors_mask=[..]) -- the reference's calling convention (single_stage.py:110, tools/test.py:31)."""
    from sassd.voxel_generator import VoxelGenerator
    from oracle import nets as onets
    model, c = _model()
    model = model.to(dev)
    an, bv = _anchors()
    gen = VoxelGenerator(**{k: v for k, v in c.data.val.generator.items() if k != "type"})
    clouds = [H.frame("small", 5), H.frame("small", 6)]
    kw = dict(voxels=[], coordinates=[], num_points=[], anchors=[], anchors_mask=[])
    for p in clouds:
        v, co, n = gen.generate(p)                      # numpy API, HIP voxelizer underneath
        m = onets.anchors_mask(co, bv, gen.voxel_size, gen.point_cloud_range, gen.grid_size, 1)
        kw["voxels"].append(torch.from_numpy(v).to(dev)); kw["coordinates"].append(torch.from_numpy(co).to(dev))
        kw["num_points"].append(torch.from_numpy(n).to(dev)); kw["anchors"].append(torch.from_numpy(an).to(dev))
        kw["anchors_mask"].append(torch.from_numpy(m).to(dev))
    out = model(None, [dict(sample_idx=0), dict(sample_idx=1)], return_loss=False, **kw)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    ref = H.oracle_forward(sd, clouds, an, bv, CFG)
    for b in range(2):
        d = ref["dets"][b]
        if d is None:
            assert out[b]["boxes_lidar"] is None
        else:
            assert len(out[b]["boxes_lidar"]) == len(d[0])
            assert _abs_err(out[b]["boxes_lidar"], d[0]) <= 1e-4 and _abs_err(out[b]["scores"], d[1]) <= 1e-4


def test_spconv_facade_matches_fused_plan(dev):
    """The module-by-module spconv facade (SparseConvTensor / SubMConv3d / SparseConv3d / SparseSequential with
    torch BatchNorm1d+ReLU) and the fused plan compute the same backbone."""
    model, c = _model(seed=3)
    model = model.to(dev)
    sd = model.state_dict()
    pts = H.frame("small", 2)
    plan = InferencePlan(sd, batch_size=1, device=dev, anchors=_anchors()[0])
    plan.voxelize([torch.from_numpy(pts).to(dev)])
    plan.backbone()
    n0, n3 = int(plan.n[0].item()), int(plan.n[3].item())
    with torch.no_grad():
        x, conv6 = model.neck(plan.mean[:n0].clone(), plan.idx[0][:n0].clone(), 1, is_test=True)
    plan.bev_and_heads()
    torch.cuda.synchronize()
    assert (x - plan.x).abs().max().item() < 2e-4 * max(1.0, plan.x.abs().max().item())
    assert (conv6 - plan.conv6).abs().max().item() < 2e-4 * max(1.0, plan.conv6.abs().max().item())


@pytest.mark.parametrize("frames", [("small", "small"), ("k21",) * 8])
def test_multi_class_batch(dev, frames):
    """configs[3]: multi_cfg (Car + Pedestrian + Cyclist, 211200 anchors) -- a two-sample batch of sparse frames and the
    stated size, batch 8 of K21 frames."""
    model, c = _model("configs/multi_cfg.py", seed=4)
    names = c.data.val.class_names
    assert len(names) == 3 and model.rpn_head.conv_cls.out_channels == 18
    an, bv = _anchors(names)
    assert an.shape[0] == 211200
    B = len(frames)
    clouds = [H.frame(f, 20 + i) for i, f in enumerate(frames)]
    if frames[0] == "k21":                       # keep K ~ 10^2 per sample on full frames (capK 4096)
        H.calibrate_cls_head(model, clouds[0], _anchors(names)[1], CFG, target_count=300)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ref, rpn_thr, score_thr = H.oracle_forward_safe(sd, clouds, an, bv, CFG, num_class=3)
    plan = InferencePlan(sd, batch_size=B, num_class=3, anchors=an, anchors_bv=bv, device=dev, cap_k=4096, cap_d=1024,
                         rpn_thr=rpn_thr, score_thr=score_thr)
    plan.run_from_points([torch.from_numpy(p).to(dev) for p in clouds])
    res = plan.results()
    assert int(plan.status.item()) == 0
    assert np.array_equal(plan.mask.cpu().numpy().astype(bool), ref["masks"])
    n3 = int(plan.n[3].item())
    assert n3 == len(ref["idx3"]) and np.array_equal(plan.idx[3][:n3].cpu().numpy(), ref["idx3"])
    errs = {}
    ndet = sum(_check_sample("multi", plan, res, ref, b, errs) for b in range(B))
    print("max abs errors vs the CPU oracle (multi_cfg, batch %d): %s" % (B, {k: (["%.1e" % x for x in v] if isinstance(v, np.ndarray) else "%.2e" % v) for k, v in errs.items()}))
    assert ndet >= 1


WAYMO = dict(voxel_size=synth.WAYMO_VOXEL, pc_range=synth.WAYMO_RANGE, max_points=5, max_voxels=150000,
             sparse_shape=(40, 1504, 1504), grid_xyz=(1504, 1504, 40))


@pytest.mark.parametrize("batch", [1, 4])
def test_waymo_scale_frame(dev, batch):
    """configs[4] shape on one GPU, inference side: 180k points, 0.1 x 0.1 x 0.15 m voxels (grid 40x1504x1504,
    ~79k active voxels per frame, BEV 188x188), batch 1 and the stated batch 4.  Stresses the hash / bitmap-rank tables
    at 5-20x the KITTI row counts."""
    c = Config.fromfile("configs/car_cfg.py")
    mcfg = dict(c.model)
    mcfg["neck"] = dict(mcfg["neck"], output_shape=[40, 1504, 1504])
    mcfg["extra_head"] = dict(mcfg["extra_head"], grid_offsets=(75.2, 75.2), featmap_stride=0.8)
    model = H.randomize_detector(build_detector(mcfg, c.train_cfg, c.test_cfg).eval(), 7, sparse_fan_div=1)
    an = A.AnchorGeneratorStride(sizes=[1.6, 3.9, 1.56], anchor_strides=[.8, .8, 1.], anchor_offsets=[-74.8, -74.8, -1.0],
                                 rotations=[0, 1.57])([1, 188, 188]).reshape(-1, 7)
    bv = A.rbbox2d_to_near_bbox(an[:, [0, 1, 3, 4, 6]]).astype(np.float32)
    clouds = [synth.waymo_synth(s)[:180000] for s in range(batch)]
    H.calibrate_cls_head(model, clouds[0], bv, WAYMO, target_count=600)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    wcfg = dict(WAYMO, grid_offsets=(75.2, 75.2), featmap_stride=0.8)
    ref, rpn_thr, score_thr = H.oracle_forward_safe(sd, clouds, an, bv, wcfg)
    plan = InferencePlan(sd, batch_size=batch, anchors=an, anchors_bv=bv, device=dev, voxel_size=WAYMO["voxel_size"],
                         point_cloud_range=WAYMO["pc_range"], max_voxels=150000, sparse_shape=WAYMO["sparse_shape"],
                         grid_offsets=(75.2, 75.2), featmap_stride=0.8, cap_k=4096, cap_d=2048, rpn_thr=rpn_thr,
                         score_thr=score_thr)
    plan.run_from_points([torch.from_numpy(p).to(dev) for p in clouds])
    torch.cuda.synchronize()
    assert int(plan.status.item()) == 0
    n0, n3 = int(plan.n[0].item()), int(plan.n[3].item())
    assert n0 == len(ref["coors"]) and (batch > 1 or n0 == 79302)
    assert np.array_equal(plan.idx[0][:n0].cpu().numpy(), ref["coors"])
    assert np.array_equal(plan.mean[:n0].cpu().numpy(), ref["feats"])
    assert n3 == len(ref["idx3"]) and np.array_equal(plan.idx[3][:n3].cpu().numpy(), ref["idx3"])
    e = (plan.sp_out[:n3].cpu() - ref["x3"]).abs().max().item()
    m3 = max(1.0, ref["x3"].abs().max().item())
    assert e < 1e-5 * m3, (e, m3)
    errs = {"sparse_features": e, "sparse_features/max": e / m3}
    e = (plan.x.cpu() - ref["x"]).abs().max().item()
    mx = max(1.0, ref["x"].abs().max().item())
    errs["bev_x"] = e
    errs["bev_x/max"] = e / mx
    assert e < 2e-5 * mx, (e, mx)
    assert np.array_equal(plan.mask.cpu().numpy().astype(bool), ref["masks"])
    res = plan.results()
    ndet = sum(_check_sample("waymo", plan, res, ref, b, errs) for b in range(batch))
    print("max abs errors vs the CPU oracle (waymo-scale, batch %d): %s" % (batch, {k: (["%.1e" % x for x in v] if isinstance(v, np.ndarray) else "%.2e" % v) for k, v in errs.items()}))
    assert ndet >= 1


# ---- the SHIPPED thresholds (configs/car_cfg.py:71-81: guided-anchor threshold 0.1 literal in single_stage.py:122,
# test_cfg.extra.score_thr 0.3) with a tolerance-aware candidate-set comparison ------------------------------------------
MARGIN = 1e-4          # a candidate whose oracle score lies within MARGIN of a threshold may fall on either side


def _subsequence(small, big, tol=1e-4):
    """Indices into `big` of the rows of `small` (both in ascending anchor order), rows matched within `tol` absolute on
    every field; None if `small` is not a subsequence of `big`."""
    idx, j = [], 0
    for row in small:
        while j < len(big) and not np.all(np.abs(big[j].astype(np.float64) - row) <= tol):
            j += 1
        if j == len(big):
            return None
        idx.append(j)
        j += 1
    return idx


def _check_at_configured_thresholds(tag, plan, res, ft, b, rpn_thr, score_thr, iou_thr, stats):
    """Sample b of a plan run at the CONFIGURED thresholds against the oracle: every candidate whose oracle score is more
    than MARGIN away from a threshold must be selected / rejected exactly like the oracle does; the (few) candidates
    within MARGIN may go either way, and everything downstream is then compared strictly (1e-4 absolute on boxes and
    scores, labels exact) against the oracle evaluated on the candidate set the GPU actually chose."""
    from oracle import nets as onets
    lo, hi = ft["lo"], ft["hi"]
    g_lo, l_lo, s_lo = lo["guided"][b]
    g_hi = hi["guided"][b][0]
    k = int(plan.df["counts"][b].item())
    got = plan.df["guided"][b, :k].cpu().numpy().astype(np.float64)
    sel = _subsequence(got, g_lo.numpy())                    # GPU candidates  <=  oracle candidates at thr - MARGIN
    assert sel is not None, (tag, b, "a guided anchor of the GPU is not an oracle candidate at %.5f" % (rpn_thr - MARGIN))
    must = _subsequence(g_hi.numpy().astype(np.float64), got)          # oracle candidates at thr + MARGIN  <=  GPU candidates
    assert must is not None, (tag, b, "an oracle candidate above %.5f is missing on the GPU" % (rpn_thr + MARGIN))
    stats["borderline_guided"] += len(g_lo) - len(g_hi)
    stats["guided"] += k
    sel_t = torch.as_tensor(sel, dtype=torch.int64)
    assert np.array_equal(plan.df["labels"][b, :k].cpu().numpy(), l_lo[sel_t].numpy())
    logits = lo["logits"][b][sel_t]
    e = _abs_err(plan.logits[b, :k].cpu().numpy(), logits.numpy())
    assert e <= max(1e-4, 2e-5 * float(lo["psfeat"].abs().max())), (tag, "logits", e)
    # rescoring + NMS of the oracle on exactly these candidates; rescoring scores within MARGIN of score_thr: both ways
    sc = torch.sigmoid(logits).view(-1)
    border = torch.nonzero((sc - score_thr).abs() <= MARGIN).view(-1).tolist()
    stats["borderline_scores"] += len(border)
    assert len(border) <= 6, "too many borderline rescoring candidates to enumerate"
    options = []
    thr_eff = score_thr - MARGIN if border else score_thr   # admits every borderline candidate; each is then kept or forced out
    for bits in range(1 << len(border)):
        lg = logits.clone()
        for i, r in enumerate(border):
            if not (bits >> i) & 1:
                lg[r] = -50.0
        options.append(onets.rescore(g_lo[sel_t], lg, l_lo[sel_t], thr_eff, iou_thr))
    ok = False
    for d in options:
        if d is None:
            ok = ok or res[b][0] is None
            continue
        if res[b][0] is None or len(res[b][0]) != len(d[0]):
            continue
        if (_abs_err(res[b][0], d[0]) <= 1e-4 and _abs_err(res[b][1], d[1]) <= 1e-4 and np.array_equal(res[b][2], d[2])):
            ok = True
            stats["det_err"] = max(stats["det_err"], _abs_err(res[b][0], d[0]), _abs_err(res[b][1], d[1]))
    assert ok, (tag, b, "detections differ from the oracle's on the same candidate set")
    return 0 if res[b][0] is None else len(res[b][0])


@pytest.mark.parametrize("cfgfile,frames", [("configs/car_cfg.py", ("k21",)), ("configs/car_cfg.py", ("k21", "k17")),
                                            ("configs/multi_cfg.py", ("k21",) * 8)])
def test_pipeline_at_configured_thresholds(dev, cfgfile, frames):
    """The shipped configuration itself: guided-anchor threshold 0.1 (single_stage.py:122), score_thr 0.3 and NMS 0.1
    (car_cfg.py:71-81 / multi_cfg.py) on full K21 / K17 frames, car_cfg and multi_cfg at its stated batch 8 -- no
    threshold is moved.  GPU expf / fp32 summation order move a score by ~1e-6, so a candidate sitting within MARGIN of
    a threshold may legitimately land on either side; everything else must match the oracle."""
    model, c = _model(cfgfile, seed=5)
    names = c.data.val.class_names
    nc = len(names)
    an, bv = _anchors(names)
    B = len(frames)
    clouds = [H.frame(f, 40 + i) for i, f in enumerate(frames)]
    H.calibrate_cls_head(model, clouds[0], bv, CFG, target_count=300)
    tc = c.test_cfg.extra
    rpn_thr, score_thr, iou_thr = 0.1, float(tc.score_thr), float(tc.nms.iou_thr)
    assert (score_thr, iou_thr) == (0.3, 0.1)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ft = H.oracle_features(sd, clouds, an, bv, CFG, nc)
    ft["lo"] = H.oracle_select(ft, rpn_thr - MARGIN)
    ft["hi"] = H.oracle_select(ft, rpn_thr + MARGIN)
    plan = InferencePlan(sd, batch_size=B, num_class=nc, anchors=an, anchors_bv=bv, device=dev, cap_k=4096,
                         cap_d=1024)                          # default thresholds = the configured ones
    assert (plan.rpn_thr, plan.score_thr, plan.iou_thr) == (0.1, 0.3, 0.1)
    plan.run_from_points([torch.from_numpy(p).to(dev) for p in clouds])
    res = plan.results()
    assert int(plan.status.item()) == 0
    stats = dict(guided=0, borderline_guided=0, borderline_scores=0, det_err=0.0)
    ndet = sum(_check_at_configured_thresholds(cfgfile, plan, res, ft, b, rpn_thr, score_thr, iou_thr, stats)
               for b in range(B))
    print("configured thresholds (%s, batch %d): %d detections, %s" % (cfgfile, B, ndet, stats))
    assert ndet >= 1 and stats["guided"] >= 50 * B
