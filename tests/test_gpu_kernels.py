"""-m gpu parity tests: every HIP kernel (through the C ABI) against the CPU oracle / golden fixtures."""
import glob
import os

import numpy as np
import pytest
import torch

import sassd
from sassd import kernels as K, synth
from oracle import clib, nets as onets, rulebook as orb
import helpers as H

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_mfma_lane_maps(dev):
    g = torch.Generator().manual_seed(0)
    ks = 5
    a32 = torch.randn(32, 2 * ks, generator=g); b32 = torch.randn(2 * ks, 32, generator=g)
    a16 = torch.randn(16, 4 * ks, generator=g); b16 = torch.randn(4 * ks, 16, generator=g)
    d32, d16 = K.mfma_probe(a32.to(dev), b32.to(dev), a16.to(dev), b16.to(dev), ks)
    assert torch.allclose(d32.cpu(), a32 @ b32, atol=1e-5), (d32.cpu() - a32 @ b32).abs().max()
    assert torch.allclose(d16.cpu(), a16 @ b16, atol=1e-5), (d16.cpu() - a16 @ b16).abs().max()


def _run_vox(dev, pts, vs, cr, t, mv, **kw):
    p = torch.from_numpy(np.ascontiguousarray(pts)).to(dev)
    st = K.new_status(dev)
    r = K.voxelize(p, vs, cr, t, mv, status=st, **kw)
    m = int(r["voxel_num"].item())
    assert int(st.item()) == 0
    return m, r


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "voxelizer_*.npz"))))
def test_voxelizer_golden(dev, path):
    g = np.load(path)
    if "points" not in g:
        pytest.skip("digest file")
    m, r = _run_vox(dev, g["points"], g["voxel_size"], g["coors_range"], int(g["max_points"]), int(g["max_voxels"]))
    assert m == len(g["coors"])
    assert np.array_equal(r["coors"][:m].cpu().numpy(), g["coors"])
    assert np.array_equal(r["num_points"][:m].cpu().numpy(), g["num_points"])
    assert np.array_equal(r["voxels"][:m].cpu().numpy(), g["voxels"])          # bit exact payload
    mean = clib.voxel_mean(g["voxels"], g["num_points"]) if m else np.zeros((0, 4), np.float32)
    assert np.array_equal(r["mean"][:m].cpu().numpy(), mean)


@pytest.mark.parametrize("name,seed", [("k21", 0), ("k17", 0), ("k21", 3)])
def test_voxelizer_full_frame(dev, name, seed):
    pts = H.frame(name, seed)
    v, c, n = clib.points_to_voxel(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 20000)
    m, r = _run_vox(dev, pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 20000, coors_cols=4, batch_idx=3)
    assert m == len(c)
    co = r["coors"][:m].cpu().numpy()
    assert np.all(co[:, 0] == 3) and np.array_equal(co[:, 1:], c)
    assert np.array_equal(r["num_points"][:m].cpu().numpy(), n)
    assert np.array_equal(r["voxels"][:m].cpu().numpy(), v)
    assert np.array_equal(r["mean"][:m].cpu().numpy(), clib.voxel_mean(v, n))
    if name == "k21" and seed == 0:
        assert m == 16111


@pytest.mark.parametrize("T", [1, 5, 8])
@pytest.mark.parametrize("nfeat,cols", [(4, 4), (3, 3), (4, 3)])
def test_voxelizer_emit_kernels_agree(dev, T, nfeat, cols):
    """the level-by-level emit kernel (vox_emit8_kernel: ndim 4, max_points <= 8, 16-byte aligned points) against the
    point-by-point one (vox_emit_kernel, the generic path) on the SAME cloud -- the generic path is forced by handing the points
    over 4 bytes off a 16-byte boundary -- voxels, coordinates, point counts, means and the voxel count bit for bit, for
    max_points 1 / 5 / 8, 3 and 4 mean features, 3 and 4 coordinate columns, with and without the max_voxels break."""
    pts = H.frame("k21", 3)
    n = len(pts)
    flat = torch.zeros(n * 4 + 8, device=dev)
    assert flat.data_ptr() % 16 == 0
    aligned = flat[4:4 + 4 * n].view(n, 4)
    aligned.copy_(torch.from_numpy(pts))
    off = torch.zeros(n * 4 + 8, device=dev)[1:1 + 4 * n].view(n, 4)
    off.copy_(torch.from_numpy(pts))
    assert aligned.data_ptr() % 16 == 0 and off.data_ptr() % 16 == 4
    for max_voxels in (20000, 3000):
        res = []
        for p in (aligned, off):
            st = K.new_status(dev)
            r = K.voxelize(p, synth.KITTI_VOXEL, synth.KITTI_RANGE, T, max_voxels, batch_idx=2, coors_cols=cols,
                           want_voxels=True, want_mean=True, nfeat=nfeat, status=st)
            torch.cuda.synchronize()
            m = int(r["voxel_num"].item())
            res.append((m, r["voxels"][:m].clone(), r["coors"][:m].clone(), r["num_points"][:m].clone(), r["mean"][:m].clone()))
        a, b = res
        assert a[0] == b[0] and a[0] > 0 and (max_voxels != 3000 or a[0] == 3000)
        for i, name in ((1, "voxels"), (2, "coors"), (3, "num_points"), (4, "mean")):
            assert torch.equal(a[i], b[i]), (name, T, nfeat, cols, max_voxels)


def test_voxelizer_waymo_scale_break(dev):
    pts = synth.waymo_synth(0)[:180000]
    v, c, n = clib.points_to_voxel(pts, synth.WAYMO_VOXEL, synth.WAYMO_RANGE, 5, True, 60000)   # break triggers
    m, r = _run_vox(dev, pts, synth.WAYMO_VOXEL, synth.WAYMO_RANGE, 5, 60000)
    assert m == 60000 == len(c)
    assert np.array_equal(r["coors"][:m].cpu().numpy(), c)
    assert np.array_equal(r["num_points"][:m].cpu().numpy(), n)
    assert np.array_equal(r["voxels"][:m].cpu().numpy(), v)


def _pairs_set(pairs, num, k):
    return set(zip(pairs[k, 0, :num[k]].tolist(), pairs[k, 1, :num[k]].tolist()))


def _level0(dev, name="k21", seed=0, batch=1):
    idx = []
    for b in range(batch):
        _, c, _ = clib.points_to_voxel(H.frame(name, seed + b), synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 20000)
        idx.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))
    return np.concatenate(idx, 0)


@pytest.mark.parametrize("batch", [1, 2])
def test_rulebooks(dev, batch):
    idx = _level0(dev, "k21", 0, batch)
    shape = (40, 1600, 1408)
    for level in range(3):
        n = len(idx)
        cap = n + 100
        di = torch.zeros(cap, 4, dtype=torch.int32, device=dev)
        di[:n] = torch.from_numpy(idx).to(dev)
        nptr = torch.tensor([n], dtype=torch.int32, device=dev)
        st = K.new_status(dev)
        tab = K.HashTable(cap, dev).build(di, nptr, shape, batch, st)
        nbr = K.rulebook_subm(di, nptr, cap, shape, batch, tab)
        _, onbr = orb.subm_rulebook(idx, shape)
        assert np.array_equal(nbr[:n].cpu().numpy(), onbr), "subm level %d" % level
        # spconv-format pairs derived on device
        pairs, num = K.rulebook_pairs(nbr, nptr, cap)
        op, onum = orb.nbr_to_pairs(onbr)
        assert np.array_equal(num.cpu().numpy(), onum)
        pc = pairs.cpu().numpy()
        for k in (0, 13, 26):
            assert _pairs_set(pc, onum, k) == set(zip(op[k][0].tolist(), op[k][1].tolist()))
        # strided conv
        cap_out = 2 * n + 64
        oi, on, onb = K.rulebook_conv(di, nptr, cap, shape, batch, tab, cap_out, status=st)
        ref_oi, ref_nbr, oshape = orb.conv_rulebook(idx, shape, batch)
        m = int(on.item())
        assert int(st.item()) == 0
        assert m == len(ref_oi), "down level %d" % level
        assert np.array_equal(oi[:m].cpu().numpy(), ref_oi)
        assert np.array_equal(onb[:m].cpu().numpy(), ref_nbr)
        idx, shape = ref_oi, oshape
    assert shape == (5, 200, 176)


@pytest.mark.parametrize("cin,cout", [(4, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64)])
def test_spconv_layer(dev, cin, cout):
    idx = _level0(dev, "small", 1)
    shape = (40, 1600, 1408)
    # use a coarser level so neighbourhoods are dense enough to exercise multi-chunk offsets
    idx, nbr1, shape = orb.conv_rulebook(idx, shape, 1)
    idx, nbr2, shape = orb.conv_rulebook(idx, shape, 1)
    _, nbr = orb.subm_rulebook(idx, shape)
    n = len(idx)
    g = torch.Generator().manual_seed(cin * 100 + cout)
    x = torch.randn(n, cin, generator=g)
    w = torch.randn(27, cin, cout, generator=g) * 0.2
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ref = torch.relu(onets.sparse_conv(x, nbr, w) * scale + shift)
    cap = n + 37
    nb = torch.full((cap, 27), -1, dtype=torch.int32, device=dev)
    nb[:n] = torch.from_numpy(nbr).to(dev)
    nptr = torch.tensor([n], dtype=torch.int32, device=dev)
    wp = K.spconv_pack_weight(w.to(dev))
    y = K.spconv_fwd(x.to(dev), nb, nptr, cap, wp, 27, cin, cout, scale.to(dev), shift.to(dev), True)
    err = (y[:n].cpu() - ref).abs().max().item()
    assert err < 2e-4 * max(1.0, ref.abs().max().item()), err
    # raw conv (no epilogue) and the 1x1x1 identity path
    y2 = K.spconv_fwd(x.to(dev), nb, nptr, cap, wp, 27, cin, cout)
    assert (y2[:n].cpu() - onets.sparse_conv(x, nbr, w)).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())
    w1 = torch.randn(1, cin, cout, generator=g) * 0.2
    y3 = K.spconv_fwd(x.to(dev), None, nptr, cap, K.spconv_pack_weight(w1.to(dev)), 1, cin, cout)
    assert (y3[:n].cpu() - x @ w1[0]).abs().max().item() < 1e-4


def test_densify(dev):
    g = torch.Generator().manual_seed(0)
    n, c, shape, b = 500, 64, (5, 200, 176), 2
    lin = torch.randperm(b * 5 * 200 * 176, generator=g)[:n].sort()[0]
    idx = torch.stack([lin // (5 * 200 * 176), (lin // (200 * 176)) % 5, (lin // 176) % 200, lin % 176], 1).int()
    f = torch.randn(n, c, generator=g)
    ref = onets.densify(f, idx.numpy(), shape, b)
    nptr = torch.tensor([n], dtype=torch.int32, device=dev)
    out = K.densify(f.to(dev), idx.to(dev), nptr, n, shape, b, 0)
    assert torch.equal(out.cpu(), ref)
    out1 = K.densify(f.to(dev), idx.to(dev), nptr, n, shape, b, 1).cpu()
    ref1 = ref.view(b, c, 5, 200, 176).permute(0, 2, 1, 3, 4).reshape(b, c * 5, 200, 176)
    assert torch.equal(out1, ref1)


@pytest.mark.parametrize("cin,cout,ks,h,w,b", [(16, 256, 3, 20, 176, 1), (320, 256, 3, 11, 176, 2), (256, 256, 1, 9, 176, 1),
                                               (256, 28, 3, 13, 176, 1), (256, 20, 1, 7, 50, 2), (8, 128, 3, 200, 176, 1),
                                               (24, 72, 1, 5, 31, 1), (28, 28, 1, 9, 176, 1), (5, 40, 3, 6, 17, 1)])
def test_conv2d(dev, cin, cout, ks, h, w, b):
    g = torch.Generator().manual_seed(cin + cout + ks)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, ks, ks, generator=g) * (1.0 / (cin * ks * ks)) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ref = torch.relu(torch.nn.functional.conv2d(x, wt, None, 1, ks // 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    wp = K.conv2d_pack_weight(wt.to(dev))
    y = K.conv2d_fwd(x.to(dev), wp, cout, ks, scale.to(dev), shift.to(dev), True).cpu()
    err = (y - ref).abs().max().item()
    assert err < 1e-4, err
    y2 = K.conv2d_fwd(x.to(dev), wp, cout, ks).cpu()
    assert (y2 - torch.nn.functional.conv2d(x, wt, None, 1, ks // 2)).abs().max().item() < 1e-4


@pytest.mark.parametrize("cin,cout,h,w,b", [(256, 20, 200, 176, 1), (28, 28, 200, 176, 2), (256, 20, 7, 50, 2), (30, 1, 5, 31, 1),
                                            (7, 32, 3, 70, 3), (64, 9, 188, 188, 1)])
def test_conv1x1_narrow(dev, cin, cout, h, w, b):
    """1x1 convolution with <= 32 output channels on the streaming vector-ALU kernel (fused SSD head 256 -> 20, part-sensitive
    28 -> 28; ragged pixel counts, Cin not a multiple of the four waves' split, every padded channel count) against float64
    conv2d at the fp32 bar of the MFMA kernel it replaces."""
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) * (1.0 / cin) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    assert K.conv1x1_narrow_supported(cin, cout) and not K.conv1x1_narrow_supported(cin, 33)
    raw = torch.nn.functional.conv2d(x.double(), wt.double())
    ref = torch.relu(raw * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    wp = K.conv1x1_narrow_pack_weight(wt.to(dev))
    y = K.conv1x1_narrow_fwd(x.to(dev), wp, cout, scale.to(dev), shift.to(dev), True).cpu().double()
    assert (y - ref).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item())
    y2 = torch.full((b, cout, h, w), 7.0, device=dev)
    K.conv1x1_narrow_fwd(x.to(dev), wp, cout, None, None, False, y2)
    assert (y2.cpu().double() - raw).abs().max().item() < 2e-6 * max(1.0, raw.abs().max().item())
    assert torch.equal(y2, K.conv1x1_narrow_fwd(x.to(dev), wp, cout))              # a fixed summation order


def test_iou_matrices_and_nms(dev):
    g = np.load(os.path.join(GOLD, "iou_ref.npz"))
    a, b = torch.from_numpy(g["a"]).to(dev), torch.from_numpy(g["b"]).to(dev)
    ov = K.boxes_overlap_bev(a, b).cpu().numpy()
    iou = K.boxes_iou_bev(a, b).cpu().numpy()
    assert np.abs(ov - g["overlap"]).max() < 1e-4, np.abs(ov - g["overlap"]).max()
    assert np.abs(iou - g["iou"]).max() < 1e-5, np.abs(iou - g["iou"]).max()
    rng = np.random.default_rng(1)
    for n in (1, 63, 64, 65, 200, 700):
        boxes = H.rand_bev_boxes(rng, n, spread=25.0)
        keep_ref, mask_ref = clib.nms_rotated(boxes, 0.1, return_mask=True)
        keep, num = K.nms_gpu(torch.from_numpy(boxes).to(dev), 0.1)
        k = int(num.item())
        got = keep[:k].cpu().numpy()
        if not np.array_equal(got, keep_ref):
            # A keep list other than the oracle's is accepted only as a VALID greedy outcome under the 1e-5 IoU tolerance,
            # decision by decision: a kept box overlaps no earlier kept box by more than thr + 1e-5, a suppressed box
            # overlaps one by at least thr - 1e-5 -- so only pairs within 1e-5 of the threshold can have gone the other
            # way, and the first differing index must be decided by such a pair.
            iou_m = clib.boxes_iou_bev(boxes, boxes)
            kept, gs = [], set(got.tolist())
            for i in range(n):
                ov_i = iou_m[kept, i] if kept else np.zeros(0)
                if i in gs:
                    assert ov_i.size == 0 or ov_i.max() <= 0.1 + 1e-5, (n, i, float(ov_i.max()))
                    kept.append(i)
                else:
                    assert ov_i.size and ov_i.max() >= 0.1 - 1e-5, (n, i, float(ov_i.max()) if ov_i.size else None)
            first = next(i for i in range(n) if (i in gs) != (i in set(keep_ref.tolist())))
            pre = [j for j in got.tolist() if j < first]
            assert np.any(np.abs(iou_m[pre, first] - 0.1) < 1e-5), (n, first)


def test_pswarp_extreme_boxes_do_not_fault(dev):
    """Random-weight heads decode boxes with lengths up to 1e18 m: samples far outside the map must read as zero
    padding (grid_sample semantics), not fault (regression: signed-overflow UB in the bounds test)."""
    feat = torch.randn(1, 28, 200, 176, device=dev)
    for big in (1e2, 1e10, 1e18, 3e38, float("inf"), float("nan")):
        g = torch.zeros(1, 64, 7, device=dev)
        g[0, :, 0] = torch.linspace(-10, 80, 64); g[0, :, 1] = torch.linspace(-50, 50, 64)
        g[0, :, 3] = 1.6; g[0, :, 4] = big; g[0, :, 5] = 1.5; g[0, :, 6] = torch.linspace(-3, 50, 64)
        cnt = torch.tensor([64], dtype=torch.int32, device=dev)
        lg = K.pswarp_sample(feat, g, cnt, 64, (0., 40.), 2.5)
        torch.cuda.synchronize()
        if big > 1e6:
            assert torch.isfinite(lg[0, :64]).all() or big != big


@pytest.mark.parametrize("cin,cout,strided", [(4, 16, False), (16, 16, False), (16, 32, True), (32, 32, False),
                                              (32, 64, True), (64, 64, False), (64, 64, True)])
def test_spconv_backward(dev, cin, cout, strided):
    """a15: data and weight gradients of the sparse conv vs torch autograd through the oracle's gather/mm/index_add."""
    idx = _level0(dev, "small", 3)
    shape = (40, 1600, 1408)
    idx, _, shape = orb.conv_rulebook(idx, shape, 1)
    idx, _, shape = orb.conv_rulebook(idx, shape, 1)
    if strided:
        out_idx, nbr, _ = orb.conv_rulebook(idx, shape, 1)
    else:
        out_idx, nbr = orb.subm_rulebook(idx, shape)
    nin, nout = len(idx), len(out_idx)
    g = torch.Generator().manual_seed(cin + 7 * cout)
    x = torch.randn(nin, cin, generator=g, requires_grad=True)
    w = (torch.randn(27, cin, cout, generator=g) * 0.2).requires_grad_(True)
    dy = torch.randn(nout, cout, generator=g)
    onets.sparse_conv(x, nbr, w).backward(dy)
    cap_o, cap_i = nout + 11, nin + 5
    nb = torch.full((cap_o, 27), -1, dtype=torch.int32, device=dev)
    nb[:nout] = torch.from_numpy(nbr).to(dev)
    n_o = torch.tensor([nout], dtype=torch.int32, device=dev)
    n_i = torch.tensor([nin], dtype=torch.int32, device=dev)
    dyd = torch.zeros(cap_o, cout, device=dev); dyd[:nout] = dy.to(dev)
    xd = torch.zeros(cap_i, cin, device=dev); xd[:nin] = x.detach().to(dev)
    dw = K.spconv_bwd_weight(xd, dyd, nb, n_o, cap_o, cin, cout)
    e = (dw.cpu() - w.grad).abs().max().item()
    assert e < 2e-4 * max(1.0, w.grad.abs().max().item()), e
    dw2 = K.spconv_bwd_weight(xd, dyd, nb, n_o, cap_o, cin, cout, dw=dw.clone(), accumulate=True)
    assert torch.allclose(dw2, 2 * dw, rtol=1e-5, atol=1e-4 * max(1.0, dw.abs().max().item()))
    if cin >= 16:                      # the first layer's input needs no gradient (and Cout'=4 is not a kernel shape)
        nbT = K.rulebook_transpose(nb, n_o, cap_o, cap_i)
        ref_t = np.full((nin, 27), -1, np.int32)
        oo, kk = np.nonzero(nbr >= 0)
        ref_t[nbr[oo, kk], kk] = oo
        assert np.array_equal(nbT[:nin].cpu().numpy(), ref_t)
        dx = K.spconv_bwd_data(dyd, nbT, n_i, cap_i, K.spconv_pack_weight_t(w.detach().to(dev)), 27, cin, cout)
        e = (dx[:nin].cpu() - x.grad).abs().max().item()
        assert e < 2e-4 * max(1.0, x.grad.abs().max().item()), e


@pytest.mark.parametrize("b,cin,cout,hw,relu", [(1, 32, 32, (4, 64), False), (2, 64, 64, (10, 72), True),
                                               (1, 320, 256, (200, 176), True), (2, 256, 256, (50, 88), False),
                                               (1, 96, 96, (2, 64), True), (3, 64, 160, (6, 132), True)])
def test_conv2d_winograd(dev, b, cin, cout, hw, relu):
    """Winograd F(2x2,3x3) conv against torch-CPU conv2d (fp32 reference of the same op) and against the direct HIP
    kernel: odd tile counts, tiles wrapping rows / images inside a 32-tile group, cout not a multiple of 64, borders."""
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(b, cin, *hw, generator=g)
    x[:, :, : hw[0] // 2] *= (torch.rand(b, cin, hw[0] // 2, hw[1], generator=g) > 0.7).float()    # sparse like BEV
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ref = torch.nn.functional.conv2d(x, w, None, 1, 1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    ref = torch.relu(ref) if relu else ref
    assert K.conv2d_wino_supported(cin, cout, *hw)
    xd, wd = x.to(dev), w.to(dev)
    y = K.conv2d_wino_fwd(xd, K.conv2d_wino_pack_weight(wd), cout, sc.to(dev), sh.to(dev), relu)
    err = (y.cpu() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err
    if hw[1] >= 4:
        yd = K.conv2d_fwd(xd, K.conv2d_pack_weight(wd), cout, 3, sc.to(dev), sh.to(dev), relu)
        assert (y - yd).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    y2 = K.conv2d_wino_fwd(xd, K.conv2d_wino_pack_weight(wd), cout)              # no epilogue
    assert (y2.cpu() - torch.nn.functional.conv2d(x, w, None, 1, 1)).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    assert not K.conv2d_wino_supported(28, 28, 200, 176) and not K.conv2d_wino_supported(256, 256, 199, 176)
    assert not K.conv2d_wino_supported(256, 256, 200, 44) and not K.conv2d_wino_supported(256, 256, 200, 66)
    assert not K.conv2d_wino_supported(48, 256, 200, 176)


def test_rotate_iou_eval(dev):
    """KITTI-eval rotated IoU kernel vs the C oracle (same algorithm) and vs golden values produced by the reference's
    own numba device functions; numpy-level mirror of rotate_iou_gpu_eval included."""
    import os
    from oracle import clib
    from sassd import eval_ops
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval_iou_ref.npz"))
    b, q = torch.from_numpy(G["boxes"]).to(dev), torch.from_numpy(G["query"]).to(dev)
    for crit in (-1, 0, 1, 2):
        got = K.rotate_iou_eval(b, q, crit).cpu().numpy()
        assert np.abs(got - clib.rotate_iou_eval(G["boxes"], G["query"], crit)).max() < 1e-6, crit
        if crit < 2:
            assert np.abs(got - G["iou_%d" % crit]).max() < 3e-6, crit
    rng = np.random.default_rng(4)
    big = np.stack([rng.uniform(0, 60, 700), rng.uniform(-30, 30, 700), rng.uniform(1, 3, 700), rng.uniform(2, 6, 700),
                    rng.uniform(-4, 4, 700)], 1).astype(np.float64)
    out = eval_ops.rotate_iou_gpu_eval(big, big[:300], -1)
    assert out.dtype == np.float64 and out.shape == (700, 300)
    ref = clib.rotate_iou_eval(big.astype(np.float32), big[:300].astype(np.float32), -1)
    assert np.abs(out - ref).max() < 1e-6 and (out >= 0).all()
    assert eval_ops.rotate_iou_gpu_eval(big[:0], big, 0).shape == (0, 700)


def test_voxelizer_reference_defaults(dev):
    """points_to_voxel with the reference's own defaults: max_points=35 (points_ops.py:104-109) and the xyz-order variant
    reverse_index=False (:53-101), bit-exact against the C oracle on a cloud dense enough to fill 35-point voxels."""
    from sassd import points_ops
    rng = np.random.default_rng(5)
    pts = np.concatenate([H.frame("small", 7), rng.uniform([10, -2, -1.5, 0], [11, 2, -1.0, 1], (6000, 4)).astype(np.float32)])
    vs, cr = [0.2, 0.2, 0.4], list(synth.KITTI_RANGE)
    for rev in (True, False):
        v, c, n = points_ops.points_to_voxel(pts, vs, cr, max_points=35, reverse_index=rev, max_voxels=20000)
        rv, rc, rn = clib.points_to_voxel(pts, vs, cr, 35, True, 20000)
        assert n.max() == 35 and len(v) == len(rv)
        assert np.array_equal(v, rv) and np.array_equal(n, rn)
        assert np.array_equal(c, rc if rev else rc[:, ::-1])


def test_nms_normal_gpu(dev):
    """iou3d_utils.nms_normal_gpu (axis-aligned IoU, rotation ignored: iou3d_kernel.cu:295-348, iou3d.cpp:123-172)
    against a plain numpy greedy loop in fp32."""
    from sassd import iou3d_utils
    rng = np.random.default_rng(3)
    for n in (1, 64, 65, 300):
        b = H.rand_bev_boxes(rng, n, spread=20.0)
        sc = rng.random(n).astype(np.float32)
        order = np.argsort(-sc, kind="stable")
        bs = b[order]
        keep, alive = [], np.ones(n, bool)
        for i in range(n):
            if not alive[i]:
                continue
            keep.append(i)
            a = bs[i]
            w = np.maximum(np.minimum(a[2], bs[:, 2]) - np.maximum(a[0], bs[:, 0]), np.float32(0))
            h = np.maximum(np.minimum(a[3], bs[:, 3]) - np.maximum(a[1], bs[:, 1]), np.float32(0))
            inter = (w * h).astype(np.float32)
            sa = np.float32((a[2] - a[0]) * (a[3] - a[1]))
            sb = ((bs[:, 2] - bs[:, 0]) * (bs[:, 3] - bs[:, 1])).astype(np.float32)
            iou = inter / np.maximum(sa + sb - inter, np.float32(1e-8))
            alive &= ~((iou > np.float32(0.3)) & (np.arange(n) > i))
        got = iou3d_utils.nms_normal_gpu(torch.from_numpy(b).to(dev), torch.from_numpy(sc).to(dev), 0.3).cpu().numpy()
        assert np.array_equal(got, order[np.asarray(keep)]), n


def test_iou3d_utils_nms_gpu_facade(dev):
    """iou3d_utils.nms_gpu (reference :114-128: sort by score, rotated NMS, indices in the caller's order) against the C
    oracle's greedy loop."""
    from sassd import iou3d_utils
    rng = np.random.default_rng(9)
    b = H.rand_bev_boxes(rng, 150, spread=18.0)
    sc = rng.random(150).astype(np.float32)
    order = np.argsort(-sc, kind="stable")
    keep_ref = clib.nms_rotated(b[order], 0.1)
    got = iou3d_utils.nms_gpu(torch.from_numpy(b).to(dev), torch.from_numpy(sc).to(dev), 0.1).cpu().numpy()
    assert np.array_equal(got, order[np.asarray(keep_ref)])


@pytest.mark.parametrize("seed", [0, 3])
def test_anchor_mask_single_and_batch_vs_oracle(dev, seed):
    """f-1: sassd_anchor_mask (per sample, what sassd.train.device_batch / the dataset call) and sassd_anchor_mask_batch
    against the oracle's integral-image formulation (kitti.py:333-343, geometry.py:676-710), bit-exact."""
    from sassd import synth
    w = synth.workload("car")
    bv = torch.from_numpy(w["anchors_bv"]).to(dev)
    H0, W0 = 1600, 1408
    refs, coors_all, offs = [], [], [0]
    for b in range(2):
        pts = synth.k21(seed + b)
        r = K.voxelize(torch.from_numpy(pts).to(dev), synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 20000, batch_idx=0,
                       coors_cols=4, want_mean=False)
        n = int(r["voxel_num"])
        co = r["coors"][:n]
        ref = onets.anchors_mask(co[:, 1:].cpu().numpy(), w["anchors_bv"], synth.KITTI_VOXEL, synth.KITTI_RANGE,
                                 (W0, H0, 40), 1)
        zero = torch.zeros(1, dtype=torch.int32, device=dev)
        got = K.anchor_mask(r["coors"], zero, r["voxel_num"], H0, W0, bv, synth.KITTI_VOXEL, synth.KITTI_RANGE, 1)
        assert 0.1 < ref.mean() < 0.9
        assert np.array_equal(got.cpu().numpy().astype(bool), ref), (int(got.sum()), int(ref.sum()))
        refs.append(ref)
        coors_all.append(co)
        offs.append(offs[-1] + n)
    mask = torch.empty(2, bv.shape[0], dtype=torch.uint8, device=dev)
    K.anchor_mask_batch(torch.cat(coors_all).contiguous(), torch.tensor(offs, dtype=torch.int32, device=dev), 2, H0, W0, bv,
                        synth.KITTI_VOXEL, synth.KITTI_RANGE, 1, mask)
    assert np.array_equal(mask.cpu().numpy().astype(bool), np.stack(refs))
