"""-m gpu parity of the training-side point operators against the C oracle."""
import numpy as np
import pytest
import torch

import sassd
from sassd import kernels as K, pointnet2_utils as P2, points_ops
from oracle import clib

pytestmark = pytest.mark.gpu


def _cloud(rng, n, nb):
    b = rng.integers(0, nb, n).astype(np.float32)
    xyz = rng.uniform([0, -40, -3], [70, 40, 1], (n, 3)).astype(np.float32)
    return np.concatenate([b[:, None], xyz], 1)


@pytest.mark.parametrize("n,m,nb", [(1, 5, 1), (700, 3, 2), (5000, 4000, 2), (3000, 2500, 3)])
def test_three_nn(dev, n, m, nb):
    rng = np.random.default_rng(n + m)
    u, k = _cloud(rng, n, nb), _cloud(rng, m, nb)
    k[: min(m, 50)] = u[: min(m, 50)] if n >= 50 else k[: min(m, 50)]     # exact hits -> distance 0 and ties
    d_ref, i_ref = clib.three_nn(u, k)
    d, i = K.three_nn(torch.from_numpy(u).to(dev), torch.from_numpy(k).to(dev))
    assert np.array_equal(i.cpu().numpy(), i_ref)
    assert np.array_equal(d.cpu().numpy(), d_ref)


def test_three_interpolate_and_grad(dev):
    rng = np.random.default_rng(3)
    n, m, c = 4000, 1500, 64
    feats = rng.standard_normal((m, c)).astype(np.float32)
    idx = rng.integers(0, m, (n, 3)).astype(np.int32)
    w = rng.random((n, 3)).astype(np.float32)
    w /= w.sum(1, keepdims=True)
    out = K.three_interpolate(torch.from_numpy(feats).to(dev), torch.from_numpy(idx).to(dev), torch.from_numpy(w).to(dev))
    assert np.abs(out.cpu().numpy() - clib.three_interpolate(feats, idx, w)).max() < 1e-6
    g = rng.standard_normal((n, c)).astype(np.float32)
    gp = K.three_interpolate_grad(torch.from_numpy(g).to(dev), torch.from_numpy(idx).to(dev), torch.from_numpy(w).to(dev), m)
    ref = clib.three_interpolate_grad(g, idx, w, m)
    assert np.abs(gp.cpu().numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    # autograd wrappers (pointnet2_utils.py:9-86 surface)
    f = torch.from_numpy(feats).to(dev).requires_grad_(True)
    unknown, known = _cloud(rng, 300, 2), _cloud(rng, m, 2)
    y = P2.nearest_neighbor_interpolate(torch.from_numpy(unknown).to(dev), torch.from_numpy(known).to(dev), f)
    y.sum().backward()
    assert f.grad is not None and f.grad.shape == f.shape and torch.isfinite(f.grad).all()
    assert abs(float(f.grad.sum()) - 300 * c) < 1e-1            # the three weights of every point sum to one


def test_pts_in_boxes3d(dev):
    rng = np.random.default_rng(5)
    n, m = 6000, 12
    pts = rng.uniform([0, -20, -3], [40, 20, 1], (n, 3)).astype(np.float32)
    boxes = np.stack([rng.uniform(5, 35, m), rng.uniform(-15, 15, m), rng.uniform(-2.5, -1, m), rng.uniform(1.4, 2.2, m),
                      rng.uniform(3, 5, m), rng.uniform(1.3, 1.8, m), rng.uniform(-3.2, 3.2, m)], 1).astype(np.float32)
    pts[:m] = boxes[:, :3] + np.array([0, 0, 0.3], np.float32)      # guaranteed hits
    f_ref, r_ref = clib.pts_in_boxes3d(pts, boxes)
    f, r = K.pts_in_boxes3d(torch.from_numpy(pts).to(dev), torch.from_numpy(boxes).to(dev))
    fg, rg = f.cpu().numpy(), r.cpu().numpy()
    diff = fg != f_ref
    assert diff.sum() <= 2, diff.sum()            # libm vs device sin/cos may flip a point lying exactly on a face
    same = ~diff.any(0)
    assert f_ref.sum() > 20
    assert np.abs(rg[same] - r_ref[same]).max() < 1e-6
    f2, r2 = points_ops.pts_in_boxes3d(torch.from_numpy(pts), torch.from_numpy(boxes))       # CPU-tensor API
    assert f2.device.type == "cpu" and np.array_equal(f2.numpy(), fg)


def test_three_nn_binned_bit_identical(dev):
    """The binned search returns exactly what the brute-force kernel (and the CPU oracle) returns: real multi-level
    voxel centres for a batch of 2, plus adversarial queries (isolated, outside the scene, empty batch element,
    fewer than three known points) and several bin sizes."""
    from sassd import spconv, synth
    import helpers as H
    shape0, vs0, off = (40, 1600, 1408), np.array([.05, .05, .1], np.float32), (0., -40., -3.)
    coors, pts = [], []
    for b, name in enumerate(("small", "k17")):
        p = H.frame(name, 40 + b)
        v, co, n = clib.points_to_voxel(p, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 20000)
        coors.append(np.concatenate([np.full((len(co), 1), b, np.int32), co], 1))
        pts.append(np.concatenate([np.full((len(co), 1), b, np.float32), clib.voxel_mean(v, n)[:, :3]], 1))
    coors, unknown = np.concatenate(coors), np.concatenate(pts)
    extra = np.array([[0, 69.9, 39.9, 0.9], [1, 0.01, -39.99, -2.99], [0, -5.0, 0.0, 0.0], [1, 35.0, 90.0, 5.0],
                      [0, 35.2, 0.0, -1.0], [1, 12.8, -6.4, -1.4], [2, 1.0, 1.0, 1.0]], np.float32)
    unknown = np.concatenate([unknown, extra]).astype(np.float32)
    idx = torch.from_numpy(coors).to(dev)
    x = spconv.SparseConvTensor(torch.zeros(len(coors), 16, device=dev), idx, shape0, 2)
    down = spconv.SparseConv3d(16, 16, 3, (2, 2, 2), padding=1, bias=False).to(dev)
    u = torch.from_numpy(unknown).to(dev)
    rng = (0., -40., 70.4, 40.)
    with torch.no_grad():
        for level in range(1, 4):
            x = down(x)
            vs = torch.from_numpy(vs0 * (2 ** level)).to(dev)
            ii = x.indices.float()
            known = ii.clone()
            known[:, 1:] = ii[:, 1:].flip(1) * vs + torch.tensor(off, device=dev) + .5 * vs
            known = known.contiguous()
            d_ref, i_ref = K.three_nn(u, known)
            od, oi = clib.three_nn(unknown, known.cpu().numpy())
            assert np.array_equal(oi, i_ref.cpu().numpy()) and np.array_equal(od, d_ref.cpu().numpy())
            for cell in (1.6, 0.4, 7.3, 200.0):
                d_got, i_got = K.three_nn_binned(u, known, rng, cell, 3)
                assert torch.equal(i_got, i_ref), (level, cell, int((i_got != i_ref).sum()))
                assert torch.equal(d_got, d_ref), (level, cell)
    # fewer than three known points in one batch element, an empty batch element, known points outside the range
    known = torch.tensor([[0, 1.0, -39., -2.], [0, 1.4, -39., -2.], [1, 0.2, -39.8, -2.6], [1, 500., 500., 0.],
                          [1, -500., 3., 0.], [1, 20., 3., 0.]], device=dev)
    q = torch.tensor([[0, 1.5, -39., -2.], [1, 0.2, -39.8, -2.6], [2, 1.0, 1.0, 1.0], [1, 499., 499., 0.]],
                     device=dev)
    d_ref, i_ref = K.three_nn(q, known)
    for cell in (1.6, 30.0):
        d_got, i_got = K.three_nn_binned(q, known, rng, cell, 3)
        assert torch.equal(i_got, i_ref) and torch.equal(d_got, d_ref), cell


def test_pts_in_boxes3d_vs_reference_golden(dev):
    """HIP pts_in_boxes3d against output of the reference's own compiled C++ (tests/golden/pts_in_boxes_ref.npz)."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pts_in_boxes_ref.npz"))
    flag, reg = K.pts_in_boxes3d(torch.from_numpy(G["pts"]).to(dev), torch.from_numpy(G["boxes"]).to(dev))
    assert np.array_equal(flag.cpu().numpy(), G["flag"])
    assert np.array_equal(reg.cpu().numpy(), G["reg"])


def test_interpolation_vs_reference_golden(dev):
    """HIP three_nn (brute force and binned) / three_interpolate against outputs of the reference's own kernels
    (tests/golden/interp_ref.npz); the gradient up to fp32 summation order."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "interp_ref.npz"))
    u, k = torch.from_numpy(G["unknown"]).to(dev), torch.from_numpy(G["known"]).to(dev)
    for d, i in (K.three_nn(u, k), K.three_nn_binned(u, k, (0., -40., 70., 40.), 1.6, 3)):
        assert np.array_equal(i.cpu().numpy(), G["idx"]) and np.array_equal(d.cpu().numpy(), G["dist2"])
    idx, w = torch.from_numpy(G["idx"]).to(dev), torch.from_numpy(G["weight"]).to(dev)
    out = K.three_interpolate(torch.from_numpy(G["feat"]).to(dev), idx, w)
    assert np.abs(out.cpu().numpy() - G["out"]).max() < 1e-6          # fused multiply-adds on the device: last ulp
    gp = K.three_interpolate_grad(torch.from_numpy(G["grad_out"]).to(dev), idx, w, len(G["known"]))
    assert np.abs(gp.cpu().numpy() - G["grad_points"]).max() < 1e-5
