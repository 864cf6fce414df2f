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
