"""CPU: the two arbiters of the training-step parity tests (oracle/train_ref.py, round 5) and the floors they define.

* RoundedConv2d rounds exactly the operands the HIP step rounds (its rule equals the product's dispatch, checked against
  libsassd's own sassd_conv2d_bf16_supported), and its three products are the textbook ones;
* the float64 step and the fp32 step are the same function (tiny workload: 1e-5);
* from the stored four-variant golden of the bench's training workload (tests/golden/train_k21_ref.npz): the fp32 oracle sits
  1.4e-3 (worst tensor 6.4e-3) from its float64 arbiter -- the floor the fp32 GPU bars are multiples of -- and the two CPU
  evaluations of the ROUNDED step (fp32 / float64 accumulation, identical rounding rule) are two orders of magnitude
  farther apart, about as far as the rounded step is from the unrounded one: bf16 rounding is discontinuous, so a whole-step
  comparison of a bf16 step against a rounded-operand oracle cannot be tight for any implementation.  That is why the bf16
  step is pinned launch by launch (tests/test_gpu_train.py::test_bf16_step_launches_vs_rounded_reference)."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import train_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_k21_ref.npz")


def test_rounding_rule_equals_the_products_dispatch():
    from sassd import _C
    L = _C.lib()
    for cin, cout, ks, h, w in [(320, 256, 3, 200, 176), (256, 256, 3, 200, 176), (256, 256, 1, 200, 176), (256, 28, 3, 200, 176),
                                (28, 28, 1, 200, 176), (256, 20, 1, 200, 176), (256, 256, 3, 188, 188), (64, 64, 3, 30, 14),
                                (64, 64, 3, 30, 18), (40, 96, 3, 16, 20), (256, 72, 1, 200, 88), (40, 256, 1, 8, 16),
                                (20, 256, 1, 200, 176), (300, 64, 1, 8, 16), (64, 300, 1, 8, 16), (32, 32, 1, 3, 6)]:
        rf, rd, rw = train_ref.bf16_conv_rule(cin, cout, ks, w, h)
        if ks == 1:                                      # (round 6) the 1x1 layers: sassd_conv1x1_bf16_supported both ways
            assert rf == bool(L.sassd_conv1x1_bf16_supported(cin, cout, h * w)), (cin, cout, ks, h, w)
            assert rd == bool(L.sassd_conv1x1_bf16_supported(cout, cin, h * w)), (cin, cout, ks, h, w)
            assert rw == (w % 2 == 0)
            continue
        assert rf == (ks == 3 and bool(L.sassd_conv2d_bf16_supported(cin, (cout + 31) // 32 * 32, h, w))), (cin, cout, ks, w)  # bf16_cout_pad
        assert rd == (ks == 3 and bool(L.sassd_conv2d_bf16_supported(cout, cin, h, w))), (cin, cout, ks, w)   # Conv2dFn.backward
        assert rw == (w % 2 == 0)                                                                               # kernels.conv2d_bwd_weight


def test_rounded_conv_products():
    g = torch.Generator().manual_seed(3)
    rb = train_ref.round_bf16
    for cin, cout, ks, w_ in [(32, 32, 3, 16), (32, 20, 3, 16), (20, 32, 3, 16), (32, 32, 1, 16), (32, 32, 3, 18)]:
        x = torch.randn(2, cin, 6, w_, generator=g, dtype=torch.float64, requires_grad=True)
        w = torch.randn(cout, cin, ks, ks, generator=g, dtype=torch.float64, requires_grad=True)
        b = torch.randn(cout, generator=g, dtype=torch.float64, requires_grad=True)
        dy = torch.randn(2, cout, 6, w_, generator=g, dtype=torch.float64)
        rf, rd, rw = train_ref.bf16_conv_rule(cin, cout, ks, w_, 6)
        y = train_ref.conv2d(x, w, b, ks // 2, True)
        y.backward(dy)
        xd, wd = x.detach(), w.detach()
        assert torch.equal(y.detach(), F.conv2d(rb(xd) if rf else xd, rb(wd) if rf else wd, b.detach(), 1, ks // 2))
        # data gradient = correlation of (rounded) dy with the (rounded) flipped, transposed weights
        wq, dq = (rb(wd), rb(dy)) if rd else (wd, dy)
        ref_dx = F.conv2d(dq, wq.transpose(0, 1).flip(2, 3), None, 1, ks // 2)
        assert (x.grad - ref_dx).abs().max() < 1e-12 * max(1.0, float(ref_dx.abs().max()))
        xq, dq = (rb(xd), rb(dy)) if rw else (xd, dy)
        xp = F.pad(xq, (ks // 2,) * 4)
        ref_dw = torch.stack([torch.stack([torch.einsum("bohw,bihw->oi", dq, xp[:, :, i:i + 6, j:j + w_]) for j in range(ks)], -1)
                              for i in range(ks)], -2)
        assert (w.grad - ref_dw).abs().max() < 1e-12 * max(1.0, float(ref_dw.abs().max()))
        assert (b.grad - dy.sum((0, 2, 3))).abs().max() < 1e-12            # the bias gradient sums the UNROUNDED dy
        # rounding really happens where the rule says so, and only there
        assert (rb(xd) != xd).any()
        y0 = F.conv2d(xd, wd, b.detach(), 1, ks // 2)
        assert torch.equal(y.detach(), y0) == (not rf)


def _stored(G, tag):
    out = {}
    for k in G.files:
        if k.startswith(tag + "grad:") or k.startswith(tag + "grad8:"):
            out[k[len(tag):]] = torch.from_numpy(G[k]).double()
    return out


def _dist(a, b):
    ks = [k for k in b if float(b[k].norm()) > 1e-7]
    num = sum(float((a[k] - b[k]).pow(2).sum()) for k in ks)
    den = sum(float(b[k].pow(2).sum()) for k in ks)
    per = {k: float((a[k] - b[k]).norm() / b[k].norm()) for k in ks}
    return (num / den) ** 0.5, max(per.values()), max(per, key=per.get), len(ks)


def test_k21_golden_floors():
    G = np.load(GOLD)
    f32, f64, b32, b64 = (_stored(G, t) for t in ("", "f64/", "b32/", "b64/"))
    assert set(f32) == set(f64) == set(b32) == set(b64) and len(f32) >= 60
    d32, w32, k32, n = _dist(f32, f64)
    db, wb, kb, _ = _dist(b32, b64)
    dr, wr, kr, _ = _dist(b64, f64)
    print("K21 x 2 training step, stored layers (%d tensors), relative L2 -- fp32 oracle vs float64: together %.2e, worst %.2e "
          "(%s) | rounded fp32 vs rounded float64: %.2e, worst %.2e (%s) | rounded float64 vs float64: %.2e, worst %.2e (%s)"
          % (n, d32, w32, k32, db, wb, kb, dr, wr, kr))
    # the fp32 floor: what "no farther from float64 than the fp32 oracle" means on this workload
    assert 1e-4 < d32 < 3e-3 and 1e-3 < w32 < 1.5e-2, (d32, w32)
    # the rounded step: two evaluations that round identically are >= 30 x farther apart than the fp32 pair ...
    assert db > 30 * d32, (db, d32)
    # ... and about as far apart as rounding itself moves the step (no "signal" a whole-step bar could hold on to)
    assert db > 0.3 * dr, (db, dr)
    # every parameter has its distance to the arbiter stored (the norm / projection bars of the GPU test)
    for tag in ("", "b32/"):
        assert len(G[tag + "grad_dist"]) == len(G[tag + "grad_names"]) >= 75
        assert (G[tag + "grad_dist"] >= 0).all()
    # all four variants differentiate one candidate set, and it is the fp32 oracle's own selection at the stored threshold
    assert len(G["sel0"]) + len(G["sel1"]) + 16 == int(G["n_guided"])
    ln = [str(k) for k in G["loss_names"]]
    assert len(ln) == 6 and all(np.isfinite(G[t + "losses"]).all() for t in ("", "f64/", "b32/", "b64/"))
    rel = np.abs(G["losses"] - G["f64/losses"]) / np.abs(G["f64/losses"])
    assert rel.max() < 1e-5, dict(zip(ln, rel))


def test_float64_step_is_the_same_function():
    """one small cloud, quarter-size grid: the float64 step reproduces the fp32 step to fp32 accuracy (same discrete decisions:
    targets, labels, 3-NN, candidate set), the rounded step moves it by the bf16 operand rounding, not by orders more"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_gpu_train as T
    Q = dict(T.HALF, sparse_shape=(40, 1600, 352), grid_xyz=(352, 1600, 40), pc_range=[0, -40., -3., 17.6, 40., 1.], bev_w=44,
             xmax=16.0)
    case = T.oracle_case("configs/car_cfg.py", ["Car"], Q)
    l32, g32, ex = train_ref.train_step(*case["ref_args"])
    l64, g64, ex64 = train_ref.train_step(*case["ref_args"], guided_sel=ex["guided_sel"], dtype=torch.float64)
    lb, gb, _ = train_ref.train_step(*case["ref_args"], guided_sel=ex["guided_sel"], bf16=("bev",))
    for b in range(2):
        assert np.array_equal(ex["guided_free"][b], ex64["guided_free"][b])
    assert torch.equal(ex["labels"], ex64["labels"]) and torch.equal(ex["ext_labels"], ex64["ext_labels"])
    for k in l32:
        assert abs(l32[k] - l64[k]) <= 1e-5 * max(1.0, abs(l64[k])), (k, l32[k], l64[k])
        assert abs(lb[k] - l64[k]) <= 3e-2 * max(1.0, abs(l64[k])), (k, lb[k], l64[k])
    ks = [k for k, v in g64.items() if v is not None and float(v.norm()) > 1e-7]
    assert len(ks) >= 75 and all(g64[k].dtype == torch.float64 and g32[k].dtype == torch.float32 for k in ks)
    num = sum(float((g32[k].double() - g64[k]).pow(2).sum()) for k in ks)
    numb = sum(float((gb[k].double() - g64[k]).pow(2).sum()) for k in ks)
    den = sum(float(g64[k].pow(2).sum()) for k in ks)
    print("quarter grid: fp32 vs float64 %.2e, rounded fp32 vs float64 %.2e" % ((num / den) ** 0.5, (numb / den) ** 0.5))
    assert (num / den) ** 0.5 < 1e-4
    assert 1e-4 < (numb / den) ** 0.5 < 0.3
