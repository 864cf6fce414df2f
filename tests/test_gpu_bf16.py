"""-m gpu parity of the bf16 BEV training kernels (BASELINE configs[2]: "training ... bf16").

The bf16 kernels round their MFMA operands to bf16 (round-to-nearest-even) and accumulate in fp32, so the sharp check
is against a float64 convolution of the SAME ROUNDED operands: only the fp32 summation order differs (1e-5 relative).
The loose check against the unrounded fp32 result states the tolerance the bf16 step is held to: 1e-2 relative L2 per
tensor (operand rounding 2^-9 per factor, averaged over the contraction)."""
import pytest
import torch
import torch.nn.functional as F

from sassd import kernels as K
from sassd import autograd as AG

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / max(b.norm().item(), 1e-12))


def _bf(t):
    return t.bfloat16().float()


def _wgrad_ref(x, dy, ks):
    """dW of conv2d(x, w, pad=k/2) for upstream gradient dy, float64 on the CPU."""
    x, dy = x.double(), dy.double()
    w = torch.zeros(dy.shape[1], x.shape[1], ks, ks, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w, None, 1, ks // 2).backward(dy)
    return w.grad


@pytest.mark.parametrize("b,cin,cout,ks,hw", [(2, 320, 256, 3, (50, 88)), (1, 256, 256, 3, (13, 180)),
                                             (2, 72, 136, 1, (9, 44)), (2, 64, 64, 3, (7, 46)),
                                             (1, 128, 256, 3, (24, 176)), (3, 256, 28, 1, (16, 24))])
def test_wgrad_bf16(dev, b, cin, cout, ks, hw):
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(b, cin, *hw, generator=g)
    dy = torch.randn(b, cout, *hw, generator=g)
    got = K.conv2d_bwd_weight(x.to(dev), dy.to(dev), ks, bf16=True)
    sharp = _rel(got, _wgrad_ref(_bf(x), _bf(dy), ks))
    loose = _rel(got, _wgrad_ref(x, dy, ks))
    print("wgrad bf16 %s: vs rounded operands %.2e, vs fp32 operands %.2e" % ((b, cin, cout, ks, hw), sharp, loose))
    assert sharp < 1e-5
    assert loose < 1e-2
    # accumulate=True adds into dw
    again = K.conv2d_bwd_weight(x.to(dev), dy.to(dev), ks, dw=got.clone(), accumulate=True, bf16=True)
    assert _rel(again, 2 * got) < 1e-6


def test_wgrad_bf16_full_bev_layer(dev):
    """The bench shape (B=2, 256->256, 200x176): bf16 kernel against the fp32 split-K kernel on the device."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 256, 200, 176, generator=g).to(dev)
    dy = (torch.randn(2, 256, 200, 176, generator=g) * 0.01).to(dev)
    ref = K.conv2d_bwd_weight(x, dy, 3)
    got = K.conv2d_bwd_weight(x, dy, 3, bf16=True)
    exact = K.conv2d_bwd_weight(_bf(x), _bf(dy), 3)               # fp32 kernel on the rounded operands
    print("full layer: bf16 vs fp32-on-rounded %.2e, vs fp32 %.2e" % (_rel(got, exact), _rel(got, ref)))
    assert _rel(got, exact) < 2e-5
    assert _rel(got, ref) < 1e-2
    again = K.conv2d_bwd_weight(x, dy, 3, bf16=True)
    assert torch.equal(got, again)                                # fixed split order: bit-reproducible


@pytest.mark.parametrize("b,cin,cout,hw", [(2, 256, 256, (24, 32)), (1, 128, 256, (17, 48)), (2, 256, 128, (8, 16)),
                                          (1, 32, 128, (5, 176)), (1, 64, 512, (9, 16)), (2, 28, 256, (12, 32)),
                                          (1, 70, 128, (6, 16)), (2, 256, 320, (16, 32)), (1, 64, 96, (8, 16)),
                                          (1, 64, 256, (11, 188)), (2, 32, 64, (23, 20)), (1, 256, 256, (188, 188))])
def test_conv3x3_bf16(dev, b, cin, cout, hw):
    g = torch.Generator().manual_seed(cin + 3 * cout)
    x = torch.randn(b, cin, *hw, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (cin * 9)) ** 0.5
    bias = torch.randn(cout, generator=g)
    pk = K.conv2d_bf16_pack_weight(w.to(dev))
    got = K.conv2d_bf16_fwd(x.to(dev), pk, cout, bias.to(dev))
    sharp = _rel(got, F.conv2d(_bf(x).double(), _bf(w).double(), bias.double(), 1, 1))
    loose = _rel(got, F.conv2d(x.double(), w.double(), bias.double(), 1, 1))
    print("conv3x3 bf16 %s: vs rounded operands %.2e, vs fp32 operands %.2e" % ((b, cin, cout, hw), sharp, loose))
    assert sharp < 1e-5
    assert loose < 1e-2
    assert _rel(K.conv2d_bf16_fwd(x.to(dev), pk, cout), F.conv2d(_bf(x).double(), _bf(w).double(), None, 1, 1)) < 1e-5


@pytest.mark.parametrize("b,cin,cout,hw", [(2, 64, 320, (37, 52)), (1, 96, 512, (41, 36)), (3, 32, 128, (15, 20)),
                                          (2, 64, 256, (51, 48)), (1, 32, 256, (2, 16)), (2, 40, 256, (1, 64))])
@pytest.mark.parametrize("nwg", [8, 24])
def test_conv3x3_bf16_long_runs(dev, b, cin, cout, hw, nwg):
    """The persistent kernel with few workgroups: every run holds many tiles (1..5 blocks each), crosses strip, image and
    cout-tile boundaries, ends in an odd last row -- same numbers as the reference and as the default launch."""
    from sassd import _C
    g = torch.Generator().manual_seed(cin + 5 * cout + hw[0])
    x = torch.randn(b, cin, *hw, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (cin * 9)) ** 0.5
    bias = torch.randn(cout, generator=g)
    pk = K.conv2d_bf16_pack_weight(w.to(dev))
    auto = K.conv2d_bf16_fwd(x.to(dev), pk, cout, bias.to(dev))
    try:
        _C.lib().sassd_debug_set_bf16(nwg << 8)
        got = K.conv2d_bf16_fwd(x.to(dev), pk, cout, bias.to(dev))
        torch.cuda.synchronize()
    finally:
        _C.lib().sassd_debug_set_bf16(0)
    assert _rel(got, F.conv2d(_bf(x).double(), _bf(w).double(), bias.double(), 1, 1)) < 1e-5
    assert torch.equal(got, auto)           # the tiling does not change a pixel's summation order


def test_conv3x3_bf16_full_bev_layer(dev):
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 256, 200, 176, generator=g).to(dev)
    w = (torch.randn(256, 256, 3, 3, generator=g) / 48).to(dev)
    got = K.conv2d_bf16_fwd(x, K.conv2d_bf16_pack_weight(w), 256)
    exact = K.conv2d_wino_fwd(_bf(x), K.conv2d_wino_pack_weight(_bf(w)), 256, None, None)   # fp32 F(2x2) on rounded data
    ref = K.conv2d_wino_fwd(x, K.conv2d_wino_pack_weight(w), 256, None, None)
    print("full layer conv: bf16 vs fp32-on-rounded %.2e, vs fp32 %.2e" % (_rel(got, exact), _rel(got, ref)))
    assert _rel(got, exact) < 2e-5
    assert _rel(got, ref) < 1e-2
    assert float((got - exact).abs().max()) < 1e-3


def test_conv2dfn_bf16_matches_torch_autograd(dev):
    """forward + data gradient + weight gradient of Conv2dFn in bf16 mode against float64 autograd on rounded operands."""
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 128, 16, 32, generator=g)
    w = torch.randn(256, 128, 3, 3, generator=g) / 34
    dy = torch.randn(2, 256, 16, 32, generator=g)
    xd, wd = x.to(dev).requires_grad_(), w.to(dev).requires_grad_()
    try:
        AG.set_bev_precision("bf16")
        y = AG.Conv2dFn.apply(xd, wd, None, None, None)
        y.backward(dy.to(dev))
    finally:
        AG.set_bev_precision("fp32")
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    F.conv2d(xr, wr, None, 1, 1).backward(dy.double())
    assert _rel(y, F.conv2d(_bf(x).double(), _bf(w).double(), None, 1, 1)) < 1e-5
    assert _rel(xd.grad, xr.grad) < 1e-2 and _rel(wd.grad, wr.grad) < 1e-2
    # data gradient = conv of rounded dy with rounded, mirrored weights
    dx_sharp = F.conv_transpose2d(_bf(dy).double(), _bf(w).double(), None, 1, 1)
    assert _rel(xd.grad, dx_sharp) < 1e-5


def test_odd_width_falls_back_to_fp32(dev):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 32, 6, 45, generator=g).to(dev)
    dy = torch.randn(1, 64, 6, 45, generator=g).to(dev)
    assert torch.equal(K.conv2d_bwd_weight(x, dy, 3, bf16=True), K.conv2d_bwd_weight(x, dy, 3))


def test_conv2dfn_precision_switch(dev):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 64, 16, 48, generator=g).to(dev)                    # Cout = 64: fp32 forward, bf16 wgrad only
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev).requires_grad_()
    dy = torch.randn(2, 64, 16, 48, generator=g).to(dev)
    grads = {}
    try:
        for prec in ("fp32", "bf16"):
            AG.set_bev_precision(prec)
            w.grad = None
            AG.Conv2dFn.apply(x, w, None, None, None).backward(dy)
            grads[prec] = w.grad.clone()
    finally:
        AG.set_bev_precision("fp32")
    r = _rel(grads["bf16"], grads["fp32"])
    assert 1e-5 < r < 1e-2, r                                     # bf16 really ran, and stays inside the tolerance
    with pytest.raises(ValueError):
        AG.set_bev_precision("fp16")
