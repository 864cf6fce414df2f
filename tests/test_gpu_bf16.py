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
    g = torch.Generator().manual_seed(cin + 5 * cout + hw[0])
    x = torch.randn(b, cin, *hw, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (cin * 9)) ** 0.5
    bias = torch.randn(cout, generator=g)
    pk = K.conv2d_bf16_pack_weight(w.to(dev))
    auto = K.conv2d_bf16_fwd(x.to(dev), pk, cout, bias.to(dev))
    got = K.conv2d_bf16_fwd(x.to(dev), pk, cout, bias.to(dev), cfg=nwg << 8)      # per-call word: forced workgroup count
    torch.cuda.synchronize()
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


@pytest.mark.parametrize("b,cin,cout,hw", [(2, 256, 256, (24, 32)), (1, 256, 256, (9, 44)), (2, 20, 256, (12, 32)),
                                          (1, 28, 28, (7, 20)), (3, 128, 128, (5, 28)), (2, 256, 20, (16, 24)),
                                          (1, 60, 192, (3, 36)), (1, 16, 64, (1, 4)), (2, 250, 70, (11, 12)),
                                          (1, 256, 256, (200, 176))])
def test_conv1x1_bf16(dev, b, cin, cout, hw):
    """Round 6: 1x1 convolution on the bf16 MFMA with register-stationary weights (sassd_conv1x1_bf16_fwd) -- forward and, with
    the transposed pack, the data gradient of the same layer -- against float64 on the SAME rounded operands (only the fp32
    summation order differs) and, loosely, against the unrounded product.  Ragged Cin (20, 28, 60, 250), a masked last
    64-channel group (20, 28, 70), partial last tiles (HW % 64 != 0), many tiles per wave (the full BEV map)."""
    assert K.conv1x1_bf16_supported(cin, cout, hw[0] * hw[1])
    g = torch.Generator().manual_seed(cin + 3 * cout + hw[1])
    x = torch.randn(b, cin, *hw, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) * (1.0 / cin) ** 0.5
    bias = torch.randn(cout, generator=g)
    pk = K.conv1x1_bf16_pack_weight(w.to(dev))
    got = K.conv1x1_bf16_fwd(x.to(dev), pk, cout, bias.to(dev))
    sharp = _rel(got, F.conv2d(_bf(x).double(), _bf(w).double(), bias.double()))
    loose = _rel(got, F.conv2d(x.double(), w.double(), bias.double()))
    print("conv1x1 bf16 %s: vs rounded operands %.2e, vs fp32 operands %.2e" % ((b, cin, cout, hw), sharp, loose))
    assert sharp < 1e-5 and loose < 1e-2
    assert torch.equal(got, K.conv1x1_bf16_fwd(x.to(dev), pk, cout, bias.to(dev)))
    # data gradient: dy [b, cout] -> dx [b, cin] with the SAME weight array packed transposed
    dy = torch.randn(b, cout, *hw, generator=g)
    if K.conv1x1_bf16_supported(cout, cin, hw[0] * hw[1]):
        pkt = K.conv1x1_bf16_pack_weight(w.to(dev), transposed=True)
        dx = K.conv1x1_bf16_fwd(dy.to(dev), pkt, cin)
        assert _rel(dx, F.conv_transpose2d(_bf(dy).double(), _bf(w).double())) < 1e-5


def test_conv1x1_bf16_unsupported_shapes():
    assert not K.conv1x1_bf16_supported(257, 256, 64) and not K.conv1x1_bf16_supported(256, 256, 66)
    assert not K.conv1x1_bf16_supported(40, 256, 64)          # K = 64 with two ragged k-steps: stays on the fp32 kernel
    assert K.conv1x1_bf16_supported(49, 256, 64) and K.conv1x1_bf16_supported(241, 8, 4)


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


@pytest.mark.parametrize("b,cin,cout,hw", [(2, 256, 256, (200, 176)), (1, 320, 256, (188, 188)), (3, 40, 128, (9, 20)),
                                           (2, 72, 288, (13, 36))])
def test_bn_relu_in_the_conv_loaders_is_bit_identical(dev, b, cin, cout, hw):
    """Round 6: BatchNorm2d (batch statistics) + ReLU applied by the LOADER WAVES of the bf16 convolution and of its weight
    gradient (sassd_bn2d_stats + sassd_conv2d_bf16_bnrelu_fwd / sassd_conv2d_bwd_weight_bf16_bnrelu) against the stand-alone
    pass they replace (sassd_bn2d_relu_fwd, then the plain kernels on the normalised map): the same fp32 expression feeds the
    same rounding, so outputs, statistics, running statistics and weight gradients must be EQUAL BIT FOR BIT.  Shapes: the BEV
    layer at batch 2, the Waymo-scale map with a partial last tile column and 320 input channels, channel counts that are not
    multiples of the 32-channel chunk / the 64-channel weight-gradient tile, 128- and 288-cout workgroups."""
    g = torch.Generator().manual_seed(cin + hw[0])
    x = (torch.randn(b, cin, *hw, generator=g) * 1.3 + 0.2).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).to(dev)
    dy = torch.randn(b, cout, *hw, generator=g).to(dev)
    gam, bet = (torch.rand(cin, generator=g) + 0.5).to(dev), (torch.randn(cin, generator=g) * 0.3).to(dev)
    rm0, rv0 = torch.zeros(cin, device=dev), torch.ones(cin, device=dev)
    rm1, rv1 = rm0.clone(), rv0.clone()
    a, mean0, is0 = K.bn2d_relu_fwd(x, gam, bet, rm0, rv0, 0.01, 1e-3)
    mean1, is1, aff = K.bn2d_stats(x, gam, bet, rm1, rv1, 0.01, 1e-3)
    assert torch.equal(mean0, mean1) and torch.equal(is0, is1) and torch.equal(rm0, rm1) and torch.equal(rv0, rv1)
    assert torch.equal(aff[0], mean1) and torch.equal(aff[1], is1 * gam) and torch.equal(aff[2], bet)
    pk = K.conv2d_bf16_pack_weight(w)
    bias = torch.randn(cout, generator=g).to(dev)
    y0 = K.conv2d_bf16_fwd(a, pk, cout, bias)
    y1 = K.conv2d_bf16_fwd(x, pk, cout, bias, in_affine=aff)
    assert torch.equal(y0, y1), float((y0 - y1).abs().max())
    dw0 = K.conv2d_bwd_weight(a, dy, 3, bf16=True)
    dw1 = K.conv2d_bwd_weight(x, dy, 3, bf16=True, x_affine=aff)
    assert torch.equal(dw0, dw1), float((dw0 - dw1).abs().max())


def test_bevnet_with_fused_bn_is_bit_identical(dev):
    """The BEV stack of the bf16 training step with BatchNorm + ReLU of conv0 .. conv5 folded into the next layer's loaders
    (BEVNet.fuse_bn_into_conv, autograd.BnReluConvBf16Fn) against the layer-by-layer formulation: both outputs, every parameter
    gradient, the input gradient and every running statistic equal bit for bit; the fused run must not launch the stand-alone
    apply for those six layers (counted through the autograd functions it goes through)."""
    from sassd.detector import BEVNet
    torch.manual_seed(5)
    net = BEVNet(in_features=320, num_filters=256).to(dev).train()
    x0 = torch.relu(torch.randn(2, 320, 40, 48, device=dev))
    dys = torch.randn(2, 256, 40, 48, device=dev), torch.randn(2, 256, 40, 48, device=dev)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    res = {}
    calls = {"fused": 0}
    f0 = AG.BnReluConvBf16Fn.forward

    def counted(ctx, *a):
        calls["fused"] += 1
        return f0(ctx, *a)
    try:
        AG.set_bev_precision("bf16")
        AG.BnReluConvBf16Fn.forward = staticmethod(counted)
        for fused in (False, True):
            net.load_state_dict(state)
            net.zero_grad()
            BEVNet.fuse_bn_into_conv = fused
            x = x0.clone().requires_grad_()
            out, c6 = net(x)
            (out * dys[0]).sum().backward(retain_graph=True)
            (c6 * dys[1]).sum().backward()
            res[fused] = (out.detach().clone(), c6.detach().clone(), x.grad.clone(),
                          {n: p.grad.clone() for n, p in net.named_parameters()},
                          {n: bf.clone() for n, bf in net.named_buffers() if "running" in n})
            assert calls["fused"] == (6 if fused else 0), calls
    finally:
        AG.BnReluConvBf16Fn.forward = staticmethod(f0)
        AG.set_bev_precision("fp32")
        BEVNet.fuse_bn_into_conv = True
    a, b_ = res[False], res[True]
    assert torch.equal(a[0], b_[0]) and torch.equal(a[1], b_[1]) and torch.equal(a[2], b_[2])
    for n in a[3]:
        assert torch.equal(a[3][n], b_[3][n]), n
    for n in a[4]:
        assert torch.equal(a[4][n], b_[4][n]), n
