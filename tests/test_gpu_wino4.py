"""-m gpu parity of the Winograd F(4x4,3x3) BEV convolution (input transform -> 36 fp32-MFMA GEMMs -> output
transform) against torch-CPU conv2d (plain fp32 reference of the same op)."""
import pytest
import torch

import sassd
from sassd import kernels as K

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("b,cin,cout,hw,relu", [(1, 32, 256, (8, 12), True), (2, 64, 256, (20, 44), False),
                                                (1, 320, 256, (24, 176), True), (1, 256, 256, (200, 176), True),
                                                (3, 96, 256, (4, 4), False), (2, 64, 512, (188, 188), True)])
def test_conv2d_wino4(dev, b, cin, cout, hw, relu):
    """tile counts that are not multiples of the GEMM's 64-column block, several images, borders (every tile of a 4x4
    map touches the zero padding), the two BEV layer shapes; bar 1e-4 * max(1, |ref|) (F(4x4) rounds ~6x the direct
    kernel: measured 4e-5 at 256 input channels, tools/wino4_numerics.py)."""
    g = torch.Generator().manual_seed(cin * 7 + cout + hw[0])
    x = torch.randn(b, cin, *hw, generator=g)
    x[:, :, : hw[0] // 2] *= (torch.rand(b, cin, hw[0] // 2, hw[1], generator=g) > 0.7).float()    # sparse like BEV
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    raw = torch.nn.functional.conv2d(x, w, None, 1, 1)
    ref = raw * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    ref = torch.relu(ref) if relu else ref
    assert K.conv2d_wino4_supported(cin, cout, *hw)
    xd, wd = x.to(dev), w.to(dev)
    wp = K.conv2d_wino4_pack_weight(wd)
    y = K.conv2d_wino4_fwd(xd, wp, cout, sc.to(dev), sh.to(dev), relu)
    tol = 1e-4 * max(1.0, raw.abs().max().item())
    err = (y.cpu() - ref).abs().max().item()
    print("wino4 %s max abs err %.2e (tol %.2e)" % ((b, cin, cout, hw), err, tol))
    assert err <= tol, err
    y2 = K.conv2d_wino4_fwd(xd, wp, cout)              # no epilogue
    assert (y2.cpu() - raw).abs().max().item() <= tol
    assert not K.conv2d_wino4_supported(28, 28, 200, 176) and not K.conv2d_wino4_supported(256, 256, 198, 176)
    assert not K.conv2d_wino4_supported(256, 28, 200, 176) and not K.conv2d_wino4_supported(48, 256, 200, 176)
    assert not K.conv2d_wino4_supported(256, 128, 200, 176)


@pytest.mark.parametrize("b,cin,cout,hw,relu", [(1, 256, 256, (200, 176), True), (2, 64, 128, (12, 16), False),
                                                (3, 32, 384, (10, 32), True)])
def test_conv1x1_gemm(dev, b, cin, cout, hw, relu):
    """1x1 convolution on the MFMA GEMM kernel (one problem per image) against torch-CPU conv2d."""
    g = torch.Generator().manual_seed(cin + cout + hw[1])
    x = torch.randn(b, cin, *hw, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    raw = torch.nn.functional.conv2d(x, w)
    ref = raw * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    ref = torch.relu(ref) if relu else ref
    assert K.conv1x1_gemm_supported(cin, cout, *hw)
    wp = K.conv1x1_gemm_pack_weight(w.to(dev))
    y = K.conv1x1_gemm_fwd(x.to(dev), wp, cout, sc.to(dev), sh.to(dev), relu)
    assert (y.cpu() - ref).abs().max().item() <= 1e-5 * max(1.0, raw.abs().max().item())
    y2 = K.conv1x1_gemm_fwd(x.to(dev), wp, cout)
    assert (y2.cpu() - raw).abs().max().item() <= 1e-5 * max(1.0, raw.abs().max().item())
    assert not K.conv1x1_gemm_supported(256, 28, 200, 176) and not K.conv1x1_gemm_supported(256, 256, 188, 188)
