"""-m gpu parity of the Winograd F(4x4,3x3) BEV convolution (input transform -> 36 GEMMs -> output transform) against
torch-CPU conv2d (plain fp32 reference of the same op).  The GEMMs run their fp32 products on the bf16 MFMA over exactly
split operands by default; the fp32-MFMA form and the other workgroup shapes are held to the same bar against fp64."""
import numpy as np
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


@pytest.mark.parametrize("cfg", [0, 11, 13, 14])
def test_conv2d_wino4_gemm_geometries(dev, cfg):
    """The launch geometries of the Winograd GEMM selected by the per-call cfg word -- 1 = the fp32 MFMA, 0 (default) and 11..14 =
    fp32 products on the bf16 MFMA over exactly-split operands (three bf16 pieces per fp32 value, eight of the nine piece
    products) --
    against torch-CPU conv2d at the bar of the default path, on the KITTI BEV layer and on a ragged multi-image shape with
    512 output channels.  The error of each is printed next to the fp32 MFMA's: the split path must not be the looser one
    by more than 2x (it computes every product to 2^-30 and accumulates in fp32 like the fp32 instruction)."""
    errs = {}
    for shape in ((1, 256, 256, (200, 176)), (3, 96, 512, (20, 44))):
        b, cin, cout, hw = shape
        g = torch.Generator().manual_seed(cin + cout)
        x = torch.randn(b, cin, *hw, generator=g)
        x *= (torch.rand(b, cin, *hw, generator=g) > 0.5).float()
        w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
        raw = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
        wp = K.conv2d_wino4_pack_weight(w.to(dev))
        tol = 1e-4 * max(1.0, raw.abs().max().item())
        for c in (1, cfg):
            with K.default_cfg(wino4=K.wino4_cfg(c)):
                y = K.conv2d_wino4_fwd(x.to(dev), wp, cout)
                torch.cuda.synchronize()
            errs[c] = (y.cpu().double() - raw).abs().max().item()
            assert errs[c] <= tol, (c, shape, errs[c])
        print("wino4 geometry %d %s: max abs err vs fp64 %.2e (fp32 MFMA %.2e, tol %.2e)" % (cfg, shape, errs[cfg], errs[1], tol))
        assert errs[cfg] <= 2.0 * errs[1] + 1e-7


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


def test_split_operand_gemm_dynamic_range(dev):
    """The split-operand GEMM without a Winograd transform around it (1x1 convolution, 256-column map -> the split kernel):
    operands whose magnitudes span 2^-40 .. 2^40 per input channel, zeros, negative zeros, values with all 24 significand bits
    set and values whose low pieces vanish (exact bf16 numbers).  Every product is formed from bf16 pieces, so the bar is the
    fp32 dot product's: |y - y64| <= 2e-6 * sum |w| |x| per output, and the fp32-MFMA geometry must not be tighter by more
    than 2x.  (An Inf operand gives NaN instead of Inf -- Inf - Inf in the split -- which the reference's convolution
    produces one layer later anyway; not exercised.)"""
    g = torch.Generator().manual_seed(11)
    b, cin, cout, hw = 2, 256, 128, (16, 16)
    x = torch.randn(b, cin, *hw, generator=g)
    scale = torch.pow(2.0, torch.randint(-40, 41, (cin,), generator=g).float())
    x = x * scale.view(1, -1, 1, 1)
    x[:, 0] = 0.0
    x[:, 1] = -0.0
    x[:, 2] = torch.tensor(16777215.0)                       # 2^24 - 1: all significand bits set
    x[:, 3] = x[:, 3].bfloat16().float()                     # exact bf16 values: the lower pieces are zero
    w = torch.randn(cout, cin, 1, 1, generator=g) / scale.view(1, -1, 1, 1) / 16.0
    w[:, 2] = torch.tensor(1.0 / 16777215.0)
    assert K.conv1x1_gemm_supported(cin, cout, *hw) and (hw[0] * hw[1]) % 128 == 0
    wp = K.conv1x1_gemm_pack_weight(w.to(dev))
    y64 = torch.nn.functional.conv2d(x.double(), w.double())
    bound = torch.nn.functional.conv2d(x.double().abs(), w.double().abs())
    errs = {}
    for c in (1, 0):
        with K.default_cfg(wino4=K.wino4_cfg(c)):
            y = K.conv1x1_gemm_fwd(x.to(dev), wp, cout)
            torch.cuda.synchronize()
        assert torch.isfinite(y).all()
        errs[c] = ((y.cpu().double() - y64).abs() / bound.clamp(min=1e-300)).max().item()
    print("split-operand GEMM, operands over 2^-40..2^40: max |err| / sum|w||x| = %.2e (fp32 MFMA %.2e)" % (errs[0], errs[1]))
    assert errs[0] <= 2e-6 and errs[0] <= 2.0 * errs[1] + 1e-8


@pytest.mark.parametrize("b,cin0,hw", [(1, 320, (200, 176)), (3, 64, (20, 44)), (2, 256, (188, 188)), (2, 96, (8, 12))])
def test_conv2d_wino4_chain(dev, b, cin0, hw):
    """Three chained 3x3 layers (cin0 -> 256 -> 256 -> 256, folded scale / shift + ReLU after each) through
    sassd_conv2d_wino4_chain -- the maps between the layers stay in the transform domain (fused output -> input transform,
    one LDS plane per (channel, image)) -- against the same three layers through sassd_conv2d_wino4_fwd (three launches per
    layer, maps in HBM), which the test above holds to torch-CPU conv2d.  The fused kernel performs the same fp32 operations
    in the same order: the results are compared at 1e-6 relative (bit equality is reported).  Shapes: the KITTI BEV map
    (tile count 2200), a multi-image batch whose images do not start on a 4-tile boundary (11 x 5 = 55 tiles per image),
    the Waymo-scale map (47 x 47 tiles), a tiny map where every tile touches the padding."""
    g = torch.Generator().manual_seed(cin0 + hw[0])
    x = torch.randn(b, cin0, *hw, generator=g)
    x *= (torch.rand(b, cin0, *hw, generator=g) > 0.6).float()
    cins = [cin0, 256, 256]
    ws = [torch.randn(256, c, 3, 3, generator=g) * (2.0 / (c * 9)) ** 0.5 for c in cins]
    scs = [(torch.rand(256, generator=g) + 0.5).to(dev) for _ in cins]
    shs = [(torch.randn(256, generator=g) * 0.1).to(dev) for _ in cins]
    wps = [K.conv2d_wino4_pack_weight(w.to(dev)) for w in ws]
    assert all(K.conv2d_wino4_chain_supported(c, 256, *hw) for c in cins)
    xd = x.to(dev)
    ref = xd
    for wp, sc, sh in zip(wps, scs, shs):
        ref = K.conv2d_wino4_fwd(ref, wp, 256, sc, sh, True)
    cmax = max(cin0, 256)
    wsb = K.conv2d_wino4_chain_workspace(b, cmax, hw[0], hw[1], dev)
    y = torch.empty(b, 256, *hw, device=dev)
    for i, (wp, sc, sh) in enumerate(zip(wps, scs, shs)):
        K.conv2d_wino4_chain(xd if i == 0 else None, None if i == 0 else (scs[i - 1], shs[i - 1], True), wp, cins[i], 256,
                             cmax, b, hw[0], hw[1], sc, sh, True, y if i == 2 else None, wsb)
    torch.cuda.synchronize()
    err = (y - ref).abs().max().item()
    print("wino4 chain %s: max abs difference to the unchained layers %.2e (bit-equal: %s)" % ((b, cin0, hw), err,
                                                                                          torch.equal(y, ref)))
    assert err <= 1e-6 * max(1.0, ref.abs().max().item()), err
    assert not K.conv2d_wino4_chain_supported(256, 256, 400, 352)          # plane does not fit the LDS


@pytest.mark.parametrize("b,cin0,hw,frac", [(1, 320, (200, 176), 0.002), (3, 64, (20, 44), 0.01), (2, 320, (188, 188), 0.0),
                                            (1, 96, (8, 8), 1.0)])
def test_conv2d_wino4_chain_on_active_tiles_is_bit_identical(dev, b, cin0, hw, frac):
    """conv0 of the BEV stack reads the densified sparse map; with a tile map (sassd_wino4_tile_map, built from the sparse rows'
    coordinates) its Winograd launch transforms and multiplies only the tiles with an occupied pixel in their 6 x 6 patch.
    An inactive tile's products are exactly zero, so both forms of the layer -- with its own output transform, and chained
    into a second layer through the fused transform -- must equal the dense launch BIT FOR BIT; the map itself is checked
    against a numpy restatement (patch rows 4 t - 1 .. 4 t + 4).  Cases: a KITTI-sized map with clustered occupancy, a batch
    of small images, an EMPTY map, a map where every pixel is occupied."""
    g = torch.Generator().manual_seed(int(frac * 1000) + hw[0])
    H_, W_ = hw
    occ = torch.rand(b, H_, W_, generator=g) < frac
    if 0.0 < frac < 1.0:                                    # clustered: occupancy only in a band of rows
        occ[:, : H_ // 3] = False
    x = torch.randn(b, cin0, H_, W_, generator=g) * occ.unsqueeze(1).float()
    nz = occ.nonzero()                                       # (b, y, x)
    n = nz.shape[0]
    cap = n + 37
    idx = torch.zeros(cap, 4, dtype=torch.int32)
    idx[:n, 0], idx[:n, 2], idx[:n, 3] = nz[:, 0].int(), nz[:, 1].int(), nz[:, 2].int()
    idx[n:] = torch.tensor([0, 0, 1, 1], dtype=torch.int32)  # rows past the count must be ignored
    nptr = torch.tensor([n], dtype=torch.int32, device=dev)
    tmap = K.wino4_tile_map(idx.to(dev), nptr, cap, b, H_, W_)
    TH, TW = H_ // 4, W_ // 4
    T = b * TH * TW
    want = np.zeros((b, TH, TW), bool)
    for bb, yy, xx in nz.numpy():
        for ty in {yy // 4, (yy - 1) // 4 if yy % 4 == 0 else yy // 4, (yy + 1) // 4 if yy % 4 == 3 else yy // 4}:
            for tx in {xx // 4, (xx - 1) // 4 if xx % 4 == 0 else xx // 4, (xx + 1) // 4 if xx % 4 == 3 else xx // 4}:
                if 0 <= ty < TH and 0 <= tx < TW:
                    want[bb, ty, tx] = True
    tm = tmap.cpu().numpy()
    act = np.flatnonzero(want.reshape(-1))
    assert tm[0] == len(act) and tm[1] == T
    assert np.array_equal(tm[4 + T:4 + T + len(act)], act)
    pos = np.full(T, -1, np.int64)
    pos[act] = np.arange(len(act))
    assert np.array_equal(tm[4:4 + T], pos)
    ws = [torch.randn(256, c, 3, 3, generator=g) * (2.0 / (c * 9)) ** 0.5 for c in (cin0, 256)]
    scs = [(torch.rand(256, generator=g) + 0.5).to(dev) for _ in ws]
    shs = [(torch.randn(256, generator=g) * 0.1).to(dev) for _ in ws]
    wps = [K.conv2d_wino4_pack_weight(w.to(dev)) for w in ws]
    cmax = max(cin0, 256)
    wsb = K.conv2d_wino4_chain_workspace(b, cmax, H_, W_, dev)
    xd = x.to(dev)

    def one_layer(tile_map):
        y = torch.empty(b, 256, H_, W_, device=dev)
        wsb.view(torch.float32).fill_(float("nan"))          # stale columns must never be read
        K.conv2d_wino4_chain(xd, None, wps[0], cin0, 256, cmax, b, H_, W_, scs[0], shs[0], True, y, wsb, tile_map=tile_map)
        return y

    def two_layers(tile_map):
        y = torch.empty(b, 256, H_, W_, device=dev)
        wsb.view(torch.float32).fill_(float("nan"))
        K.conv2d_wino4_chain(xd, None, wps[0], cin0, 256, cmax, b, H_, W_, scs[0], shs[0], True, None, wsb, tile_map=tile_map)
        K.conv2d_wino4_chain(None, (scs[0], shs[0], True), wps[1], 256, 256, cmax, b, H_, W_, scs[1], shs[1], True, y, wsb,
                             prev_tile_map=tile_map)
        return y
    for fn in (one_layer, two_layers):
        dense, sparse = fn(None), fn(tmap)
        torch.cuda.synchronize()
        assert torch.isfinite(sparse).all()
        assert torch.equal(dense, sparse), (fn.__name__, (dense - sparse).abs().max().item())
    print("wino4 active tiles %s: %d of %d tiles active, outputs bit-identical" % ((b, cin0, hw), len(act), T))


@pytest.mark.parametrize("b,hw,cout", [(1, (200, 176), 28), (2, (188, 188), 28), (3, (20, 44), 64), (1, (8, 8), 5)])
def test_conv2d_wino4_chain_tail_narrow_layer(dev, b, hw, cout):
    """Round 6: a narrow 3x3 layer (the part-sensitive head's 256 -> 28 conv) as the TAIL of a Winograd chain
    (sassd_conv2d_wino4_chain_tail: fused transform of the previous layer's products, 36 GEMMs on a 64-channel block, output
    transform) -- against float64 conv2d on the previous layer's map at the bar of the other F(4x4) layers, and next to the direct
    fp32-MFMA kernel it replaces.  The previous layer's activation map stored by the fused transform (`y_prev`) must equal the one
    its own output-transform launch writes bit for bit."""
    g = torch.Generator().manual_seed(cout + hw[0])
    H_, W_ = hw
    x = torch.randn(b, 256, H_, W_, generator=g)
    x *= (torch.rand(b, 256, H_, W_, generator=g) > 0.5).float()
    wa = torch.randn(256, 256, 3, 3, generator=g) * (2.0 / (256 * 9)) ** 0.5
    wt = torch.randn(cout, 256, 3, 3, generator=g) * (2.0 / (256 * 9)) ** 0.5
    sca, sha = (torch.rand(256, generator=g) + 0.5).to(dev), (torch.randn(256, generator=g) * 0.1).to(dev)
    sct, sht = (torch.rand(cout, generator=g) + 0.5).to(dev), (torch.randn(cout, generator=g) * 0.1).to(dev)
    wpa = K.conv2d_wino4_pack_weight(wa.to(dev))
    wpt = K.conv2d_wino4_pack_weight_narrow(wt.to(dev))
    wsb = K.conv2d_wino4_chain_workspace(b, 256, H_, W_, dev)
    xd = x.to(dev)
    ya = torch.empty(b, 256, H_, W_, device=dev)
    K.conv2d_wino4_chain(xd, None, wpa, 256, 256, 256, b, H_, W_, sca, sha, True, ya, wsb)            # its own output transform
    wsb.view(torch.float32).fill_(float("nan"))
    K.conv2d_wino4_chain(xd, None, wpa, 256, 256, 256, b, H_, W_, sca, sha, True, None, wsb)          # products stay
    yp = torch.full((b, 256, H_, W_), 7.0, device=dev)
    yt = torch.full((b, cout, H_, W_), 7.0, device=dev)
    K.conv2d_wino4_chain_tail((sca, sha, True), yp, wpt, 256, cout, 256, b, H_, W_, sct, sht, True, yt, wsb)
    torch.cuda.synchronize()
    assert torch.equal(yp, ya)
    ref = torch.relu(torch.nn.functional.conv2d(ya.cpu().double(), wt.double(), None, 1, 1) * sct.cpu().double().view(1, -1, 1, 1)
                     + sht.cpu().double().view(1, -1, 1, 1))
    direct = K.conv2d_fwd(ya, K.conv2d_pack_weight(wt.to(dev)), cout, 3, sct, sht, True)
    tol = 1e-4 * max(1.0, ref.abs().max().item())
    e_t, e_d = (yt.cpu().double() - ref).abs().max().item(), (direct.cpu().double() - ref).abs().max().item()
    print("wino4 chain tail %s -> %d: max abs err vs fp64 %.2e (direct fp32-MFMA kernel %.2e, tol %.2e)" % ((b, hw), cout, e_t, e_d, tol))
    assert torch.isfinite(yt).all() and e_t <= tol
    yt2 = torch.empty_like(yt)                                                                          # y_prev is optional
    K.conv2d_wino4_chain(xd, None, wpa, 256, 256, 256, b, H_, W_, sca, sha, True, None, wsb)
    K.conv2d_wino4_chain_tail((sca, sha, True), None, wpt, 256, cout, 256, b, H_, W_, sct, sht, True, yt2, wsb)
    assert torch.equal(yt, yt2)
