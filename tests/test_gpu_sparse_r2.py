"""-m gpu parity of the round-2 sparse path: fused rulebook pyramid, gather-GEMM-scatter sparse conv (both row-slice
sizes + the legacy register-stationary kernel), device-side point count of the voxelizer, hipGraph replay of a frame."""
import numpy as np
import pytest
import torch

import sassd
from sassd import kernels as K, synth
from oracle import clib, nets as onets, rulebook as orb
import helpers as H

pytestmark = pytest.mark.gpu


def _level0(name="k21", seed=0, batch=1):
    idx = []
    for b in range(batch):
        _, c, _ = clib.points_to_voxel(H.frame(name, seed + b), synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 20000)
        idx.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))
    return np.concatenate(idx, 0)


@pytest.mark.parametrize("batch,staged,persistent", [(1, False, False), (2, True, False), (8, False, False),
                                                     (1, False, True), (2, False, True), (8, False, True)])
def test_rulebook_pyramid_bit_exact(dev, batch, staged, persistent):
    """All seven VxNet rulebooks + the three down-sampled coordinate sets from ONE pyramid build, bit-exact against
    the oracle (same bar as the per-op chain); batch 8 exercises the chained scan past the resident-grid size.
    `persistent`: the one-launch form (phases separated by in-launch grid barriers)."""
    idx0 = _level0("k21" if batch < 8 else "k17", 0, batch)
    shape = (40, 1600, 1408)
    n0 = len(idx0)
    caps = [n0 + 11] + [2 * n0 + 64] * 3
    i32 = torch.int32
    idx = [torch.zeros(c, 4, dtype=i32, device=dev) for c in caps]
    idx[0][:n0] = torch.from_numpy(idx0).to(dev)
    n = [torch.tensor([n0 if l == 0 else -7], dtype=i32, device=dev) for l in range(4)]
    subm = [torch.full((c, 27), -5, dtype=i32, device=dev) for c in caps]
    down = [None] + [torch.full((c, 27), -5, dtype=i32, device=dev) for c in caps[1:]]
    st = K.new_status(dev)
    pyr = K.RulebookPyramid(idx, n, caps, shape, batch, subm, down, st)
    for rep in range(2):                       # the second build checks that the workspace reset is complete
        if staged:
            for l in range(4):
                pyr.build(l, l + 1)
        else:
            pyr.build(persistent=persistent)
    torch.cuda.synchronize()
    assert int(st.item()) == 0
    ref_idx, ref_shape = idx0, shape
    for l in range(4):
        m = int(n[l].item())
        assert m == len(ref_idx), (l, m, len(ref_idx))
        assert np.array_equal(idx[l][:m].cpu().numpy(), ref_idx), "coordinates level %d" % l
        _, onbr = orb.subm_rulebook(ref_idx, ref_shape)
        assert np.array_equal(subm[l][:m].cpu().numpy(), onbr), "subm level %d" % l
        if l < 3:
            nxt_idx, nbr_d, nxt_shape = orb.conv_rulebook(ref_idx, ref_shape, batch)
            torch.cuda.synchronize()
            md = int(n[l + 1].item())
            assert md == len(nxt_idx)
            assert np.array_equal(down[l + 1][:md].cpu().numpy(), nbr_d), "down level %d" % l
            ref_idx, ref_shape = nxt_idx, nxt_shape
    assert ref_shape == (5, 200, 176)


def test_rulebook_pyramid_persistent_under_load(dev):
    """The one-launch pyramid hands data between workgroups INSIDE a launch (agent-scope release / acquire grid barriers):
    its outputs must equal the phase-per-launch form's bit for bit on every one of 30 builds while two other streams keep
    the chip unevenly busy (a bandwidth-bound copy and a short-kernel stream), with consumers' caches warm from the
    previous build -- idle chips and cold caches hide stale hand-offs (MI355X_MICROARCH.md, inter-workgroup visibility)."""
    idx0 = _level0("k21", 0, 2)
    shape = (40, 1600, 1408)
    n0 = len(idx0)
    caps = [n0 + 11] + [2 * n0 + 64] * 3
    i32 = torch.int32

    def mk():
        idx = [torch.zeros(c, 4, dtype=i32, device=dev) for c in caps]
        idx[0][:n0] = torch.from_numpy(idx0).to(dev)
        n = [torch.tensor([n0 if l == 0 else -7], dtype=i32, device=dev) for l in range(4)]
        subm = [torch.full((c, 27), -5, dtype=i32, device=dev) for c in caps]
        down = [None] + [torch.full((c, 27), -5, dtype=i32, device=dev) for c in caps[1:]]
        st = K.new_status(dev)
        return idx, n, subm, down, st, K.RulebookPyramid(idx, n, caps, shape, 2, subm, down, st)
    ref = mk()
    ref[5].build()
    torch.cuda.synchronize()
    tst = mk()
    big = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    small = torch.zeros(4096, device=dev)
    s1, s2, s3 = (torch.cuda.Stream(device=dev) for _ in range(3))
    for rep in range(30):
        with torch.cuda.stream(s1):
            for _ in range(1 + rep % 3):
                big.add_(1.0)
        with torch.cuda.stream(s2):
            for _ in range(20):
                small.add_(1.0)
        with torch.cuda.stream(s3):
            tst[5].build(persistent=True, wgs_per_cu=1 + rep % 4)
        torch.cuda.synchronize()
        assert int(tst[4].item()) == 0, "status 0x%x at rep %d" % (int(tst[4].item()), rep)
        for l in range(4):
            m = int(ref[1][l].item())
            assert int(tst[1][l].item()) == m
            assert torch.equal(tst[0][l][:m], ref[0][l][:m]), "coordinates level %d rep %d" % (l, rep)
            assert torch.equal(tst[2][l][:m], ref[2][l][:m]), "subm level %d rep %d" % (l, rep)
            if l:
                assert torch.equal(tst[3][l][:m], ref[3][l][:m]), "down level %d rep %d" % (l, rep)
        for l in range(1, 4):                      # poison the outputs: a stale read must not pass on old data
            tst[2][l].fill_(-5)
            tst[3][l].fill_(-5)
            tst[0][l].zero_()


def test_rulebook_pyramid_overflow_flag(dev):
    idx0 = _level0("small", 3, 1)
    n0 = len(idx0)
    caps = [n0, n0 // 4, n0, n0]                     # level 1 cannot hold its rows
    i32 = torch.int32
    idx = [torch.zeros(c, 4, dtype=i32, device=dev) for c in caps]
    idx[0][:n0] = torch.from_numpy(idx0).to(dev)
    n = [torch.tensor([n0], dtype=i32, device=dev)] + [torch.zeros(1, dtype=i32, device=dev) for _ in range(3)]
    subm = [torch.zeros(c, 27, dtype=i32, device=dev) for c in caps]
    down = [None] + [torch.zeros(c, 27, dtype=i32, device=dev) for c in caps[1:]]
    st = K.new_status(dev)
    K.RulebookPyramid(idx, n, caps, (40, 1600, 1408), 1, subm, down, st).build()
    torch.cuda.synchronize()
    assert int(st.item()) & 1 and int(n[1].item()) == caps[1]


# every geometry the default dispatch can pick (by layer shape and capacity), forced onto every channel pair: none is "slow"
@pytest.mark.parametrize("mode", ["default", "legacy", "gq16x4", "gq4x4", "r3", "rw64x4", "rw64x8"])
@pytest.mark.parametrize("cin,cout", [(16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (32, 16), (64, 32)])
def test_spconv_gather_gemm_scatter(dev, cin, cout, mode):
    """subm and strided gather tables, ragged row counts (not multiples of the slice), every forward / data-gradient
    shape, against the CPU oracle; bar 2e-4 * max|y| (fp32 sums in a different order)."""
    flags = {"default": 0, "rw64x4": 1 << 16, "rw64x8": 5 << 16, "legacy": 256, "gq16x4": 9 << 16, "gq4x4": 8 << 16,
             "r3": 10 << 16}[mode]
    if mode == "legacy" and (cin, cout) in ((32, 16), (64, 32)):
        pytest.skip("covered by the backward tests")
    idx = _level0("small", 1)
    shape = (40, 1600, 1408)
    idx1, nbr_d1, shape1 = orb.conv_rulebook(idx, shape, 1)
    idx2, nbr_d2, shape2 = orb.conv_rulebook(idx1, shape1, 1)
    _, nbr_s = orb.subm_rulebook(idx2, shape2)
    with K.default_cfg(spconv=flags):
        for nbr, n_in in ((nbr_s, len(idx2)), (nbr_d2, len(idx1))):
            n = len(nbr)
            g = torch.Generator().manual_seed(cin * 100 + cout)
            x = torch.randn(n_in, cin, generator=g)
            w = torch.randn(27, cin, cout, generator=g) * 0.2
            scale = torch.rand(cout, generator=g) + 0.5
            shift = torch.randn(cout, generator=g) * 0.1
            raw = onets.sparse_conv(x, nbr, w)
            ref = torch.relu(raw * scale + shift)
            cap = n + 37
            nb = torch.full((cap, 27), -1, dtype=torch.int32, device=dev)
            nb[:n] = torch.from_numpy(nbr).to(dev)
            nptr = torch.tensor([n], dtype=torch.int32, device=dev)
            wp = K.spconv_pack_weight(w.to(dev))
            y = torch.full((cap, cout), 7.0, device=dev)
            K.spconv_fwd(x.to(dev), nb, nptr, cap, wp, 27, cin, cout, scale.to(dev), shift.to(dev), True, y)
            tol = 2e-4 * max(1.0, raw.abs().max().item())
            err = (y[:n].cpu() - ref).abs().max().item()
            assert err < tol, (mode, err)
            assert bool((y[n:] == 7.0).all()), "rows past the device row count must stay untouched"
            y2 = K.spconv_fwd(x.to(dev), nb, nptr, cap, wp, 27, cin, cout)
            assert (y2[:n].cpu() - raw).abs().max().item() < tol


@pytest.mark.parametrize("mode", [0, 1, 5, 8, 9, 10])
@pytest.mark.parametrize("cin,cout", [(64, 64), (32, 32), (16, 32)])
def test_spconv_balanced_kernel_many_blocks(dev, cin, cout, mode):
    """The balanced kernel's row -> (block, interleaved slice) map past 16384 rows (more than 8 blocks), on a batch of
    two K21 frames at level 1 (36 k rows) with the device row count below the capacity; every geometry switch."""
    idx = _level0("k21", 0, 2)
    idx1, _, shape1 = orb.conv_rulebook(idx, (40, 1600, 1408), 2)
    _, nbr = orb.subm_rulebook(idx1, shape1)
    n = len(nbr)
    assert n > 32768
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(n, cin, generator=g)
    w = torch.randn(27, cin, cout, generator=g) * 0.2
    raw = onets.sparse_conv(x, nbr, w)
    cap = n + 5000
    nb = torch.full((cap, 27), -1, dtype=torch.int32, device=dev)
    nb[:n] = torch.from_numpy(nbr).to(dev)
    nptr = torch.tensor([n], dtype=torch.int32, device=dev)
    wp = K.spconv_pack_weight(w.to(dev))
    with K.default_cfg(spconv=mode << 16):
        y = torch.full((cap, cout), 7.0, device=dev)
        K.spconv_fwd(x.to(dev), nb, nptr, cap, wp, 27, cin, cout, None, None, False, y)
        y2 = y.clone()
        K.spconv_fwd(x.to(dev), nb, nptr, cap, wp, 27, cin, cout, None, None, False, y2)
    tol = 2e-4 * max(1.0, raw.abs().max().item())
    assert (y[:n].cpu() - raw).abs().max().item() < tol
    assert bool((y[n:] == 7.0).all())
    assert torch.equal(y, y2)


@pytest.mark.parametrize("cin,cout", [(64, 64), (16, 32), (64, 32)])
@pytest.mark.parametrize("n", [0, 1, 17, 4099, 40000])
def test_spconv_1x1x1_streaming_kernel(dev, cin, cout, n):
    """The 1 x 1 x 1 layer (cmn.py:208-212 `extra_conv`, K = 1, no rulebook) on the streaming kernel: ragged row counts,
    folded scale / shift / ReLU, rows past the device row count untouched, against a float64 product."""
    cap = n + 33
    g = torch.Generator().manual_seed(n + cin)
    x = torch.randn(max(cap, 1), cin, generator=g)
    w = torch.randn(1, cin, cout, generator=g) * 0.3
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ref = torch.relu((x[:n].double() @ w[0].double()) * scale.double() + shift.double()).float()
    nptr = torch.tensor([n], dtype=torch.int32, device=dev)
    y = torch.full((cap, cout), 5.0, device=dev)
    K.spconv_fwd(x.to(dev), None, nptr, cap, K.spconv_pack_weight(w.to(dev)), 1, cin, cout, scale.to(dev), shift.to(dev),
                 True, y)
    torch.cuda.synchronize()
    assert bool((y[n:] == 5.0).all())
    if n:
        assert (y[:n].cpu() - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())
        y2 = K.spconv_fwd(x.to(dev), None, nptr, cap, K.spconv_pack_weight(w.to(dev)), 1, cin, cout)
        assert (y2[:n].cpu() - (x[:n].double() @ w[0].double()).float()).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


def test_spconv_empty_and_tiny(dev):
    """0 rows and fewer rows than one MFMA tile."""
    for n in (0, 1, 5):
        cap = 64
        x = torch.randn(max(n, 1), 32, device=dev)
        nb = torch.full((cap, 27), -1, dtype=torch.int32, device=dev)
        for r in range(n):
            nb[r, 13] = r
            if r + 1 < n:
                nb[r, 14] = r + 1
        w = torch.randn(27, 32, 32, device=dev) * 0.1
        y = torch.full((cap, 32), 3.0, device=dev)
        K.spconv_fwd(x, nb, torch.tensor([n], dtype=torch.int32, device=dev), cap, K.spconv_pack_weight(w), 27, 32, 32,
                     None, None, False, y)
        torch.cuda.synchronize()
        assert bool((y[n:] == 3.0).all())
        if n:
            ref = x[:n] @ w[13]
            ref[:n - 1] += x[1:n] @ w[14]
            assert (y[:n] - ref).abs().max().item() < 1e-4


def test_voxelize_device_count(dev):
    """sassd_voxelize_dev on a capacity-sized staging buffer == sassd_voxelize on the exact cloud (bit-exact)."""
    pts = H.frame("k21", 3)
    p = torch.from_numpy(pts).to(dev)
    ref = K.voxelize(p, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 20000, coors_cols=4)
    stage = torch.full((30000, 4), 1e9, device=dev)
    stage[:len(pts)] = p
    stage[len(pts):len(pts) + 50] = p[:50]            # garbage past the count must be ignored
    ndev = torch.tensor([len(pts)], dtype=torch.int32, device=dev)
    st = K.new_status(dev)
    got = K.voxelize(stage, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 20000, coors_cols=4, n_dev=ndev, status=st)
    m = int(ref["voxel_num"].item())
    assert int(got["voxel_num"].item()) == m and int(st.item()) == 0
    for k in ("coors", "num_points", "mean", "voxels"):
        assert torch.equal(got[k][:m], ref[k][:m]), k
    ndev.fill_(0)
    got = K.voxelize(stage, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 20000, coors_cols=4, n_dev=ndev)
    assert int(got["voxel_num"].item()) == 0


def test_spconv_is_bit_reproducible(dev):
    """Static offset -> wave assignment + slab sums in wave order: the same inputs give the same bits on every launch,
    on a busy and on an idle GPU (the ticket-counter assignment, debug bit 4, would not guarantee that)."""
    idx = _level0("k21", 0)
    shape = (40, 1600, 1408)
    for _ in range(3):
        idx, _, shape = orb.conv_rulebook(idx, shape, 1)
    _, nbr = orb.subm_rulebook(idx, shape)
    n = len(nbr)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, 64, generator=g).to(dev)
    w = (torch.randn(27, 64, 64, generator=g) * 0.1).to(dev)
    nb = torch.from_numpy(nbr).to(dev)
    nptr = torch.tensor([n], dtype=torch.int32, device=dev)
    wp = K.spconv_pack_weight(w)
    ref = K.spconv_fwd(x, nb, nptr, n, wp, 27, 64, 64).clone()
    busy = torch.randn(4096, 4096, device=dev)
    for i in range(20):
        if i % 2:
            busy = busy @ busy * 1e-3                 # another kernel in flight changes the wave timing
        y = K.spconv_fwd(x, nb, nptr, n, wp, 27, 64, 64)
        assert torch.equal(y, ref), i


def test_weight_gradient_formulations_agree(dev):
    """The offset-per-wave weight gradient (default) against the tile-per-wave kernel (debug bit 5) on the K21 input
    level (oracle rulebook): same partial-sum layout, different order inside a 128-row chunk -> equal to fp32 rounding;
    both bit-reproducible."""
    idx0 = _level0("k21")
    _, nbr_np = orb.subm_rulebook(idx0, (40, 1600, 1408))
    n = len(idx0)
    cap = n + 37
    nbr = torch.full((cap, 27), -1, dtype=torch.int32, device=dev)
    nbr[:n] = torch.from_numpy(nbr_np).to(dev)
    n_ptr = torch.tensor([n], dtype=torch.int32, device=dev)
    g = torch.Generator().manual_seed(0)
    for cin, cout in ((64, 64), (16, 32), (4, 16)):
        x = torch.zeros(cap, cin, device=dev)
        dy = torch.zeros(cap, cout, device=dev)
        x[:n] = torch.randn(n, cin, generator=g).to(dev)
        dy[:n] = torch.randn(n, cout, generator=g).to(dev)
        new = K.spconv_bwd_weight(x, dy, nbr, n_ptr, cap, cin, cout)
        assert torch.equal(new, K.spconv_bwd_weight(x, dy, nbr, n_ptr, cap, cin, cout))
        with K.default_cfg(spconv=32):
            old = K.spconv_bwd_weight(x, dy, nbr, n_ptr, cap, cin, cout)
        err = float((new - old).abs().max()) / float(old.abs().max())
        assert err < 1e-5, (cin, cout, err)


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 32), (64, 64)])
def test_submanifold_data_gradient_on_forward_rulebook(dev, cin, cout):
    """dx of a submanifold layer through the FORWARD table with the offset-reversed transposed weights
    (nbr[j][k] = i <=> nbr[i][26-k] = j) against the transposed-table formulation and against torch autograd through the
    oracle's gather / mm / index_add."""
    from sassd.autograd import SparseConvFn
    idx = _level0("small", 2)
    shape = (40, 1600, 1408)
    idx, _, shape = orb.conv_rulebook(idx, shape, 1)
    idx, _, shape = orb.conv_rulebook(idx, shape, 1)
    _, nbr = orb.subm_rulebook(idx, shape)
    n = len(nbr)
    g = torch.Generator().manual_seed(cin * 3 + cout)
    x = torch.randn(n, cin, generator=g)
    w = torch.randn(27, cin, cout, generator=g) * 0.2
    dy = torch.randn(n, cout, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    onets.sparse_conv(xr, nbr, wr).backward(dy)
    nb = torch.from_numpy(nbr).to(dev)
    got = {}
    for on_fwd in (True, False):
        SparseConvFn.subm_on_forward_table = on_fwd
        try:
            xd, wd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
            y = SparseConvFn.apply(xd, wd, nb, n, K.spconv_pack_weight(wd.detach()), True)
            y.backward(dy.to(dev))
            got[on_fwd] = (xd.grad.cpu(), wd.grad.cpu())
        finally:
            SparseConvFn.subm_on_forward_table = True
    tol = 2e-4 * max(1.0, xr.grad.abs().max().item())
    assert (got[True][0] - xr.grad).abs().max().item() < tol
    assert (got[False][0] - xr.grad).abs().max().item() < tol
    assert (got[True][1] - wr.grad).abs().max().item() < 2e-4 * max(1.0, wr.grad.abs().max().item())


def test_packed_weight_cache_and_address_reuse(dev):
    """Round-5 regression: the transposed-pack cache of the sparse data gradient is keyed by the weight's address + shape and
    validated by (version, address, device, generation).  A fresh weight on the address of a freed one has the SAME key and
    generation -- before the cache pinned its source tensors, the second layer below was differentiated with the first
    layer's weights (relative error 1.4-2.5 in the float64 guard when the suite ran in one process).  Four same-shape layers
    are created, used and dropped in turn; every data gradient must match its own float64 reference."""
    from sassd.autograd import SparseConvFn
    import gc
    idx = _level0("small", 4)
    _, nbr = orb.subm_rulebook(idx, (40, 1600, 1408))
    n = len(nbr)
    nb = torch.from_numpy(nbr).to(dev)
    seen = []
    for trial in range(4):
        g = torch.Generator().manual_seed(50 + trial)
        x = torch.randn(n, 16, generator=g)
        w = torch.randn(27, 16, 16, generator=g) * 0.2
        dy = torch.randn(n, 16, generator=g)
        xr, wr = x.double().requires_grad_(True), w.double()
        _f64_sparse_conv(xr, nbr, wr).backward(dy.double())
        xd, wd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
        seen.append(wd.data_ptr())
        SparseConvFn.apply(xd, wd, nb, n, K.spconv_pack_weight(wd.detach()), True).backward(dy.to(dev))
        torch.cuda.synchronize()
        err = float((xd.grad.double().cpu() - xr.grad).norm() / xr.grad.norm())
        assert err < 1e-5, (trial, err, seen)
        del xd, wd
        gc.collect()


def _f64_sparse_conv(x, nbr, w):
    """float64 gather / mm / index_add over a gather table (the oracle's sparse conv restated in double)"""
    nbt = torch.from_numpy(nbr).long()
    y = torch.zeros(len(nbr), w.shape[2], dtype=torch.float64)
    for k in range(w.shape[0]):
        o = torch.nonzero(nbt[:, k] >= 0).view(-1)
        if o.numel():
            y = y.index_add(0, o, x[nbt[o, k]] @ w[k])
    return y


def test_sparse_layers_k21x2_vs_float64(dev):
    """Kernel-level guard under the whole-step bars: EVERY sparse layer shape of the network on the rulebooks of the
    bench's training workload (two K21 frames: 32 k / 37 k / 29 k / 27 k rows), through the DEFAULT dispatch -- forward, data
    gradient (forward-table form for the submanifold layers, transposed table for the strided ones) and weight gradient --
    against a float64 gather / matmul reference: 1e-5 relative L2 each (fp32 sums of <= 27 x 64 products; measured ~1e-6).  A
    dropped pair in a split offset or a wrong unit boundary in the balanced partition is a 1e-2 error here; the whole-step
    tests alone would only see it as "fp32 noise"."""
    from sassd.autograd import SparseConvFn
    idx = _level0("k21", 0, 2)
    shape = (40, 1600, 1408)
    levels = []                                           # (rows idx, subm table, strided table into the level, input rows)
    n_prev = None
    nbr_down = None
    for lvl in range(4):
        if lvl > 0:
            n_prev = len(idx)
            idx, nbr_down, shape = orb.conv_rulebook(idx, shape, 2)
        _, nbr_s = orb.subm_rulebook(idx, shape)
        levels.append((nbr_s, nbr_down, n_prev))
    layers = [(0, "subm", 4, 16), (0, "subm", 16, 16), (1, "down", 16, 32), (1, "subm", 32, 32), (2, "down", 32, 64),
              (2, "subm", 64, 64), (3, "down", 64, 64), (3, "subm", 64, 64), (3, "1x1", 64, 64)]
    worst = {}
    for lvl, kind, cin, cout in layers:
        nbr_s, nbr_d, n_in_down = levels[lvl]
        nbr = nbr_s if kind == "subm" else (nbr_d if kind == "down" else None)
        n_out = len(nbr_s)
        n_in = n_in_down if kind == "down" else n_out
        g = torch.Generator().manual_seed(100 * cin + cout + lvl)
        K_ = 1 if kind == "1x1" else 27
        x = torch.randn(n_in, cin, generator=g)
        w = torch.randn(K_, cin, cout, generator=g) * (0.5 / (cin ** 0.5))
        dy = torch.randn(n_out, cout, generator=g)
        xr, wr = x.double().requires_grad_(cin >= 16), w.double().requires_grad_(True)
        ref = (xr @ wr[0]) if nbr is None else _f64_sparse_conv(xr, nbr, wr)
        ref.backward(dy.double())
        xd, wd = x.to(dev).requires_grad_(cin >= 16), w.to(dev).requires_grad_(True)
        nb = None if nbr is None else torch.from_numpy(nbr).to(dev)
        y = SparseConvFn.apply(xd, wd, nb, n_out, K.spconv_pack_weight(wd.detach()), kind == "subm")
        y.backward(dy.to(dev))
        torch.cuda.synchronize()
        tag = "%s %d->%d @ %d rows" % (kind, cin, cout, n_out)

        def rel(a, b):
            return float((a.detach().double().cpu() - b).norm() / b.norm())
        worst[tag] = (rel(y, ref.detach()), rel(xd.grad, xr.grad) if cin >= 16 else 0.0, rel(wd.grad, wr.grad))
    print("sparse layers on the K21 x 2 rulebooks vs float64 (forward, data gradient, weight gradient):",
          {k: tuple("%.1e" % e for e in v) for k, v in worst.items()})
    bad = {k: v for k, v in worst.items() if not max(v) < 1e-5}
    assert not bad, bad


@pytest.mark.parametrize("legacy", [False, True])
def test_spconv_input_layer_kernel(dev, legacy):
    """The 4-channel input layer (cmn.py:197, SubMConv3d(4, 16, 3)) on the row-per-16-threads kernel (and on the
    register-stationary MFMA kernel it replaced, debug bit 8): full K21 level-0 table, device row count below the capacity,
    rows past it untouched, folded scale / shift / ReLU."""
    idx = _level0("k21", 0)
    _, nbr = orb.subm_rulebook(idx, (40, 1600, 1408))
    n = len(nbr)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, 4, generator=g)
    w = torch.randn(27, 4, 16, generator=g) * 0.3
    scale, shift = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.1
    raw = onets.sparse_conv(x, nbr, w)
    cap = n + 777
    nb = torch.full((cap, 27), -1, dtype=torch.int32, device=dev)
    nb[:n] = torch.from_numpy(nbr).to(dev)
    nptr = torch.tensor([n], dtype=torch.int32, device=dev)
    with K.default_cfg(spconv=256 if legacy else 0):
        y = torch.full((cap, 16), 9.0, device=dev)
        K.spconv_fwd(x.to(dev), nb, nptr, cap, K.spconv_pack_weight(w.to(dev)), 27, 4, 16, scale.to(dev), shift.to(dev), True, y)
    ref = torch.relu(raw * scale + shift)
    assert (y[:n].cpu() - ref).abs().max().item() < 2e-4 * max(1.0, raw.abs().max().item())
    assert bool((y[n:] == 9.0).all())
