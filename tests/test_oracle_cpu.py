"""CPU tests (-m "not gpu"): the oracle against the reference-generated golden fixtures, oracle self-consistency
(sparse conv vs torch conv3d on a dense crop), and the C ABI library surface (load + every declared symbol)."""
import glob
import hashlib
import os
import re

import numpy as np
import pytest
import torch

import sassd
from sassd import synth
from oracle import clib, nets as onets, rulebook as orb
import helpers as H

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "voxelizer_*.npz"))))
def test_voxelizer_oracle_vs_reference_golden(path):
    g = np.load(path)
    if "points" not in g:
        pytest.skip("digest")
    v, c, n = clib.points_to_voxel(g["points"], g["voxel_size"], g["coors_range"], int(g["max_points"]), True,
                                   int(g["max_voxels"]))
    assert np.array_equal(v, g["voxels"]) and np.array_equal(c, g["coors"]) and np.array_equal(n, g["num_points"])


def test_voxelizer_oracle_k21_digest():
    g = np.load(os.path.join(GOLD, "voxelizer_k21_digest.npz"))
    v, c, n = clib.points_to_voxel(synth.k21(0), synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 20000)
    h = hashlib.sha256()
    for a in (v, c, n):
        h.update(np.ascontiguousarray(a).tobytes())
    assert len(c) == int(g["m"]) == 16111
    assert h.hexdigest() == str(g["sha"])


def test_iou_oracle_vs_reference_device_code():
    g = np.load(os.path.join(GOLD, "iou_ref.npz"))
    ov = clib.boxes_overlap_bev(g["a"], g["b"])
    iou = clib.boxes_iou_bev(g["a"], g["b"])
    assert np.abs(ov - g["overlap"]).max() < 1e-6
    assert np.abs(iou - g["iou"]).max() < 1e-6
    if clib.ref() is not None:          # live cross-check where oracle/_ref is present
        rng = np.random.default_rng(9)
        a, b = H.rand_bev_boxes(rng, 30), H.rand_bev_boxes(rng, 30)
        assert np.abs(clib.boxes_iou_bev(a, b) - clib.boxes_iou_bev(a, b, use_ref=True)).max() < 1e-6


def test_head_oracle_vs_reference_functions():
    g = np.load(os.path.join(GOLD, "head_fns.npz"))
    dec = onets.box_decode(torch.from_numpy(g["enc"]), torch.from_numpy(g["anchors"]))
    assert np.array_equal(dec.numpy(), g["dec"])
    assert np.array_equal(onets.boxes3d_to_bev(dec).numpy(), g["bev"])
    img = torch.from_numpy(np.random.default_rng(int(g["img_seed"])).standard_normal((28, 200, 176)).astype(np.float32))
    # PSWarp sampling with identity convs == reference gen_sample_grid + grid_sample on `img`
    params = {"conv0": dict(weight=torch.zeros(28, 28, 3, 3), bn=dict(weight=torch.ones(28), bias=torch.zeros(28),
                                                                     running_mean=torch.zeros(28),
                                                                     running_var=torch.ones(28) - 1e-3)),
              "conv1": dict(weight=torch.eye(28).view(28, 28, 1, 1))}
    params["conv0"]["weight"][torch.arange(28), torch.arange(28), 1, 1] = 1.0
    # relu would clip negatives: shift the image positive and undo
    off = 10.0
    scores, _ = onets.pswarp_forward((img + off).unsqueeze(0), params, [dec])
    # zero padding makes "+off" not exactly removable for border samples; compare only fully inside boxes
    sx, sy = g["sx"], g["sy"]
    inside = ((sx > 1) & (sx < 174) & (sy > 1) & (sy < 198)).all(0)
    assert inside.sum() > 10
    assert np.abs((scores[0].numpy() - off)[inside] - g["score"][inside]).max() < 2e-5


def test_sparse_conv_oracle_vs_dense_conv3d():
    rng = np.random.default_rng(0)
    shape, b = (8, 16, 12), 2
    n = 300
    lin = np.sort(rng.choice(b * 8 * 16 * 12, n, replace=False))
    idx = np.stack([lin // (8 * 16 * 12), (lin // (16 * 12)) % 8, (lin // 12) % 16, lin % 12], 1).astype(np.int32)
    perm = rng.permutation(n)
    idx = idx[perm]                                   # unsorted rows, like voxelizer output
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, 5, generator=g)
    w = torch.randn(27, 5, 7, generator=g)
    dense = torch.zeros(b, 5, *shape)
    ii = torch.from_numpy(idx).long()
    dense[ii[:, 0], :, ii[:, 1], ii[:, 2], ii[:, 3]] = x
    wt = w.view(3, 3, 3, 5, 7).permute(4, 3, 0, 1, 2)
    # submanifold: same active set
    _, nbr = orb.subm_rulebook(idx, shape)
    y = onets.sparse_conv(x, nbr, w)
    yd = torch.nn.functional.conv3d(dense, wt, padding=1)
    assert torch.allclose(y, yd[ii[:, 0], :, ii[:, 1], ii[:, 2], ii[:, 3]], atol=1e-4)
    # strided: active outputs = any active input in the receptive field, ascending order
    oi, nbr2, oshape = orb.conv_rulebook(idx, shape, b)
    y2 = onets.sparse_conv(x, nbr2, w)
    yd2 = torch.nn.functional.conv3d(dense, wt, stride=2, padding=1)
    occ = torch.nn.functional.conv3d((dense.abs().sum(1, keepdim=True) > 0).float(), torch.ones(1, 1, 3, 3, 3), stride=2, padding=1) > 0
    assert tuple(yd2.shape[2:]) == oshape
    oo = torch.from_numpy(oi).long()
    assert int(occ.sum()) == len(oi)
    assert bool(occ[oo[:, 0], 0, oo[:, 1], oo[:, 2], oo[:, 3]].all())
    assert torch.allclose(y2, yd2[oo[:, 0], :, oo[:, 1], oo[:, 2], oo[:, 3]], atol=1e-4)
    lin_o = ((oo[:, 0] * oshape[0] + oo[:, 1]) * oshape[1] + oo[:, 2]) * oshape[2] + oo[:, 3]
    assert bool((lin_o[1:] > lin_o[:-1]).all())
    # pairs form
    pairs, num = orb.nbr_to_pairs(nbr2)
    assert int(num.sum()) == int((nbr2 >= 0).sum())


def test_k21_layer_sizes_match_survey():
    _, c, _ = clib.points_to_voxel(synth.k21(0), synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 20000)
    idx = np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)
    shape = (40, 1600, 1408)
    sizes, pairs = [len(idx)], []
    for _ in range(3):
        _, nbr = orb.subm_rulebook(idx, shape)
        pairs.append(int((nbr >= 0).sum()))
        idx, nbr, shape = orb.conv_rulebook(idx, shape, 1)
        pairs.append(int((nbr >= 0).sum()))
        sizes.append(len(idx))
    _, nbr = orb.subm_rulebook(idx, shape)
    pairs.append(int((nbr >= 0).sum()))
    assert sizes == [16111, 18355, 14579, 13287]
    assert pairs == [58581, 37543, 103841, 45840, 104955, 56270, 176667]


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    from sassd import _C
    hdr = open(os.path.join(ROOT, "include", "sassd.h")).read()
    declared = set(re.findall(r"\b(sassd_\w+)\s*\(", hdr))
    assert declared == set(_C.EXPORTS)
    lib = _C.lib()                         # dlopen + resolve every symbol (no compute without a GPU)
    assert b"gfx950" in lib.sassd_version()
    assert lib.sassd_voxelize_workspace_bytes(21500, 5) > 0
    assert lib.sassd_conv2d_packed_floats(256, 28, 3) == 256 * 9 * 32
    assert lib.sassd_conv2d_packed_floats(28, 28, 1) == 32 * 32


def test_eval_rotate_iou_oracle_vs_reference_functions():
    """oracle rotate_iou_eval vs the reference's own numba device functions executed as plain Python
    (tests/golden/make_golden_eval.py): all three criteria, incl. coincident / contained / disjoint boxes."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval_iou_ref.npz"))
    for crit in (-1, 0, 1):
        got = clib.rotate_iou_eval(G["boxes"], G["query"], crit)
        assert np.abs(got - G["iou_%d" % crit]).max() < 2e-6, crit
    assert abs(float(G["iou_-1"][9, 9]) - 0.25) < 1e-6            # contained box of a quarter of the area
    assert float(G["iou_-1"][10, 10]) == 0.0                       # disjoint


def test_pts_in_boxes3d_oracle_vs_reference_cpp():
    """oracle pts_in_boxes3d vs the reference's own C++ (points_op.cpp compiled from /root/reference by
    tests/golden/make_golden_points_op.py): flags and centre offsets bit-exact, incl. overlapping boxes (last box wins)
    and the 10 m coarse-reject distance."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pts_in_boxes_ref.npz"))
    flag, reg = clib.pts_in_boxes3d(G["pts"], G["boxes"])
    assert np.array_equal(flag, G["flag"]) and np.array_equal(reg, G["reg"])
    assert int(G["flag"].max(0).sum()) > 300


def test_interpolation_oracle_vs_reference_kernels():
    """oracle three_nn / three_interpolate / grad vs the reference's own CUDA kernels compiled for the host
    (oracle/build.py::build_ref_interp -> tests/golden/interp_ref.npz): bit-exact."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "interp_ref.npz"))
    d, i = clib.three_nn(G["unknown"], G["known"])
    assert np.array_equal(i, G["idx"]) and np.array_equal(d, G["dist2"])
    assert np.array_equal(clib.three_interpolate(G["feat"], G["idx"], G["weight"]), G["out"])
    assert np.array_equal(clib.three_interpolate_grad(G["grad_out"], G["idx"], G["weight"], len(G["known"])),
                          G["grad_points"])


def test_anchor_mask_and_near_bbox_vs_reference_functions():
    """oracle anchors_mask (+ sassd.anchors.rbbox2d_to_near_bbox) vs the reference's own numba functions run as plain
    Python (tests/golden/make_golden_anchor_mask.py): masks bit-exact for two frames and both thresholds."""
    import os
    from oracle import nets as onets
    from sassd import anchors as A, synth
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "anchor_mask_ref.npz"))
    an = A.AnchorGeneratorStride(sizes=[1.6, 3.9, 1.56], anchor_strides=[.4, .4, 1.], anchor_offsets=[.2, -39.8, -1.78],
                                 rotations=[0, 1.57])([1, 200, 176]).reshape(-1, 7)
    bv = A.rbbox2d_to_near_bbox(an[:, [0, 1, 3, 4, 6]]).astype(np.float32)
    assert np.array_equal(bv, G["anchors_bv"])
    for name in ("small", "k17"):
        for thr in (1, 0):
            m = onets.anchors_mask(G["coors_" + name], bv, synth.KITTI_VOXEL, synth.KITTI_RANGE, (1408, 1600, 40), thr)
            ref = np.unpackbits(G["mask_%s_thr%d" % (name, thr)])[:len(bv)].astype(bool)
            assert np.array_equal(m, ref), (name, thr, int((m != ref).sum()))


def test_guided_anchors_inference_vs_reference_method():
    """oracle guided_anchors (inference path) vs the reference's own SSDRotateHead.get_guided_anchors called without
    ground truth (tests/golden/make_golden_train.py), one class and the three-class variant."""
    import os
    from oracle import nets as onets
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_fns.npz"))
    t = lambda k: torch.from_numpy(G[k])                                  # noqa: E731
    anc = torch.stack([t("anchors"), t("a2")])
    msk = torch.stack([t("anchor_mask"), t("m2")])
    got = onets.guided_anchors(t("rpn_box"), t("rpn_cls"), t("rpn_dir"), anc, msk, 1, 0.1)
    for i in range(2):
        assert got[i][0].shape == G["test_guided%d" % i].shape
        assert np.abs(got[i][0].numpy() - G["test_guided%d" % i]).max() < 1e-6
        assert np.array_equal(got[i][1].numpy(), G["test_guided_labels%d" % i])
    got3 = onets.guided_anchors(t("mc_box"), t("mc_cls"), t("mc_dir"), t("mc_anchors"), t("mc_mask"), 3, 0.1)
    assert np.abs(got3[0][0].numpy() - G["mc_guided"]).max() < 1e-6
    assert np.array_equal(got3[0][1].numpy(), G["mc_labels"]) and set(np.unique(G["mc_labels"])) == {0, 1, 2}


def test_rescore_vs_reference_method():
    """oracle rescore (sigmoid, 0.3 score threshold, BEV boxes, rotated NMS at 0.1, gather) vs the reference's own
    PSWarpHead.get_rescore_bboxes -> rotate_nms_torch -> iou3d_utils.nms_gpu Python, with the compiled nms step served
    by a greedy pass over the reference's own iou_bev device function (tests/golden/make_golden_train.py)."""
    import os
    from oracle import nets as onets
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_fns.npz"))
    t = lambda k: torch.from_numpy(G[k])                                  # noqa: E731
    d = onets.rescore(t("rs_boxes0"), t("rs_logits0"), t("rs_labels0"), 0.3, 0.1)
    assert d is not None and len(d[0]) == len(G["rs_det_boxes"]) > 10
    assert np.abs(d[0] - G["rs_det_boxes"]).max() < 1e-6 and np.abs(d[1] - G["rs_det_scores"]).max() < 1e-6
    assert np.array_equal(d[2], G["rs_det_labels"])
    assert onets.rescore(t("rs_boxes0")[:700], t("rs_logits1"), t("rs_labels0")[:700], 0.3, 0.1) is None
    assert onets.rescore(t("rs_boxes0")[:0], t("rs_logits0")[:0], t("rs_labels0")[:0], 0.3, 0.1) is None


def test_voxel_mean_vs_reference_simplevoxel():
    """oracle voxel_mean vs the reference's own SimpleVoxel.forward (tests/golden/make_golden_voxel_mean.py): bit-exact
    for max_points 3 / 5 / 8."""
    import os
    here = os.path.join(os.path.dirname(__file__), "golden")
    R = np.load(os.path.join(here, "voxel_mean_ref.npz"))
    for case in ("small", "oob", "dense_t3", "t8"):
        G = np.load(os.path.join(here, "voxelizer_%s.npz" % case))
        assert np.array_equal(clib.voxel_mean(G["voxels"], G["num_points"]), R[case]), case


def test_submanifold_rulebook_is_its_own_transpose_under_offset_reversal():
    """For a submanifold layer nbr[j][k] = i  <=>  nbr[i][26 - k] = j (offset k and 26 - k are opposite), so the data
    gradient is the forward operator on the SAME rulebook with the offset-reversed, transposed weights:
    dx[i] = sum_k dy[nbr[i][k]] @ W[26 - k]^T.  (lead 1: no transposed rulebook for those layers.)"""
    import numpy as np
    import torch
    from oracle import nets, rulebook
    rng = np.random.default_rng(5)
    shape = (6, 9, 8)
    idx = np.stack([rng.integers(0, 2, 90), rng.integers(0, shape[0], 90), rng.integers(0, shape[1], 90),
                    rng.integers(0, shape[2], 90)], 1).astype(np.int32)
    idx = np.unique(idx, axis=0)
    idx, nbr = rulebook.subm_rulebook(idx, shape)
    n = idx.shape[0]
    for j in range(n):
        for k in range(27):
            i = nbr[j, k]
            if i >= 0:
                assert nbr[i, 26 - k] == j
    assert (nbr[:, 13] == np.arange(n)).all()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, 8, generator=g, requires_grad=True)
    w = torch.randn(27, 8, 5, generator=g)
    dy = torch.randn(n, 5, generator=g)
    nb = torch.from_numpy(nbr.astype(np.int64))
    nets.sparse_conv(x, nb, w).backward(dy)
    dx = nets.sparse_conv(dy, nb, w.flip(0).transpose(1, 2).contiguous())
    assert torch.allclose(dx, x.grad, rtol=1e-5, atol=1e-5)
