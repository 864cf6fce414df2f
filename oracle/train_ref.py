"""CPU oracle of ONE SA-SSD training step (TEST INFRASTRUCTURE ONLY -- never imported by the product).

A plain torch-CPU fp32 restatement with autograd of what the reference computes in
mmdet/models/detectors/single_stage.py:75-108 (forward_train), following

  VxNet / BEVNet / aux head, train-mode BN     mmdet/models/necks/cmn.py:106-135,192-282
  aux targets + aux loss                       cmn.py:45-104
  tensor2points                                mmdet/core/bbox/transforms.py:218-223
  anchor targets                               mmdet/core/bbox3d/target_ops.py:139-277
  nearest / rotated-3D similarity              mmdet/ops/iou3d/iou3d_utils.py:9-45,79-111
  rpn loss, guided anchors (GT prepended)      mmdet/models/single_stage_heads/ssd_rotate_head.py:128-388
  PSWarp forward + rescoring loss              ssd_rotate_head.py:431-490
  focal / smooth-L1 / CE                       mmdet/core/loss/losses.py:13-96

It is written independently of sassd.train_ops (numpy target assignment, closed-form losses) and pinned against the
same reference-generated golden vectors (tests/test_train_cpu.py::test_train_ref_pieces_pinned).  Sparse convolutions gather
through the oracle rulebooks, 3-NN / point-in-box / rotated overlap come from oracle/sassd_oracle.c.

Two switches turn the fp32 step into the two ARBITERS the GPU step is judged by (round 5):

  dtype=torch.float64   every floating-point operation of the step in double (weights, inputs and the C helpers' fp32
                        results are the same numbers, widened): the reference value two fp32 implementations that sum in
                        different orders are both measured against -- a GPU gradient passes when it is no farther from
                        this one than the fp32 oracle itself is (x a stated factor).
  bf16=("bev",)         BASELINE configs[2] trains in bf16: the operands of the dense convolutions are rounded to bf16
                        (round-to-nearest-even, torch.bfloat16) exactly where the HIP kernels round them
                        (sa-ssd_amd/autograd.py Conv2dFn, csrc/conv2d_bf16.hip, csrc/conv2d_wgrad.hip): forward of a 3x3
                        layer with Cout % 32 == 0 -- x and w; data gradient of a 3x3 layer with Cin % 32 == 0 -- dy and w;
                        EVERY weight gradient (1x1 layers and heads included) -- x and dy; map width >= 16 and % 4 == 0
                        (% 2 for the weight gradient).  Products and sums stay in `dtype`; biases, BatchNorm, ReLU and
                        everything outside the dense convolutions are untouched, as on the GPU.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import clib
from . import nets
from . import rulebook as rb

EPS = 1e-3


# ---- layers (train mode: batch statistics) ------------------------------------------------------------------------
def bn_train(x, w, b):
    dims = [0] + list(range(2, x.dim()))
    shp = (1, -1) + (1,) * (x.dim() - 2)
    mu = x.mean(dims, keepdim=True)
    var = ((x - mu) ** 2).mean(dims, keepdim=True)
    return (x - mu) / torch.sqrt(var + EPS) * w.view(shp) + b.view(shp)


def round_bf16(t):
    """round-to-nearest-even to bf16, returned in t's own dtype (what v_cvt_pk_bf16_f32 / (__bf16) do to an MFMA operand)"""
    return t.to(torch.bfloat16).to(t.dtype)


def _conv1x1_bf16_ok(cin, cout, hw):
    """restates sassd_conv1x1_bf16_supported (csrc/conv1x1_bf16.hip): K <= 256 with at most its last 16-channel step ragged"""
    ksteps = 1 if cin <= 16 else 2 if cin <= 32 else 4 if cin <= 64 else 8 if cin <= 128 else 16
    return 1 <= cin <= 256 and cin > 16 * (ksteps - 1) and cout >= 1 and hw >= 4 and hw % 4 == 0


def bf16_conv_rule(cin, cout, ks, w_, h_):
    """(forward, data gradient, weight gradient) -> does the GPU step under set_bev_precision("bf16") round that launch's
    operands?  Restates sassd_conv2d_bf16_supported (csrc/conv2d_bf16.hip), sassd_conv1x1_bf16_supported (round 6: the 1x1
    layers' forward and data gradient run on the bf16 MFMA too) and Conv2dFn / conv2d_bwd_weight's dispatch."""
    if ks == 1:
        return (_conv1x1_bf16_ok(cin, cout, h_ * w_), _conv1x1_bf16_ok(cout, cin, h_ * w_), w_ % 2 == 0)
    shape_ok = w_ >= 16 and w_ % 4 == 0
    # (round 6: a 3x3 forward with Cout % 32 != 0 -- the part-sensitive head's 256 -> 28 -- runs on the bf16 kernel with zero
    # weight rows up to the next multiple of 32, sassd.autograd.bf16_cout_pad: every 3x3 forward of a supported map rounds)
    return (ks == 3 and shape_ok, ks == 3 and cin % 32 == 0 and shape_ok, w_ % 2 == 0)


class RoundedConv2d(torch.autograd.Function):
    """conv2d (stride 1, `pad`) whose three products each see bf16-rounded operands when `rule` says so"""

    @staticmethod
    def forward(ctx, x, w, b, pad, rule):
        ctx.save_for_backward(x, w)
        ctx.pad, ctx.rule, ctx.has_b = pad, rule, b is not None
        return F.conv2d(round_bf16(x), round_bf16(w), b, 1, pad) if rule[0] else F.conv2d(x, w, b, 1, pad)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        _, rd, rw = ctx.rule
        dyr = round_bf16(dy) if (rd or rw) else dy
        dx = torch.nn.grad.conv2d_input(x.shape, round_bf16(w) if rd else w, dyr if rd else dy, padding=ctx.pad)
        dw = torch.nn.grad.conv2d_weight(round_bf16(x) if rw else x, w.shape, dyr if rw else dy, padding=ctx.pad)
        return dx, dw, (dy.sum((0, 2, 3)) if ctx.has_b else None), None, None


def conv2d(x, w, b, pad, bf16):
    if not bf16:
        return F.conv2d(x, w, b, 1, pad)
    return RoundedConv2d.apply(x, w, b, pad, bf16_conv_rule(w.shape[1], w.shape[0], w.shape[2], x.shape[3], x.shape[2]))


def gather_conv(x, nbr, w):
    """y[o] = sum_k x[nbr[o,k]] @ w[k]; -1 entries read a zero row."""
    n = x.shape[0]
    xp = torch.cat([x, x.new_zeros(1, x.shape[1])], 0)
    idx = torch.as_tensor(np.where(nbr < 0, n, nbr), dtype=torch.int64)
    y = x.new_zeros(nbr.shape[0], w.shape[2])
    for k in range(nbr.shape[1]):
        if (nbr[:, k] >= 0).any():
            y = y + xp[idx[:, k]] @ w[k]
    return y


# ---- losses ---------------------------------------------------------------------------------------------------------
def focal_sum(x, t, w, gamma=2.0, alpha=0.25):
    p = torch.sigmoid(x)
    ce = torch.clamp(x, min=0) - x * t + torch.log1p(torch.exp(-torch.abs(x)))
    mod = ((1 - p) * t + p * (1 - t)) ** gamma
    return (ce * mod * (alpha * t + (1 - alpha) * (1 - t)) * w).sum()


def smooth_l1_sum(a, b, w, beta):
    d = (a - b).abs()
    return (torch.where(d < beta, d * d * (0.5 / beta), d - 0.5 * beta) * w).sum()


# ---- similarity + target assignment (numpy, fp32) -------------------------------------------------------------------
def near_boxes(b):
    b = np.asarray(b, np.float32)
    r = b[:, 6]
    lim = np.abs(r - np.floor(r / np.float32(math.pi) + np.float32(0.5)) * np.float32(math.pi))
    swap = lim > np.float32(math.pi / 4)
    w = np.where(swap, b[:, 4], b[:, 3])
    l = np.where(swap, b[:, 3], b[:, 4])
    return np.stack([b[:, 0] - w / 2, b[:, 1] - l / 2, b[:, 0] + w / 2, b[:, 1] + l / 2], 1).astype(np.float32)


def nearest_iou(a, b):
    a, b = near_boxes(a), near_boxes(b)
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    br = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip(br - lt, 0, None)
    ov = wh[..., 0] * wh[..., 1]
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return ov / (aa[:, None] + ab[None] - ov)


def rotated_iou3d(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    ov = clib.boxes_overlap_bev(nets.boxes3d_to_bev(torch.from_numpy(a)).numpy(),
                                nets.boxes3d_to_bev(torch.from_numpy(b)).numpy())
    top = np.minimum((a[:, 2] + a[:, 5])[:, None], (b[:, 2] + b[:, 5])[None])
    bot = np.maximum(a[:, 2][:, None], b[:, 2][None])
    o3 = ov * np.clip(top - bot, 0, None)
    va = (a[:, 3] * a[:, 4] * a[:, 5])[:, None]
    vb = (b[:, 3] * b[:, 4] * b[:, 5])[None]
    return o3 / np.clip(va + vb - o3, np.float32(1e-7), None)


def box_encode(g, a):
    g, a = np.asarray(g, np.float32), np.asarray(a, np.float32)
    diag = np.sqrt(a[:, 4] ** 2 + a[:, 3] ** 2)
    zg, za = g[:, 2] + g[:, 5] / 2, a[:, 2] + a[:, 5] / 2
    return np.stack([(g[:, 0] - a[:, 0]) / diag, (g[:, 1] - a[:, 1]) / diag, (zg - za) / a[:, 5],
                     np.log(g[:, 3] / a[:, 3]), np.log(g[:, 4] / a[:, 4]), np.log(g[:, 5] / a[:, 5]),
                     g[:, 6] - a[:, 6]], 1).astype(np.float32)


def assign(all_anchors, mask, gt, gt_cls, pos_thr, neg_thr, sim):
    """-> labels [A] int64 (-1 ignore / 0 negative / class), targets [A,7], per-anchor best overlap (masked rows)."""
    all_anchors = np.asarray(all_anchors, np.float32)
    sel = np.arange(len(all_anchors)) if mask is None else np.nonzero(np.asarray(mask))[0]
    anc = all_anchors[sel]
    n = len(anc)
    lab = np.full(n, -1, np.int64)
    tar = np.zeros((n, 7), np.float32)
    best = np.zeros(n, np.float32)
    if len(gt) and n:
        ov = sim(anc, gt)
        arg = ov.argmax(1)
        best = ov[np.arange(n), arg]
        gmax = ov.max(0)
        gmax = np.where(gmax == 0, np.float32(-1), gmax)
        forced = np.nonzero((ov == gmax[None]).any(1))[0]
        lab[best >= pos_thr] = gt_cls[arg[best >= pos_thr]]
        lab[best < neg_thr] = 0
        lab[forced] = gt_cls[arg[forced]]
        fg = np.nonzero(lab > 0)[0]
        tar[fg] = box_encode(gt[arg[fg]], anc[fg])
    else:
        lab[:] = 0
    full_l = np.full(len(all_anchors), -1, np.int64)
    full_t = np.zeros((len(all_anchors), 7), np.float32)
    full_l[sel], full_t[sel] = lab, tar
    return full_l, full_t, best


# ---- the step ---------------------------------------------------------------------------------------------------------
VX = [("conv0.0", "conv0.1", "subm", "subm0"), ("conv0.3", "conv0.4", "subm", "subm0"),
      ("down0.0", "down0.1", "down", None),
      ("conv1.0", "conv1.1", "subm", "subm1"), ("conv1.3", "conv1.4", "subm", "subm1"),
      ("down1.0", "down1.1", "down", None),
      ("conv2.0", "conv2.1", "subm", "subm2"), ("conv2.3", "conv2.4", "subm", "subm2"),
      ("conv2.6", "conv2.7", "subm", "subm2"),
      ("down2.0", "down2.1", "down", None),
      ("conv3.0", "conv3.1", "subm", "subm3"), ("conv3.3", "conv3.4", "subm", "subm3"),
      ("conv3.6", "conv3.7", "subm", "subm3"),
      ("extra_conv.0", "extra_conv.1", "1x1", None)]


def decode(enc, a):
    za = a[..., 2] + a[..., 5] / 2
    diag = torch.sqrt(a[..., 4] ** 2 + a[..., 3] ** 2)
    w, l, h = torch.exp(enc[..., 3]) * a[..., 3], torch.exp(enc[..., 4]) * a[..., 4], torch.exp(enc[..., 5]) * a[..., 5]
    return torch.stack([enc[..., 0] * diag + a[..., 0], enc[..., 1] * diag + a[..., 1],
                        enc[..., 2] * a[..., 5] + za - h / 2, w, l, h, enc[..., 6] + a[..., 6]], -1)


def train_step(sd, feats, coors, batch_size, sparse_shape, gt_bboxes, gt_types, class_names, anchors, anchors_mask,
               assign_cfg, anchor_thr=0.1, extra_thr=0.7, grid_offsets=(0., 40.), featmap_stride=0.4,
               aux_offset=(0., -40., -3.), aux_voxel_size=(.05, .05, .1), grad_exclude=(), dtype=torch.float32, bf16=(),
               guided_sel=None):
    """sd: detector state_dict (CPU tensors); feats [N,4] voxel means; coors [N,4] (b,z,y,x); gt_bboxes: list of
    [G,7]; gt_types: list of str arrays; anchors / anchors_mask: {class: [B, A, 7] / [B, A]};
    assign_cfg: {class: (pos_thr, neg_thr)}.  Returns (losses {name: float}, grads {param name: tensor},
    extras).  aux_offset / aux_voxel_size: the auxiliary head's voxel-centre geometry, cmn.py:121-127 literals by default
    (KITTI: offset (0, -40, -3), level voxel sizes 2 / 4 / 8 x (.05, .05, .1)).
    dtype / bf16: see the module header (float64 arbiter; "bev" in bf16 = rounded dense-conv operands).
    guided_sel: None, or per sample the int64 indices (into the sample's MASKED anchors, ascending) of the anchors that
    pass the guided-anchor threshold -- teacher-forces the one discrete decision of the step that depends on network
    outputs, so that two implementations whose scores differ by more than any threshold margin (bf16) can still be
    compared on the full objective; extras["guided_sel"] returns the selection this run differentiated, extras["guided_free"]
    the one its own scores make at anchor_thr, extras["masked_top"] those scores (per sample, masked anchors, float64)."""
    dt = dtype
    rbev = "bev" in bf16
    P = {k: v.detach().clone().to(dt).requires_grad_(v.dtype.is_floating_point and "running" not in k)
         for k, v in sd.items() if torch.is_tensor(v) and v.dtype.is_floating_point}
    B = batch_size
    x = torch.as_tensor(feats, dtype=torch.float32).to(dt)
    idx = np.asarray(coors, np.int32)
    shape = tuple(sparse_shape)
    books, middle = {}, []
    for wn, bn, kind, key in VX:
        w = P["neck.backbone.%s.weight" % wn]
        if kind == "subm":
            if key not in books:
                books[key] = rb.subm_rulebook(idx, shape)[1]
            x = gather_conv(x, books[key], w.reshape(27, w.shape[3], w.shape[4]))
        elif kind == "down":
            oi, nbr, oshape = rb.conv_rulebook(idx, shape, B)
            x = gather_conv(x, nbr, w.reshape(27, w.shape[3], w.shape[4]))
            idx, shape = oi, oshape
        else:
            x = x @ w.reshape(w.shape[3], w.shape[4])
        x = torch.relu(bn_train(x, P["neck.backbone.%s.weight" % bn], P["neck.backbone.%s.bias" % bn]))
        if wn in ("conv1.3", "conv2.6", "conv3.6"):
            middle.append((x, idx.copy()))
    d, h, w_ = shape
    dense = x.new_zeros(B, d, h, w_, x.shape[1])
    ii = torch.as_tensor(idx, dtype=torch.int64)
    dense = dense.index_put((ii[:, 0], ii[:, 1], ii[:, 2], ii[:, 3]), x)
    y = dense.permute(0, 4, 1, 2, 3).reshape(B, -1, h, w_)
    conv6 = None
    for i in range(8):
        wt = P["neck.fcn.conv%d.weight" % i]
        y = conv2d(y, wt, None, 1 if wt.shape[-1] == 3 else 0, rbev)
        y = torch.relu(bn_train(y, P["neck.fcn.bn%d.weight" % i], P["neck.fcn.bn%d.bias" % i]))
        if i == 6:
            conv6 = y
    losses = {}
    # -- auxiliary head
    pm = np.concatenate([np.asarray(coors, np.float32)[:, :1], np.asarray(feats, np.float32)[:, :3]], 1)
    ps = []
    for (mf, mi), mult in zip(middle, (2, 4, 8)):
        vsz, off = np.asarray([v * mult for v in aux_voxel_size], np.float32), np.asarray(aux_offset, np.float32)
        known = mi.astype(np.float32)
        known[:, 1:] = mi[:, [3, 2, 1]].astype(np.float32) * vsz + off + np.float32(.5) * vsz
        d2, nn = clib.three_nn(pm, known)
        rec = 1.0 / (torch.sqrt(torch.from_numpy(d2).to(dt)) + 1e-8)
        wgt = rec / rec.sum(1, keepdim=True)
        nn = torch.from_numpy(nn.astype(np.int64))
        ps.append((mf[nn] * wgt[..., None]).sum(1))
    pw = torch.cat(ps, -1) @ P["neck.point_fc.weight"].t()
    pcls, preg = pw @ P["neck.point_cls.weight"].t(), pw @ P["neck.point_reg.weight"].t()
    labs, offs = [], []
    for b in range(B):
        pts = pm[pm[:, 0] == b, 1:]
        if len(gt_bboxes[b]) and len(pts):
            flag, reg = clib.pts_in_boxes3d(pts, np.asarray(gt_bboxes[b], np.float32))
            labs.append(flag.max(0).astype(np.float32))
            offs.append(reg)
        else:
            labs.append(np.zeros(len(pts), np.float32))
            offs.append(np.zeros((len(pts), 3), np.float32))
    lab, off = torch.from_numpy(np.concatenate(labs)).to(dt), torch.from_numpy(np.concatenate(offs)).to(dt)
    norm = torch.clamp((lab > 0).to(dt).sum(), min=1.0)
    losses["aux_loss_cls"] = focal_sum(pcls.view(-1), lab, torch.ones_like(lab) / norm) / B
    losses["aux_loss_reg"] = smooth_l1_sum(preg, off, ((lab > 0).to(dt) / norm)[:, None], 1 / 9.) / B
    # -- rpn head
    nc = len(class_names)
    outs = []
    for n in ("conv_box", "conv_cls", "conv_dir_cls"):
        o = conv2d(y, P["rpn_head.%s.weight" % n], P["rpn_head.%s.bias" % n], 0, rbev)
        outs.append(o.view(B, nc, -1, h, w_).permute(0, 1, 3, 4, 2))
    box, cls, dr = outs[0].reshape(B, -1, 7), outs[1].reshape(B, -1, nc), outs[2].reshape(B, -1, 2)
    L, Tg = [], []
    for c in class_names:
        lc, tc = [], []
        for b in range(B):
            gm = np.asarray(gt_types[b]) == c
            l_, t_, _ = assign(anchors[c][b], anchors_mask[c][b], np.asarray(gt_bboxes[b], np.float32)[gm],
                               np.full(int(gm.sum()), class_names.index(c) + 1, np.int64), assign_cfg[c][0],
                               assign_cfg[c][1], nearest_iou)
            lc.append(l_)
            tc.append(t_)
        L.append(np.stack(lc))
        Tg.append(np.stack(tc))
    labels = torch.from_numpy(np.stack(L, 1).reshape(B, -1))
    targets = torch.from_numpy(np.stack(Tg, 1).reshape(B, -1, 7)).to(dt)
    anc_all = torch.cat([torch.as_tensor(anchors[c], dtype=torch.float32) for c in class_names], 1).view(B, -1, 7).to(dt)
    msk_all = torch.cat([torch.as_tensor(anchors_mask[c]).bool() for c in class_names], 1).view(B, -1)
    pos = (labels > 0).to(dt)
    pn = torch.clamp(pos.sum(1, keepdim=True), min=1.0)
    cw, rw = (labels >= 0).to(dt) / pn, pos / pn
    onehot = torch.zeros(B, labels.shape[1], nc, dtype=dt)
    for c in range(nc):
        onehot[..., c] = (labels == c + 1).to(dt)
    bp = torch.cat([box[..., :6], torch.sin(box[..., 6:]) * torch.cos(targets[..., 6:])], -1)
    tp = torch.cat([targets[..., :6], torch.cos(box[..., 6:]) * torch.sin(targets[..., 6:])], -1)
    losses["rpn_loc_loss"] = smooth_l1_sum(bp, tp, rw[..., None], 1 / 9.) / B * 2
    losses["rpn_cls_loss"] = focal_sum(cls, onehot, cw[..., None]) / B
    dlab = ((targets[..., 6] + anc_all[..., 6]) > 0).long().view(-1)
    dwt = (pos / torch.clamp(pos.sum(-1, keepdim=True), min=1.0)).view(-1)
    losses["rpn_dir_loss"] = (F.cross_entropy(dr.reshape(-1, 2), dlab, reduction="none") * dwt).sum() / B * .2
    # -- guided anchors (+ ground truth) and the part-sensitive rescoring loss
    dec = decode(box, anc_all)
    f = conv2d(conv6, P["extra_head.convs.0.weight"], None, 1, rbev)
    f = torch.relu(bn_train(f, P["extra_head.convs.1.weight"], P["extra_head.convs.1.bias"]))
    f = conv2d(f, P["extra_head.convs.3.weight"], None, 0, rbev)
    scores, elabels, guided_all, sel_all, free_all, top_all = [], [], [], [], [], []
    for b in range(B):
        m = msk_all[b]
        bx, sc, dl = dec[b][m], torch.sigmoid(cls[b][m]), dr[b][m].argmax(-1)
        top = sc.squeeze(-1) if nc == 1 else sc.max(-1)[0]
        s = top > anchor_thr
        free_all.append(torch.nonzero(s).view(-1).numpy())
        top_all.append(top.detach().double().numpy())
        if guided_sel is not None:
            s = torch.zeros_like(s)
            s[torch.as_tensor(guided_sel[b], dtype=torch.int64)] = True
        sel_all.append(torch.nonzero(s).view(-1).numpy())
        bx, dl = bx[s], dl[s]
        flip = ((bx[:, 6] > 0) ^ dl.bool()).to(dt)
        bx = torch.cat([bx[:, :6], (bx[:, 6] + flip * math.pi)[:, None]], 1)
        ga = torch.cat([torch.as_tensor(gt_bboxes[b], dtype=torch.float32).to(dt), bx], 0)
        guided_all.append(ga.detach())
        n = ga.shape[0]
        ct, st = torch.cos(ga[:, 6]).view(n, 1, 1), torch.sin(ga[:, 6]).view(n, 1, 1)
        xx = torch.linspace(-.5, .5, 4, dtype=dt).view(1, 4, 1) * ga[:, 3].view(n, 1, 1)
        yy = torch.linspace(-.5, .5, 7, dtype=dt).view(1, 1, 7) * ga[:, 4].view(n, 1, 1)
        sx = (xx * ct + yy * st + ga[:, 0].view(n, 1, 1) + grid_offsets[0]) / featmap_stride
        sy = (yy * ct - xx * st + ga[:, 1].view(n, 1, 1) + grid_offsets[1]) / featmap_stride
        g = torch.stack([sx.reshape(n, 28).t() / (w_ - 1), sy.reshape(n, 28).t() / (h - 1)], -1).view(28, n, 1, 2)
        o = F.grid_sample(f[b].unsqueeze(1), g * 2 - 1, align_corners=True)
        scores.append(o.mean(0).view(-1))
        el, _, _ = assign(ga.detach().numpy(), None, np.asarray(gt_bboxes[b], np.float32),
                          np.ones(len(gt_bboxes[b]), np.int64), extra_thr, extra_thr, rotated_iou3d)
        elabels.append(el)
    el = torch.from_numpy(np.concatenate(elabels))
    ew = (el >= 0).to(dt) / torch.clamp((el > 0).to(dt).sum(), min=1.0)
    losses["loss_cls"] = focal_sum(torch.cat(scores), (el > 0).to(dt), ew) / B
    # grad_exclude: loss terms left out of the differentiated sum (tests: "loss_cls" removes the only term whose gradient
    # depends on the discrete guided-anchor selection); every term is still reported
    total = sum(v for k, v in losses.items() if k not in grad_exclude)
    names = [k for k, v in P.items() if v.requires_grad]
    grads = torch.autograd.grad(total, [P[k] for k in names], allow_unused=True)
    return ({k: float(v.detach()) for k, v in losses.items()}, dict(zip(names, grads)),
            dict(guided=guided_all, labels=labels, ext_labels=el, box=box.detach(), cls=cls.detach(), guided_sel=sel_all,
                 guided_free=free_all, masked_top=top_all))
