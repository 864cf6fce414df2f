"""CPU oracle (torch fp32, CPU) of the SA-SSD network-side hot path (TEST INFRASTRUCTURE ONLY).

Restates, with file:line citations relative to /root/reference:
  sparse conv + BN1d + ReLU   mmdet/models/necks/cmn.py:138-173,192-231  (spconv v1.0 semantics, App. A)
  dense()                     cmn.py:112-114
  BEVNet                      cmn.py:233-282
  SSDRotateHead.forward       mmdet/models/single_stage_heads/ssd_rotate_head.py:218-235
  second_box_decode           ssd_rotate_head.py:53-91
  get_guided_anchors          ssd_rotate_head.py:307-372
  gen_sample_grid / PSWarp    ssd_rotate_head.py:374-447
  get_rescore_bboxes          ssd_rotate_head.py:487-533 (+ iou3d_utils.py:47-60,114-128, bbox_nms.py:4-26)
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import clib
from . import rulebook as rb

BN_EPS = 1e-3


def sparse_conv(x, nbr, weight):
    """Y[o] = sum_k X[nbr[o,k]] @ W[k]   (W: [K, Cin, Cout], K=27 in (kz,ky,kx) order; k ascending)."""
    nout = nbr.shape[0]
    y = torch.zeros(nout, weight.shape[2], dtype=torch.float32)
    nbr_t = torch.as_tensor(nbr, dtype=torch.int64)
    for k in range(nbr.shape[1]):
        o = torch.nonzero(nbr_t[:, k] >= 0).view(-1)
        if o.numel() == 0:
            continue
        y.index_add_(0, o, x[nbr_t[o, k]] @ weight[k])
    return y


def bn_relu(x, bn):
    """eval-mode BatchNorm (eps 1e-3, cmn.py:141) + ReLU on [N,C] or [N,C,H,W]."""
    g, b, m, v = bn["weight"], bn["bias"], bn["running_mean"], bn["running_var"]
    shape = (1, -1) + (1,) * (x.dim() - 2)
    y = (x - m.view(shape)) / torch.sqrt(v.view(shape) + BN_EPS) * g.view(shape) + b.view(shape)
    return torch.relu(y)


# (name, kind, Cin, Cout, indice_key) -- VxNet.__init__ cmn.py:197-212
VXNET_LAYERS = [
    ("conv0.0", "subm", 4, 16, "subm0"), ("conv0.1", "subm", 16, 16, "subm0"),
    ("down0", "down", 16, 32, "down0"),
    ("conv1.0", "subm", 32, 32, "subm1"), ("conv1.1", "subm", 32, 32, "subm1"),
    ("down1", "down", 32, 64, "down1"),
    ("conv2.0", "subm", 64, 64, "subm2"), ("conv2.1", "subm", 64, 64, "subm2"), ("conv2.2", "subm", 64, 64, "subm2"),
    ("down2", "down", 64, 64, "down2"),
    ("conv3.0", "subm", 64, 64, "subm3"), ("conv3.1", "subm", 64, 64, "subm3"), ("conv3.2", "subm", 64, 64, "subm3"),
    ("extra", "1x1", 64, 64, None),
]


def vxnet_forward(feats, coors, spatial_shape, batch_size, params, return_all=False):
    """VxNet.forward (cmn.py:214-231). params[name] = dict(weight=[K,Cin,Cout], bn=dict(...)).
    Returns (features[N3,64], indices[N3,4], shape3, middle list, rulebooks dict)."""
    x = torch.as_tensor(feats, dtype=torch.float32)
    idx = np.asarray(coors, np.int32)
    shape = tuple(spatial_shape)
    books = {}
    middle = []
    acts = {}
    for name, kind, cin, cout, key in VXNET_LAYERS:
        if kind == "subm":
            if key not in books:
                books[key] = rb.subm_rulebook(idx, shape)
            _, nbr = books[key]
            x = sparse_conv(x, nbr, params[name]["weight"])
        elif kind == "down":
            out_idx, nbr, oshape = rb.conv_rulebook(idx, shape, batch_size)
            books[key] = (out_idx, nbr)
            x = sparse_conv(x, nbr, params[name]["weight"])
            idx, shape = out_idx, oshape
        else:                                   # 1x1x1 conv == plain mm (spconv shortcut)
            x = x @ params[name]["weight"][0]
        x = bn_relu(x, params[name]["bn"])
        acts[name] = x
        if name in ("conv1.1", "conv2.2", "conv3.2"):
            middle.append((x, idx.copy(), shape))
    if return_all:
        return x, idx, shape, middle, books, acts
    return x, idx, shape, middle, books


def densify(feats, idx, shape, batch_size):
    """SparseConvTensor.dense() + view (cmn.py:112-114): [B, C*D, H, W], channel = c*D + d."""
    d, h, w = shape
    c = feats.shape[1]
    out = torch.zeros(batch_size, d, h, w, c, dtype=torch.float32)
    i = torch.as_tensor(idx, dtype=torch.int64)
    out[i[:, 0], i[:, 1], i[:, 2], i[:, 3]] = feats
    out = out.permute(0, 4, 1, 2, 3).contiguous()
    return out.view(batch_size, c * d, h, w)


def bevnet_forward(x, params):
    """BEVNet.forward (cmn.py:264-282). params['convN'] = dict(weight=[Cout,Cin,kh,kw], bn=...)."""
    conv6 = None
    for i in range(8):
        p = params["conv%d" % i]
        pad = 1 if p["weight"].shape[-1] == 3 else 0
        x = F.conv2d(x, p["weight"], None, 1, pad)
        x = bn_relu(x, p["bn"])
        if i == 6:
            conv6 = x.clone()
    return x, conv6


def ssd_head_forward(x, params, num_class=1):
    """SSDRotateHead.forward (ssd_rotate_head.py:218-235)."""
    n, _, h, w = x.shape
    box = F.conv2d(x, params["conv_box"]["weight"], params["conv_box"]["bias"])
    cls = F.conv2d(x, params["conv_cls"]["weight"], params["conv_cls"]["bias"])
    dirp = F.conv2d(x, params["conv_dir_cls"]["weight"], params["conv_dir_cls"]["bias"])
    box = box.view(n, num_class, -1, h, w).permute(0, 1, 3, 4, 2).contiguous()
    cls = cls.view(n, num_class, -1, h, w).permute(0, 1, 3, 4, 2).contiguous()
    dirp = dirp.view(n, num_class, -1, h, w).permute(0, 1, 3, 4, 2).contiguous()
    return box, cls, dirp


def box_decode(enc, anchors):
    """second_box_decode (ssd_rotate_head.py:53-91), default flags."""
    xa, ya, za, wa, la, ha, ra = torch.split(anchors, 1, dim=-1)
    xt, yt, zt, wt, lt, ht, rt = torch.split(enc, 1, dim=-1)
    za = za + ha / 2
    diag = torch.sqrt(la ** 2 + wa ** 2)
    xg = xt * diag + xa
    yg = yt * diag + ya
    zg = zt * ha + za
    lg = torch.exp(lt) * la
    wg = torch.exp(wt) * wa
    hg = torch.exp(ht) * ha
    rg = rt + ra
    zg = zg - hg / 2
    return torch.cat([xg, yg, zg, wg, lg, hg, rg], dim=-1)


def guided_anchors(box, cls, dirp, anchors, anchors_mask, num_class=1, thr=0.1):
    """get_guided_anchors, test path (ssd_rotate_head.py:307-372). anchors [B,A,7], mask [B,A] bool.
    Returns per-sample (boxes[K,7], labels[K] int64, scores[K] (extra, for tolerance-aware tests))."""
    bsz = box.shape[0]
    bbox = box_decode(box.view(bsz, -1, 7), anchors)
    bcls = cls.view(bsz, -1, num_class)
    bdir = dirp.view(bsz, -1, 2)
    amask = anchors_mask.view(bsz, -1)
    out = []
    for b in range(bsz):
        bp, cp, dp = bbox[b][amask[b]], bcls[b][amask[b]], bdir[b][amask[b]]
        dl = torch.max(dp, dim=-1)[1]
        ts = torch.sigmoid(cp)
        if num_class == 1:
            top_s = ts.squeeze(-1)
            top_l = torch.zeros(ts.shape[0], dtype=torch.int64)
        else:
            top_s, top_l = torch.max(ts, dim=-1)
        sel = top_s > thr
        bp = bp[sel].clone()
        top_l = top_l[sel]
        dl = dl[sel]
        opp = (bp[..., -1] > 0) ^ dl.bool()
        bp[opp, -1] += np.pi
        out.append((bp, top_l, top_s[sel]))
    return out


def pswarp_forward(conv6, params, guided, grid_offsets=(0.0, 40.0), featmap_stride=0.4):
    """PSWarpHead.forward (ssd_rotate_head.py:431-447) with gen_sample_grid (:374-398) and
    bilinear_interpolate_torch_gridsample (:400-414). guided: list of [K,7] per sample."""
    x = F.conv2d(conv6, params["conv0"]["weight"], None, 1, 1)
    x = bn_relu(x, params["conv0"]["bn"])
    x = F.conv2d(x, params["conv1"]["weight"], None)
    scale = 1.0 / featmap_stride
    scores = []
    for i, ga in enumerate(guided):
        if len(ga) == 0:
            scores.append(torch.empty(0))
            continue
        b5 = ga[:, [0, 1, 3, 4, 6]]
        n = b5.shape[0]
        xg, yg, wg, lg, rg = [b5[:, j] for j in range(5)]
        ct, st = torch.cos(rg), torch.sin(rg)
        xx = torch.linspace(-.5, .5, 4).view(1, 4, 1) * wg.view(n, 1, 1)      # [n,4,1]
        yy = torch.linspace(-.5, .5, 7).view(1, 1, 7) * lg.view(n, 1, 1)      # [n,1,7]
        sx = xx * ct.view(n, 1, 1) + yy * st.view(n, 1, 1) + xg.view(n, 1, 1)  # [n,4,7]
        sy = yy * ct.view(n, 1, 1) - xx * st.view(n, 1, 1) + yg.view(n, 1, 1)
        sx = ((sx.permute(1, 2, 0).contiguous() + grid_offsets[0]) * scale).view(28, n)
        sy = ((sy.permute(1, 2, 0).contiguous() + grid_offsets[1]) * scale).view(28, n)
        im = x[i].unsqueeze(1)                                                 # [28,1,H,W]
        h, w = im.shape[-2:]
        g = torch.stack([sx / (w - 1), sy / (h - 1)], -1).view(28, n, 1, 2) * 2 - 1
        o = F.grid_sample(im, g, align_corners=True)                          # [28,1,n,1]
        scores.append(torch.mean(o, 0).view(-1))
    return scores, x


def boxes3d_to_bev(b):
    """iou3d_utils.py:47-60: (x - w/2, y - l/2, x + w/2, y + l/2, r) from cols 0,1,3,4,6."""
    out = b.new_zeros(b.shape[0], 5)
    out[:, 0] = b[:, 0] - b[:, 3] / 2
    out[:, 1] = b[:, 1] - b[:, 4] / 2
    out[:, 2] = b[:, 0] + b[:, 3] / 2
    out[:, 3] = b[:, 1] + b[:, 4] / 2
    out[:, 4] = b[:, 6]
    return out


def rescore(guided, logits, labels, score_thr=0.3, iou_thr=0.1):
    """get_rescore_bboxes for one sample (ssd_rotate_head.py:487-533). Stable descending sort."""
    if logits.numel() == 0:
        return None
    s = torch.sigmoid(logits).view(-1)
    sel = s > score_thr
    bp, s, lb = guided[sel], s[sel], labels[sel]
    if s.numel() == 0:
        return None
    bev = boxes3d_to_bev(bp)
    order = torch.sort(s, descending=True, stable=True)[1]
    keep = clib.nms_rotated(bev[order].numpy(), iou_thr)
    k = order[torch.as_tensor(keep)]
    return bp[k].numpy(), s[k].numpy(), lb[k].numpy()


def anchors_mask(coors_zyx, anchors_bv, voxel_size, pc_range, grid_size_xyz, area_threshold=1):
    """kitti.py:333-343 + geometry.py:676-710: occupancy integral image + per-anchor area > threshold.
    coors_zyx [M,3] int32; anchors_bv [A,4] f32 (xmin,ymin,xmax,ymax); grid_size_xyz (W0,H0,D0)."""
    w0, h0 = int(grid_size_xyz[0]), int(grid_size_xyz[1])
    dense = np.zeros((h0, w0), np.float32)
    np.add.at(dense, (coors_zyx[:, 1], coors_zyx[:, 2]), 1.0)
    dense = dense.cumsum(0).cumsum(1)
    vs = np.asarray(voxel_size, np.float32)
    off = np.asarray(pc_range, np.float32)
    bv = np.asarray(anchors_bv, np.float32)
    c0 = np.floor((bv[:, 0] - off[0]) / vs[0]).astype(np.int32)
    c1 = np.floor((bv[:, 1] - off[1]) / vs[1]).astype(np.int32)
    c2 = np.floor((bv[:, 2] - off[0]) / vs[0]).astype(np.int32)
    c3 = np.floor((bv[:, 3] - off[1]) / vs[1]).astype(np.int32)
    c0 = np.maximum(c0, 0); c1 = np.maximum(c1, 0)
    c2 = np.minimum(c2, w0 - 1); c3 = np.minimum(c3, h0 - 1)
    area = dense[c3, c2] - dense[c3, c0] - dense[c1, c2] + dense[c1, c0]
    return area > area_threshold
