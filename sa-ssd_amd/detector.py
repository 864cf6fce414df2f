"""Host-side mirror of the reference model/config API for the hot path (same class names, constructor
arguments, parameter names -> reference checkpoints load with `load_state_dict`):

  SimpleVoxel          mmdet/models/backbones/vxnet.py:99-116
  SpMiddleFHD / VxNet / BEVNet   mmdet/models/necks/cmn.py:12-282
  SSDRotateHead        mmdet/models/single_stage_heads/ssd_rotate_head.py:93-372   (test path)
  PSWarpHead           ssd_rotate_head.py:416-533                                    (test path)
  SingleStageDetector  mmdet/models/detectors/single_stage.py:13-131, base.py:77-81
  build_detector       mmdet/models/builder.py:54-56

`model(img, img_meta, return_loss=False, voxels=[..], coordinates=[..], num_points=[..], anchors=[..],
anchors_mask=[..])` runs the fused HIP pipeline (sassd.pipeline.InferencePlan).  `return_loss=True` runs the
training branch (single_stage.py:75-108): module-by-module forward through the same HIP kernels recorded by
sassd.autograd (sparse / dense conv, densify, part-sensitive warp, 3-NN interpolation all have HIP backward
kernels), torch BatchNorm / ReLU / Linear in between, losses and target assignment from sassd.train_ops.
"""
import sys

import numpy as np
import torch
from torch import nn

from . import iou3d_utils
from . import kernels as K
from . import spconv
from . import train_ops as T
from .autograd import bf16_cout_pad, bn_relu_conv, bn_relu_conv_fusable, AuxHeadFn, Conv2dFn, FocalLossFn, GuidedDecodeFn, PSWarpBatchFn, PSWarpFn, RpnLossFn, bev_precision, bn_relu_2d
from .config import _wrap, obj_from_dict
from .kitti_common import kitti_bbox2results
from .pipeline import InferencePlan
from .pointnet2_utils import nearest_neighbor_interpolate


_CONSTS = {}


def _const(dev, values):
    """Small fp32 device constant, uploaded once per (device, values) instead of one blocking H2D copy per use."""
    key = (str(dev), tuple(float(v) for v in values))
    if key not in _CONSTS:
        _CONSTS[key] = torch.tensor(key[1], dtype=torch.float32, device=dev)
    return _CONSTS[key]


_CONSTS_I32 = {}


_CONSTS_I64 = {}


def _const_i64(dev, values):
    key = (str(dev), tuple(int(v) for v in values))
    if key not in _CONSTS_I64:
        if len(_CONSTS_I64) > 4096:
            _CONSTS_I64.clear()
        _CONSTS_I64[key] = torch.tensor(key[1], dtype=torch.int64, device=dev)
    return _CONSTS_I64[key]


def _const_i32(dev, values):
    key = (str(dev), tuple(int(v) for v in values))
    if key not in _CONSTS_I32:
        if len(_CONSTS_I32) > 4096:
            _CONSTS_I32.clear()
        _CONSTS_I32[key] = torch.tensor(key[1], dtype=torch.int32, device=dev)
    return _CONSTS_I32[key]


_CONSTS_F32 = {}


def _const_f32(dev, values):
    key = (str(dev), tuple(float(v) for v in values))
    if key not in _CONSTS_F32:
        if len(_CONSTS_F32) > 4096:
            _CONSTS_F32.clear()
        _CONSTS_F32[key] = torch.tensor(key[1], dtype=torch.float32, device=dev)
    return _CONSTS_F32[key]


_ARANGE_I32 = {}


def _arange_i32(dev, n):
    """cached torch.arange(n, int32) (read-only): the padded-row masks of a step compare it with the device counts"""
    key = (str(dev), int(n))
    if key not in _ARANGE_I32:
        if len(_ARANGE_I32) > 64:
            _ARANGE_I32.clear()
        _ARANGE_I32[key] = torch.arange(int(n), device=dev, dtype=torch.int32)
    return _ARANGE_I32[key]


def _scaled_terms(sums, scales):
    """Loss sums [n] x per-term constants -> n loss tensors of shape [1] in TWO launches each way (one multiply by a cached constant
    vector; unbind forward = views, backward = one stack).  `sums[i:i+1] / b * w` per term was two to three elementwise
    launches forward and a division, a multiplication and a slice backward (zeros + copy) per term backward: ~25 of the
    training step's small torch launches (round 5)."""
    return tuple(t.view(1) for t in (sums * _const_f32(sums.device, scales)).unbind(0))      # ([1] each, like the module path)


_GT_CAT = [None, None]


def _gt_cat(gt_bboxes):
    """Ground-truth boxes of a batch as one contiguous fp32 [T, 7] tensor (None when the batch has none).  The four
    consumers of a training step (RPN assignment, aux targets, guided anchors, rescoring targets) share one concatenation:
    the last result is kept, keyed on the identity and autograd version of the per-sample tensors."""
    key = tuple((id(g), g._version, tuple(g.shape)) for g in gt_bboxes)
    if _GT_CAT[0] != key:
        tot = sum(int(g.shape[0]) for g in gt_bboxes)
        _GT_CAT[1] = torch.cat([g.float() for g in gt_bboxes], 0).contiguous() if tot else None
        _GT_CAT[0] = key
        _GT_CAT.append(list(gt_bboxes))          # (keeps the keyed tensors alive: their ids cannot be recycled)
        del _GT_CAT[2:-1]
    return _GT_CAT[1]


def change_default_args(**kwargs):
    """mmdet/models/utils/__init__.py:41."""
    def layer_wrapper(layer_class):
        class DefaultArgLayer(layer_class):
            def __init__(self, *args, **kw):
                for k, v in kwargs.items():
                    kw.setdefault(k, v)
                super().__init__(*args, **kw)
        return DefaultArgLayer
    return layer_wrapper


class SimpleVoxel(nn.Module):
    def __init__(self, num_input_features=4, use_norm=True, num_filters=(32, 128), with_distance=False,
                 name='VoxelFeatureExtractor'):
        super().__init__()
        self.name, self.num_input_features = name, num_input_features

    def forward(self, features, num_voxels):
        return K.voxel_mean(features.contiguous().float(), num_voxels.int().contiguous(), self.num_input_features)


def _bn1d(c):
    return nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)


def single_conv(cin, cout, indice_key=None):
    return spconv.SparseSequential(spconv.SubMConv3d(cin, cout, 1, bias=False, indice_key=indice_key), _bn1d(cout),
                                   nn.ReLU())


def double_conv(cin, cout, indice_key=None):
    return spconv.SparseSequential(
        spconv.SubMConv3d(cin, cout, 3, bias=False, indice_key=indice_key), _bn1d(cout), nn.ReLU(),
        spconv.SubMConv3d(cout, cout, 3, bias=False, indice_key=indice_key), _bn1d(cout), nn.ReLU())


def triple_conv(cin, cout, indice_key=None):
    return spconv.SparseSequential(
        spconv.SubMConv3d(cin, cout, 3, bias=False, indice_key=indice_key), _bn1d(cout), nn.ReLU(),
        spconv.SubMConv3d(cout, cout, 3, bias=False, indice_key=indice_key), _bn1d(cout), nn.ReLU(),
        spconv.SubMConv3d(cout, cout, 3, bias=False, indice_key=indice_key), _bn1d(cout), nn.ReLU())


def stride_conv(cin, cout, indice_key=None):
    return spconv.SparseSequential(
        spconv.SparseConv3d(cin, cout, 3, (2, 2, 2), padding=1, bias=False, indice_key=indice_key), _bn1d(cout),
        nn.ReLU())


class VxNet(nn.Module):
    def __init__(self, num_input_features):
        super().__init__()
        self.conv0 = double_conv(num_input_features, 16, 'subm0')
        self.down0 = stride_conv(16, 32, 'down0')
        self.conv1 = double_conv(32, 32, 'subm1')
        self.down1 = stride_conv(32, 64, 'down1')
        self.conv2 = triple_conv(64, 64, 'subm2')
        self.down2 = stride_conv(64, 64, 'down2')
        self.conv3 = triple_conv(64, 64, 'subm3')
        self.extra_conv = spconv.SparseSequential(spconv.SparseConv3d(64, 64, (1, 1, 1), (1, 1, 1), bias=False),
                                                  _bn1d(64), nn.ReLU())

    def precompute_rulebooks(self, coors, spatial_shape, batch_size):
        """All seven rulebooks of a batch from its voxel coordinates alone [N,4] (b,z,y,x): the indice_dict to hand to
        `forward` later.  Lets a training loop build them (with their three host syncs) for batch i+1 before the
        backward of batch i is queued, so that the next forward never waits on the GPU."""
        x = spconv.SparseConvTensor(None, coors.int().contiguous(), spatial_shape, batch_size)
        for block in (self.conv0, self.down0, self.conv1, self.down1, self.conv2, self.down2, self.conv3):
            for m in block:
                if isinstance(m, spconv.SparseConvolution) and not m.conv1x1:
                    out_idx, nbr, oshape, otable = m.book(x)
                    if not m.subm:
                        nxt = spconv.SparseConvTensor(None, out_idx, oshape, batch_size)
                        nxt.indice_dict = x.indice_dict
                        x = nxt
        return x.indice_dict

    def forward(self, x):
        middle = []
        x = self.conv1(self.down0(self.conv0(x)))
        middle.append(x)
        x = self.conv2(self.down1(x))
        middle.append(x)
        x = self.conv3(self.down2(x))
        middle.append(x)
        return self.extra_conv(x), middle


class _HipConv2d(nn.Conv2d):
    """nn.Conv2d parameters, HIP fp32-MFMA forward (sassd_conv2d_fwd) with an optional fused affine + ReLU."""

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)      # .to() / .cuda() on this layer alone: its packed images are stale
        K.bump_weights_generation()
        return out

    def packed_weight(self):
        v = K.weight_key(self.weight)
        if getattr(self, "_pk", None) is None or self._pkv != v:
            self._pk, self._pkv = K.conv2d_pack_weight(self.weight.detach().float().contiguous()), v
        return self._pk

    def packed_wino(self, h, w):
        """Winograd-packed weights when this layer / feature-map shape supports the F(2x2,3x3) kernel, else None."""
        if self.kernel_size[0] != 3 or not K.conv2d_wino_supported(self.in_channels, self.out_channels, h, w):
            return None
        v = K.weight_key(self.weight)
        if getattr(self, "_pkw", None) is None or self._pkwv != v:
            self._pkw, self._pkwv = K.conv2d_wino_pack_weight(self.weight.detach().float().contiguous()), v
        return self._pkw

    def packed_wino4(self, h, w):
        """Winograd F(4x4,3x3) weights (G g G^T, [36][Cin][Cout]) when the layer / feature-map shape supports it."""
        if self.kernel_size[0] != 3 or not K.conv2d_wino4_supported(self.in_channels, self.out_channels, h, w):
            return None
        v = K.weight_key(self.weight)
        if getattr(self, "_pk4", None) is None or self._pk4v != v:
            self._pk4, self._pk4v = K.conv2d_wino4_pack_weight(self.weight.detach().float().contiguous()), v
        return self._pk4

    def hip_forward(self, x, scale=None, shift=None, relu=False):
        if shift is None and self.bias is not None:
            shift = self.bias.detach().float().contiguous()
        p4 = self.packed_wino4(x.shape[2], x.shape[3])
        if p4 is not None:
            return K.conv2d_wino4_fwd(x.contiguous().float(), p4, self.out_channels, scale, shift, relu)
        pw = self.packed_wino(x.shape[2], x.shape[3])
        if pw is not None:
            return K.conv2d_wino_fwd(x.contiguous().float(), pw, self.out_channels, scale, shift, relu)
        self.packed_weight()
        if shift is None and self.bias is not None:
            shift = self.bias.detach().float().contiguous()
        return K.conv2d_fwd(x.contiguous().float(), self._pk, self.out_channels, self.kernel_size[0], scale, shift,
                            relu)

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            if (bev_precision() == "bf16" and self.kernel_size[0] == 3 and
                    K.conv2d_bf16_supported(self.in_channels, bf16_cout_pad(self.out_channels), x.shape[2], x.shape[3])):
                return Conv2dFn.apply(x.float(), self.weight, self.bias, None, None, None)   # packs its own bf16 image
            if (bev_precision() == "bf16" and self.kernel_size[0] == 1 and
                    K.conv1x1_bf16_supported(self.in_channels, self.out_channels, x.shape[2] * x.shape[3])):
                return Conv2dFn.apply(x.float(), self.weight, self.bias, None, None, None)   # (round 6) bf16 1x1 kernel
            p4 = self.packed_wino4(x.shape[2], x.shape[3])
            pw = None if p4 is not None else self.packed_wino(x.shape[2], x.shape[3])
            pk = None if (p4 is not None or pw is not None) else self.packed_weight()
            return Conv2dFn.apply(x.float(), self.weight, self.bias, pk, pw, p4)
        return self.hip_forward(x)


def _bn_affine(bn):
    scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
    return scale.float().contiguous(), (bn.bias.detach() - bn.running_mean * scale).float().contiguous()


class BEVNet(nn.Module):
    def __init__(self, in_features, num_filters=256):
        super().__init__()
        BatchNorm2d = change_default_args(eps=1e-3, momentum=0.01)(nn.BatchNorm2d)
        Conv2d = change_default_args(bias=False)(_HipConv2d)
        for i in range(8):
            cin = in_features if i == 0 else num_filters
            setattr(self, 'conv%d' % i, Conv2d(cin, num_filters, 3, padding=1) if i < 7 else Conv2d(cin, num_filters, 1))
            setattr(self, 'bn%d' % i, BatchNorm2d(num_filters))

    # bf16 training (BASELINE configs[2]): BatchNorm + ReLU of layer i is applied by the LOADER WAVES of layer i + 1's convolution
    # and weight gradient (autograd.BnReluConvBf16Fn) wherever the normalised map has no other reader -- conv0 .. conv5; conv6's
    # output also feeds the part-sensitive head and conv7 is 1x1, so bn6 / bn7 keep the stand-alone kernels.  False: every layer
    # as bn_relu_2d(bn, conv(x)) (A/B; bit-identical by construction, tests/test_gpu_bf16.py).
    fuse_bn_into_conv = True

    def forward(self, x):
        conv6 = None
        raw, raw_bn = None, None                            # a conv output whose BatchNorm + ReLU is still pending
        for i in range(8):
            conv, bn = getattr(self, 'conv%d' % i), getattr(self, 'bn%d' % i)
            if self.training or (torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad)):
                if raw is not None:
                    y = bn_relu_conv(raw_bn, conv, raw)     # relu(bn_{i-1}(raw)) is formed inside conv i
                else:
                    y = conv(x)
                nxt = getattr(self, 'conv%d' % (i + 1)) if i + 1 < 8 else None
                if (self.fuse_bn_into_conv and self.training and i != 6 and nxt is not None
                        and bn_relu_conv_fusable(bn, nxt, y)):
                    raw, raw_bn, x = y, bn, None            # deferred into the next layer
                else:
                    raw, raw_bn = None, None
                    x = bn_relu_2d(bn, y)                   # batch statistics + ReLU: sassd_bn2d_relu_* (two launches)
            else:
                s, b = _bn_affine(bn)
                x = conv.hip_forward(x, s, b, True)
            if i == 6:
                conv6 = x
        return x, conv6


class SpMiddleFHD(nn.Module):
    def __init__(self, output_shape, num_input_features=4, num_hidden_features=128, aux_offset=(0., -40., -3.),
                 aux_voxel_size=(.05, .05, .1)):
        """aux_offset / aux_voxel_size: lower corner of the point-cloud range and the voxel size the auxiliary head's
        voxel centres are computed with.  The reference hard-codes KITTI's (cmn.py:121-127: offset (0, -40, -3), voxel
        sizes (.1, .1, .2) / (.2, .2, .4) / (.4, .4, .8) = 2, 4, 8 x the input voxel) -- the defaults; a config for
        another range (the Waymo-scale workload) passes its own."""
        super().__init__()
        self.sparse_shape = output_shape
        self.aux_offset = tuple(float(v) for v in aux_offset)
        self.aux_voxel_size = tuple(float(v) for v in aux_voxel_size)
        self.backbone = VxNet(num_input_features)
        self.fcn = BEVNet(in_features=num_hidden_features, num_filters=256)
        self.point_fc = nn.Linear(160, 64, bias=False)        # training-only auxiliary head (cmn.py:27-29)
        self.point_cls = nn.Linear(64, 1, bias=False)
        self.point_reg = nn.Linear(64, 3, bias=False)
        # scene extent / bin size for the exact binned 3-NN of the aux head (any values give identical results; the
        # extent of the voxel centres keeps the bins evenly filled)
        ox, oy, vx, vy = self.aux_offset[0], self.aux_offset[1], self.aux_voxel_size[0], self.aux_voxel_size[1]
        self.aux_xy_range = (ox, oy, ox + float(output_shape[2]) * vx, oy + float(output_shape[1]) * vy)
        self.aux_bin = 32 * vx

    def build_aux_target(self, nxyz, gt_boxes3d, enlarge=1.0):
        """cmn.py:45-72 with the point-in-box test on the device (sassd_pts_in_boxes3d)."""
        pts = nxyz[:, 1:].contiguous()
        n = pts.shape[0]
        labels = torch.zeros(n, dtype=torch.uint8, device=pts.device)
        offsets = torch.zeros(n, 3, device=pts.device)
        for i, boxes3d in enumerate(gt_boxes3d):
            if boxes3d.shape[0] == 0 or n == 0:
                continue
            boxes3d = boxes3d.clone().float()
            boxes3d[:, 3:6] *= enlarge
            # every point against sample i's boxes, kept only for the points of sample i (no boolean compaction)
            flag, off = K.pts_in_boxes3d(pts, boxes3d.to(pts.device).contiguous())
            mine = nxyz[:, 0] == i
            inside = (flag.max(0)[0] > 0) & mine
            labels = torch.where(inside, torch.ones_like(labels), labels)
            offsets = torch.where(inside[:, None], off, offsets)
        return labels, offsets

    # The auxiliary head in training as three fused launches + the three 3-NN searches (sassd_aux_*), instead of ~150
    # small torch / library launches; False keeps the module-by-module formulation (A/B, parity tests).
    fused_aux = True

    def _aux_loss_fused(self, ctx, gt_bboxes):
        """aux_loss on the fused kernels: `ctx` = (voxel_features, coors, middle tensors, batch size) left by forward."""
        voxel_features, coors, middle, batch_size = ctx
        dev = voxel_features.device
        counts = [int(g.shape[0]) for g in gt_bboxes]
        gt_all = _gt_cat(gt_bboxes)
        points, known, label, target, npos = K.aux_prepare(
            voxel_features.detach().float().contiguous(), coors.int().contiguous(),
            [m.indices.int().contiguous() for m in middle], self.aux_voxel_size, self.aux_offset, gt_all,
            K.gt_offsets(counts, dev), batch_size)
        nn_idx, nn_d2 = [], []
        for kn, mult in zip(known, (2, 4, 8)):
            d2, idx = K.three_nn_binned(points, kn, self.aux_xy_range,
                                        min(self.aux_bin, 4 * self.aux_voxel_size[0] * mult), batch_size)
            nn_idx.append(idx)
            nn_d2.append(d2)
        sums = AuxHeadFn.apply(middle[0].features, middle[1].features, middle[2].features, self.point_fc.weight,
                               self.point_cls.weight, self.point_reg.weight, nn_idx, nn_d2, label, target, npos)
        n = len(gt_bboxes)
        cls_term, reg_term = _scaled_terms(sums, (1.0 / n, 1.0 / n))
        return dict(aux_loss_cls=cls_term, aux_loss_reg=reg_term)

    def aux_loss(self, points, point_cls, point_reg, gt_bboxes):
        """cmn.py:74-104."""
        if point_cls is None and isinstance(points, tuple):      # forward() left the fused context
            return self._aux_loss_fused(points, gt_bboxes)
        n = len(gt_bboxes)
        pts_labels, center_targets = self.build_aux_target(points, gt_bboxes)
        pos, neg = (pts_labels > 0).float(), (pts_labels == 0).float()
        norm = torch.clamp(pos.sum(), min=1.0)
        aux_cls = T.weighted_sigmoid_focal_loss(point_cls.view(-1), pts_labels.float(), weight=(pos + neg) / norm,
                                                avg_factor=1.) / n
        aux_reg = T.weighted_smoothl1(point_reg, center_targets, beta=1 / 9., weight=(pos / norm)[..., None],
                                      avg_factor=1.) / n
        return dict(aux_loss_cls=aux_cls, aux_loss_reg=aux_reg)

    @staticmethod
    def tensor2points(tensor, offset=(0., -40., -3.), voxel_size=(.05, .05, .1)):
        """mmdet/core/bbox/transforms.py:218-223: voxel centres [b,x,y,z] of a sparse tensor's rows."""
        ind = tensor.indices.float()
        off, vs = _const(ind.device, offset), _const(ind.device, voxel_size)
        out = ind.clone()
        out[:, 1:] = ind[:, 1:].flip(1) * vs + off + .5 * vs          # columns (3,2,1) without an index tensor upload
        return tensor.features, out

    def forward(self, voxel_features, coors, batch_size, is_test=False, indice_dict=None):
        x = spconv.SparseConvTensor(voxel_features, coors.int(), self.sparse_shape, batch_size)
        if indice_dict is not None:                  # rulebooks built ahead of time (VxNet.precompute_rulebooks)
            x.indice_dict = indice_dict
        x, middle = self.backbone(x)
        x = x.dense()
        n, c, d, h, w = x.shape
        x, conv6 = self.fcn(x.view(n, c * d, h, w))
        if is_test:
            return x, conv6
        # auxiliary network (cmn.py:121-135): multi-scale voxel features interpolated back to the voxel means
        if (self.fused_aux and voxel_features.is_cuda and torch.is_grad_enabled() and voxel_features.shape[0] > 0
                and all(m.features.shape[0] > 0 for m in middle)
                and [m.features.shape[1] for m in middle] == [32, 64, 64]):
            return x, conv6, ((voxel_features, coors, middle, batch_size), None, None)
        points_mean = torch.zeros_like(voxel_features)
        points_mean[:, 0] = coors[:, 0]
        points_mean[:, 1:] = voxel_features[:, :3]
        ps = []
        for m, mult in zip(middle, (2, 4, 8)):
            vs = tuple(v * mult for v in self.aux_voxel_size)
            feat, nxyz = self.tensor2points(m, self.aux_offset, vs)
            # exact for any bin edge (the ring search widens until the third neighbour is inside its reach); four
            # voxels of the scale per bin keeps the 3x3 ring at a few hundred candidates instead of ~1000
            grid = (self.aux_xy_range, min(self.aux_bin, 4 * vs[0]), batch_size)
            ps.append(nearest_neighbor_interpolate(points_mean, nxyz, feat, grid))
        pointwise = self.point_fc(torch.cat(ps, dim=-1))
        return x, conv6, (points_mean, self.point_cls(pointwise), self.point_reg(pointwise))


class SSDRotateHead(nn.Module):
    def __init__(self, num_class=1, num_output_filters=768, num_anchor_per_loc=2, use_sigmoid_cls=True,
                 encode_rad_error_by_sin=True, use_direction_classifier=True, box_coder='GroundBox3dCoder',
                 box_code_size=7):
        super().__init__()
        num_anchor_per_loc *= num_class
        self._num_class = num_class if use_sigmoid_cls else num_class + 1
        self._num_anchor_per_loc = num_anchor_per_loc
        self._use_direction_classifier, self._use_sigmoid_cls = use_direction_classifier, use_sigmoid_cls
        self._encode_rad_error_by_sin, self._box_code_size = encode_rad_error_by_sin, box_code_size
        self.conv_cls = _HipConv2d(num_output_filters, num_anchor_per_loc * self._num_class, 1)
        self.conv_box = _HipConv2d(num_output_filters, num_anchor_per_loc * box_code_size, 1)
        if use_direction_classifier:
            self.conv_dir_cls = _HipConv2d(num_output_filters, num_anchor_per_loc * 2, 1)

    # training: the three 1x1 head convolutions as ONE convolution over the concatenated weights (one forward, one data
    # gradient into the shared input, one weight gradient, one bias reduction instead of three each plus two 72 MB
    # gradient accumulations); the inference plan fuses them the same way.  False: three separate convolutions (A/B).
    fused_head = True

    def forward(self, x):
        n, _, h, w = x.shape
        convs = [self.conv_box, self.conv_cls] + ([self.conv_dir_cls] if self._use_direction_classifier else [])
        if (self.fused_head and x.is_cuda and torch.is_grad_enabled() and all(c.bias is not None for c in convs)
                and (x.requires_grad or convs[0].weight.requires_grad)):
            y = Conv2dFn.apply(x.float(), torch.cat([c.weight for c in convs], 0), torch.cat([c.bias for c in convs], 0),
                               None, None, None)
            ys = torch.split(y, [c.out_channels for c in convs], 1)
        else:
            ys = [conv(x) for conv in convs]
        return tuple(y.reshape(n, self._num_class, -1, h, w).permute(0, 1, 3, 4, 2).contiguous() for y in ys)

    # ---- training (ssd_rotate_head.py:128-314) -----------------------------------------------------------------
    @staticmethod
    def add_sin_difference(boxes1, boxes2):
        r1 = torch.sin(boxes1[..., -1:]) * torch.cos(boxes2[..., -1:])
        r2 = torch.cos(boxes1[..., -1:]) * torch.sin(boxes2[..., -1:])
        return torch.cat((boxes1[..., :-1], r1), dim=-1), torch.cat((boxes2[..., :-1], r2), dim=-1)

    @staticmethod
    def get_direction_target(anchors, reg_targets, use_one_hot=True):
        b = reg_targets.shape[0]
        rot_gt = reg_targets[..., -1] + anchors.view(b, -1, 7)[..., -1]
        t = (rot_gt > 0).long()
        return T.one_hot(t, 2, dtype=anchors.dtype) if use_one_hot else t

    @staticmethod
    def prepare_loss_weights(labels, pos_cls_weight=1.0, neg_cls_weight=1.0, dtype=torch.float32):
        """'NormByNumPositives' (the only normalisation the reference's loss() requests)."""
        cared, positives, negatives = labels >= 0, labels > 0, labels == 0
        cls_w = negatives.type(dtype) * neg_cls_weight + pos_cls_weight * positives.type(dtype)
        reg_w = positives.type(dtype)
        norm = torch.clamp(positives.sum(1, keepdim=True).type(dtype), min=1.0)
        return cls_w / norm, reg_w / norm, cared

    def create_loss(self, box_preds, cls_preds, cls_targets, cls_weights, reg_targets, reg_weights, num_class,
                    use_sigmoid_cls=True, encode_rad_error_by_sin=True, box_code_size=7):
        b = int(box_preds.shape[0])
        box_preds = box_preds.view(b, -1, box_code_size)
        cls_preds = cls_preds.view(b, -1, num_class if use_sigmoid_cls else num_class + 1)
        oh = T.one_hot(cls_targets, depth=num_class + 1, dtype=box_preds.dtype)
        if use_sigmoid_cls:
            oh = oh[..., 1:]
        if encode_rad_error_by_sin:
            box_preds, reg_targets = self.add_sin_difference(box_preds, reg_targets)
        loc = T.weighted_smoothl1(box_preds, reg_targets, beta=1 / 9., weight=reg_weights[..., None], avg_factor=1.)
        cls = T.weighted_sigmoid_focal_loss(cls_preds, oh, weight=cls_weights[..., None], avg_factor=1.)
        return loc, cls

    def _loss_fused(self, box_preds, cls_preds, dir_cls_preds, gt_bboxes, gt_labels, gt_types, anchors, anchors_mask,
                    cfg):
        """The same loss through sassd_assign_targets + sassd_rpn_loss: per class one assignment call for the whole
        batch writing straight into the [B, classes, A] label / target tensors, then ONE kernel for the three loss sums
        and their gradients (instead of ~250 elementwise launches and as many autograd nodes)."""
        b = box_preds.shape[0]
        dev = box_preds.device
        names = list(anchors.keys())
        ncls = len(names)
        a_c = anchors[names[0]].shape[1]
        counts = [int(g.shape[0]) for g in gt_bboxes]
        gt_off = K.gt_offsets(counts, dev)
        gt_all = _gt_cat(gt_bboxes)
        cls_all = torch.cat([l.to(dev).long() for l in gt_labels], 0).contiguous() if sum(counts) else None
        flat = np.concatenate([np.asarray(c) == n for n in names for c in gt_types] or [np.zeros(0, bool)])
        up = torch.from_numpy(flat).pin_memory().to(dev, non_blocking=True)
        labels = torch.empty(b, ncls, a_c, dtype=torch.int64, device=dev)
        targets = torch.empty(b, ncls, a_c, self._box_code_size, dtype=torch.float32, device=dev)
        num_pos = torch.empty(b, dtype=torch.int32, device=dev)
        tot = sum(counts)
        for i, name in enumerate(names):
            mask = anchors_mask[name]
            mask = mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8).contiguous()
            K.assign_targets(anchors[name].contiguous(), mask, gt_all, cls_all,
                             up[i * tot:(i + 1) * tot].view(torch.uint8), gt_off, cfg.assigner[name].pos_iou_thr,
                             cfg.assigner[name].neg_iou_thr, labels[:, i], targets[:, i], num_pos, zero_num_pos=i == 0,
                             out_stride=ncls * a_c)
        all_anchors = anchors[names[0]] if ncls == 1 else torch.stack([anchors[n] for n in names], 1)
        all_anchors = all_anchors.contiguous().view(b, ncls * a_c, 7)
        sums = RpnLossFn.apply(box_preds.view(b, -1, self._box_code_size),
                               cls_preds.view(b, -1, self._num_class),
                               dir_cls_preds.view(b, -1, 2) if self._use_direction_classifier else None,
                               labels.view(b, -1), targets.view(b, -1, self._box_code_size), all_anchors, num_pos)
        terms = _scaled_terms(sums, (2.0 / b, 1.0 / b, 0.2 / b)[:int(sums.shape[0])])
        out = dict(rpn_loc_loss=terms[0], rpn_cls_loss=terms[1])
        if self._use_direction_classifier:
            out['rpn_dir_loss'] = terms[2]
        return out

    def loss(self, box_preds, cls_preds, dir_cls_preds, gt_bboxes, gt_labels, gt_types, anchors, anchors_mask, cfg):
        b = box_preds.shape[0]
        if (box_preds.is_cuda and cfg.get('fused_loss', True) and cfg.assigner.similarity_fn == 'NearestIouSimilarity'
                and self._use_sigmoid_cls and self._encode_rad_error_by_sin and self._box_code_size == 7):
            return self._loss_fused(box_preds, cls_preds, dir_cls_preds, gt_bboxes, gt_labels, gt_types, anchors,
                                    anchors_mask, cfg)
        multi_labels, multi_targets, multi_anchors = [], [], []
        sim = getattr(T, cfg.assigner.similarity_fn, None) or getattr(iou3d_utils, cfg.assigner.similarity_fn)
        dev = box_preds.device
        # class membership of every ground truth, all (class, sample) masks in ONE pinned non-blocking upload
        flat = np.concatenate([np.asarray(c) == n for n in anchors for c in gt_types] or [np.zeros(0, bool)])
        up = torch.from_numpy(flat)
        up = up.pin_memory().to(dev, non_blocking=True) if dev.type == "cuda" else up
        off = 0
        for cls_name, cls_anchor in anchors.items():
            gt_mask = []
            for c in gt_types:
                gt_mask.append(up[off:off + len(c)])
                off += len(c)
            labels, targets, _ = T.multi_apply(
                T.create_target_torch, cls_anchor, anchors_mask[cls_name], gt_bboxes, gt_labels, gt_mask,
                similarity_fn=sim(), box_encoding_fn=T.second_box_encode,
                matched_threshold=cfg.assigner[cls_name].pos_iou_thr,
                unmatched_threshold=cfg.assigner[cls_name].neg_iou_thr, box_code_size=self._box_code_size)
            multi_labels.append(torch.stack(labels))
            multi_targets.append(torch.stack(targets))
            multi_anchors.append(cls_anchor)
        labels = torch.stack(multi_labels, 1).view(b, -1)
        targets = torch.stack(multi_targets, 1).view(b, -1, self._box_code_size)
        anchors = torch.stack(multi_anchors, 1).view(b, -1, self._box_code_size)
        cls_weights, reg_weights, cared = self.prepare_loss_weights(labels)
        cls_targets = labels * cared.type_as(labels)
        loc, cls = self.create_loss(box_preds, cls_preds, cls_targets, cls_weights, targets, reg_weights,
                                    num_class=self._num_class, use_sigmoid_cls=self._use_sigmoid_cls,
                                    encode_rad_error_by_sin=self._encode_rad_error_by_sin,
                                    box_code_size=self._box_code_size)
        out = dict(rpn_loc_loss=loc / b * 2, rpn_cls_loss=cls / b)
        if self._use_direction_classifier:
            dir_labels = self.get_direction_target(anchors, targets, use_one_hot=False).view(-1)
            w = (labels > 0).type_as(dir_cls_preds)
            w = w / torch.clamp(w.sum(-1, keepdim=True), min=1.0)
            out['rpn_dir_loss'] = T.weighted_cross_entropy(dir_cls_preds.view(-1, 2), dir_labels, weight=w.view(-1),
                                                           avg_factor=1.) / b * .2
        return out

    # decode / flip / ground-truth prefix of the padded guided anchors in one fused kernel each way; False keeps the torch
    # gather / split / cat formulation (A/B, parity tests)
    fused_tail = True

    def get_guided_anchors_padded(self, box_preds, cls_preds, dir_cls_preds, anchors, anchors_mask, gt_bboxes, thr=.1,
                                  cap=None):
        """The training-mode selection of get_guided_anchors without its host round trip (boolean compaction): the
        selected anchor indices come from sassd_guided_select into a fixed-capacity buffer with a DEVICE count, the
        decode / direction flip run on the padded [B, cap, 7] gather, and the result is one [B, Gmax + cap, 7] tensor
        whose rows [0, G_b) are the sample's ground truth, [G_b, G_b + K_b) the selected boxes in ascending anchor
        order and the rest padding (count[b] = G_b + K_b stays on the device).  Gradients reach box_preds through the
        gather.  -> (guided [B, Gmax+cap, 7], counts [B] int32)."""
        b = box_preds.shape[0]
        dev = box_preds.device
        if isinstance(anchors, dict):               # (one class: the tensor itself -- a cat of one tensor is a copy)
            anchors = torch.cat(list(anchors.values()), 1) if len(anchors) > 1 else next(iter(anchors.values()))
        if isinstance(anchors_mask, dict):
            anchors_mask = torch.cat(list(anchors_mask.values()), 1) if len(anchors_mask) > 1 else next(iter(anchors_mask.values()))
        a = anchors.view(b, -1, 7).shape[1]
        cap = min(int(cap), a) if cap else a        # default: every anchor fits (padding rows cost next to nothing)
        mask = anchors_mask.view(b, -1)
        mask = mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8).contiguous()
        if getattr(self, "_guided_overflow", None) is None or self._guided_overflow.device != dev:
            self._guided_overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        sel, cnt = K.guided_select(cls_preds.detach().reshape(b, a, self._num_class).contiguous(), mask, thr, cap,
                                   self._guided_overflow)
        counts = [int(g.shape[0]) for g in gt_bboxes]
        gmax = max(counts) if counts else 0
        if self.fused_tail:
            # decode + direction flip + ground-truth prefix in one kernel each way (sassd_guided_decode_*)
            gt_all = _gt_cat(gt_bboxes)
            guided, total = GuidedDecodeFn.apply(box_preds.view(b, a, self._box_code_size),
                                                 dir_cls_preds.view(b, a, 2) if self._use_direction_classifier else None,
                                                 anchors.view(b, a, 7).contiguous(), sel, cnt, gt_all,
                                                 K.gt_offsets(counts, dev), gmax)
            return guided, total
        s7 = sel.unsqueeze(-1).expand(b, cap, 7)
        box = T.second_box_decode(box_preds.view(b, a, self._box_code_size).gather(1, s7),
                                  anchors.view(b, a, 7).gather(1, s7))
        if self._use_direction_classifier:
            dirp = dir_cls_preds.view(b, a, 2).gather(1, sel.unsqueeze(-1).expand(b, cap, 2))
            opp = (box[..., -1] > 0) ^ (dirp[..., 1] > dirp[..., 0])          # argmax of two = (second > first)
            box = torch.cat([box[..., :-1], (box[..., -1] + opp.type_as(box) * np.pi).unsqueeze(-1)], dim=-1)
        rows = []
        for i in range(b):
            parts = [gt_bboxes[i].type_as(box), box[i]] if counts[i] else [box[i]]
            if gmax > counts[i]:
                parts.append(box.new_zeros(gmax - counts[i], 7))
            rows.append(torch.cat(parts, 0) if len(parts) > 1 else parts[0])
        guided = torch.stack(rows, 0)
        return guided, cnt + _const_i32(dev, counts)

    def check_guided_capacity(self):
        """Raises if an earlier get_guided_anchors_padded call overflowed its capacity (one host read: call it where
        the step synchronises anyway)."""
        f = getattr(self, "_guided_overflow", None)
        if f is not None and int(f.item()) != 0:
            f.zero_()
            raise RuntimeError("guided anchors exceeded the padded capacity: raise train_cfg.rpn.guided_cap")

    def poll_guided_capacity(self):
        """check_guided_capacity without blocking: the flag travels to pinned host memory by an asynchronous copy and is
        looked at once that copy has completed (i.e. an overflow is reported a step or two after it happened)."""
        f = getattr(self, "_guided_overflow", None)
        if f is None:
            return
        st = getattr(self, "_guided_poll", None)
        if st is not None and st[1].query():
            if int(st[0][0]) != 0:
                f.zero_()
                self._guided_poll = None
                raise RuntimeError("guided anchors exceeded the padded capacity: raise train_cfg.rpn.guided_cap")
            st = None
        if st is None:
            host = torch.zeros(1, dtype=torch.int32).pin_memory()
            host.copy_(f, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._guided_poll = (host, ev)

    def get_guided_anchors(self, box_preds, cls_preds, dir_cls_preds, anchors, anchors_mask, gt_bboxes, gt_labels,
                           thr=.1):
        """ssd_rotate_head.py:316-388, module-level torch path used by training (gradients flow into box_preds;
        ground-truth boxes are prepended).  Inference uses the fused sassd_decode_filter kernel instead."""
        b = box_preds.shape[0]
        if isinstance(anchors, dict):               # (one class: the tensor itself -- a cat of one tensor is a copy)
            anchors = torch.cat(list(anchors.values()), 1) if len(anchors) > 1 else next(iter(anchors.values()))
        if isinstance(anchors_mask, dict):
            anchors_mask = torch.cat(list(anchors_mask.values()), 1) if len(anchors_mask) > 1 else next(iter(anchors_mask.values()))
        batch_box = T.second_box_decode(box_preds.view(b, -1, self._box_code_size), anchors.view(b, -1, 7))
        batch_mask = anchors_mask.view(b, -1).bool()
        batch_cls = cls_preds.view(b, -1, self._num_class)
        batch_dir = dir_cls_preds.view(b, -1, 2)
        gt_bboxes = gt_bboxes if gt_bboxes is not None else [None] * b
        gt_labels = gt_labels if gt_labels is not None else [None] * b
        guided, labels_out = [], []
        for box, cls, dirp, m, gtb, gtl in zip(batch_box, batch_cls, batch_dir, batch_mask, gt_bboxes, gt_labels):
            scores = torch.sigmoid(cls) if self._use_sigmoid_cls else torch.softmax(cls, dim=-1)[..., 1:]
            if self._num_class == 1:
                top_scores = scores.squeeze(-1)
                top_labels = torch.zeros(scores.shape[0], dtype=torch.int64, device=scores.device)
            else:
                top_scores, top_labels = torch.max(scores, dim=-1)
            # one compaction (one host sync) for mask AND threshold; ascending anchor order like the reference's
            # two successive boolean selections
            sel = torch.nonzero(m & (top_scores > thr)).view(-1)
            box, top_labels = box[sel], top_labels[sel]
            dir_labels = torch.max(dirp[sel], dim=-1)[1]
            if self._use_direction_classifier:
                opp = (box[..., -1] > 0) ^ dir_labels.bool()
                box = torch.cat([box[:, :-1], (box[:, -1] + opp.type_as(box) * np.pi)[:, None]], dim=1)
            if gtb is not None:
                box = torch.cat([gtb.type_as(box), box], 0)
                top_labels = torch.cat([gtl.to(top_labels.device), top_labels], 0)
            guided.append(box)
            labels_out.append(top_labels)
        return guided, labels_out


class PSWarpHead(nn.Module):
    def __init__(self, grid_offsets, featmap_stride, in_channels, num_class=1, num_parts=49):
        super().__init__()
        self._num_class = num_class
        self.grid_offsets, self.featmap_stride = grid_offsets, featmap_stride
        oc = num_class * num_parts
        self.convs = nn.Sequential(_HipConv2d(in_channels, oc, 3, 1, padding=1, bias=False),
                                   nn.BatchNorm2d(oc, eps=1e-3, momentum=0.01), nn.ReLU(inplace=True),
                                   _HipConv2d(oc, oc, 1, 1, padding=0, bias=False))

    def forward(self, x, guided_anchors, is_test=False):
        """guided_anchors: list (per sample) of [K,7] device tensors -> list of [K] logits."""
        grad = torch.is_grad_enabled() and (x.requires_grad or self.convs[0].weight.requires_grad)
        if self.training or grad:
            f = self._convs_train(x)
        else:
            s, b = _bn_affine(self.convs[1])
            f = self.convs[3].hip_forward(self.convs[0].hip_forward(x, s, b, True))
        scores = []
        for i, ga in enumerate(guided_anchors):
            k = ga.shape[0]
            if k == 0:
                scores.append(torch.empty(0, device=x.device))
                continue
            if grad:
                scores.append(PSWarpFn.apply(f[i:i + 1], ga.float(), tuple(self.grid_offsets),
                                             1.0 / self.featmap_stride))
                continue
            cnt = torch.full((1,), k, dtype=torch.int32, device=x.device)
            lg = K.pswarp_sample(f[i:i + 1].contiguous(), ga.contiguous().view(1, k, 7), cnt, k, self.grid_offsets,
                                 1.0 / self.featmap_stride)
            scores.append(lg.view(-1))
        return scores if is_test else torch.cat(scores, 0)

    def _convs_train(self, x):
        """self.convs(x) with the BatchNorm2d + ReLU pair on the fused kernels."""
        return self.convs[3](bn_relu_2d(self.convs[1], self.convs[0](x)))

    fused_tail = True          # batched 3-D IoU + fused focal loss in loss_padded (False: the torch formulation, A/B)

    def forward_padded(self, x, guided, counts):
        """guided [B, capK, 7] padded, counts [B] int32 (device) -> logits [B, capK] (zero past a sample's count)."""
        return PSWarpBatchFn.apply(self._convs_train(x), guided.float(), counts, tuple(self.grid_offsets),
                                   1.0 / self.featmap_stride)

    def loss_padded(self, logits, gt_bboxes, guided, counts, cfg):
        """loss() on the padded batch: rotated 3-D IoU per sample (HIP overlap kernel), ONE sassd_assign_targets call for
        the labels of the whole batch (rows past a sample's count are masked to -1 and so carry no weight), focal loss
        normalised by the positives of the batch."""
        b, capk = logits.shape
        dev = logits.device
        g_counts = [int(g.shape[0]) for g in gt_bboxes]
        tot = sum(g_counts)
        labels = torch.empty(b, capk, dtype=torch.int64, device=dev)
        targets = torch.empty(b, capk, 7, dtype=torch.float32, device=dev)
        num_pos = torch.empty(b, dtype=torch.int32, device=dev)
        boxes = guided.detach().float().contiguous()
        row_ok = (_arange_i32(dev, capk)[None, :] < counts[:, None]).view(torch.uint8)
        if cfg.assigner.similarity_fn != 'RotateIou3dSimilarity':
            raise NotImplementedError("padded rescoring loss: RotateIou3dSimilarity only")
        offs, o = [0], 0
        for i in range(b):
            o += capk * g_counts[i]
            offs.append(o)
        gt_all = _gt_cat(gt_bboxes)
        if self.fused_tail and tot:
            # every sample's rotated 3-D IoU matrix in one launch (sassd_boxes_iou3d_batch)
            ov = K.boxes_iou3d_batch(boxes, counts.contiguous(), gt_all, K.gt_offsets(g_counts, dev), max(g_counts),
                                     _const_i64(dev, offs), o)
        else:
            ovs = [iou3d_utils.boxes_iou3d_gpu(boxes[i], gt_bboxes[i].float()).reshape(-1) for i in range(b) if g_counts[i]]
            ov = torch.cat(ovs) if len(ovs) > 1 else (ovs[0] if ovs else None)
        K.assign_targets(boxes, row_ok.contiguous(), gt_all, None, None, K.gt_offsets(g_counts, dev),
                         cfg.assigner.pos_iou_thr, cfg.assigner.neg_iou_thr,
                         labels, targets, num_pos, overlaps=ov.contiguous() if ov is not None else boxes,
                         overlap_offsets=_const_i64(dev, offs))
        if self.fused_tail and self._num_class == 1:
            return dict(loss_cls=FocalLossFn.apply(logits, labels, num_pos) / b)
        labels = labels.view(-1, 1)
        cared, positives = labels >= 0, labels > 0
        w = cared.float() / torch.clamp(num_pos.sum().float(), min=1.0)
        cls_targets = labels * cared.type_as(labels)
        cls = T.weighted_sigmoid_focal_loss(logits.reshape(-1, self._num_class), cls_targets.float(), weight=w,
                                            avg_factor=1.)
        return dict(loss_cls=cls / b)

    def loss(self, cls_preds, gt_bboxes, gt_labels, anchors, cfg):
        """ssd_rotate_head.py:456-490: class-agnostic rescoring targets from the rotated 3-D IoU (HIP overlap kernel)."""
        b = len(anchors)
        none = (None,) * b
        sim = getattr(T, cfg.assigner.similarity_fn, None) or getattr(iou3d_utils, cfg.assigner.similarity_fn)
        labels, _, _ = T.multi_apply(T.create_target_torch, [a.detach() for a in anchors], none, gt_bboxes, none,
                                     none, similarity_fn=sim(), box_encoding_fn=T.second_box_encode,
                                     matched_threshold=cfg.assigner.pos_iou_thr,
                                     unmatched_threshold=cfg.assigner.neg_iou_thr)
        labels = torch.cat(labels).unsqueeze(1)
        cared, positives, negatives = labels >= 0, labels > 0, labels == 0
        w = negatives.float() + positives.float()
        w = w / torch.clamp(positives.sum().float(), min=1.0)
        cls_targets = labels * cared.type_as(labels)
        cls = T.weighted_sigmoid_focal_loss(cls_preds.view(-1, self._num_class), cls_targets.float(), weight=w,
                                            avg_factor=1.)
        return dict(loss_cls=cls / b)


class SingleStageDetector(nn.Module):
    def __init__(self, backbone, neck=None, bbox_head=None, extra_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None):
        super().__init__()
        me = sys.modules[__name__]
        self.backbone = obj_from_dict(backbone, me)
        if neck is None:
            raise NotImplementedError
        self.neck = obj_from_dict(neck, me)
        if bbox_head is not None:
            self.rpn_head = obj_from_dict(bbox_head, me)
        if extra_head is not None:
            self.extra_head = obj_from_dict(extra_head, me)
        self.train_cfg, self.test_cfg = _wrap(train_cfg), _wrap(test_cfg)
        self.class_names = None
        self._plan, self._plan_key = None, None
        self._cfg = dict(num_class=bbox_head.get('num_class', 1) if bbox_head else 1,
                         sparse_shape=neck['output_shape'],
                         grid_offsets=extra_head['grid_offsets'] if extra_head else (0., 40.),
                         featmap_stride=extra_head['featmap_stride'] if extra_head else .4)
        if isinstance(pretrained, str):
            from .train import load_params_from_file      # accepts the reference's 'module.'-prefixed checkpoints
            load_params_from_file(self, pretrained, to_cpu=True)

    def _apply(self, fn, *args, **kwargs):
        """.to() / .cuda() / .float(): the parameters become new tensors -- cached weight images and the plan are stale."""
        out = super()._apply(fn, *args, **kwargs)
        K.bump_weights_generation()
        self._plan, self._plan_key = None, None
        return out

    def train(self, mode=True):
        if mode:
            self._plan, self._plan_key = None, None      # weights are about to change: the folded plan is stale
        else:
            from .autograd import flush_bn_counters
            flush_bn_counters()
        return super().train(mode)

    def state_dict(self, *args, **kwargs):
        from .autograd import flush_bn_counters
        flush_bn_counters()                              # deferred num_batches_tracked increments land first
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        from .autograd import flush_bn_counters
        flush_bn_counters()                              # ... and must not be added on top of loaded counters later
        return super().load_state_dict(*args, **kwargs)

    @property
    def with_rpn(self):
        return hasattr(self, 'rpn_head') and self.rpn_head is not None

    def merge_second_batch(self, batch_args):
        """single_stage.py:52-73."""
        ret = {}
        merged = batch_args.get('sassd_merged')          # sassd.train.device_batch: the batch is already one buffer
        for key, elems in batch_args.items():
            if key == 'sassd_merged':
                continue
            if merged is not None and key in merged:
                ret[key] = merged[key]
            elif key in ('voxels', 'num_points'):
                ret[key] = torch.cat(elems, dim=0)
            elif key == 'coordinates':
                ret[key] = torch.cat([nn.functional.pad(c, [1, 0, 0, 0], mode='constant', value=i)
                                      for i, c in enumerate(elems)], dim=0)
            elif key in ('img_meta', 'gt_labels', 'gt_bboxes', 'gt_types', 'sassd_rulebooks'):
                ret[key] = elems
            elif isinstance(elems, dict):
                ret[key] = {k: torch.stack(v, dim=0) for k, v in elems.items()}
            else:
                ret[key] = torch.stack(list(elems), dim=0)
        return ret

    def plan(self, batch_size, anchors, device, anchors_src=None, **kw):
        """Build (and cache) the fused inference plan for this batch size / anchor set.  `anchors_src`: the caller's own
        anchor tensor (the dataset hands the same object to every frame); when given, a host-side fingerprint of it
        (identity, address, autograd version, shape) decides whether the cached plan still matches -- no device
        comparison, no host sync per frame.  Without it the anchor CONTENT is compared (merge_second_batch stacks a
        fresh tensor per call, so the stacked tensor's address says nothing)."""
        key = (batch_size, str(device), K.weights_generation(),
               sum(t._version for t in list(self.parameters()) + list(self.buffers())))
        an_t = anchors if torch.is_tensor(anchors) else torch.as_tensor(np.asarray(anchors))
        same = self._plan is not None and self._plan_key == key and self._plan.anchors.shape == an_t.reshape(-1, 7).shape
        fp = None
        if anchors_src is not None and torch.is_tensor(anchors_src):
            fp = (id(anchors_src), anchors_src.data_ptr(), anchors_src._version, tuple(anchors_src.shape))
        if same and not (fp is not None and fp == getattr(self, "_plan_anchor_fp", None)):
            same = torch.equal(self._plan.anchors, an_t.reshape(-1, 7).to(self._plan.anchors))
        if not same:
            tc = self.test_cfg.get('extra', self.test_cfg) if self.test_cfg else {}
            an = an_t.detach().cpu().numpy()
            self._plan = InferencePlan(self.state_dict(), batch_size=batch_size, anchors=an.reshape(-1, 7),
                                       score_thr=tc.get('score_thr', 0.3),
                                       iou_thr=tc.get('nms', {}).get('iou_thr', 0.1), device=device, **self._cfg, **kw)
            self._plan_key = key
        # (the source tensor is kept alive so that its id / address cannot be recycled by another tensor)
        self._plan_anchor_fp, self._plan_anchor_src = fp, anchors_src
        return self._plan

    def forward_train(self, img, img_meta, **kwargs):
        """single_stage.py:75-108 -> dict of loss tensors."""
        batch_size = len(img_meta)
        ret = self.merge_second_batch(kwargs)
        vx = self.backbone(ret['voxels'], ret['num_points'])
        x, conv6, point_misc = self.neck(vx, ret['coordinates'], batch_size, is_test=False,
                                         indice_dict=ret.get('sassd_rulebooks'))
        losses = dict()
        losses.update(self.neck.aux_loss(*point_misc, gt_bboxes=ret['gt_bboxes']))
        if not self.with_rpn:
            raise NotImplementedError
        rpn_outs = self.rpn_head(x)
        losses.update(self.rpn_head.loss(*rpn_outs, ret['gt_bboxes'], ret['gt_labels'], ret['gt_types'],
                                         ret['anchors'], ret['anchors_mask'], self.train_cfg.rpn))
        rpn_cfg = self.train_cfg.rpn
        extra = getattr(self, 'extra_head', None)
        if (x.is_cuda and rpn_cfg.get('padded_guided', True) and self.rpn_head._use_sigmoid_cls and extra is not None
                and self.train_cfg.extra.assigner.similarity_fn == 'RotateIou3dSimilarity'):
            # no host synchronisation between the backbone and the last loss term: the next step's forward can be
            # queued while this step's backward still runs
            guided, counts = self.rpn_head.get_guided_anchors_padded(
                *rpn_outs, ret['anchors'], ret['anchors_mask'], ret['gt_bboxes'], thr=rpn_cfg.anchor_thr,
                cap=rpn_cfg.get('guided_cap'))
            score = extra.forward_padded(conv6, guided, counts)
            losses.update(extra.loss_padded(score, ret['gt_bboxes'], guided, counts, self.train_cfg.extra))
            return losses
        guided, _ = self.rpn_head.get_guided_anchors(*rpn_outs, ret['anchors'], ret['anchors_mask'], ret['gt_bboxes'],
                                                     ret['gt_labels'], thr=rpn_cfg.anchor_thr)
        if extra is not None:
            score = extra(conv6, guided)
            losses.update(extra.loss(score, ret['gt_bboxes'], ret['gt_labels'], guided, self.train_cfg.extra))
        return losses

    def forward_test(self, img, img_meta, **kwargs):
        """single_stage.py:110-131 on the fused pipeline.  A sample whose img_meta carries 'calib' (and 'img_shape')
        comes back as the reference's KITTI result annotation (kitti_bbox2results, camera frame, ready for
        sassd.kitti_eval.get_official_eval_result); without calibration the lidar-frame detections are returned as
        {boxes_lidar [k,7], scores [k], labels [k]} (numpy)."""
        batch_size = len(img_meta)
        ret = self.merge_second_batch(kwargs)
        dev = ret['voxels'].device
        anchors = ret['anchors']
        src = kwargs.get('anchors')
        plan = self.plan(batch_size, anchors[0], dev,
                         anchors_src=src[0] if isinstance(src, (list, tuple)) and len(src) else None)
        vx = self.backbone(ret['voxels'], ret['num_points'])
        plan.run_from_voxels(vx, ret['coordinates'], ret['anchors_mask'])
        out = []
        for (boxes, scores, labels), meta in zip(plan.results(), img_meta):
            if isinstance(meta, dict) and meta.get('calib') is not None:
                out.append(kitti_bbox2results(boxes, scores, labels, meta, class_names=self.class_names))
            else:
                out.append(dict(boxes_lidar=boxes, scores=scores, labels=labels))
        return out

    def forward(self, img, img_meta, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_meta, **kwargs)
        return self.forward_test(img, img_meta, **kwargs)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """mmdet/models/builder.py:54-56."""
    # a new model: packed-weight caches are keyed on (address, version, generation) -- a fresh parameter may land on the
    # address of a dead one, so start a new generation
    K.bump_weights_generation()
    return obj_from_dict(cfg, sys.modules[__name__], dict(train_cfg=train_cfg, test_cfg=test_cfg))
