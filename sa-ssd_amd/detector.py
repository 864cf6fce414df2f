"""Host-side mirror of the reference model/config API for the hot path (same class names, constructor
arguments, parameter names -> reference checkpoints load with `load_state_dict`):

  SimpleVoxel          mmdet/models/backbones/vxnet.py:99-116
  SpMiddleFHD / VxNet / BEVNet   mmdet/models/necks/cmn.py:12-282
  SSDRotateHead        mmdet/models/single_stage_heads/ssd_rotate_head.py:93-372   (test path)
  PSWarpHead           ssd_rotate_head.py:416-533                                    (test path)
  SingleStageDetector  mmdet/models/detectors/single_stage.py:13-131, base.py:77-81
  build_detector       mmdet/models/builder.py:54-56

`model(img, img_meta, return_loss=False, voxels=[..], coordinates=[..], num_points=[..], anchors=[..],
anchors_mask=[..])` runs the fused HIP pipeline (sassd.pipeline.InferencePlan).  The training branch
(forward_train: losses, target assignment, aux head) is outside this round's scope and raises.
"""
import sys

import numpy as np
import torch
from torch import nn

from . import kernels as K
from . import spconv
from .config import obj_from_dict
from .pipeline import InferencePlan


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

    def hip_forward(self, x, scale=None, shift=None, relu=False):
        v = self.weight._version
        if getattr(self, "_pk", None) is None or self._pkv != v or self._pk.device != self.weight.device:
            self._pk, self._pkv = K.conv2d_pack_weight(self.weight.detach().float().contiguous()), v
        if shift is None and self.bias is not None:
            shift = self.bias.detach().float().contiguous()
        return K.conv2d_fwd(x.contiguous().float(), self._pk, self.out_channels, self.kernel_size[0], scale, shift,
                            relu)

    def forward(self, x):
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

    def forward(self, x):
        conv6 = None
        for i in range(8):
            conv, bn = getattr(self, 'conv%d' % i), getattr(self, 'bn%d' % i)
            if self.training:
                raise NotImplementedError("training-mode BatchNorm is outside this round's scope")
            s, b = _bn_affine(bn)
            x = conv.hip_forward(x, s, b, True)
            if i == 6:
                conv6 = x
        return x, conv6


class SpMiddleFHD(nn.Module):
    def __init__(self, output_shape, num_input_features=4, num_hidden_features=128):
        super().__init__()
        self.sparse_shape = output_shape
        self.backbone = VxNet(num_input_features)
        self.fcn = BEVNet(in_features=num_hidden_features, num_filters=256)
        self.point_fc = nn.Linear(160, 64, bias=False)        # training-only auxiliary head (cmn.py:27-29)
        self.point_cls = nn.Linear(64, 1, bias=False)
        self.point_reg = nn.Linear(64, 3, bias=False)

    def forward(self, voxel_features, coors, batch_size, is_test=False):
        if not is_test:
            raise NotImplementedError("auxiliary-network training branch (cmn.py:121-135) not in scope yet")
        x = spconv.SparseConvTensor(voxel_features, coors.int(), self.sparse_shape, batch_size)
        x, middle = self.backbone(x)
        x = x.dense()
        n, c, d, h, w = x.shape
        return self.fcn(x.view(n, c * d, h, w))


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

    def forward(self, x):
        n, _, h, w = x.shape
        outs = []
        for conv in (self.conv_box, self.conv_cls, self.conv_dir_cls):
            y = conv(x)
            outs.append(y.view(n, self._num_class, -1, h, w).permute(0, 1, 3, 4, 2).contiguous())
        return tuple(outs)


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
        s, b = _bn_affine(self.convs[1])
        f = self.convs[3].hip_forward(self.convs[0].hip_forward(x, s, b, True))
        scores = []
        for i, ga in enumerate(guided_anchors):
            k = ga.shape[0]
            if k == 0:
                scores.append(torch.empty(0, device=x.device))
                continue
            cnt = torch.tensor([k], dtype=torch.int32, device=x.device)
            lg = K.pswarp_sample(f[i:i + 1].contiguous(), ga.contiguous().view(1, k, 7), cnt, k, self.grid_offsets,
                                 1.0 / self.featmap_stride)
            scores.append(lg.view(-1))
        return scores if is_test else torch.cat(scores, 0)


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
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.class_names = None
        self._plan, self._plan_key = None, None
        self._cfg = dict(num_class=bbox_head.get('num_class', 1) if bbox_head else 1,
                         sparse_shape=neck['output_shape'],
                         grid_offsets=extra_head['grid_offsets'] if extra_head else (0., 40.),
                         featmap_stride=extra_head['featmap_stride'] if extra_head else .4)
        if isinstance(pretrained, str):
            self.load_state_dict(torch.load(pretrained, map_location='cpu').get('model_state', {}), strict=False)

    @property
    def with_rpn(self):
        return hasattr(self, 'rpn_head') and self.rpn_head is not None

    def merge_second_batch(self, batch_args):
        """single_stage.py:52-73."""
        ret = {}
        for key, elems in batch_args.items():
            if key in ('voxels', 'num_points'):
                ret[key] = torch.cat(elems, dim=0)
            elif key == 'coordinates':
                ret[key] = torch.cat([nn.functional.pad(c, [1, 0, 0, 0], mode='constant', value=i)
                                      for i, c in enumerate(elems)], dim=0)
            elif key in ('img_meta', 'gt_labels', 'gt_bboxes', 'gt_types'):
                ret[key] = elems
            elif isinstance(elems, dict):
                ret[key] = {k: torch.stack(v, dim=0) for k, v in elems.items()}
            else:
                ret[key] = torch.stack(list(elems), dim=0)
        return ret

    def plan(self, batch_size, anchors, device, **kw):
        """Build (and cache) the fused inference plan for this batch size / anchor set."""
        key = (batch_size, anchors.data_ptr() if torch.is_tensor(anchors) else id(anchors), str(device))
        if self._plan is None or self._plan_key != key:
            tc = self.test_cfg.get('extra', self.test_cfg) if self.test_cfg else {}
            an = anchors.detach().cpu().numpy() if torch.is_tensor(anchors) else np.asarray(anchors)
            self._plan = InferencePlan(self.state_dict(), batch_size=batch_size, anchors=an.reshape(-1, 7),
                                       score_thr=tc.get('score_thr', 0.3),
                                       iou_thr=tc.get('nms', {}).get('iou_thr', 0.1), device=device, **self._cfg, **kw)
            self._plan_key = key
        return self._plan

    def forward_train(self, img, img_meta, **kwargs):
        raise NotImplementedError("training path (losses / target assignment / aux head) is a later round")

    def forward_test(self, img, img_meta, **kwargs):
        """single_stage.py:110-131 on the fused pipeline.  Returns per-sample dicts
        {boxes_lidar [k,7], scores [k], labels [k]} (numpy); KITTI camera-frame annos need calib files."""
        batch_size = len(img_meta)
        ret = self.merge_second_batch(kwargs)
        dev = ret['voxels'].device
        anchors = ret['anchors']
        plan = self.plan(batch_size, anchors[0], dev)
        vx = self.backbone(ret['voxels'], ret['num_points'])
        plan.run_from_voxels(vx, ret['coordinates'], ret['anchors_mask'])
        out = []
        for boxes, scores, labels in plan.results():
            out.append(dict(boxes_lidar=boxes, scores=scores, labels=labels))
        return out

    def forward(self, img, img_meta, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_meta, **kwargs)
        return self.forward_test(img, img_meta, **kwargs)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """mmdet/models/builder.py:54-56."""
    return obj_from_dict(cfg, sys.modules[__name__], dict(train_cfg=train_cfg, test_cfg=test_cfg))
