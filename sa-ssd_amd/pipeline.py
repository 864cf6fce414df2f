"""Device-resident whole-frame inference plan: raw points (or reference-style voxel inputs) -> detections.

This is the production path of the drop-in: the same kernels the per-op facades call, chained on ONE HIP stream
with every data-dependent size kept in device memory, so a frame is a fixed launch sequence with zero host syncs
between the first kernel and the final result read-back (the reference path has >= 6 syncs per sample,
SURVEY.md 3.1).  Eval-mode BatchNorm is folded into per-channel (scale, shift) applied in the conv epilogues.

Stages / reference lines:
  voxelize + mean      points_ops.py:104-164, vxnet.py:110-116          sassd_voxelize
  rulebooks x7         cmn.py:147-173 (spconv get_indice_pairs)         sassd_hash_build / rulebook_subm / rulebook_conv
  sparse convs x14     cmn.py:192-231 (+BN1d+ReLU)                      sassd_spconv_fwd
  dense()              cmn.py:112-114                                   sassd_densify (d-major channels, conv0 permuted)
  BEVNet x8            cmn.py:233-282                                   sassd_conv2d_fwd (fp32 MFMA)
  SSD head (fused 1x1) ssd_rotate_head.py:218-235                       sassd_conv2d_fwd
  anchors_mask         kitti.py:333-343                                 sassd_anchor_mask
  guided anchors       ssd_rotate_head.py:307-372                       sassd_decode_filter
  PSWarp               ssd_rotate_head.py:416-447                       sassd_conv2d_fwd x2 + sassd_pswarp_sample
  rescore + NMS        ssd_rotate_head.py:487-533                       sassd_rescore_nms
"""
import numpy as np

import torch

from . import kernels as K
from . import anchors as A

BN_EPS = 1e-3

# (state-dict prefix under neck.backbone, bn prefix, kind, Cin, Cout, rulebook key) -- VxNet, cmn.py:197-212
VXNET = [
    ("conv0.0", "conv0.1", "subm", 4, 16, "subm0"), ("conv0.3", "conv0.4", "subm", 16, 16, "subm0"),
    ("down0.0", "down0.1", "down", 16, 32, "down0"),
    ("conv1.0", "conv1.1", "subm", 32, 32, "subm1"), ("conv1.3", "conv1.4", "subm", 32, 32, "subm1"),
    ("down1.0", "down1.1", "down", 32, 64, "down1"),
    ("conv2.0", "conv2.1", "subm", 64, 64, "subm2"), ("conv2.3", "conv2.4", "subm", 64, 64, "subm2"),
    ("conv2.6", "conv2.7", "subm", 64, 64, "subm2"),
    ("down2.0", "down2.1", "down", 64, 64, "down2"),
    ("conv3.0", "conv3.1", "subm", 64, 64, "subm3"), ("conv3.3", "conv3.4", "subm", 64, 64, "subm3"),
    ("conv3.6", "conv3.7", "subm", 64, 64, "subm3"),
    ("extra_conv.0", "extra_conv.1", "1x1", 64, 64, None),
]


def fold_bn(sd, prefix, eps=BN_EPS):
    g, b = sd[prefix + ".weight"].float(), sd[prefix + ".bias"].float()
    m, v = sd[prefix + ".running_mean"].float(), sd[prefix + ".running_var"].float()
    scale = g / torch.sqrt(v + eps)
    return scale.contiguous(), (b - m * scale).contiguous()


class InferencePlan:
    """Pre-packed weights + pre-allocated buffers for a fixed (batch_size, config)."""

    def __init__(self, state_dict, *, batch_size=1, num_class=1, voxel_size=(0.05, 0.05, 0.1),
                 point_cloud_range=(0, -40., -3., 70.4, 40., 1.), max_num_points=5, max_voxels=20000,
                 sparse_shape=(40, 1600, 1408), anchors=None, anchors_bv=None, anchor_area_threshold=1,
                 anchors_per_loc=2, grid_offsets=(0., 40.), featmap_stride=0.4, rpn_thr=0.1, score_thr=0.3,
                 iou_thr=0.1, cap_k=4096, cap_d=512, device=None, level_cap_factor=2, overlap=True, winograd=True,
                 fused_rulebooks=True, chain_bev=True, pyramid_persistent=False, spconv_cfg=None, wino4_cfg=None,
                 skip_inactive_tiles=True, rb_sync_levels=(0, 1, 2, 3), ps_tail=False):
        dev = torch.device(device if device is not None else "cuda:0")
        self.dev, self.B, self.ncls, self.A = dev, int(batch_size), int(num_class), int(anchors_per_loc)
        self.voxel_size = np.asarray(voxel_size, np.float32)
        self.pc_range = np.asarray(point_cloud_range, np.float32)
        self.T, self.max_voxels = int(max_num_points), int(max_voxels)
        self.shape0 = tuple(int(s) for s in sparse_shape)
        self.rpn_thr, self.score_thr, self.iou_thr = float(rpn_thr), float(score_thr), float(iou_thr)
        self.grid_offsets, self.spatial_scale = grid_offsets, 1.0 / featmap_stride
        self.area_thr = anchor_area_threshold
        self.capK, self.capD = int(cap_k), int(cap_d)
        sd = {k: v.detach().to(dev) for k, v in state_dict.items()}

        # ---- level geometry / capacities ----------------------------------------------------------
        self.shapes = [self.shape0]
        for _ in range(3):
            self.shapes.append(K.conv_out_shape(self.shapes[-1]))
        cap0 = self.B * self.max_voxels
        self.caps = [cap0] + [cap0 * level_cap_factor] * 3
        D3, self.H, self.W = self.shapes[3]
        self.D3 = D3

        # ---- sparse weights -----------------------------------------------------------------------
        self.sp = []
        for wname, bnname, kind, cin, cout, key in VXNET:
            w = sd["neck.backbone.%s.weight" % wname].float()
            k = 1 if kind == "1x1" else 27
            wp = K.spconv_pack_weight(w.reshape(k, cin, cout).contiguous())
            scale, shift = fold_bn(sd, "neck.backbone.%s" % bnname)
            self.sp.append((kind, cin, cout, key, wp, scale, shift))

        # ---- dense weights ------------------------------------------------------------------------
        self.bev = []
        for i in range(8):
            w = sd["neck.fcn.conv%d.weight" % i].float()
            if i == 0:      # densify writes d-major channels (d*C + c); reference order is c*D + d (cmn.py:113-114)
                cout, cin = w.shape[:2]
                c = cin // D3
                w = w.view(cout, c, D3, 3, 3).permute(0, 2, 1, 3, 4).reshape(cout, cin, 3, 3)
            scale, shift = fold_bn(sd, "neck.fcn.bn%d" % i)
            # 3x3 layers: Winograd F(4x4,3x3) (transform + 36 MFMA GEMMs + transform: 4x fewer multiplications) when the
            # shape allows, else the fused F(2x2,3x3) kernel (2.25x fewer), else the direct kernel.  `winograd` = 2
            # forces F(2x2), 0 / False the direct kernel (A/B)
            wino = 0
            if winograd and w.shape[2] == 3:
                if winograd != 2 and K.conv2d_wino4_supported(w.shape[1], w.shape[0], self.H, self.W):
                    wino = 4
                elif K.conv2d_wino_supported(w.shape[1], w.shape[0], self.H, self.W):
                    wino = 2
            if w.shape[2] == 1 and winograd and K.conv1x1_gemm_supported(w.shape[1], w.shape[0], self.H, self.W):
                wino = 1                             # 1x1 layer (conv7): the same MFMA GEMM kernel, one problem per image
            wp = (K.conv2d_wino4_pack_weight(w.contiguous()) if wino == 4 else
                  K.conv2d_wino_pack_weight(w.contiguous()) if wino == 2 else
                  K.conv1x1_gemm_pack_weight(w.contiguous()) if wino == 1 else K.conv2d_pack_weight(w.contiguous()))
            self.bev.append((wp, w.shape[0], w.shape[2], scale, shift, wino))
        self.wino4_ws = None
        # consecutive F(4x4) layers are chained: the map between two of them stays in the transform domain (fused output
        # -> input transform, sassd_conv2d_wino4_chain); `chain_bev=False` keeps three launches per layer (A/B)
        self.bev_cin = [int(sd["neck.fcn.conv%d.weight" % i].shape[1]) for i in range(8)]
        self.chain = [False] * 8                         # chain[i]: layer i reads the products layer i-1 left behind
        if any(l[5] == 4 for l in self.bev):
            self.cmax = max(max(self.bev_cin[i], self.bev[i][1]) for i in range(8) if self.bev[i][5] == 4)
            self.wino4_ws = K.conv2d_wino4_chain_workspace(self.B, self.cmax, self.H, self.W, dev)
            for i in range(1, 8):
                # (equal Cout on both sides of a boundary: the two layers then pad their tile count to one plane stride)
                self.chain[i] = bool(chain_bev and self.bev[i][5] == 4 and self.bev[i - 1][5] == 4 and i - 1 != 6 and
                                     self.bev[i - 1][1] == self.bev_cin[i] and self.bev[i - 1][1] == self.bev[i][1] and
                                     K.conv2d_wino4_chain_supported(self.bev_cin[i], self.bev[i][1], self.H, self.W))
        hw = torch.cat([sd["rpn_head.conv_box.weight"], sd["rpn_head.conv_cls.weight"],
                        sd["rpn_head.conv_dir_cls.weight"]], 0).float().contiguous()
        hb = torch.cat([sd["rpn_head.conv_box.bias"], sd["rpn_head.conv_cls.bias"],
                        sd["rpn_head.conv_dir_cls.bias"]], 0).float().contiguous()
        self.n_box = sd["rpn_head.conv_box.weight"].shape[0]
        self.n_cls = sd["rpn_head.conv_cls.weight"].shape[0]
        self.n_dir = sd["rpn_head.conv_dir_cls.weight"].shape[0]
        self.head_c = hw.shape[0]
        # 1x1 convs with <= 32 output channels (the single-class fused head, the second part-sensitive conv) stream on the
        # vector ALU (sassd_conv1x1_narrow_fwd); wider ones (three-class head: 60 maps) stay on the MFMA kernel
        self.head_narrow = K.conv1x1_narrow_supported(hw.shape[1], hw.shape[0])
        self.head_w = K.conv1x1_narrow_pack_weight(hw) if self.head_narrow else K.conv2d_pack_weight(hw)
        self.head_b = hb
        w0 = sd["extra_head.convs.0.weight"].float().contiguous()
        self.ps_parts = w0.shape[0]
        self.ps_w0 = K.conv2d_pack_weight(w0)
        # the part-sensitive 3x3 conv (256 -> 28) as a narrow TAIL of the Winograd chain on conv6's products (fused transform +
        # 64-channel GEMM block + output transform; conv6's own NCHW map is stored by the fused transform): when conv6 is a
        # chained F(4x4) layer with the default GEMM geometry.  OPT-IN (ps_tail=True): measured in round 6 at 205 us for conv6 +
        # tail against 143 + 77 us for conv6 + the direct fp32-MFMA conv -- 15 us per frame, 0.3 % of the frame rate, for 5 x the
        # direct kernel's error on the part-sensitive features (1.5e-5 vs 2.9e-6): not the default
        self.ps_tail = bool(ps_tail and chain_bev and self.bev[6][5] == 4 and self.chain[6] and self.ps_parts <= 64 and
                            w0.shape[2] == 3 and w0.shape[1] == self.bev[6][1] and not ((wino4_cfg or 0) & 0xff))
        self.ps_w0t = K.conv2d_wino4_pack_weight_narrow(w0) if self.ps_tail else None
        self.ps_s0, self.ps_b0 = fold_bn(sd, "extra_head.convs.1")
        w1 = sd["extra_head.convs.3.weight"].float().contiguous()
        self.ps_narrow = K.conv1x1_narrow_supported(w1.shape[1], w1.shape[0])
        self.ps_w1 = K.conv1x1_narrow_pack_weight(w1) if self.ps_narrow else K.conv2d_pack_weight(w1)

        # ---- anchors --------------------------------------------------------------------------------
        self.Atot = self.ncls * self.H * self.W * self.A
        if anchors is not None:
            an = np.asarray(anchors, np.float32).reshape(-1, 7)
            assert an.shape[0] == self.Atot, (an.shape, self.Atot)
            self.anchors = torch.from_numpy(np.ascontiguousarray(an)).to(dev)
            bv = anchors_bv if anchors_bv is not None else A.rbbox2d_to_near_bbox(an[:, [0, 1, 3, 4, 6]])
            self.anchors_bv = torch.from_numpy(np.ascontiguousarray(bv, np.float32)).to(dev)
        else:
            self.anchors = self.anchors_bv = None

        # ---- buffers --------------------------------------------------------------------------------
        B, H, W = self.B, self.H, self.W
        i32, f32 = torch.int32, torch.float32
        z = lambda *s, dt=f32: torch.zeros(*s, dtype=dt, device=dev)       # noqa: E731
        self.status = z(1, dt=i32)
        self.row_off = z(B + 1, dt=i32)
        self.vnum = z(B, dt=i32)
        self.n = [self.row_off[B:B + 1]] + [z(1, dt=i32) for _ in range(3)]       # device row counts per level
        self.idx = [z(c, 4, dt=i32) for c in self.caps]
        self.tables = [K.HashTable(c, dev) for c in self.caps]
        self.nbr = {key: z(self.caps[lvl], 27, dt=i32)
                    for key, lvl in (("subm0", 0), ("down0", 1), ("subm1", 1), ("down1", 2), ("subm2", 2),
                                     ("down2", 3), ("subm3", 3))}
        self.feat = [z(max(self.caps), 64), z(max(self.caps), 64)]
        self.mean = z(self.caps[0], 4)
        self.dense = z(B, 64 * D3, H, W)
        self.act = [z(B, 256, H, W) for _ in range(3)]
        self.head_out = z(B, self.head_c, H, W)
        self.ps_t = [z(B, self.ps_parts, H, W), z(B, self.ps_parts, H, W)]
        self.mask = z(B, self.Atot, dt=torch.uint8)
        self.df = dict(guided=z(B, self.capK, 7), labels=z(B, self.capK, dt=i32), scores=z(B, self.capK),
                       counts=z(B, dt=i32))
        self.logits = z(B, self.capK)
        self.det = dict(boxes=z(B, self.capD, 7), scores=z(B, self.capD), labels=z(B, self.capD, dt=i32),
                        counts=z(B, dt=i32))
        self.middle = {}
        # all seven rulebooks through the fused pyramid (1 fill + 8 launches); the per-op chain (30) stays selectable for A/B
        self.pyr = None
        if fused_rulebooks:
            self.pyr = K.RulebookPyramid(self.idx, self.n, self.caps, self.shape0, B,
                                         [self.nbr["subm%d" % l] for l in range(4)],
                                         [None] + [self.nbr["down%d" % l] for l in range(3)], self.status)
        # True: the whole pyramid as ONE persistent launch (in-launch grid barriers) instead of two launches per level
        self.pyramid_persistent = bool(pyramid_persistent)
        # per-call kernel-selection words of the C ABI (None: the binding's default, 0 in production)
        self.spconv_cfg, self.wino4_cfg = spconv_cfg, wino4_cfg
        self.graph = None
        self._wsid = id(self)
        # coordinate-only work (rulebooks, anchors_mask) runs on a side stream, overlapping the feature path
        self.overlap = bool(overlap)
        self.side = torch.cuda.Stream(device=dev) if self.overlap else None
        self.rb_ev = {k: torch.cuda.Event() for k in self.nbr}
        self.mask_ev = torch.cuda.Event()
        # BEV conv0 reads the densified sparse map: its Winograd launch runs on the tiles that have an occupied pixel in their
        # 6x6 patch only (56 % on a KITTI frame; bit-identical, the others' products are exactly zero).  The map is built from
        # the level-3 coordinates on the side stream.
        self.tile_map = None
        if skip_inactive_tiles and self.bev[0][5] == 4:
            n_ints = K._C.lib().sassd_wino4_tile_map_ints(B, H, W)
            if n_ints:
                self.tile_map = z(n_ints, dt=i32)
        self.tmap_ev = torch.cuda.Event()
        self.rb_sync_levels = tuple(sorted(int(l) for l in rb_sync_levels))   # levels whose completion the main stream waits
        assert self.rb_sync_levels and self.rb_sync_levels[-1] == 3           # for ((0, 1, 2, 3): one wait per level)
        self.prof = None           # set to {} to collect (name, start_event, end_event) tuples per frame

    def _ev(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()                 # current stream == the stream every sassd kernel is launched on
        return e

    def _seg(self, name, e0):
        if self.prof is not None:
            self.prof.setdefault(name, []).append((e0, self._ev()))

    # ------------------------------------------------------------------------------------------------
    def voxelize(self, clouds, n_dev=None):
        """clouds: list of B device tensors [Ni, ndim] f32.  Fills idx[0] (b,z,y,x), mean, row_off.  With `n_dev`
        (device int32 [B]) the clouds are capacity-sized staging buffers holding n_dev[b] points each."""
        assert len(clouds) == self.B
        e0 = self._ev() if self.prof is not None else None
        self.row_off.fill_(0)                   # (a fill KERNEL: torch's zero_() is a memset, i.e. a memset node in a captured frame)
        for b, pts in enumerate(clouds):
            K.voxelize(pts, self.voxel_size, self.pc_range, self.T, self.max_voxels, batch_idx=b, coors_cols=4,
                       want_voxels=False, want_mean=True, nfeat=4,
                       out=dict(coors=self.idx[0], mean=self.mean, voxel_num=self.vnum[b:b + 1],
                                num_points=self._numpts()),
                       row_offset=self.row_off[b:b + 2], status=self.status, cap=self.caps[0],
                       n_dev=None if n_dev is None else n_dev[b:b + 1])
        self._seg("voxelize", e0)

    def _numpts(self):
        if not hasattr(self, "_np"):
            self._np = torch.zeros(self.caps[0], dtype=torch.int32, device=self.dev)
        return self._np

    def load_voxels(self, voxel_feats, coors4):
        """Reference-style inputs (single_stage.py:110-117): mean features [M,4] and coords [M,4] (b,z,y,x)."""
        m = voxel_feats.shape[0]
        assert m <= self.caps[0]
        self.mean[:m].copy_(voxel_feats)
        self.idx[0][:m].copy_(coors4.int())
        self.row_off.fill_(0)
        bc = torch.bincount(coors4[:, 0].long(), minlength=self.B).cumsum(0).int()
        self.row_off[1:].copy_(bc)

    def rulebooks(self):
        """All 7 rulebooks + 3 next-level hash tables.  They depend only on voxel COORDINATES, so they are issued on
        a side HIP stream and overlap the feature path; `self.rb_ev[key]` fires when a rulebook is ready."""
        B = self.B
        if self.pyr is not None:
            if self.pyramid_persistent:
                self.pyr.build(persistent=True)
                for ev in self.rb_ev.values():
                    ev.record()
                return
            for lvl in range(4):
                self._rulebook_level(lvl)
            return
        self.tables[0].build(self.idx[0], self.n[0], self.shapes[0], B, self.status)
        for lvl in range(4):
            K.rulebook_subm(self.idx[lvl], self.n[lvl], self.caps[lvl], self.shapes[lvl], B, self.tables[lvl],
                            self.nbr["subm%d" % lvl])
            self.rb_ev["subm%d" % lvl].record()
            if lvl < 3:
                K.rulebook_conv(self.idx[lvl], self.n[lvl], self.caps[lvl], self.shapes[lvl], B, self.tables[lvl],
                                self.caps[lvl + 1], self.idx[lvl + 1], self.n[lvl + 1], self.nbr["down%d" % lvl],
                                self.status)
                self.rb_ev["down%d" % lvl].record()
                self.tables[lvl + 1].build(self.idx[lvl + 1], self.n[lvl + 1], self.shapes[lvl + 1], B, self.status)

    def _rulebook_level(self, lvl):
        """level `lvl` of the fused pyramid on the CURRENT stream: the level's submanifold table (`subm<lvl>`) and, for lvl > 0,
        the strided table into it (`down<lvl-1>`) + its coordinates; fires their events."""
        self.pyr.build(lvl, lvl + 1)                    # level 0: clears + hash; l >= 1: ordered emit of level l; tables
        self.rb_ev["subm%d" % lvl].record()
        if lvl > 0:
            self.rb_ev["down%d" % (lvl - 1)].record()

    def backbone(self, keep_middle=False, anchors_mask=None, densify=True, masks=True, rulebooks=True, convs=True):
        """7 rulebooks + 14 sparse convs (+ densify).  Two streams: coordinate-only work (rulebook pyramid, anchors_mask) on the
        side stream, features on the main stream, which waits for each rulebook event once.  (Round 5 measured the alternative
        issue order -- pyramid level l+1 issued behind the first conv of level l, so that a conv never waits for more of the
        pyramid than it needs: 703 against 748 frames/s and 0.365 against 0.347 ms for the segment's graph on one box,
        profiles/r05_pyramid_issue_order.txt.  The pyramid in front it is.)"""
        main = torch.cuda.current_stream(self.dev)
        e0 = self._ev() if self.prof is not None else None
        if self.overlap:
            self.side.wait_stream(main)                 # voxel coordinates are ready
            with torch.cuda.stream(self.side):
                if rulebooks:
                    self.rulebooks()
                if self.tile_map is not None and densify:
                    K.wino4_tile_map(self.idx[3], self.n[3], self.caps[3], self.B, self.H, self.W, out=self.tile_map)
                    self.tmap_ev.record()
                if masks:
                    self.anchor_masks(anchors_mask)     # also coordinate-only work
                self.mask_ev.record()
        else:
            if rulebooks:
                self.rulebooks()
            if self.tile_map is not None and densify:
                K.wino4_tile_map(self.idx[3], self.n[3], self.caps[3], self.B, self.H, self.W, out=self.tile_map)
        if not convs:                                   # (measurement: the rulebook pyramid alone)
            return
        x = self.mean
        lvl = 0
        cur = 0
        # cross-stream waits: the main stream joins the side stream at `rb_sync_levels` -- a wait on level L's event covers
        # every level <= L (stream order).  Round 6 measured one wait per level (the default) against two (after levels 1 and
        # 3) and one (after level 3): sparse segment 0.320 / 0.332 / 0.334 ms, frames/s equal -- fewer graph edges do not pay
        # for the later start of the first convs (profiles/r06_pyramid_forms.txt).
        covered = -1
        for li, (kind, cin, cout, key, wp, scale, shift) in enumerate(self.sp):
            y = self.feat[cur]
            if kind == "down":
                lvl += 1
            if key is not None and self.overlap and rulebooks and lvl > covered:
                upto = min([l for l in self.rb_sync_levels if l >= lvl] or [3])
                main.wait_event(self.rb_ev["subm%d" % upto])
                covered = upto
            if kind == "subm" or kind == "down":
                K.spconv_fwd(x, self.nbr[key], self.n[lvl], self.caps[lvl], wp, 27, cin, cout, scale, shift, True, y,
                             cfg=self.spconv_cfg)
            else:
                K.spconv_fwd(x, None, self.n[lvl], self.caps[lvl], wp, 1, cin, cout, scale, shift, True, y, cfg=self.spconv_cfg)
            if keep_middle:
                self.middle[li] = (y.clone(), lvl, cout)
            x = y
            cur ^= 1
        self.sp_out = x
        self._seg("sparse", e0)
        if not densify:
            return
        e1 = self._ev() if self.prof is not None else None
        K.densify(x, self.idx[3], self.n[3], self.caps[3], self.shapes[3], self.B, 1, self.dense)
        self._seg("densify", e1)

    def bev_and_heads(self):
        x = self.dense
        for i, (wp, cout, ks, scale, shift, wino) in enumerate(self.bev):
            y = self.act[i % 2] if i < 7 else self.act[2]
            e0 = self._ev() if self.prof is not None else None
            if wino == 4:
                # layer i hands its products to layer i+1 (no NCHW map in between) unless its output is needed: conv6
                # feeds the part-sensitive head, the last F(4x4) layer feeds a 1x1 / direct layer
                keep = (i + 1 < 8 and self.chain[i + 1]) or (i == 6 and self.ps_tail)
                prev = self.bev[i - 1][3:5] + (True,) if self.chain[i] else None
                tmap = self.tile_map if i == 0 else None                     # conv0: the active tiles of the sparse map only
                ptmap = self.tile_map if (i == 1 and self.chain[1]) else None   # conv1 reads conv0's compacted products
                if tmap is not None and self.overlap:
                    torch.cuda.current_stream(self.dev).wait_event(self.tmap_ev)
                K.conv2d_wino4_chain(None if self.chain[i] else x, prev, wp, self.bev_cin[i], cout, self.cmax, self.B,
                                     self.H, self.W, scale, shift, True, None if keep else y, self.wino4_ws, cfg=self.wino4_cfg,
                                     tile_map=tmap, prev_tile_map=ptmap)
            elif wino == 2:
                K.conv2d_wino_fwd(x, wp, cout, scale, shift, True, y)
            elif wino == 1:
                K.conv1x1_gemm_fwd(x, wp, cout, scale, shift, True, y, cfg=self.wino4_cfg)
            else:
                K.conv2d_fwd(x, wp, cout, ks, scale, shift, True, y)
            if i == 6 and self.ps_tail:
                # conv6's products stay in the workspace: the tail call stores conv6's activation map (for conv7) and runs the
                # part-sensitive 3x3 conv on them
                K.conv2d_wino4_chain_tail(self.bev[6][3:5] + (True,), y, self.ps_w0t, cout, self.ps_parts, self.cmax, self.B,
                                          self.H, self.W, self.ps_s0, self.ps_b0, True, self.ps_t[0], self.wino4_ws,
                                          cfg=self.wino4_cfg)
            self._seg("bev_conv%d" % i, e0)
            x = y
            if i == 6:
                self.conv6 = y
        self.x = x
        if self.head_narrow:
            K.conv1x1_narrow_fwd(x, self.head_w, self.head_c, None, self.head_b, False, self.head_out)
        else:
            K.conv2d_fwd(x, self.head_w, self.head_c, 1, None, self.head_b, False, self.head_out)
        if not self.ps_tail:
            K.conv2d_fwd(self.conv6, self.ps_w0, self.ps_parts, 3, self.ps_s0, self.ps_b0, True, self.ps_t[0])
        if self.ps_narrow:
            K.conv1x1_narrow_fwd(self.ps_t[0], self.ps_w1, self.ps_parts, None, None, False, self.ps_t[1])
        else:
            K.conv2d_fwd(self.ps_t[0], self.ps_w1, self.ps_parts, 1, None, None, False, self.ps_t[1])

    def anchor_masks(self, masks=None):
        if masks is not None:
            self.mask.copy_(masks.view(self.B, -1).to(torch.uint8))
            return
        H0, W0 = self.shape0[1], self.shape0[2]
        K.anchor_mask_batch(self.idx[0], self.row_off, self.B, H0, W0, self.anchors_bv, self.voxel_size, self.pc_range,
                            self.area_thr, self.mask)

    def post(self):
        e0 = self._ev() if self.prof is not None else None
        HW = self.H * self.W
        ho = self.head_out
        base = ho.view(-1)
        box = base
        cls = base[self.n_box * HW:]
        dirp = base[(self.n_box + self.n_cls) * HW:]
        K.decode_filter(box, cls, dirp, self.head_c * HW, self.B, self.ncls, self.A, self.H, self.W, self.anchors,
                        self.mask, self.rpn_thr, self.capK, self.df, self.status)
        K.pswarp_sample(self.ps_t[1], self.df["guided"], self.df["counts"], self.capK, self.grid_offsets,
                        self.spatial_scale, self.logits)
        K.rescore_nms(self.df["guided"], self.logits, self.df["labels"], self.df["counts"], self.score_thr,
                      self.iou_thr, self.capD, self.det, self.status)
        self._seg("post", e0)

    def sparse_work(self):
        """Algorithmic work of the sparse path for the CURRENT frame (SURVEY.md 8d formulas): per layer
        B_gs = 4P(Cin+Cout) + 8P + 4K*Cin*Cout + 4*Nout*Cout, B_min, flops = 2*P*Cin*Cout; rulebook bytes."""
        n = [int(t.item()) for t in self.n]
        pairs = {}
        lvl_of = dict(subm0=0, down0=1, subm1=1, down1=2, subm2=2, down2=3, subm3=3)
        for key, t in self.nbr.items():
            pairs[key] = int((t[:n[lvl_of[key]]] >= 0).sum().item())
        bgs = bmin = flops = rb = 0
        lvl = 0
        seen = set()
        for kind, cin, cout, key, *_ in self.sp:
            if kind == "down":
                nin, lvl = n[lvl], lvl + 1
            else:
                nin = n[lvl]
            nout = n[lvl]
            p = pairs[key] if key else nout
            k = 27 if key else 1
            bgs += 4 * p * (cin + cout) + 8 * p + 4 * k * cin * cout + 4 * nout * cout
            bmin += 4 * nin * cin + 4 * nout * cout + 4 * k * cin * cout + 8 * p
            flops += 2 * p * cin * cout
            if key and key not in seen:
                seen.add(key)
                rb += (16 * nin + 8 * p) if kind == "subm" else (16 * nin + 16 * nout + 8 * p)
        return dict(n=n, pairs=pairs, bytes_gs=bgs, bytes_min=bmin, flops=flops, rulebook_bytes=rb)

    # ------------------------------------------------------------------------------------------------
    def _tail(self, anchors_mask):
        self.bev_and_heads()
        if self.overlap:
            torch.cuda.current_stream(self.dev).wait_event(self.mask_ev)
        else:
            self.anchor_masks(anchors_mask)
        self.post()
        return self.det

    def run_from_points(self, clouds, anchors_mask=None):
        """One frame batch, raw device point clouds in -> device detection buffers out (no host sync)."""
        with K.ws_scope(self._wsid):
            self.voxelize(clouds)
            self.backbone(anchors_mask=anchors_mask)
            return self._tail(anchors_mask)

    def run_from_voxels(self, voxel_feats, coors4, anchors_mask=None):
        with K.ws_scope(self._wsid):
            self.load_voxels(voxel_feats, coors4)
            self.backbone(anchors_mask=anchors_mask)
            return self._tail(anchors_mask)

    # ---- hipGraph: the whole frame as ONE launch ---------------------------------------------------------
    def capture(self, points_cap, ndim=4, stages=("voxelize", "backbone", "tail")):
        """Capture the frame (raw points -> detections, side stream included) into a hipGraph.  Input staging:
        `self.pts_in[b]` [points_cap, ndim] f32 and `self.npts` int32 [B]; `run_graph(clouds)` fills them and replays.
        `stages` lets a caller capture a sub-range (e.g. only the sparse backbone for a roofline measurement)."""
        dev = self.dev
        self.pts_cap = int(points_cap)
        self.pts_in = [torch.zeros(self.pts_cap, ndim, dtype=torch.float32, device=dev) for _ in range(self.B)]
        self.npts = torch.zeros(self.B, dtype=torch.int32, device=dev)
        assert self.prof is None, "per-stage event timing and graph capture exclude each other"

        def frame():
            with K.ws_scope(self._wsid):
                if "voxelize" in stages:
                    self.voxelize(self.pts_in, n_dev=self.npts)
                if "backbone" in stages:
                    self.backbone()
                elif "sparse" in stages:              # rulebooks + the 14 sparse convs only (roofline measurement)
                    self.backbone(densify=False, masks=False)
                elif "sparse_convs" in stages:        # the 14 sparse convs on the rulebooks of the previous pass (floor model)
                    self.backbone(densify=False, masks=False, rulebooks=False)
                elif "pyramid" in stages:             # the rulebook pyramid alone
                    self.backbone(densify=False, masks=False, convs=False)
                if "tail" in stages:
                    self._tail(None)
                elif self.overlap:
                    torch.cuda.current_stream(self.dev).wait_event(self.mask_ev)      # join the side stream

        # capture needs a non-default stream (the legacy null stream cannot be captured)
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(cap):
            frame()                               # warm-up: every workspace / lazily created event exists
            cap.synchronize()
            self.graph = K.Graph().capture(frame)
        # the graph holds raw pointers into this plan's grow-only kernel workspaces: keep the tensors it was captured
        # with alive, so that a later eager call that regrows a workspace cannot free memory the graph still replays on
        self._graph_ws = K.scoped_workspaces(self._wsid)
        torch.cuda.current_stream(dev).wait_stream(cap)
        return self.graph

    def stage_inputs(self, clouds):
        for b, pts in enumerate(clouds or ()):
            n = pts.shape[0]
            if n > self.pts_cap:            # never detect on a silently truncated cloud
                raise ValueError("cloud %d has %d points, the captured graph was sized for %d (plan.capture(points_cap))"
                                 % (b, n, self.pts_cap))
            self.pts_in[b][:n].copy_(pts, non_blocking=True)
            self.npts[b:b + 1].fill_(n)

    def run_graph(self, clouds=None):
        """Replay the captured frame (after staging `clouds`, unless the caller filled pts_in / npts itself)."""
        if clouds is not None:
            self.stage_inputs(clouds)
        self.graph.launch()
        return self.det

    def results(self):
        """The only host sync of a frame: D2H of the (small) detection buffers, like
        ssd_rotate_head.py:529-531.  Returns per-sample (boxes[k,7], scores[k], labels[k]) numpy or None."""
        counts = self.det["counts"].cpu().numpy()
        st = int(self.status.item())
        if st:
            raise RuntimeError("sassd pipeline status flags 0x%x (capacity overflow / hash full)" % st)
        out = []
        for b in range(self.B):
            k = int(counts[b])
            if k == 0:
                out.append((None, None, None))
                continue
            out.append((self.det["boxes"][b, :k].cpu().numpy(), self.det["scores"][b, :k].cpu().numpy(),
                        self.det["labels"][b, :k].cpu().numpy().astype(np.int64)))
        return out
