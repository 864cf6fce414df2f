"""torch.autograd glue for the training path: every forward and every data / weight gradient below
runs in a hand-written HIP kernel of libsassd (BatchNorm / ReLU / Linear between them are torch ops, as in the
reference).  There is no CPU fallback."""
import os

import torch
from torch.autograd import Function

from . import kernels as K


_BEV_PRECISION = "fp32"


def set_bev_precision(precision):
    """Arithmetic of the dense BEV / head convolutions in TRAINING (BASELINE configs[2] trains in bf16): "fp32" keeps the
    fp32-MFMA kernels (bit-comparable with the fp32 oracle), "bf16" rounds the MFMA operands to bf16 (fp32 accumulation,
    fp32 master weights, fp32 activations between layers) -- what torch autocast gives the reference's cuDNN convs."""
    global _BEV_PRECISION
    if precision not in ("fp32", "bf16"):
        raise ValueError("precision must be 'fp32' or 'bf16', got %r" % (precision,))
    _BEV_PRECISION = precision


def bev_precision():
    return _BEV_PRECISION


_n_ptr_cache = {}


def _n_ptr(n, dev):
    """[1] int32 device scalar holding n.  The same row counts recur through the layers of one step (and across steps
    for fixed-size inputs): cached per (n, device), filled by a kernel (no blocking H2D copy), never written again."""
    key = (int(n), dev)
    t = _n_ptr_cache.get(key)
    if t is None:
        if len(_n_ptr_cache) > 512:
            _n_ptr_cache.clear()
        t = torch.full((1,), int(n), dtype=torch.int32, device=dev)
        _n_ptr_cache[key] = t
    return t


_ident_nbr_cache = {}


def _ident_nbr(n, dev):
    """[n, 27] gather table of a 1x1x1 convolution seen as a 27-offset one: the centre offset (13) maps every row to
    itself, all other offsets are empty.  Lets the 1x1x1 layer's weight gradient run on sassd_spconv_bwd_weight instead of
    a library GEMM.  Grow-only cache per device (the rows are independent of the data)."""
    t = _ident_nbr_cache.get(dev)
    if t is None or t.shape[0] < n:
        cap = max(int(n * 1.25), 1024)
        t = torch.full((cap, 27), -1, dtype=torch.int32, device=dev)
        t[:, 13] = torch.arange(cap, dtype=torch.int32, device=dev)
        _ident_nbr_cache[dev] = t
    return t[:n]


class _PackCache(dict):
    """Packed-weight images keyed by the source weight's (address, shape, ...) and validated by its generation
    (kernels.weight_key: version counter, address, device, global generation).  A FRESH tensor can land on the address of a
    dead one with the same shape, version 0 and the same global generation -- its key and generation then equal the dead
    tensor's and the lookup would return the dead tensor's image (round 5: the sparse data gradient of the third 64 -> 64 layer
    a test created was computed with the first one's transposed weights; models built through build_detector bump the
    generation, hand-made layers and tests do not).  So an entry PINS the tensor it was packed from: while the entry exists
    that storage cannot be freed, hence no other tensor can sit on its address.  Bounded, LEAST RECENTLY USED first (ADVICE r05:
    insertion order dropped the long-lived parameter packs a PackPlan installs first once 256 transient keys had piled up):
    every hit and every put moves the key to the young end; a dropped entry only costs a re-pack."""
    MAX = 256

    def get(self, key, default=None):
        hit = dict.get(self, key, default)
        if hit is not default and key in self:
            dict.__delitem__(self, key)                     # re-insert: dicts keep insertion order, the front is the oldest
            dict.__setitem__(self, key, hit)
        return hit

    def put(self, key, gen, pack, source):
        if key in self:
            dict.__delitem__(self, key)
        elif len(self) >= self.MAX:
            for k in list(self)[:self.MAX // 4]:
                dict.__delitem__(self, k)
        self[key] = (gen, pack, source)


_sp_t_packs = _PackCache()


def _spconv_t_pack(weight, reverse=False):
    """Transposed packed image of a sparse-conv weight [K, Cin, Cout] for the data gradient, cached per parameter
    storage and weight generation (sassd.train.PackPlan refreshes the entry right after the optimizer step).
    reverse: the image of offset k holds W[K-1-k]^T (submanifold layers, see SparseConvFn.backward)."""
    key = (weight.data_ptr(), tuple(weight.shape)) + ((True,) if reverse else ())
    gen = K.weight_key(weight)
    hit = _sp_t_packs.get(key)
    if hit is not None and hit[0] == gen:
        return hit[1]
    w = weight.detach()
    pack = K.spconv_pack_weight_t((w.flip(0) if reverse else w).contiguous())
    # the entry pins the STORAGE (so that no other tensor can land on the address), not the autograd graph: the weight handed
    # in is a view of the module parameter (grad_fn = a view node) -- pinning it would keep that graph alive
    _sp_t_packs.put(key, gen, pack, w if weight.grad_fn is not None else weight)
    return pack


class SparseConvFn(Function):
    """y = sum_k x[nbr[:, k]] @ w[k]  (raw conv; BatchNorm / ReLU stay separate modules in training mode)."""

    # submanifold layers take their data gradient on the FORWARD rulebook (no transposed table, no memset): for a
    # submanifold table nbr[j][k] = i  <=>  nbr[i][26-k] = j, hence dx[i] = sum_k dy[nbr[i][k]] . W[26-k]^T -- the forward
    # kernel with the offset-reversed transposed weight image (identity tested on the oracle, tests/test_oracle_cpu.py).
    # The identity needs UNIQUE voxel coordinates (two rows on one cell make the hash last-writer-wins and the table
    # asymmetric); the voxelizer and the strided-conv emit produce unique coordinates by construction, and spconv itself
    # leaves duplicate indices undefined -- a caller feeding hand-made indices with duplicates through the facade must set
    # SASSD_SUBM_FWD_TABLE=0 (the transposed table is exact for any table).
    # False: the transposed-table formulation for every layer (A/B, tests).
    subm_on_forward_table = os.environ.get("SASSD_SUBM_FWD_TABLE", "1") != "0"

    @staticmethod
    def forward(ctx, x, weight, nbr, n_out, packed, subm=False):
        # x [Nin, Cin]; weight [K, Cin, Cout] view of the module parameter; nbr [Nout, 27] or None (1x1x1)
        k, cin, cout = weight.shape
        x = x.contiguous()
        dev = x.device
        y = K.spconv_fwd(x, nbr, _n_ptr(n_out, dev), max(n_out, 1), packed, k, cin, cout)
        ctx.save_for_backward(x, weight)
        ctx.nbr, ctx.n_out = nbr, n_out
        ctx.subm = bool(subm) and nbr is not None and k == 27 and x.shape[0] == n_out
        return y[:n_out]

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        k, cin, cout = weight.shape
        nbr, n_out = ctx.nbr, ctx.n_out
        n_in = x.shape[0]
        dev = x.device
        if n_out > 0:
            dyc = dy.contiguous()
        else:
            dyc = torch.zeros(1, cout, dtype=torch.float32, device=dev)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if cin < 16:
                raise NotImplementedError("sparse data gradient needs Cin >= 16 (the 4-channel input layer has none)")
            on_fwd = ctx.subm and SparseConvFn.subm_on_forward_table and n_in > 0
            wt = _spconv_t_pack(weight, reverse=on_fwd)
            if nbr is None:
                dx = K.spconv_bwd_data(dyc, None, _n_ptr(n_in, dev), max(n_in, 1), wt, 1, cin, cout)[:n_in]
            elif on_fwd:
                dx = K.spconv_bwd_data(dyc, nbr, _n_ptr(n_in, dev), nbr.shape[0], wt, 27, cin, cout)[:n_in]
            else:
                # the transposed table depends on the rulebook alone: layers sharing an `indice_key` (conv2.0-2, ...)
                # build it once per batch (kept on the rulebook tensor, which lives as long as the batch)
                cached = getattr(nbr, "_sassd_transposed", None)
                if cached is not None and cached[0] == (n_out, n_in):
                    nbr_t = cached[1]
                else:
                    nbr_t = K.rulebook_transpose(nbr, _n_ptr(n_out, dev), nbr.shape[0], max(n_in, 1))
                    nbr._sassd_transposed = ((n_out, n_in), nbr_t)
                dx = K.spconv_bwd_data(dyc, nbr_t, _n_ptr(n_in, dev), max(n_in, 1), wt, 27, cin, cout)[:n_in]
        if ctx.needs_input_grad[1]:
            if nbr is None and n_in > 0 and (cin, cout) in _SP_PAIRS:
                # 1x1x1 layer: X^T dY through the sparse weight-gradient kernel over an identity table (centre offset only)
                dw = K.spconv_bwd_weight(x, dyc, _ident_nbr(n_in, dev), _n_ptr(n_in, dev), n_in, cin, cout)[13:14]
            elif nbr is None:
                dw = (x.t() @ dyc[:n_in]).view(1, cin, cout)
            else:
                xc = x if n_in > 0 else torch.zeros(1, cin, device=dev)
                dw = K.spconv_bwd_weight(xc, dyc, nbr, _n_ptr(n_out, dev), nbr.shape[0], cin, cout)
        return dx, dw, None, None, None, None


def _conv_any(x, weight, ks, packed=None, wino=None, shift=None, wino4=None):
    """3x3 layers take Winograd F(4x4,3x3) (transforms + 36 MFMA GEMMs) when the shape allows, else the fused F(2x2,3x3)
    kernel, everything else the direct kernel.  Pre-packed weights may be handed in (`wino4` / `wino` / `packed`)."""
    cout, cin = weight.shape[0], weight.shape[1]
    h, w = x.shape[2], x.shape[3]
    if ks == 3 and K.conv2d_wino4_supported(cin, cout, h, w):
        return K.conv2d_wino4_fwd(x, wino4 if wino4 is not None else K.conv2d_wino4_pack_weight(weight), cout, None, shift)
    if ks == 3 and K.conv2d_wino_supported(cin, cout, h, w):
        return K.conv2d_wino_fwd(x, wino if wino is not None else K.conv2d_wino_pack_weight(weight), cout, None, shift)
    return K.conv2d_fwd(x, packed if packed is not None else K.conv2d_pack_weight(weight), cout, ks, None, shift)


# channel pairs the sparse kernels are instantiated for (csrc/spconv.hip SP_DISPATCH); anything else takes the GEMM
_SP_PAIRS = {(4, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (32, 16), (64, 32)}

_dgrad_packs = _PackCache()
_dgrad_direct = {}


def _cacheable(weight):
    """Packed images are cached per parameter STORAGE and weight generation.  A derived tensor (the fused RPN head's
    `torch.cat` of three conv weights: fresh storage every step, `_version` always 0, and the caching allocator hands the
    same address back) has no generation of its own -- after load_state_dict / a torch.optim step / an in-place edit of
    its sources its cached image would be silently stale: such weights are packed on every call."""
    return weight.is_leaf and weight.grad_fn is None


def _dgrad_pack(weight, h, w):
    """Packed image of the data-gradient conv's weights (transposed, taps mirrored), cached per parameter storage and
    weight generation: the pack is a pure function of the weights, which change once per optimizer step."""
    ks = weight.shape[2]
    gen = K.weight_key(weight)
    cache = _cacheable(weight)
    hit = _dgrad_direct.get((weight.data_ptr(), tuple(weight.shape))) if cache else None   # installed by sassd.train.PackPlan
    if hit is not None and hit[0] == gen:
        return hit[1]
    key = (weight.data_ptr(), tuple(weight.shape), h, w)
    hit = _dgrad_packs.get(key) if cache else None
    if hit is not None and hit[0] == gen:
        return hit[1]
    wt = weight.detach().transpose(0, 1).flip(2, 3).contiguous()          # [Cin, Cout, k, k]
    cin_g, cout_g = wt.shape[1], wt.shape[0]
    if ks == 3 and K.conv2d_wino4_supported(cin_g, cout_g, h, w):
        pack = dict(wino4=K.conv2d_wino4_pack_weight(wt))
    elif ks == 3 and K.conv2d_wino_supported(cin_g, cout_g, h, w):
        pack = dict(wino=K.conv2d_wino_pack_weight(wt))
    else:
        pack = dict(packed=K.conv2d_pack_weight(wt))
    pack["wt"] = wt
    if cache:
        _dgrad_packs.put(key, gen, pack, weight)
    return pack


_bf16_packs = _PackCache()


def bf16_cout_pad(cout):
    """A 3x3 layer whose Cout is not a multiple of 32 (the part-sensitive head's 256 -> 28, ssd_rotate_head.py:424-429) runs on
    the bf16 kernel with zero weight rows up to the next multiple; the caller keeps the first Cout output maps (round 6: 53 us
    + a 5 us slice against 147 us on the fp32-MFMA kernel at batch 2)."""
    return (cout + 31) // 32 * 32


def _bf16_pack(weight, transposed):
    """bf16 [tap][Cin/8][Cout][8] image of the weights (forward; Cout zero-padded to a multiple of 32) or of their transposed,
    tap-mirrored form (data gradient), cached per parameter storage and weight generation like _dgrad_pack."""
    key = (weight.data_ptr(), tuple(weight.shape), transposed)
    gen = K.weight_key(weight)
    cache = _cacheable(weight)
    hit = _bf16_packs.get(key) if cache else None
    if hit is not None and hit[0] == gen:
        return hit[1]
    w = weight.detach()
    if transposed:
        w = w.transpose(0, 1).flip(2, 3)
    elif w.shape[0] % 32:
        w = torch.cat([w, w.new_zeros((bf16_cout_pad(w.shape[0]) - w.shape[0],) + tuple(w.shape[1:]))], 0)
    pack = K.conv2d_bf16_pack_weight(w.contiguous())
    if cache:
        _bf16_packs.put(key, gen, pack, weight)
    return pack


_bf16_1x1_packs = _PackCache()


def _bf16_pack_1x1(weight, transposed):
    """bf16 MFMA-fragment image of a 1x1 conv's weights (forward) or of their transpose (data gradient), cached like
    _bf16_pack (sassd.train.PackPlan refreshes the entries of the module parameters after the optimizer step)."""
    key = (weight.data_ptr(), tuple(weight.shape), transposed)
    gen = K.weight_key(weight)
    cache = _cacheable(weight)
    hit = _bf16_1x1_packs.get(key) if cache else None
    if hit is not None and hit[0] == gen:
        return hit[1]
    pack = K.conv1x1_bf16_pack_weight(weight.detach(), transposed)
    if cache:
        _bf16_1x1_packs.put(key, gen, pack, weight)
    return pack


class Conv2dFn(Function):
    """NCHW fp32 conv (3x3 pad 1 / 1x1) + optional bias on the fp32-MFMA kernels; data gradient = the same kernels
    with flipped, transposed weights; weight gradient = the split-K MFMA kernel.  Under set_bev_precision("bf16") the
    3x3 forward / data gradient (shapes sassd_conv2d_bf16_supported), the 1x1 forward / data gradient (round 6: shapes
    sassd_conv1x1_bf16_supported -- 150 us -> 33 us for BEVNet's conv7 at batch 2) and every weight gradient run on the
    bf16 MFMA kernels instead (fp32 accumulation, fp32 tensors in and out)."""

    @staticmethod
    def forward(ctx, x, weight, bias, packed, wino, wino4=None):
        x = x.contiguous()
        ks = weight.shape[2]
        cout, cin = weight.shape[0], weight.shape[1]
        shift = bias.detach().contiguous() if bias is not None else None
        ctx.bf16 = _BEV_PRECISION == "bf16"
        if ctx.bf16 and ks == 3 and K.conv2d_bf16_supported(cin, bf16_cout_pad(cout), x.shape[2], x.shape[3]):
            cp = bf16_cout_pad(cout)
            if cp != cout and shift is not None:
                shift = torch.cat([shift, shift.new_zeros(cp - cout)])
            y = K.conv2d_bf16_fwd(x, _bf16_pack(weight, False), cp, shift)
            if cp != cout:
                y = y[:, :cout].contiguous()
        elif ctx.bf16 and ks == 1 and K.conv1x1_bf16_supported(cin, cout, x.shape[2] * x.shape[3]):
            y = K.conv1x1_bf16_fwd(x, _bf16_pack_1x1(weight, False), cout, shift)
        else:
            y = _conv_any(x, weight.detach(), ks, packed, wino, shift, wino4)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        ks = weight.shape[2]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            cout, cin = weight.shape[0], weight.shape[1]
            if ctx.bf16 and ks == 3 and K.conv2d_bf16_supported(cout, cin, dy.shape[2], dy.shape[3]):
                dx = K.conv2d_bf16_fwd(dy, _bf16_pack(weight, True), cin)
            elif ctx.bf16 and ks == 1 and K.conv1x1_bf16_supported(cout, cin, dy.shape[2] * dy.shape[3]):
                dx = K.conv1x1_bf16_fwd(dy, _bf16_pack_1x1(weight, True), cin)
            else:
                pk = _dgrad_pack(weight, dy.shape[2], dy.shape[3])                  # [Cin, Cout, k, k], taps mirrored
                dx = _conv_any(dy, pk["wt"], ks, pk.get("packed"), pk.get("wino"), None, pk.get("wino4"))
        if ctx.needs_input_grad[1]:
            # (Round 5 measured this launch on a side HIP stream -- beside the data gradient, or fenced so that it only
            # overlaps the next layer's BatchNorm backward: 247 / 241 against 257.6 samples/s for the one-stream step, the
            # concurrent kernels slow each other by more than the overlap hides; profiles/r05_wgrad_side_stream.txt.)
            dw = K.conv2d_bwd_weight(x, dy, ks, bf16=ctx.bf16)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum((0, 2, 3))
        return dx, dw, db, None, None, None


class DensifyFn(Function):
    """SparseConvTensor.dense(): [N,C] rows -> [B, C, D, H, W]; backward gathers the rows back."""

    @staticmethod
    def forward(ctx, feats, indices, shape, batch_size):
        d, h, w = shape
        n, c = feats.shape
        out = K.densify(feats.contiguous(), indices, _n_ptr(n, feats.device), max(n, 1), shape, batch_size, 0)
        ctx.indices = indices
        return out.view(batch_size, c, d, h, w)

    @staticmethod
    def backward(ctx, g):
        i = ctx.indices.long()
        return g[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]].contiguous(), None, None, None


class PSWarpFn(Function):
    """Part-sensitive bilinear sampling of one sample: feat [1,28,H,W], boxes [K,7] -> logits [K]."""

    @staticmethod
    def forward(ctx, feat, boxes, grid_offsets, spatial_scale):
        k = boxes.shape[0]
        feat = feat.contiguous()
        boxes = boxes.contiguous()
        cnt = _n_ptr(k, feat.device)
        lg = K.pswarp_sample(feat, boxes.view(1, k, 7), cnt, k, grid_offsets, spatial_scale)
        ctx.save_for_backward(feat, boxes)
        ctx.args = (grid_offsets, spatial_scale)
        return lg.view(-1)

    @staticmethod
    def backward(ctx, dlog):
        feat, boxes = ctx.saved_tensors
        k = boxes.shape[0]
        cnt = _n_ptr(k, feat.device)
        dfeat, dg = K.pswarp_sample_bwd(feat, boxes.view(1, k, 7), cnt, k, ctx.args[0], ctx.args[1],
                                        dlog.contiguous().view(1, k))
        return dfeat, dg.view(k, 7), None, None


class RpnLossFn(Function):
    """(loc, cls, dir) loss sums of the RPN head in one kernel (sassd_rpn_loss); the gradients with respect to the three
    prediction tensors come out of the same pass, backward scales them by the upstream gradient."""

    @staticmethod
    def forward(ctx, box_preds, cls_preds, dir_preds, labels, targets, anchors, num_pos):
        sums, gbox, gcls, gdir = K.rpn_loss(box_preds.contiguous(), cls_preds.contiguous(),
                                            dir_preds.contiguous() if dir_preds is not None else None, labels, targets,
                                            anchors, num_pos)
        ctx.save_for_backward(gbox, gcls, gdir if gdir is not None else gbox.new_zeros(0))
        ctx.has_dir = gdir is not None
        return sums

    @staticmethod
    def backward(ctx, g):
        gbox, gcls, gdir = ctx.saved_tensors
        return (gbox * g[0], gcls * g[1], gdir * g[2] if ctx.has_dir else None, None, None, None, None)


class PSWarpBatchFn(Function):
    """Part-sensitive sampling of a whole batch on padded boxes: feat [B,28,H,W], boxes [B,capK,7], counts [B] int32
    (device) -> logits [B,capK], zero for the rows past a sample's count (their gradients are zero too)."""

    @staticmethod
    def forward(ctx, feat, boxes, counts, grid_offsets, spatial_scale):
        feat, boxes = feat.contiguous(), boxes.contiguous()
        lg = K.pswarp_sample(feat, boxes, counts, boxes.shape[1], grid_offsets, spatial_scale)
        ctx.save_for_backward(feat, boxes, counts)
        ctx.args = (grid_offsets, spatial_scale)
        return lg

    @staticmethod
    def backward(ctx, dlog):
        feat, boxes, counts = ctx.saved_tensors
        dfeat, dg = K.pswarp_sample_bwd(feat, boxes, counts, boxes.shape[1], ctx.args[0], ctx.args[1], dlog.contiguous())
        return dfeat, dg, None, None, None


class BnReluFn(Function):
    """Training-mode BatchNorm1d + ReLU over sparse features [N, C] in two launches each way (sassd_bn_relu_fwd / _bwd);
    the running statistics of the module are updated in place like torch.nn.functional.batch_norm(training=True)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps):
        x = x.contiguous()
        y, mean, invstd = K.bn_relu_fwd(x, gamma.detach().contiguous(), beta.detach().contiguous(), running_mean,
                                        running_var, momentum, eps)
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        dx, dg, db = K.bn_relu_bwd(x, dy.contiguous(), gamma.detach().contiguous(), beta.detach().contiguous(), mean,
                                   invstd)
        return dx, dg, db, None, None, None, None


class BnRelu2dFn(Function):
    """Training-mode BatchNorm2d + ReLU over NCHW maps in two launches each way (sassd_bn2d_relu_fwd / _bwd); the
    module's running statistics are updated in place like torch.nn.functional.batch_norm(training=True)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps):
        x = x.contiguous()
        y, mean, invstd = K.bn2d_relu_fwd(x, gamma.detach().contiguous(), beta.detach().contiguous(), running_mean,
                                          running_var, momentum, eps)
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        dx, dg, db = K.bn2d_relu_bwd(x, dy.contiguous(), gamma.detach().contiguous(), beta.detach().contiguous(), mean,
                                     invstd)
        return dx, dg, db, None, None, None, None


class BnReluConvBf16Fn(Function):
    """relu(batchnorm2d(x)) -> 3x3 conv on the bf16 MFMA as ONE unit (round 6; cmn.py:236-237 + the next layer's conv): the
    batch statistics come from sassd_bn2d_stats, the normalisation + ReLU is applied by the LOADER WAVES of the convolution (and
    of its weight gradient) to the raw map x, with the expression of the stand-alone apply kernel -- so forward, data gradient and
    weight gradient are bit-identical to BnRelu2dFn followed by Conv2dFn, and the normalised map is never written to HBM
    (72 MB written + read back per 256-channel layer at batch 2: the bn2d_apply_kernel pass).  Backward: weight gradient with the
    same fused operand, data gradient as before, then sassd_bn2d_relu_bwd on the raw map."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, weight, bias):
        x = x.contiguous()
        cout = weight.shape[0]
        mean, invstd, aff = K.bn2d_stats(x, gamma.detach().contiguous(), beta.detach().contiguous(), running_mean, running_var,
                                         momentum, eps)
        shift = bias.detach().contiguous() if bias is not None else None
        y = K.conv2d_bf16_fwd(x, _bf16_pack(weight, False), cout, shift, in_affine=aff)
        ctx.save_for_backward(x, gamma, beta, mean, invstd, aff, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, invstd, aff, weight = ctx.saved_tensors
        dy = dy.contiguous()
        cout, cin = weight.shape[0], weight.shape[1]
        dw = K.conv2d_bwd_weight(x, dy, 3, bf16=True, x_affine=aff) if ctx.needs_input_grad[7] else None
        db = dy.sum((0, 2, 3)) if (ctx.has_bias and ctx.needs_input_grad[8]) else None
        dx = dg = dbt = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            if K.conv2d_bf16_supported(cout, cin, dy.shape[2], dy.shape[3]):
                da = K.conv2d_bf16_fwd(dy, _bf16_pack(weight, True), cin)
            else:
                pk = _dgrad_pack(weight, dy.shape[2], dy.shape[3])
                da = _conv_any(dy, pk["wt"], 3, pk.get("packed"), pk.get("wino"), None, pk.get("wino4"))
            dx, dg, dbt = K.bn2d_relu_bwd(x, da, gamma.detach().contiguous(), beta.detach().contiguous(), mean, invstd)
        return dx, dg, dbt, None, None, None, None, dw, db


def bn_relu_conv_fusable(bn, conv, x):
    """Can relu(bn(x)) -> conv run as BnReluConvBf16Fn?  bf16 BEV precision, training-mode affine BatchNorm with running
    statistics, a 3x3 / stride 1 / pad 1 convolution of a shape the bf16 kernels take (W even for its weight gradient)."""
    return (_BEV_PRECISION == "bf16" and bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None
            and torch.is_grad_enabled() and K.bn2d_relu_supported(x) and conv.kernel_size[0] == 3 and conv.stride[0] == 1
            and conv.padding[0] == 1 and x.shape[1] <= 1024 and x.shape[3] % 2 == 0
            and K.conv2d_bf16_supported(conv.in_channels, conv.out_channels, x.shape[2], x.shape[3]))


def bn_relu_conv(bn, conv, x):
    """conv(relu(bn(x))) through BnReluConvBf16Fn (caller checked bn_relu_conv_fusable)."""
    y = BnReluConvBf16Fn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, conv.weight,
                               conv.bias)
    count_bn_batch(bn)
    return y


import weakref

_pending_nbt = weakref.WeakKeyDictionary()      # BatchNorm module -> increments not yet applied (no strong references:
                                                # a discarded model takes its pending counts with it)


def _nbt_state_dict_hook(module, prefix, keep_vars):      # a module-level function: the layer stays picklable
    flush_bn_counters()


def _nbt_load_hook(module, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
    # load_state_dict on ANY module that contains the layer: pending increments land first (and are then overwritten by
    # the loaded counter) instead of being added on top of it later
    flush_bn_counters()


def count_bn_batch(bn):
    """`bn.num_batches_tracked += 1`, deferred: the 23 BatchNorm layers of a training step would each launch a one-element
    int64 add (torch._foreach_add_ on 0-dim int64 tensors takes the per-tensor path too: 23 launches a step in the round-3
    profile).  The increments are counted on the host, per MODULE (whatever tensor the module holds when they land:
    `.to(device)` in between is fine), and land when somebody can look at them: flush_bn_counters -- a state_dict pre-hook
    and a load_state_dict pre-hook on every counted layer, plus eval / load_state_dict of the detector.  A direct read of
    `bn.num_batches_tracked` between flushes sees the last flushed value; nothing in a training step reads the counter
    (every BatchNorm here has a fixed momentum)."""
    if not getattr(bn, "_sassd_nbt_hook", False):
        # whoever serialises / loads this layer -- directly or through any parent -- sees the flushed counter
        bn.register_state_dict_pre_hook(_nbt_state_dict_hook)
        bn._register_load_state_dict_pre_hook(_nbt_load_hook, with_module=True)
        bn._sassd_nbt_hook = True
    _pending_nbt[bn] = _pending_nbt.get(bn, 0) + 1


def flush_bn_counters():
    if len(_pending_nbt):
        todo = list(_pending_nbt.items())
        _pending_nbt.clear()
        for bn, n in todo:
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked += n


def bn_relu_2d(bn, x):
    """`relu(bn(x))` for a training-mode nn.BatchNorm2d on the fused HIP kernels when the layer / tensor allow it, else
    the torch ops (eval mode, no affine, cumulative-average momentum, exotic shapes)."""
    if (bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None and torch.is_grad_enabled()
            and K.bn2d_relu_supported(x)):
        y = BnRelu2dFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps)
        count_bn_batch(bn)
        return y
    return torch.relu(bn(x))


class AuxHeadFn(Function):
    """The auxiliary head's interpolation + three Linear layers + both loss sums in one forward kernel, and its whole
    backward (feature gradients of the three scales, weight gradients) in three (sassd_aux_head_fwd / _bwd).
    Inputs: the three middle feature tensors [M_s, C_s], point_fc / point_cls / point_reg weights, then the
    non-differentiable context (3-NN indices / squared distances per scale, labels, targets, positive count).
    -> loss sums [2] = (focal, smooth-L1), both already divided by max(#positive points, 1)."""

    @staticmethod
    def forward(ctx, f0, f1, f2, w_fc, w_cls, w_reg, nn_idx, nn_d2, label, target, npos):
        feats = [f0.contiguous(), f1.contiguous(), f2.contiguous()]
        w1 = w_fc.detach().contiguous()
        w2 = torch.cat([w_cls.detach(), w_reg.detach()], 0).contiguous()
        sums, wgt, h, out, gout = K.aux_head_fwd(feats, nn_idx, nn_d2, w1, w2, label, target, npos)
        ctx.save_for_backward(feats[0], feats[1], feats[2], w1, w2, wgt, h, gout)
        ctx.nn_idx = nn_idx
        return sums

    @staticmethod
    def backward(ctx, g):
        f0, f1, f2, w1, w2, wgt, h, gout = ctx.saved_tensors
        gf, dw1, dw2 = K.aux_head_bwd([f0, f1, f2], ctx.nn_idx, w1, w2, wgt, h, gout, g.contiguous())
        return gf[0], gf[1], gf[2], dw1, dw2[0:1], dw2[1:4], None, None, None, None, None


class GuidedDecodeFn(Function):
    """Padded guided anchors of the training step (sassd_guided_decode_fwd / _bwd): ground truth first, then the decoded,
    direction-flipped boxes of the selected anchors; gradients reach box_preds through the decode."""

    @staticmethod
    def forward(ctx, box_preds, dir_preds, anchors, sel, sel_count, gt_boxes, gt_off, gmax):
        box_preds = box_preds.contiguous()
        guided, counts = K.guided_decode_fwd(box_preds, dir_preds.detach().contiguous() if dir_preds is not None else None,
                                             anchors, sel, sel_count, gt_boxes, gt_off, gmax)
        ctx.save_for_backward(box_preds, anchors, sel, sel_count, gt_off)
        ctx.gmax = gmax
        ctx.mark_non_differentiable(counts)
        return guided, counts

    @staticmethod
    def backward(ctx, dguided, _dcounts):
        box_preds, anchors, sel, sel_count, gt_off = ctx.saved_tensors
        dbox = K.guided_decode_bwd(box_preds, anchors, sel, sel_count, gt_off, ctx.gmax, dguided.contiguous())
        return dbox, None, None, None, None, None, None, None


class FocalLossFn(Function):
    """Sigmoid focal loss sum of [n] logits with its gradient from the same pass (sassd_focal_loss)."""

    @staticmethod
    def forward(ctx, logits, labels, num_pos):
        out, grad = K.focal_loss(logits.contiguous().view(-1), labels.contiguous().view(-1), num_pos)
        ctx.save_for_backward(grad)
        ctx.shape = logits.shape
        return out

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g).view(ctx.shape), None, None
