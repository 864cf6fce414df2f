"""Mirror of the spconv v1.0 surface the reference uses (mmdet/models/necks/cmn.py:1,109-112,138-173,208-212;
mmdet/core/bbox/transforms.py:218-223): SparseConvTensor(.features/.indices/.dense()), SubMConv3d, SparseConv3d,
SparseSequential -- same constructor arguments, same weight layout [kz,ky,kx,Cin,Cout] (checkpoint compatible),
same `indice_key` rulebook sharing.  All arithmetic runs in libsassd (hash/bitmap rulebook + MFMA gather conv).

This module-by-module facade is the drop-in/compatibility path (one host read of the output row count per strided
conv, because the reference API exposes exact-size tensors); the fused, sync-free production path is
sassd.pipeline.InferencePlan, which calls the same kernels with folded BatchNorm.
"""
import math

import numpy as np
import torch
from torch import nn

from . import kernels as K
from .autograd import BnReluFn, DensifyFn, SparseConvFn, count_bn_batch, _n_ptr as _cached_n_ptr


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size):
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {}          # indice_key -> (out_indices, nbr, out_shape, hash table)
        self._table = None

    @property
    def spatial_size(self):
        return int(np.prod(self.spatial_shape))

    def _n_ptr(self):
        return _cached_n_ptr(self.indices.shape[0], self.indices.device)        # read-only [1] int32, cached per value

    def table(self):
        if self._table is None:
            idx = self.indices.int().contiguous()
            cap = max(idx.shape[0], 1)
            self._table = K.HashTable(cap, idx.device).build(idx, self._n_ptr(), self.spatial_shape, self.batch_size)
        return self._table

    def dense(self, channels_first=True):
        """[B, C, D, H, W] like spconv (scatter_nd + permute)."""
        d, h, w = self.spatial_shape
        c = self.features.shape[1]
        n = self.indices.shape[0]
        if torch.is_grad_enabled() and self.features.requires_grad:
            out = DensifyFn.apply(self.features.float(), self.indices.int().contiguous(), (d, h, w), self.batch_size)
            return out if channels_first else out.permute(0, 2, 3, 4, 1).contiguous()
        out = K.densify(self.features.contiguous(), self.indices.int().contiguous(), self._n_ptr(), max(n, 1),
                        (d, h, w), self.batch_size, 0)
        out = out.view(self.batch_size, c, d, h, w)
        return out if channels_first else out.permute(0, 2, 3, 4, 1).contiguous()


def _triple(v):
    return tuple(v) if isinstance(v, (list, tuple)) else (v, v, v)


class SparseConvolution(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False, subm=False,
                 indice_key=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _triple(kernel_size), _triple(stride), _triple(padding)
        self.subm, self.indice_key = subm, indice_key
        self.conv1x1 = int(np.prod(self.kernel_size)) == 1
        self.weight = nn.Parameter(torch.empty(*self.kernel_size, in_channels, out_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()
        self._packed = None
        self._packed_version = None

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)      # .to() / .cuda() on this layer alone: its packed image is stale
        K.bump_weights_generation()
        return out

    def reset_parameters(self):
        n = self.in_channels * int(np.prod(self.kernel_size))
        stdv = 1.0 / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def packed_weight(self):
        v = K.weight_key(self.weight)
        if self._packed is None or self._packed_version != v:
            k = int(np.prod(self.kernel_size))
            w = self.weight.detach().reshape(k, self.in_channels, self.out_channels).contiguous().float()
            self._packed = K.spconv_pack_weight(w)
            self._packed_version = v
        return self._packed

    def _conv(self, feats, nbr, n_out, k, grad):
        """Raw conv (+bias) over a gather table: autograd-recording HIP path when gradients are wanted."""
        cin, cout = self.in_channels, self.out_channels
        if grad:
            y = SparseConvFn.apply(feats, self.weight.view(k, cin, cout), nbr, n_out, self.packed_weight(),
                                   isinstance(self, SubMConv3d))
            return y + self.bias if self.bias is not None else y
        bias = self.bias.detach() if self.bias is not None else None
        n_ptr = _cached_n_ptr(n_out, feats.device)
        return K.spconv_fwd(feats, nbr, n_ptr, max(n_out, 1), self.packed_weight(), k, cin, cout, None, bias)[:n_out]

    def book(self, inp):
        """Rulebook of this layer for `inp`'s coordinates: (out_indices, nbr, out_shape, out_table).  Coordinate-only
        work (hash / bitmap kernels + ONE host read of the output row count for strided layers); cached per
        `indice_key` in `inp.indice_dict`, so it can be built ahead of the forward pass (VxNet.precompute_rulebooks)."""
        idx = inp.indices.int().contiguous()
        n = idx.shape[0]
        if self.kernel_size != (3, 3, 3):
            raise NotImplementedError("only k=3 and k=1 sparse convs exist on the SA-SSD path")
        book = inp.indice_dict.get(self.indice_key) if self.indice_key is not None else None
        if book is not None:
            return book
        if self.subm:
            nbr = K.rulebook_subm(idx, inp._n_ptr(), max(n, 1), inp.spatial_shape, inp.batch_size, inp.table())
            book = (idx, nbr, inp.spatial_shape, inp._table)
        else:
            if self.stride != (2, 2, 2) or self.padding != (1, 1, 1):
                raise NotImplementedError("strided sparse conv is k=3,s=2,p=1 on the SA-SSD path (cmn.py:170)")
            cap_out = max(8 * n, 1)
            oi, on, nbr = K.rulebook_conv(idx, inp._n_ptr(), max(n, 1), inp.spatial_shape, inp.batch_size,
                                          inp.table(), cap_out)
            n_out = int(on.item())            # compatibility path: exact-size tensors need the count
            book = (oi[:n_out].contiguous(), nbr[:max(n_out, 1)].contiguous(),
                    list(K.conv_out_shape(inp.spatial_shape)), None)
        if self.indice_key is not None:
            inp.indice_dict[self.indice_key] = book
        return book

    def forward(self, inp):
        assert isinstance(inp, SparseConvTensor)
        feats = inp.features.contiguous().float()
        n = inp.indices.shape[0]
        grad = torch.is_grad_enabled() and (feats.requires_grad or self.weight.requires_grad)
        if self.conv1x1:
            # spconv shortcut: features @ weight.view(Cin, Cout), indices unchanged
            y = self._conv(feats, None, n, 1, grad)
            out = SparseConvTensor(y, inp.indices, inp.spatial_shape, inp.batch_size)
            out.indice_dict, out._table = inp.indice_dict, inp._table
            return out
        out_idx, nbr, oshape, otable = self.book(inp)
        n_out = out_idx.shape[0]
        y = self._conv(feats, nbr, n_out, 27, grad)
        out = SparseConvTensor(y, out_idx, oshape, inp.batch_size)
        out.indice_dict = inp.indice_dict
        out._table = otable if not self.subm else inp._table
        return out


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, bias, True, indice_key)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, bias, False, indice_key)


class SparseSequential(nn.Sequential):
    """Sparse modules consume/produce SparseConvTensor; plain nn.Modules are applied to `.features`."""

    # Training-mode BatchNorm1d -> ReLU pairs run as one fused HIP forward / backward pair (sassd_bn_relu_*, 2 launches
    # each way instead of torch's 6: collect / transform / clamp, threshold / reduce / elementwise).  Statistics are
    # accumulated in double and rounded once, i.e. within fp32 rounding of the exact batch statistics (torch's fp32
    # Welford and the CPU oracle's fp32 two-pass mean / variance are too; the unit test holds the kernels to torch at
    # 2e-5 / 1e-4).  Set to False for torch's own BatchNorm1d + ReLU (A/B, bench.py --torch-bn).
    fuse_bn_relu = True

    def forward(self, inp):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, SparseConvolution):
                inp = m(inp)
            elif isinstance(inp, SparseConvTensor):
                if inp.indices.shape[0] != 0:
                    f = inp.features
                    nxt = mods[i + 1] if i + 1 < len(mods) else None
                    if (self.fuse_bn_relu and isinstance(m, nn.BatchNorm1d) and isinstance(nxt, nn.ReLU) and m.training
                            and m.affine and m.track_running_stats and m.momentum is not None and f.is_cuda
                            and f.dtype == torch.float32 and torch.is_grad_enabled()
                            and K.bn_relu_supported(f.shape[0], f.shape[1])):
                        inp.features = BnReluFn.apply(f, m.weight, m.bias, m.running_mean, m.running_var, m.momentum,
                                                      m.eps)
                        count_bn_batch(m)
                        i += 2
                        continue
                    inp.features = m(f)
            else:
                inp = m(inp)
            i += 1
        return inp
