"""Mirror of mmdet/ops/pointnet2/pointnet2_utils.py (reference :9-86): ThreeNN / ThreeInterpolate autograd functions
on the HIP kernels of pointops.hip; `three_nn` returns sqrt distances like the reference (:33)."""
import torch
from torch.autograd import Function

from . import kernels as K


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        """unknown (N,4) [b,x,y,z], known (M,4) -> dist (N,3) l2 distance to the three nearest neighbours, idx (N,3)."""
        dist2, idx = K.three_nn(unknown.contiguous().float(), known.contiguous().float())
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeNNBinned(Function):
    """Same outputs as ThreeNN (bit-identical) through sassd_three_nn_binned: O(N+M)-ish instead of O(N*M)."""

    @staticmethod
    def forward(ctx, unknown, known, xy_range, cell, batch_size):
        dist2, idx = K.three_nn_binned(unknown.contiguous().float(), known.contiguous().float(), xy_range, cell,
                                       batch_size)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return (None,) * 5


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        """features (M,C), idx (N,3), weight (N,3) -> (N,C)."""
        ctx.three_interpolate_for_backward = (idx, weight, features.shape[0])
        return K.three_interpolate(features.contiguous().float(), idx.contiguous(), weight.contiguous().float())

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, m = ctx.three_interpolate_for_backward
        return K.three_interpolate_grad(grad_out.contiguous().float(), idx, weight.contiguous().float(), m), None, None


three_interpolate = ThreeInterpolate.apply


def nearest_neighbor_interpolate(unknown, known, known_feats, grid=None):
    """mmdet/models/necks/cmn.py:175-189.  grid = (xy_range, cell, batch_size) selects the binned exact search."""
    if grid is not None:
        dist, idx = ThreeNNBinned.apply(unknown, known, *grid)
    else:
        dist, idx = three_nn(unknown, known)
    dist_recip = 1.0 / (dist + 1e-8)
    weight = dist_recip / torch.sum(dist_recip, dim=1, keepdim=True)
    return three_interpolate(known_feats, idx, weight)
