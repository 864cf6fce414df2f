"""CPU oracle: spconv v1.0 rulebook semantics (TEST INFRASTRUCTURE ONLY).

spconv v1.0 (traveller59/spconv, pinned only by /root/reference/readme.md:58) is NOT vendored in the
reference, so this restates its published algorithm; PARITY UNPINNED upstream.
Call sites in the reference: mmdet/models/necks/cmn.py:147-173 (SubMConv3d k=3, SparseConv3d k=3 s=2 p=1).

Canonical form (ours):  out_indices array exact; per kernel offset k=(kz*3+ky)*3+kx the SET of
(in_row,out_row) pairs; pair_num[27] exact.  We also emit the gather table nbr[Nout,27] (in_row or -1)
which is what the HIP conv kernel consumes (each (out,k) has at most one input).
"""
import numpy as np


def _lin(idx, shape):
    d, h, w = shape
    i = idx.astype(np.int64)
    return ((i[:, 0] * d + i[:, 1]) * h + i[:, 2]) * w + i[:, 3]


def _lookup(keys_sorted, order, q):
    pos = np.searchsorted(keys_sorted, q)
    pos = np.clip(pos, 0, len(keys_sorted) - 1)
    hit = keys_sorted[pos] == q
    return np.where(hit, order[pos], -1)


def subm_rulebook(indices, spatial_shape):
    """SubMConv3d(k=3): out rows == in rows; output at c gathers input at c + (k-1) per axis (pad 1)."""
    indices = np.asarray(indices, np.int32)
    n = len(indices)
    keys = _lin(indices, spatial_shape)
    order = np.argsort(keys, kind="stable")
    ks = keys[order]
    nbr = np.full((n, 27), -1, np.int32)
    d, h, w = spatial_shape
    for kz in range(3):
        for ky in range(3):
            for kx in range(3):
                k = (kz * 3 + ky) * 3 + kx
                c = indices.astype(np.int64).copy()
                c[:, 1] += kz - 1
                c[:, 2] += ky - 1
                c[:, 3] += kx - 1
                ok = ((c[:, 1] >= 0) & (c[:, 1] < d) & (c[:, 2] >= 0) & (c[:, 2] < h) &
                      (c[:, 3] >= 0) & (c[:, 3] < w))
                q = _lin(np.where(ok[:, None], c, 0), spatial_shape)
                r = _lookup(ks, order, q)
                nbr[:, k] = np.where(ok, r, -1)
    return indices.copy(), nbr


def out_shape(spatial_shape, k=3, s=2, p=1):
    return tuple((int(x) + 2 * p - (k - 1) - 1) // s + 1 for x in spatial_shape)


def conv_rulebook(indices, spatial_shape, batch_size, k=3, s=2, p=1):
    """SparseConv3d(k=3,s=2,p=1): input i, offset kk -> output o=(i+p-kk)/s when divisible & in range.
    Out rows ascending in linear (b,z,y,x) (GPU path of spconv sorts+uniques)."""
    indices = np.asarray(indices, np.int32)
    oshape = out_shape(spatial_shape, k, s, p)
    od, oh, ow = oshape
    cand = []
    i64 = indices.astype(np.int64)
    for kz in range(k):
        for ky in range(k):
            for kx in range(k):
                oz = i64[:, 1] + p - kz
                oy = i64[:, 2] + p - ky
                ox = i64[:, 3] + p - kx
                ok = ((oz % s == 0) & (oy % s == 0) & (ox % s == 0))
                oz, oy, ox = oz // s, oy // s, ox // s
                ok &= (oz >= 0) & (oz < od) & (oy >= 0) & (oy < oh) & (ox >= 0) & (ox < ow)
                lin = ((i64[:, 0] * od + oz) * oh + oy) * ow + ox
                cand.append((ok, lin))
    allk = np.concatenate([l[o] for o, l in cand])
    uniq = np.unique(allk)                       # ascending
    nout = len(uniq)
    out_idx = np.empty((nout, 4), np.int32)
    r = uniq.copy()
    out_idx[:, 3] = r % ow; r //= ow
    out_idx[:, 2] = r % oh; r //= oh
    out_idx[:, 1] = r % od; r //= od
    out_idx[:, 0] = r
    nbr = np.full((nout, k * k * k), -1, np.int32)
    rows = np.arange(len(indices), dtype=np.int32)
    for kk, (ok, lin) in enumerate(cand):
        o = np.searchsorted(uniq, lin[ok])
        nbr[o, kk] = rows[ok]
    return out_idx, nbr, oshape


def nbr_to_pairs(nbr):
    """spconv-format rulebook from the gather table: per offset (in_rows, out_rows) and pair_num[K]."""
    pairs = []
    num = np.zeros(nbr.shape[1], np.int32)
    for k in range(nbr.shape[1]):
        o = np.nonzero(nbr[:, k] >= 0)[0].astype(np.int32)
        pairs.append((nbr[o, k].copy(), o))
        num[k] = len(o)
    return pairs, num
