"""Thin torch-tensor front-end over the C ABI (include/sassd.h).  torch is used only for device memory and
the current HIP stream; every function here launches hand-written HIP kernels from libsassd.so and raises if
the library is missing or a call fails (no CPU fallback, no eager-PyTorch fallback)."""
import numpy as np
import torch

from . import _C

_ws_cache = {}

# Packed-weight caches (SparseConvolution.packed_weight, _HipConv2d.packed_*, SingleStageDetector.plan) are keyed on
# the parameter's autograd version AND on this generation: the fused optimizer and the checkpoint loaders write the
# flat parameter buffer through raw pointers / views, which never bumps `Tensor._version`.
_weights_gen = [0]


def weights_generation():
    return _weights_gen[0]


def bump_weights_generation():
    _weights_gen[0] += 1


def weight_key(w):
    return (w._version, w.data_ptr(), str(w.device), _weights_gen[0])


_ws_scope = [None]


class ws_scope:
    """Scratch buffers handed out inside `with ws_scope(owner)` belong to `owner`: plans that run concurrently on
    different HIP streams must not share kernel workspaces."""

    def __init__(self, owner):
        self.owner = owner

    def __enter__(self):
        self.prev, _ws_scope[0] = _ws_scope[0], self.owner

    def __exit__(self, *exc):
        _ws_scope[0] = self.prev


def workspace(name, nbytes, device):
    """Grow-only cached scratch buffer (uint8) per (name, device, workspace scope)."""
    key = (name, str(device), _ws_scope[0])
    t = _ws_cache.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = t
    return t


def scoped_workspaces(owner):
    """The scratch buffers currently handed out under workspace scope `owner` (a captured hipGraph keeps them alive)."""
    return [t for (name, dev, scope), t in _ws_cache.items() if scope == owner]


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _chk_cuda(*ts):
    for t in ts:
        if t is not None and not (t.is_cuda and t.is_contiguous()):
            raise RuntimeError("sassd kernels need contiguous CUDA(HIP) tensors")


def new_status(device):
    return torch.zeros(1, dtype=torch.int32, device=device)


# --------------------------------------------------------------------------------------------------
def voxelize(points, voxel_size, coors_range, max_points, max_voxels, batch_idx=0, coors_cols=3,
             want_voxels=True, want_mean=True, nfeat=4, out=None, row_offset=None, status=None, cap=None,
             n_dev=None):
    """points [N,ndim] f32 cuda.  Returns dict(voxels, coors, num_points, mean, voxel_num) of capacity-sized
    tensors (rows [0, voxel_num) valid; voxel_num is a device int32 scalar).  With `n_dev` (device int32 scalar) the
    first min(n_dev, N) rows of `points` are the cloud (graph-capturable: the launches depend on N only)."""
    _chk_cuda(points)
    L = _C.lib()
    n, ndim = points.shape
    dev = points.device
    vs, cr = _f32(voxel_size), _f32(coors_range)
    cap = int(cap if cap is not None else max_voxels)
    out = out or {}
    voxels = out.get("voxels")
    if want_voxels and voxels is None:
        voxels = torch.empty(cap, max_points, ndim, dtype=torch.float32, device=dev)
    coors = out.get("coors")
    if coors is None:
        coors = torch.empty(cap, coors_cols, dtype=torch.int32, device=dev)
    num = out.get("num_points")
    if num is None:
        num = torch.empty(cap, dtype=torch.int32, device=dev)
    mean = out.get("mean")
    nfeat = min(nfeat, ndim)
    if want_mean and mean is None:
        mean = torch.empty(cap, nfeat, dtype=torch.float32, device=dev)
    vnum = out.get("voxel_num")
    if vnum is None:
        vnum = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = L.sassd_voxelize_workspace_bytes(n, max_points)
    ws = workspace("voxelize", wsb, dev)
    tail = (ndim, vs.ctypes.data, cr.ctypes.data, int(max_points), int(max_voxels), int(batch_idx),
            _C.ptr(voxels) if want_voxels else None, _C.ptr(coors), coors_cols, _C.ptr(num),
            _C.ptr(mean) if want_mean else None, nfeat, _C.ptr(row_offset), _C.ptr(vnum), cap, _C.ptr(status),
            _C.ptr(ws), wsb, _C.stream())
    if n_dev is None:
        rc = L.sassd_voxelize(_C.ptr(points), n, *tail)
    else:
        rc = L.sassd_voxelize_dev(_C.ptr(points), n, _C.ptr(n_dev), *tail)
    _C.check(rc, "sassd_voxelize")
    return dict(voxels=voxels, coors=coors, num_points=num, mean=mean, voxel_num=vnum)


def voxel_mean(voxels, num_points, nfeat=4):
    _chk_cuda(voxels, num_points)
    m, t, ndim = voxels.shape
    out = torch.empty(m, nfeat, dtype=torch.float32, device=voxels.device)
    rc = _C.lib().sassd_voxel_mean(_C.ptr(voxels), _C.ptr(num_points.int().contiguous()), m, t, ndim, nfeat,
                                   _C.ptr(out), _C.stream())
    _C.check(rc, "sassd_voxel_mean")
    return out


# --------------------------------------------------------------------------------------------------
class HashTable:
    """coordinate -> row lookup table for one sparse level."""

    def __init__(self, cap, device):
        self.cap = int(cap)
        self.nbytes = _C.lib().sassd_hash_bytes(self.cap)
        self.buf = torch.empty(self.nbytes, dtype=torch.uint8, device=device)

    def build(self, indices, n_ptr, shape, batch_size, status=None):
        d, h, w = shape
        rc = _C.lib().sassd_hash_build(_C.ptr(indices), _C.ptr(n_ptr), self.cap, d, h, w, batch_size,
                                       _C.ptr(self.buf), self.nbytes, _C.ptr(status), _C.stream())
        _C.check(rc, "sassd_hash_build")
        return self


def rulebook_subm(indices, n_ptr, cap, shape, batch_size, table, nbr=None):
    d, h, w = shape
    if nbr is None:
        nbr = torch.empty(cap, 27, dtype=torch.int32, device=indices.device)
    rc = _C.lib().sassd_rulebook_subm(_C.ptr(indices), _C.ptr(n_ptr), cap, d, h, w, batch_size, _C.ptr(table.buf),
                                      table.nbytes, _C.ptr(nbr), _C.stream())
    _C.check(rc, "sassd_rulebook_subm")
    return nbr


def conv_out_shape(shape):
    return tuple((int(x) - 1) // 2 + 1 for x in shape)


def rulebook_conv(indices, n_in_ptr, cap_in, shape, batch_size, table, cap_out, out_indices=None, n_out_ptr=None,
                  nbr=None, status=None):
    d, h, w = shape
    dev = indices.device
    L = _C.lib()
    if out_indices is None:
        out_indices = torch.empty(cap_out, 4, dtype=torch.int32, device=dev)
    if n_out_ptr is None:
        n_out_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
    if nbr is None:
        nbr = torch.empty(cap_out, 27, dtype=torch.int32, device=dev)
    wsb = L.sassd_rulebook_conv_workspace_bytes(d, h, w, batch_size)
    ws = workspace("rulebook_conv", wsb, dev)
    rc = L.sassd_rulebook_conv(_C.ptr(indices), _C.ptr(n_in_ptr), cap_in, d, h, w, batch_size, _C.ptr(table.buf),
                               table.nbytes, _C.ptr(out_indices), _C.ptr(n_out_ptr), cap_out, _C.ptr(nbr),
                               _C.ptr(status), _C.ptr(ws), wsb, _C.stream())
    _C.check(rc, "sassd_rulebook_conv")
    return out_indices, n_out_ptr, nbr


class RulebookPyramid:
    """Pre-resolved argument block of sassd_rulebook_pyramid for fixed buffers (a plan builds it once; a frame is then
    `build(level_begin, level_end)` calls with no per-call marshalling)."""

    def __init__(self, indices, n_ptrs, caps, shape0, batch_size, nbr_subm, nbr_down, status=None):
        import ctypes as C
        L = _C.lib()
        self.levels = len(indices)
        dev = indices[0].device
        arr = lambda ts: (C.c_void_p * self.levels)(*[(t.data_ptr() if t is not None else None) for t in ts])  # noqa: E731
        self._keep = (list(indices), list(n_ptrs), list(nbr_subm), list(nbr_down), status)
        self.indices, self.n_ptrs = arr(indices), arr(n_ptrs)
        self.nbr_subm, self.nbr_down = arr(nbr_subm), arr(nbr_down)
        self.caps = (C.c_int * self.levels)(*[int(c) for c in caps])
        self.shape0, self.B = tuple(int(v) for v in shape0), int(batch_size)
        self.status = status
        self.wsb = L.sassd_rulebook_pyramid_workspace_bytes(self.levels, self.caps, *self.shape0, self.B)
        if self.wsb == 0:
            raise ValueError("rulebook pyramid: unsupported level count / grid too large")
        self.ws = torch.empty(self.wsb, dtype=torch.uint8, device=dev)

    def build(self, level_begin=0, level_end=None, persistent=False, wgs_per_cu=0):
        """`persistent`: every phase in ONE launch with in-launch grid barriers (whole pyramid only)."""
        level_end = self.levels if level_end is None else level_end
        d, h, w = self.shape0
        flags = (1 | (int(wgs_per_cu) << 8)) if persistent else 0
        rc = _C.lib().sassd_rulebook_pyramid(self.levels, self.indices, self.n_ptrs, self.caps, d, h, w, self.B,
                                             self.nbr_subm, self.nbr_down, level_begin, level_end, flags,
                                             _C.ptr(self.status), _C.ptr(self.ws), self.wsb, _C.stream())
        _C.check(rc, "sassd_rulebook_pyramid")


class Graph:
    """hipGraph of a launch sequence issued through this module on the current stream (sassd_graph_*)."""

    def __init__(self):
        self.exe = None

    def capture(self, fn):
        import ctypes as C
        st = _C.stream()
        _C.check(_C.lib().sassd_graph_begin(st), "sassd_graph_begin")
        try:
            fn()
        except BaseException:
            junk = C.c_void_p()
            _C.lib().sassd_graph_end(st, C.byref(junk))
            raise
        exe = C.c_void_p()
        _C.check(_C.lib().sassd_graph_end(st, C.byref(exe)), "sassd_graph_end")
        self.exe = exe
        return self

    def launch(self):
        _C.check(_C.lib().sassd_graph_launch(self.exe, _C.stream()), "sassd_graph_launch")

    def __del__(self):
        try:
            if self.exe is not None:
                _C.lib().sassd_graph_destroy(self.exe)
        except Exception:
            pass


# ---- kernel selection: a per-call `cfg` word of the C ABI (include/sassd.h); 0 in production -----------------------------
# DEFAULT_CFG is what the wrappers below pass when the caller gives none: host-side state of THIS binding (the library has no
# process-wide switch any more).  SASSD_SPCONV_DEBUG / SASSD_WINO4_CFG preset it for an unmodified test / bench command.
import os as _os
DEFAULT_CFG = {"spconv": int(_os.environ.get("SASSD_SPCONV_DEBUG", "0"), 0),
               "wino4": int(_os.environ.get("SASSD_WINO4_CFG", "0"), 0)}


def spconv_cfg(geometry=0, flags=0):
    """cfg word of the sparse-conv entry points: ablation flags (low 16 bits) + forced workgroup geometry."""
    return (int(geometry) << 16) | (int(flags) & 0xFFFF)


def wino4_cfg(geometry=0, dbg=0):
    """cfg word of the Winograd entry points: GEMM geometry (bits 0-7) + ablation / stage-skip flags."""
    return (int(geometry) & 0xFF) | (int(dbg) << 8)


class default_cfg:
    """`with K.default_cfg(spconv=..., wino4=...):` -- the cfg the wrappers pass inside the block (tests, tools)."""

    def __init__(self, spconv=None, wino4=None):
        self.new = {k: int(v) for k, v in (("spconv", spconv), ("wino4", wino4)) if v is not None}

    def __enter__(self):
        self.old = dict(DEFAULT_CFG)
        DEFAULT_CFG.update(self.new)
        return self

    def __exit__(self, *exc):
        DEFAULT_CFG.update(self.old)
        return False


def rulebook_pairs(nbr, n_out_ptr, cap_out):
    dev = nbr.device
    k = nbr.shape[1]
    pairs = torch.full((k, 2, cap_out), -1, dtype=torch.int32, device=dev)
    num = torch.zeros(k, dtype=torch.int32, device=dev)
    rc = _C.lib().sassd_rulebook_pairs(_C.ptr(nbr), _C.ptr(n_out_ptr), cap_out, k, _C.ptr(pairs), _C.ptr(num),
                                       _C.stream())
    _C.check(rc, "sassd_rulebook_pairs")
    return pairs, num


# --------------------------------------------------------------------------------------------------
def spconv_pack_weight(w):
    """w [K, Cin, Cout] f32 cuda (spconv layout flattened) -> packed MFMA-fragment order."""
    _chk_cuda(w)
    k, cin, cout = w.shape
    packed = torch.empty(_C.lib().sassd_spconv_packed_floats(k, cin, cout), dtype=torch.float32, device=w.device)
    rc = _C.lib().sassd_spconv_pack_weight(_C.ptr(w), k, cin, cout, _C.ptr(packed), _C.stream())
    _C.check(rc, "sassd_spconv_pack_weight")
    return packed


def spconv_fwd(x, nbr, n_out_ptr, cap_out, w_packed, k, cin, cout, scale=None, shift=None, relu=False, y=None, cfg=None):
    _chk_cuda(x, nbr, w_packed, scale, shift)
    if y is None:
        y = torch.empty(cap_out, cout, dtype=torch.float32, device=x.device)
    rc = _C.lib().sassd_spconv_fwd(_C.ptr(x), _C.ptr(nbr), _C.ptr(n_out_ptr), cap_out, _C.ptr(w_packed), k, cin, cout,
                                   _C.ptr(scale), _C.ptr(shift), 1 if relu else 0, _C.ptr(y),
                                   DEFAULT_CFG["spconv"] if cfg is None else int(cfg), _C.stream())
    _C.check(rc, "sassd_spconv_fwd")
    return y


def rulebook_transpose(nbr, n_out_ptr, cap_out, cap_in):
    nbrT = torch.empty(cap_in, 27, dtype=torch.int32, device=nbr.device)
    _C.check(_C.lib().sassd_rulebook_transpose(_C.ptr(nbr), _C.ptr(n_out_ptr), cap_out, _C.ptr(nbrT), cap_in,
                                               _C.stream()), "sassd_rulebook_transpose")
    return nbrT


def spconv_pack_weight_t(w):
    """forward weight [K, Cin, Cout] -> packed W[k]^T image for sassd_spconv_bwd_data."""
    _chk_cuda(w)
    k, cin, cout = w.shape
    packed = torch.empty(k * cin * cout, dtype=torch.float32, device=w.device)
    _C.check(_C.lib().sassd_spconv_pack_weight_t(_C.ptr(w), k, cin, cout, _C.ptr(packed), _C.stream()),
             "sassd_spconv_pack_weight_t")
    return packed


def spconv_bwd_data(dy, nbrT, n_in_ptr, cap_in, wT_packed, k, cin, cout, cfg=None):
    _chk_cuda(dy, nbrT, wT_packed)
    dx = torch.empty(cap_in, cin, dtype=torch.float32, device=dy.device)
    _C.check(_C.lib().sassd_spconv_bwd_data(_C.ptr(dy), _C.ptr(nbrT), _C.ptr(n_in_ptr), cap_in, _C.ptr(wT_packed), k,
                                            cin, cout, _C.ptr(dx), DEFAULT_CFG["spconv"] if cfg is None else int(cfg),
                                            _C.stream()), "sassd_spconv_bwd_data")
    return dx


def spconv_bwd_weight(x, dy, nbr, n_out_ptr, cap_out, cin, cout, dw=None, accumulate=False, cfg=None):
    _chk_cuda(x, dy, nbr)
    L = _C.lib()
    if dw is None:                                  # (the reduction kernel overwrites every element unless `accumulate`)
        dw = torch.empty(27, cin, cout, dtype=torch.float32, device=x.device)
    wsb = L.sassd_spconv_bwd_weight_workspace_bytes(cap_out, 27, cin, cout)
    ws = workspace("spconv_wgrad", wsb, x.device)
    _C.check(L.sassd_spconv_bwd_weight(_C.ptr(x), _C.ptr(dy), _C.ptr(nbr), _C.ptr(n_out_ptr), cap_out, 27, cin, cout,
                                       _C.ptr(dw), 1 if accumulate else 0,
                                       DEFAULT_CFG["spconv"] if cfg is None else int(cfg), _C.ptr(ws), wsb, _C.stream()),
             "sassd_spconv_bwd_weight")
    return dw


def densify(feats, indices, n_ptr, cap, shape, batch_size, channel_order=0, out=None):
    d, h, w = shape
    c = feats.shape[1]
    if out is None:
        out = torch.empty(batch_size, c * d, h, w, dtype=torch.float32, device=feats.device)
    rc = _C.lib().sassd_densify(_C.ptr(feats), _C.ptr(indices), _C.ptr(n_ptr), cap, c, d, h, w, batch_size,
                                channel_order, _C.ptr(out), _C.stream())
    _C.check(rc, "sassd_densify")
    return out


# --------------------------------------------------------------------------------------------------
def conv2d_pack_weight(w):
    """w [Cout, Cin, k, k] f32 cuda (torch layout) -> packed [Cin/8][k*k][8][CoutPad]."""
    _chk_cuda(w)
    cout, cin, ks, _ = w.shape
    packed = torch.empty(_C.lib().sassd_conv2d_packed_floats(cin, cout, ks), dtype=torch.float32, device=w.device)
    rc = _C.lib().sassd_conv2d_pack_weight(_C.ptr(w), cout, cin, ks, _C.ptr(packed), _C.stream())
    _C.check(rc, "sassd_conv2d_pack_weight")
    return packed


def conv2d_fwd(x, w_packed, cout, ksize, scale=None, shift=None, relu=False, y=None, cfg=0):
    """cfg: per-call ablation word of the direct kernel (tools/run_conv.py), 0 in production."""
    _chk_cuda(x, w_packed, scale, shift)
    b, cin, h, w = x.shape
    if y is None:
        y = torch.empty(b, cout, h, w, dtype=torch.float32, device=x.device)
    if cfg:
        rc = _C.lib().sassd_conv2d_fwd_cfg(_C.ptr(x), _C.ptr(w_packed), _C.ptr(scale), _C.ptr(shift), 1 if relu else 0,
                                           _C.ptr(y), b, cin, cout, h, w, ksize, int(cfg), _C.stream())
    else:
        rc = _C.lib().sassd_conv2d_fwd(_C.ptr(x), _C.ptr(w_packed), _C.ptr(scale), _C.ptr(shift), 1 if relu else 0,
                                       _C.ptr(y), b, cin, cout, h, w, ksize, _C.stream())
    _C.check(rc, "sassd_conv2d_fwd")
    return y


def conv1x1_narrow_supported(cin, cout):
    return bool(_C.lib().sassd_conv1x1_narrow_supported(int(cin), int(cout)))


def conv1x1_narrow_pack_weight(w):
    """w [Cout,Cin,1,1] -> wT [Cin][CO], CO = Cout padded to the kernel's channel count with zeros (the layout
    sassd_conv1x1_narrow_fwd streams whole scalar weight rows from)."""
    cout, cin = w.shape[0], w.shape[1]
    co = _C.lib().sassd_conv1x1_narrow_pad(int(cout))
    wt = torch.zeros(cin, co, dtype=torch.float32, device=w.device)
    wt[:, :cout] = w.reshape(cout, cin).t().float()
    return wt


def conv1x1_narrow_fwd(x, wT, cout, scale=None, shift=None, relu=False, y=None):
    """1x1 conv with <= 32 output channels as an HBM stream (fused SSD head, part-sensitive head)."""
    _chk_cuda(x, wT, scale, shift)
    b, cin, h, w = x.shape
    if y is None:
        y = torch.empty(b, cout, h, w, dtype=torch.float32, device=x.device)
    _C.check(_C.lib().sassd_conv1x1_narrow_fwd(_C.ptr(x), _C.ptr(wT), _C.ptr(scale), _C.ptr(shift), 1 if relu else 0,
                                               _C.ptr(y), b, cin, cout, h, w, _C.stream()), "sassd_conv1x1_narrow_fwd")
    return y


def conv2d_wino_supported(cin, cout, h, w):
    return bool(_C.lib().sassd_conv2d_wino_supported(int(cin), int(cout), int(h), int(w)))


def conv2d_wino_pack_weight(w):
    """w [Cout,Cin,3,3] -> G g G^T packed in MFMA A-operand order (once per weight update)."""
    _chk_cuda(w)
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3
    L = _C.lib()
    n = L.sassd_conv2d_wino_packed_floats(cin, cout)
    if n == 0:
        raise ValueError("winograd conv needs Cin % 16 == 0")
    packed = torch.empty(n, dtype=torch.float32, device=w.device)
    _C.check(L.sassd_conv2d_wino_pack_weight(_C.ptr(w.contiguous()), cout, cin, _C.ptr(packed), _C.stream()),
             "sassd_conv2d_wino_pack_weight")
    return packed


def conv2d_wino_fwd(x, w_packed, cout, scale=None, shift=None, relu=False, y=None):
    """3x3 pad-1 conv through Winograd F(2x2,3x3) on the fp32 MFMA; same epilogue as conv2d_fwd."""
    _chk_cuda(x, w_packed)
    b, cin, h, w = x.shape
    if y is None:
        y = torch.empty(b, cout, h, w, dtype=torch.float32, device=x.device)
    _C.check(_C.lib().sassd_conv2d_wino_fwd(_C.ptr(x), _C.ptr(w_packed), _C.ptr(scale), _C.ptr(shift),
                                            1 if relu else 0, _C.ptr(y), b, cin, cout, h, w, _C.stream()),
             "sassd_conv2d_wino_fwd")
    return y


def conv2d_wino4_supported(cin, cout, h, w):
    return bool(_C.lib().sassd_conv2d_wino4_supported(int(cin), int(cout), int(h), int(w)))


def conv2d_wino4_pack_weight(w):
    """w [Cout,Cin,3,3] -> U = G g G^T of Winograd F(4x4,3x3) as [36][Cin][Cout] (once per weight update)."""
    _chk_cuda(w)
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3
    L = _C.lib()
    packed = torch.empty(L.sassd_conv2d_wino4_packed_floats(cin, cout), dtype=torch.float32, device=w.device)
    _C.check(L.sassd_conv2d_wino4_pack_weight(_C.ptr(w.contiguous()), cout, cin, _C.ptr(packed), _C.stream()),
             "sassd_conv2d_wino4_pack_weight")
    return packed


def conv2d_wino4_workspace(b, cin, cout, h, w, device):
    n = _C.lib().sassd_conv2d_wino4_workspace_bytes(b, cin, cout, h, w)
    return torch.empty(max(n, 256), dtype=torch.uint8, device=device)


def conv2d_wino4_fwd(x, w_packed, cout, scale=None, shift=None, relu=False, y=None, ws=None, cfg=None):
    """3x3 pad-1 conv through Winograd F(4x4,3x3): input transform, 36 GEMMs (fp32 products on the bf16 MFMA over exactly
    split operands; `cfg = wino4_cfg(1)` = the fp32 MFMA), output transform + epilogue."""
    _chk_cuda(x, w_packed)
    b, cin, h, w = x.shape
    if y is None:
        y = torch.empty(b, cout, h, w, dtype=torch.float32, device=x.device)
    L = _C.lib()
    wsb = L.sassd_conv2d_wino4_workspace_bytes(b, cin, cout, h, w)
    if ws is None or ws.numel() < wsb:
        ws = workspace("conv2d_wino4", wsb, x.device)
    _C.check(L.sassd_conv2d_wino4_fwd(_C.ptr(x), _C.ptr(w_packed), _C.ptr(scale), _C.ptr(shift), 1 if relu else 0,
                                      _C.ptr(y), b, cin, cout, h, w, DEFAULT_CFG["wino4"] if cfg is None else int(cfg),
                                      _C.ptr(ws), ws.numel(), _C.stream()),
             "sassd_conv2d_wino4_fwd")
    return y


def conv2d_wino4_chain_supported(cin, cout, h, w):
    return bool(_C.lib().sassd_conv2d_wino4_chain_supported(int(cin), int(cout), int(h), int(w)))


def conv2d_wino4_chain_workspace(b, cmax, h, w, device):
    n = _C.lib().sassd_conv2d_wino4_chain_workspace_bytes(b, cmax, h, w)
    if n == 0:
        raise ValueError("wino4 chain: unsupported shape")
    return torch.empty(n, dtype=torch.uint8, device=device)


def wino4_tile_map(indices, n_ptr, cap, batch, h, w, out=None):
    """Active-tile map of a sparse BEV input (sassd_wino4_tile_map); None when the shape has too many tiles."""
    n = _C.lib().sassd_wino4_tile_map_ints(int(batch), int(h), int(w))
    if n == 0:
        return None
    if out is None:
        out = torch.zeros(n, dtype=torch.int32, device=indices.device)
    _C.check(_C.lib().sassd_wino4_tile_map(_C.ptr(indices), _C.ptr(n_ptr), int(cap), int(batch), int(h), int(w), _C.ptr(out),
                                           _C.stream()), "sassd_wino4_tile_map")
    return out


def conv2d_wino4_chain(x, prev, w_packed, cin, cout, cmax, batch, h, w, scale, shift, relu, y, ws, cfg=None, tile_map=None,
                       prev_tile_map=None):
    """One layer of a chain of Winograd F(4x4,3x3) convolutions (sassd_conv2d_wino4_chain).  `x`: NCHW input map, or None
    to continue from the products the previous call left in `ws` (then prev = (scale, shift, relu) of that layer).
    `y`: NCHW output map, or None to leave the products in `ws` for the next call.  `tile_map` (wino4_tile_map): run on the
    active tiles of a sparse input only; the next call names the same map as `prev_tile_map`."""
    _chk_cuda(x, w_packed, scale, shift, y, ws)
    ps, pb, pr = prev if prev is not None else (None, None, False)
    _C.check(_C.lib().sassd_conv2d_wino4_chain(_C.ptr(x), 0 if x is not None else 1, _C.ptr(ps), _C.ptr(pb), 1 if pr else 0,
                                               _C.ptr(w_packed), _C.ptr(scale), _C.ptr(shift), 1 if relu else 0,
                                               _C.ptr(y), batch, cin, cout, cmax, h, w, _C.ptr(tile_map), _C.ptr(prev_tile_map),
                                               DEFAULT_CFG["wino4"] if cfg is None else int(cfg), _C.ptr(ws), ws.numel(),
                                               _C.stream()), "sassd_conv2d_wino4_chain")
    return y


def conv2d_wino4_pack_weight_narrow(w):
    """w [Cout <= 64, Cin, 3, 3] -> [36][Cin][64] (zero padded) for conv2d_wino4_chain_tail."""
    _chk_cuda(w)
    cout, cin = w.shape[0], w.shape[1]
    L = _C.lib()
    packed = torch.empty(L.sassd_conv2d_wino4_narrow_packed_floats(cin), dtype=torch.float32, device=w.device)
    _C.check(L.sassd_conv2d_wino4_pack_weight_narrow(_C.ptr(w.contiguous()), cout, cin, _C.ptr(packed), _C.stream()),
             "sassd_conv2d_wino4_pack_weight_narrow")
    return packed


def conv2d_wino4_chain_tail(prev, y_prev, w_packed64, cin, cout, cmax, batch, h, w, scale, shift, relu, y, ws, cfg=None,
                            prev_tile_map=None):
    """A narrow (<= 64 output channels) 3x3 layer on the products the previous chain call left in `ws`; `y_prev` (optional)
    receives that previous layer's NCHW activation map (prev = its (scale, shift, relu))."""
    _chk_cuda(w_packed64, scale, shift, y, ws, y_prev)
    ps, pb, pr = prev
    _C.check(_C.lib().sassd_conv2d_wino4_chain_tail(_C.ptr(ps), _C.ptr(pb), 1 if pr else 0, _C.ptr(y_prev), _C.ptr(w_packed64),
                                                    _C.ptr(scale), _C.ptr(shift), 1 if relu else 0, _C.ptr(y), batch, cin, cout,
                                                    cmax, h, w, _C.ptr(prev_tile_map),
                                                    DEFAULT_CFG["wino4"] if cfg is None else int(cfg), _C.ptr(ws), ws.numel(),
                                                    _C.stream()), "sassd_conv2d_wino4_chain_tail")
    return y


def conv1x1_gemm_supported(cin, cout, h, w):
    return bool(_C.lib().sassd_conv1x1_gemm_supported(int(cin), int(cout), int(h), int(w)))


def conv1x1_gemm_pack_weight(w):
    """w [Cout,Cin,1,1] -> [Cin][Cout] (K-major operand of the Winograd GEMM kernel)."""
    _chk_cuda(w)
    cout, cin = w.shape[0], w.shape[1]
    packed = torch.empty(cin * cout, dtype=torch.float32, device=w.device)
    _C.check(_C.lib().sassd_conv1x1_gemm_pack_weight(_C.ptr(w.contiguous()), cout, cin, _C.ptr(packed), _C.stream()),
             "sassd_conv1x1_gemm_pack_weight")
    return packed


def conv1x1_gemm_fwd(x, w_packed, cout, scale=None, shift=None, relu=False, y=None, cfg=None):
    _chk_cuda(x, w_packed, scale, shift)
    b, cin, h, w = x.shape
    if y is None:
        y = torch.empty(b, cout, h, w, dtype=torch.float32, device=x.device)
    _C.check(_C.lib().sassd_conv1x1_gemm_fwd(_C.ptr(x), _C.ptr(w_packed), _C.ptr(scale), _C.ptr(shift), 1 if relu else 0,
                                             _C.ptr(y), b, cin, cout, h, w,
                                             DEFAULT_CFG["wino4"] if cfg is None else int(cfg), _C.stream()),
             "sassd_conv1x1_gemm_fwd")
    return y


def conv2d_bf16_supported(cin, cout, h, w):
    return bool(_C.lib().sassd_conv2d_bf16_supported(cin, cout, h, w))


def conv2d_bf16_pack_weight(w):
    """w [Cout,Cin,3,3] fp32 -> bf16 [tap][Cin/8][Cout][8] (once per weight update)."""
    _chk_cuda(w)
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3
    L = _C.lib()
    packed = torch.empty(L.sassd_conv2d_bf16_packed_elems(cin, cout), dtype=torch.int16, device=w.device)
    _C.check(L.sassd_conv2d_bf16_pack_weight(_C.ptr(w.contiguous()), cout, cin, _C.ptr(packed), _C.stream()),
             "sassd_conv2d_bf16_pack_weight")
    return packed


def conv2d_bf16_fwd(x, w_packed, cout, shift=None, y=None, in_affine=None, cfg=0):
    """3x3 pad-1 conv, bf16 MFMA operands / fp32 accumulation, NCHW fp32 in and out (+ optional per-channel bias).
    in_affine [3, Cin] (bn2d_stats): the input is relu(batchnorm(x)), applied by the kernel's loader waves.
    cfg: per-call ablation / forced-geometry word (include/sassd.h, sassd_conv2d_bf16_fwd_cfg), 0 in production."""
    _chk_cuda(x, w_packed, in_affine)
    b, cin, h, w = x.shape
    if y is None:
        y = torch.empty(b, cout, h, w, dtype=torch.float32, device=x.device)
    if cfg:
        assert in_affine is None
        _C.check(_C.lib().sassd_conv2d_bf16_fwd_cfg(_C.ptr(x), _C.ptr(w_packed), _C.ptr(shift) if shift is not None else None,
                                                    _C.ptr(y), b, cin, cout, h, w, int(cfg), _C.stream()), "sassd_conv2d_bf16_fwd_cfg")
        return y
    if in_affine is not None:
        _C.check(_C.lib().sassd_conv2d_bf16_bnrelu_fwd(_C.ptr(x), _C.ptr(in_affine), _C.ptr(w_packed), _C.ptr(shift), _C.ptr(y),
                                                       b, cin, cout, h, w, _C.stream()), "sassd_conv2d_bf16_bnrelu_fwd")
        return y
    _C.check(_C.lib().sassd_conv2d_bf16_fwd(_C.ptr(x), _C.ptr(w_packed), _C.ptr(shift) if shift is not None else None,
                                            _C.ptr(y), b, cin, cout, h, w, _C.stream()), "sassd_conv2d_bf16_fwd")
    return y


def conv1x1_bf16_supported(cin, cout, hw):
    return bool(_C.lib().sassd_conv1x1_bf16_supported(cin, cout, hw))


def conv1x1_bf16_pack_weight(w, transposed=False):
    """w [Cout, Cin(,1,1)] fp32 -> bf16 MFMA A fragments of the 1x1 convolution; transposed=True packs the weights of the
    DATA GRADIENT (the same array read as [Cin, Cout]^T: output channels = the layer's input channels)."""
    _chk_cuda(w)
    w2 = w.reshape(w.shape[0], w.shape[1]).contiguous()
    cout, cin = (w2.shape[1], w2.shape[0]) if transposed else (w2.shape[0], w2.shape[1])
    L = _C.lib()
    packed = torch.empty(L.sassd_conv1x1_bf16_packed_elems(cin, cout), dtype=torch.int16, device=w.device)
    _C.check(L.sassd_conv1x1_bf16_pack_weight(_C.ptr(w2), cout, cin, 1 if transposed else 0, _C.ptr(packed), _C.stream()),
             "sassd_conv1x1_bf16_pack_weight")
    return packed


def conv1x1_bf16_fwd(x, w_packed, cout, shift=None, y=None):
    """1x1 conv over NCHW fp32, bf16 MFMA operands / fp32 accumulation (+ optional per-channel bias)."""
    _chk_cuda(x, w_packed)
    b, cin, h, w = x.shape
    if y is None:
        y = torch.empty(b, cout, h, w, dtype=torch.float32, device=x.device)
    _C.check(_C.lib().sassd_conv1x1_bf16_fwd(_C.ptr(x), _C.ptr(w_packed), _C.ptr(shift) if shift is not None else None,
                                             _C.ptr(y), b, cin, cout, h * w, _C.stream()), "sassd_conv1x1_bf16_fwd")
    return y


def conv2d_bwd_weight(x, dy, ksize, dw=None, accumulate=False, bf16=False, x_affine=None):
    """x [B,Cin,H,W], dy [B,Cout,H,W] -> dw [Cout,Cin,k,k] on the split-K MFMA kernel: fp32 operands on the fp32 pipe,
    or (bf16=True, W even) operands rounded to bf16 on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
    x_affine [3, Cin] (bn2d_stats; bf16 3x3 only): the X operand is relu(batchnorm(x)), applied inside the kernel."""
    _chk_cuda(x, dy, x_affine)
    b, cin, h, w = x.shape
    cout = dy.shape[1]
    L = _C.lib()
    if dw is None:
        dw = torch.empty(cout, cin, ksize, ksize, dtype=torch.float32, device=x.device)
    wsb = L.sassd_conv2d_wgrad_workspace_bytes(b, cin, cout, h, w, ksize)
    ws = workspace("conv2d_wgrad", wsb, x.device)
    if x_affine is not None:
        assert bf16 and w % 2 == 0 and ksize == 3
        _C.check(L.sassd_conv2d_bwd_weight_bf16_bnrelu(_C.ptr(x), _C.ptr(x_affine), _C.ptr(dy), _C.ptr(dw), b, cin, cout, h, w,
                                                       ksize, 1 if accumulate else 0, _C.ptr(ws), wsb, _C.stream()),
                 "sassd_conv2d_bwd_weight_bf16_bnrelu")
        return dw
    fn = L.sassd_conv2d_bwd_weight_bf16 if bf16 and w % 2 == 0 else L.sassd_conv2d_bwd_weight
    _C.check(fn(_C.ptr(x), _C.ptr(dy), _C.ptr(dw), b, cin, cout, h, w, ksize, 1 if accumulate else 0, _C.ptr(ws), wsb,
                _C.stream()), "sassd_conv2d_bwd_weight")
    return dw


# --------------------------------------------------------------------------------------------------
def anchor_mask(coors, row_begin_ptr, row_end_ptr, h0, w0, anchors_bv, voxel_size, coors_range, area_threshold,
                mask=None):
    _chk_cuda(coors, row_begin_ptr, row_end_ptr, anchors_bv, mask)      # raw pointers: row-major [A,4] fp32 or nothing
    assert anchors_bv.dtype == torch.float32 and anchors_bv.shape[1] == 4 and coors.dtype == torch.int32
    dev = coors.device
    L = _C.lib()
    n = anchors_bv.shape[0]
    if mask is None:
        mask = torch.empty(n, dtype=torch.uint8, device=dev)
    vs, cr = _f32(voxel_size), _f32(coors_range)
    wsb = L.sassd_anchor_mask_workspace_bytes(h0, w0)
    ws = workspace("anchor_mask", wsb, dev)
    rc = L.sassd_anchor_mask(_C.ptr(coors), _C.ptr(row_begin_ptr), _C.ptr(row_end_ptr), h0, w0, _C.ptr(anchors_bv),
                             n, vs.ctypes.data, cr.ctypes.data, float(area_threshold), _C.ptr(mask), _C.ptr(ws), wsb,
                             _C.stream())
    _C.check(rc, "sassd_anchor_mask")
    return mask


def anchor_mask_batch(coors, row_offsets, batch, h0, w0, anchors_bv, voxel_size, coors_range, area_threshold, mask):
    """anchors_mask of `batch` samples in one launch sequence (row_offsets: device int32 [batch+1])."""
    _chk_cuda(coors, row_offsets, anchors_bv, mask)
    assert anchors_bv.dtype == torch.float32 and anchors_bv.shape[1] == 4 and coors.dtype == torch.int32
    L = _C.lib()
    n = anchors_bv.shape[0]
    vs, cr = _f32(voxel_size), _f32(coors_range)
    wsb = L.sassd_anchor_mask_workspace_bytes(h0, w0) * batch
    ws = workspace("anchor_mask_batch", wsb, coors.device)
    rc = L.sassd_anchor_mask_batch(_C.ptr(coors), _C.ptr(row_offsets), batch, h0, w0, _C.ptr(anchors_bv), n,
                                   vs.ctypes.data, cr.ctypes.data, float(area_threshold), _C.ptr(mask), _C.ptr(ws), wsb,
                                   _C.stream())
    _C.check(rc, "sassd_anchor_mask_batch")
    return mask


def decode_filter(box, cls, dirp, batch_stride, batch, num_class, anchors_per_loc, h, w, anchors, mask, thr, cap_k,
                  out=None, status=None):
    dev = box.device
    L = _C.lib()
    out = out or {}
    guided = out.get("guided")
    if guided is None:
        guided = torch.empty(batch, cap_k, 7, dtype=torch.float32, device=dev)
    labels = out.get("labels")
    if labels is None:
        labels = torch.empty(batch, cap_k, dtype=torch.int32, device=dev)
    scores = out.get("scores")
    if scores is None:
        scores = torch.empty(batch, cap_k, dtype=torch.float32, device=dev)
    counts = out.get("counts")
    if counts is None:
        counts = torch.zeros(batch, dtype=torch.int32, device=dev)
    atot = num_class * h * w * anchors_per_loc
    wsb = L.sassd_decode_filter_workspace_bytes(batch, atot)
    ws = workspace("decode_filter", wsb, dev)
    rc = L.sassd_decode_filter(_C.ptr(box), _C.ptr(cls), _C.ptr(dirp), int(batch_stride), batch, num_class,
                               anchors_per_loc, h, w, _C.ptr(anchors), _C.ptr(mask), float(thr), _C.ptr(guided),
                               _C.ptr(labels), _C.ptr(scores), _C.ptr(counts), cap_k, _C.ptr(status), _C.ptr(ws), wsb,
                               _C.stream())
    _C.check(rc, "sassd_decode_filter")
    return dict(guided=guided, labels=labels, scores=scores, counts=counts)


def pswarp_sample(feat, guided, counts, cap_k, grid_offsets, spatial_scale, logits=None):
    b, parts, h, w = feat.shape
    assert parts == 28, "PSWarp window is 4x7 (ssd_rotate_head.py:374)"
    if logits is None:
        logits = torch.zeros(b, cap_k, dtype=torch.float32, device=feat.device)
    rc = _C.lib().sassd_pswarp_sample(_C.ptr(feat), b, h, w, _C.ptr(guided), _C.ptr(counts), cap_k,
                                      float(grid_offsets[0]), float(grid_offsets[1]), float(spatial_scale),
                                      _C.ptr(logits), _C.stream())
    _C.check(rc, "sassd_pswarp_sample")
    return logits


def pswarp_sample_bwd(feat, guided, counts, cap_k, grid_offsets, spatial_scale, dlogits):
    """-> (dfeat [B,28,H,W], dguided [B,capK,7])."""
    b, parts, h, w = feat.shape
    dfeat = torch.zeros_like(feat)
    dg = torch.zeros(b, cap_k, 7, dtype=torch.float32, device=feat.device)
    rc = _C.lib().sassd_pswarp_sample_bwd(_C.ptr(feat), b, h, w, _C.ptr(guided), _C.ptr(counts), cap_k,
                                          float(grid_offsets[0]), float(grid_offsets[1]), float(spatial_scale),
                                          _C.ptr(dlogits), _C.ptr(dfeat), _C.ptr(dg), _C.stream())
    _C.check(rc, "sassd_pswarp_sample_bwd")
    return dfeat, dg


def rescore_nms(guided, logits, labels, counts, score_thr, iou_thr, cap_d, out=None, status=None):
    dev = guided.device
    L = _C.lib()
    b, cap_k, _ = guided.shape
    out = out or {}
    boxes = out.get("boxes")
    if boxes is None:
        boxes = torch.empty(b, cap_d, 7, dtype=torch.float32, device=dev)
    scores = out.get("scores")
    if scores is None:
        scores = torch.empty(b, cap_d, dtype=torch.float32, device=dev)
    olabels = out.get("labels")
    if olabels is None:
        olabels = torch.empty(b, cap_d, dtype=torch.int32, device=dev)
    ocounts = out.get("counts")
    if ocounts is None:
        ocounts = torch.zeros(b, dtype=torch.int32, device=dev)
    wsb = L.sassd_rescore_nms_workspace_bytes(b, cap_k)
    ws = workspace("rescore_nms", wsb, dev)
    rc = L.sassd_rescore_nms(_C.ptr(guided), _C.ptr(logits), _C.ptr(labels), _C.ptr(counts), b, cap_k,
                             float(score_thr), float(iou_thr), _C.ptr(boxes), _C.ptr(scores), _C.ptr(olabels),
                             _C.ptr(ocounts), cap_d, _C.ptr(status), _C.ptr(ws), wsb, _C.stream())
    _C.check(rc, "sassd_rescore_nms")
    return dict(boxes=boxes, scores=scores, labels=olabels, counts=ocounts)


# --------------------------------------------------------------------------------------------------
def boxes_overlap_bev(a, b, out=None):
    _chk_cuda(a, b)
    if out is None:
        out = torch.zeros(a.shape[0], b.shape[0], dtype=torch.float32, device=a.device)
    rc = _C.lib().sassd_boxes_overlap_bev(_C.ptr(a), a.shape[0], _C.ptr(b), b.shape[0], _C.ptr(out), _C.stream())
    _C.check(rc, "sassd_boxes_overlap_bev")
    return out


def boxes_iou_bev(a, b, out=None):
    _chk_cuda(a, b)
    if out is None:
        out = torch.zeros(a.shape[0], b.shape[0], dtype=torch.float32, device=a.device)
    rc = _C.lib().sassd_boxes_iou_bev(_C.ptr(a), a.shape[0], _C.ptr(b), b.shape[0], _C.ptr(out), _C.stream())
    _C.check(rc, "sassd_boxes_iou_bev")
    return out


def nms_gpu(boxes_sorted, thresh, normal=False):
    """boxes [N,5] sorted by descending score -> (keep int64 device [N], num_keep int32 device [1]).  normal=True:
    axis-aligned IoU (iou3d_cuda.nms_normal_gpu) instead of the rotated BEV IoU."""
    _chk_cuda(boxes_sorted)
    n = boxes_sorted.shape[0]
    dev = boxes_sorted.device
    keep = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    num = torch.zeros(1, dtype=torch.int32, device=dev)
    L = _C.lib()
    wsb = L.sassd_nms_workspace_bytes(n)
    ws = workspace("nms", wsb, dev)
    fn = L.sassd_nms_normal_gpu if normal else L.sassd_nms_gpu
    rc = fn(_C.ptr(boxes_sorted), n, float(thresh), _C.ptr(keep), _C.ptr(num), _C.ptr(ws), wsb, _C.stream())
    _C.check(rc, "sassd_nms_normal_gpu" if normal else "sassd_nms_gpu")
    return keep, num


def three_nn(unknown, known):
    """unknown [N,4], known [M,4] (b,x,y,z) -> (dist2 [N,3] squared distances, idx [N,3] int32)."""
    _chk_cuda(unknown, known)
    n, m = unknown.shape[0], known.shape[0]
    d = torch.empty(n, 3, dtype=torch.float32, device=unknown.device)
    i = torch.empty(n, 3, dtype=torch.int32, device=unknown.device)
    _C.check(_C.lib().sassd_three_nn(n, m, _C.ptr(unknown), _C.ptr(known), _C.ptr(d), _C.ptr(i), _C.stream()),
             "sassd_three_nn")
    return d, i


def three_nn_binned(unknown, known, xy_range, cell, batch_size):
    """three_nn through a BEV counting-sort of `known` (exact, bit-identical results): xy_range = (x0, y0, x1, y1) of
    the scene, `cell` the bin edge in metres, batch indices 0..batch_size-1."""
    _chk_cuda(unknown, known)
    n, m = unknown.shape[0], known.shape[0]
    dev = unknown.device
    d2 = torch.empty(n, 3, dtype=torch.float32, device=dev)
    idx = torch.empty(n, 3, dtype=torch.int32, device=dev)
    x0, y0, x1, y1 = [float(v) for v in xy_range]
    nx, ny = max(1, int(np.ceil((x1 - x0) / cell))), max(1, int(np.ceil((y1 - y0) / cell)))
    L = _C.lib()
    wsb = L.sassd_three_nn_binned_workspace_bytes(m, nx, ny, int(batch_size))
    if wsb == 0:
        raise ValueError("three_nn_binned: grid %dx%dx%d too large" % (nx, ny, batch_size))
    ws = workspace("three_nn_binned", wsb, dev)
    _C.check(L.sassd_three_nn_binned(n, m, _C.ptr(unknown), _C.ptr(known), x0, y0, float(cell), nx, ny,
                                     int(batch_size), _C.ptr(d2), _C.ptr(idx), _C.ptr(ws), wsb, _C.stream()),
             "sassd_three_nn_binned")
    return d2, idx


def three_interpolate(points, idx, weight):
    _chk_cuda(points, idx, weight)
    m, c = points.shape
    n = idx.shape[0]
    out = torch.empty(n, c, dtype=torch.float32, device=points.device)
    _C.check(_C.lib().sassd_three_interpolate(c, m, n, _C.ptr(points), _C.ptr(idx), _C.ptr(weight), _C.ptr(out),
                                              _C.stream()), "sassd_three_interpolate")
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    _chk_cuda(grad_out, idx, weight)
    n, c = grad_out.shape
    gp = torch.zeros(m, c, dtype=torch.float32, device=grad_out.device)
    _C.check(_C.lib().sassd_three_interpolate_grad(c, n, m, _C.ptr(grad_out), _C.ptr(idx), _C.ptr(weight), _C.ptr(gp),
                                                   _C.stream()), "sassd_three_interpolate_grad")
    return gp


def pts_in_boxes3d(pts, boxes3d):
    """pts [N,3], boxes3d [M,7] -> (pts_in_flag [M,N] int32, reg_target [N,3]) on the device."""
    _chk_cuda(pts, boxes3d)
    n, m = pts.shape[0], boxes3d.shape[0]
    flag = torch.zeros(m, n, dtype=torch.int32, device=pts.device)
    reg = torch.zeros(n, 3, dtype=torch.float32, device=pts.device)
    _C.check(_C.lib().sassd_pts_in_boxes3d(_C.ptr(pts), n, _C.ptr(boxes3d), m, _C.ptr(flag), _C.ptr(reg),
                                           _C.stream()), "sassd_pts_in_boxes3d")
    return flag, reg


def mfma_probe(a32, b32, a16, b16, ksteps):
    dev = a32.device
    d32 = torch.zeros(32, 32, dtype=torch.float32, device=dev)
    d16 = torch.zeros(16, 16, dtype=torch.float32, device=dev)
    rc = _C.lib().sassd_mfma_probe(_C.ptr(a32), _C.ptr(b32), _C.ptr(d32), _C.ptr(a16), _C.ptr(b16), _C.ptr(d16),
                                   ksteps, _C.stream())
    _C.check(rc, "sassd_mfma_probe")
    return d32, d16


def grad_sumsq(grad_flat, out=None):
    """sum(grad^2) of a flat fp32 buffer -> device float [1] (no host sync)."""
    _chk_cuda(grad_flat)
    out = torch.empty(1, dtype=torch.float32, device=grad_flat.device) if out is None else out
    _C.check(_C.lib().sassd_grad_sumsq(_C.ptr(grad_flat), grad_flat.numel(), _C.ptr(out), _C.stream()),
             "sassd_grad_sumsq")
    return out


def adam_step(param, grad, exp_avg, exp_avg_sq, sumsq, lr, beta1, beta2, eps, weight_decay, step, max_norm=0.0,
              grad_scale=1.0):
    """In-place fused clip + decoupled weight decay + Adam update over flat fp32 buffers."""
    _chk_cuda(param, grad, exp_avg, exp_avg_sq)
    _C.check(_C.lib().sassd_adam_step(_C.ptr(param), _C.ptr(grad), _C.ptr(exp_avg), _C.ptr(exp_avg_sq),
                                      param.numel(), _C.ptr(sumsq), lr, beta1, beta2, eps, weight_decay, int(step),
                                      max_norm, grad_scale, _C.stream()), "sassd_adam_step")


def rotate_iou_eval(boxes, query_boxes, criterion=-1):
    """boxes [N,5], query_boxes [K,5] (cx, cy, x_dim, y_dim, angle) device tensors -> [N,K] overlap matrix of the KITTI
    evaluation (criterion -1 IoU, 0 / 1 intersection over the query's / the box's area)."""
    _chk_cuda(boxes, query_boxes)
    n, k = boxes.shape[0], query_boxes.shape[0]
    out = torch.zeros(n, k, dtype=torch.float32, device=boxes.device)
    _C.check(_C.lib().sassd_rotate_iou_eval(_C.ptr(boxes), n, _C.ptr(query_boxes), k, int(criterion), _C.ptr(out),
                                            _C.stream()), "sassd_rotate_iou_eval")
    return out


# ---- fused training targets / RPN loss (train_loss.hip) ----------------------------------------------------------------
_gt_offset_cache = {}


def gt_offsets(counts, device):
    """[B+1] int32 prefix of the per-sample ground-truth counts on the device, cached per (counts, device): the counts
    repeat from step to step, and a cache hit costs no upload."""
    key = (tuple(int(c) for c in counts), str(device))
    t = _gt_offset_cache.get(key)
    if t is None:
        if len(_gt_offset_cache) > 4096:
            _gt_offset_cache.clear()
        off = np.zeros(len(key[0]) + 1, np.int32)
        off[1:] = np.cumsum(key[0])
        t = torch.from_numpy(off).to(device)
        _gt_offset_cache[key] = t
    return t


def assign_targets(anchors, anchor_mask, gt_boxes, gt_classes, gt_ok, gt_off, matched, unmatched, labels, targets,
                   num_pos, zero_num_pos=True, out_stride=None, best=None, overlaps=None, overlap_offsets=None):
    """sassd_assign_targets.  anchors [A,7] or [B,A,7]; anchor_mask [B,A] (bool / uint8) or None; gt_boxes [T,7],
    gt_classes [T] int64 or None, gt_ok [T] (bool / uint8) or None, gt_off [B+1] int32 (gt_offsets()).  Writes labels
    (int64), targets (fp32, 7 per anchor) and optionally best overlap through the given (possibly strided-view) output
    tensors: element (b, a) lands at b * out_stride + a from the tensor's data pointer."""
    per_sample = anchors.dim() == 3
    a = anchors.shape[-2]
    b = gt_off.shape[0] - 1
    t = int(gt_boxes.shape[0]) if gt_boxes is not None else 0
    _chk_cuda(anchors, anchor_mask, gt_boxes, gt_classes, gt_ok, gt_off, overlaps, overlap_offsets)
    L = _C.lib()
    wsb = L.sassd_assign_targets_workspace_bytes(b, a, t)
    ws = workspace("assign_targets", wsb, anchors.device)
    _C.check(L.sassd_assign_targets(_C.ptr(anchors), 1 if per_sample else 0, _C.ptr(anchor_mask), a, b,
                                    _C.ptr(gt_boxes) if t else None, _C.ptr(gt_classes), _C.ptr(gt_ok),
                                    _C.ptr(gt_off), t, _C.ptr(overlaps), _C.ptr(overlap_offsets), float(matched),
                                    float(unmatched), _C.ptr(labels), _C.ptr(targets), _C.ptr(best),
                                    int(out_stride if out_stride is not None else a), _C.ptr(num_pos),
                                    1 if zero_num_pos else 0, _C.ptr(ws), wsb, _C.stream()), "sassd_assign_targets")


def rpn_loss(box_preds, cls_preds, dir_preds, labels, targets, anchors, num_pos):
    """sassd_rpn_loss -> (loss_sums [3] = (loc, cls, dir), grad_box, grad_cls, grad_dir) of the UNSCALED sums."""
    _chk_cuda(box_preds, cls_preds, dir_preds, labels, targets, anchors, num_pos)
    b, a = labels.shape
    nc = cls_preds.shape[-1]
    dev = box_preds.device
    gbox, gcls = torch.empty_like(box_preds), torch.empty_like(cls_preds)
    gdir = torch.empty_like(dir_preds) if dir_preds is not None else None
    sums = torch.empty(3, dtype=torch.float32, device=dev)
    L = _C.lib()
    wsb = L.sassd_rpn_loss_workspace_bytes(b, a)
    ws = workspace("rpn_loss", wsb, dev)
    _C.check(L.sassd_rpn_loss(_C.ptr(box_preds), _C.ptr(cls_preds), _C.ptr(dir_preds), nc, _C.ptr(labels),
                              _C.ptr(targets), _C.ptr(anchors), 1 if anchors.dim() == 3 else 0, _C.ptr(num_pos), a, b,
                              _C.ptr(gbox), _C.ptr(gcls), _C.ptr(gdir), _C.ptr(sums), _C.ptr(ws), wsb, _C.stream()),
             "sassd_rpn_loss")
    return sums, gbox, gcls, gdir


def gather_pack(src_flat, index_map, dst):
    """dst[i] = src_flat[index_map[i]] (0 where the map is negative); dst fp32, or int16 holding bf16 bits."""
    _chk_cuda(src_flat, index_map, dst)
    assert index_map.dtype == torch.int32 and index_map.numel() == dst.numel()
    _C.check(_C.lib().sassd_gather_pack(_C.ptr(src_flat), _C.ptr(index_map), _C.ptr(dst), dst.numel(),
                                        0 if dst.dtype == torch.float32 else 1, _C.stream()), "sassd_gather_pack")
    return dst


def guided_select(cls_preds, anchor_mask, score_thr, cap, overflow):
    """sassd_guided_select: cls_preds [B,A,NC], anchor_mask [B,A] (bool / uint8) or None -> (sel [B,cap] int64 ascending
    anchor indices, padded with sel[b][p] = p; counts [B] int32).  `overflow` is a persistent [1] int32 device flag."""
    _chk_cuda(cls_preds, anchor_mask, overflow)
    b, a, nc = cls_preds.shape
    dev = cls_preds.device
    sel = torch.empty(b, cap, dtype=torch.int64, device=dev)
    counts = torch.empty(b, dtype=torch.int32, device=dev)
    L = _C.lib()
    wsb = L.sassd_guided_select_workspace_bytes(b, a)
    ws = workspace("guided_select", wsb, dev)
    _C.check(L.sassd_guided_select(_C.ptr(cls_preds), _C.ptr(anchor_mask), a, b, nc, float(score_thr), int(cap),
                                   _C.ptr(sel), _C.ptr(counts), _C.ptr(overflow), _C.ptr(ws), wsb, _C.stream()),
             "sassd_guided_select")
    return sel, counts


# ---- fused BatchNorm1d + ReLU of the sparse blocks (bn.hip) -----------------------------------------------------------
def _bn_workspace(dev):
    return workspace("bn_relu", _C.lib().sassd_bn_relu_workspace_bytes(256), dev)


def bn_relu_supported(n, c):
    return n >= 1 and 4 <= c <= 256 and c % 4 == 0 and 256 % c == 0


def bn_relu_fwd(x, gamma, beta, running_mean, running_var, momentum, eps):
    """-> (y, save_mean, save_invstd); running statistics updated in place (pass None for both to skip)."""
    _chk_cuda(x, gamma, beta, running_mean, running_var)
    n, c = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    invstd = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = _bn_workspace(x.device)
    _C.check(_C.lib().sassd_bn_relu_fwd(_C.ptr(x), n, c, _C.ptr(gamma), _C.ptr(beta), _C.ptr(running_mean),
                                        _C.ptr(running_var), float(momentum), float(eps), _C.ptr(y), _C.ptr(mean),
                                        _C.ptr(invstd), _C.ptr(ws), ws.numel(), _C.stream()), "sassd_bn_relu_fwd")
    return y, mean, invstd


def bn_relu_bwd(x, dy, gamma, beta, mean, invstd):
    """-> (dx, dgamma, dbeta)."""
    _chk_cuda(x, dy, gamma, beta, mean, invstd)
    n, c = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty(c, dtype=torch.float32, device=x.device)
    db = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = _bn_workspace(x.device)
    _C.check(_C.lib().sassd_bn_relu_bwd(_C.ptr(x), _C.ptr(dy), n, c, _C.ptr(gamma), _C.ptr(beta), _C.ptr(mean),
                                        _C.ptr(invstd), _C.ptr(dx), _C.ptr(dg), _C.ptr(db), _C.ptr(ws), ws.numel(),
                                        _C.stream()), "sassd_bn_relu_bwd")
    return dx, dg, db


# ---- fused BatchNorm2d + ReLU over NCHW maps (bn2d.hip) -----------------------------------------------------------------
def bn2d_relu_supported(x):
    return x.dim() == 4 and x.dtype == torch.float32 and x.is_cuda and (x.shape[2] * x.shape[3]) % 4 == 0


def bn2d_stats(x, gamma, beta, running_mean, running_var, momentum, eps):
    """Batch statistics of x [B,C,H,W] only -> (save_mean, save_invstd, affine [3, C] = mean | invstd * gamma | beta); running
    statistics updated in place.  For consumers that apply BatchNorm + ReLU themselves (conv2d_bf16_fwd(in_affine=...))."""
    _chk_cuda(x, gamma, beta, running_mean, running_var)
    b, c, h, w = x.shape
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    invstd = torch.empty(c, dtype=torch.float32, device=x.device)
    aff = torch.empty(3, c, dtype=torch.float32, device=x.device)
    L = _C.lib()
    wsb = L.sassd_bn2d_relu_workspace_bytes(c)
    ws = workspace("bn2d_relu", wsb, x.device)
    _C.check(L.sassd_bn2d_stats(_C.ptr(x), b, c, h * w, _C.ptr(gamma), _C.ptr(beta), _C.ptr(running_mean),
                                _C.ptr(running_var), float(momentum), float(eps), _C.ptr(mean), _C.ptr(invstd), _C.ptr(aff),
                                _C.ptr(ws), wsb, _C.stream()), "sassd_bn2d_stats")
    return mean, invstd, aff


def bn2d_relu_fwd(x, gamma, beta, running_mean, running_var, momentum, eps):
    """x [B,C,H,W] -> (y, save_mean, save_invstd); running statistics updated in place (None for both to skip)."""
    _chk_cuda(x, gamma, beta, running_mean, running_var)
    b, c, h, w = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    invstd = torch.empty(c, dtype=torch.float32, device=x.device)
    L = _C.lib()
    wsb = L.sassd_bn2d_relu_workspace_bytes(c)
    ws = workspace("bn2d_relu", wsb, x.device)
    _C.check(L.sassd_bn2d_relu_fwd(_C.ptr(x), b, c, h * w, _C.ptr(gamma), _C.ptr(beta), _C.ptr(running_mean),
                                   _C.ptr(running_var), float(momentum), float(eps), _C.ptr(y), _C.ptr(mean),
                                   _C.ptr(invstd), _C.ptr(ws), wsb, _C.stream()), "sassd_bn2d_relu_fwd")
    return y, mean, invstd


def bn2d_relu_bwd(x, dy, gamma, beta, mean, invstd):
    """-> (dx, dgamma, dbeta)."""
    _chk_cuda(x, dy, gamma, beta, mean, invstd)
    b, c, h, w = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty(c, dtype=torch.float32, device=x.device)
    db = torch.empty(c, dtype=torch.float32, device=x.device)
    L = _C.lib()
    wsb = L.sassd_bn2d_relu_workspace_bytes(c)
    ws = workspace("bn2d_relu", wsb, x.device)
    _C.check(L.sassd_bn2d_relu_bwd(_C.ptr(x), _C.ptr(dy), b, c, h * w, _C.ptr(gamma), _C.ptr(beta), _C.ptr(mean),
                                   _C.ptr(invstd), _C.ptr(dx), _C.ptr(dg), _C.ptr(db), _C.ptr(ws), wsb, _C.stream()),
             "sassd_bn2d_relu_bwd")
    return dx, dg, db


# ---- fused auxiliary head of SpMiddleFHD in training (aux_head.hip) ----------------------------------------------------
def _ptr3(ts):
    import ctypes as C
    return (C.c_void_p * 3)(*[t.data_ptr() for t in ts])


def aux_prepare(voxel_feats, coors, indices, voxel_size, offset, gt_boxes, gt_off, batch_size):
    """-> (points [N,4], known [3 x [M_s,4]], label [N] uint8, target [N,3], npos [1] int32).  voxel_feats [N,>=3] fp32,
    coors [N,4] int32 (b,z,y,x), indices: the three middle tensors' [M_s,4] int32 coordinates, gt_boxes [T,7] (or None),
    gt_off [B+1] int32 (gt_offsets())."""
    import ctypes as C
    _chk_cuda(voxel_feats, coors, gt_boxes, gt_off, *indices)
    assert coors.dtype == torch.int32 and all(i.dtype == torch.int32 for i in indices)
    dev = voxel_feats.device
    n = voxel_feats.shape[0]
    points = torch.empty(n, 4, dtype=torch.float32, device=dev)
    known = [torch.empty(max(i.shape[0], 1), 4, dtype=torch.float32, device=dev) for i in indices]
    label = torch.empty(n, dtype=torch.uint8, device=dev)
    target = torch.empty(n, 3, dtype=torch.float32, device=dev)
    npos = torch.empty(1, dtype=torch.int32, device=dev)
    m = (C.c_int * 3)(*[int(i.shape[0]) for i in indices])
    vs, off = _f32(voxel_size), _f32(offset)
    _C.check(_C.lib().sassd_aux_prepare(_C.ptr(voxel_feats), int(voxel_feats.shape[1]), _C.ptr(coors), n, _ptr3(indices),
                                        m, vs.ctypes.data, off.ctypes.data, _C.ptr(gt_boxes), _C.ptr(gt_off),
                                        int(batch_size), _C.ptr(points), _ptr3(known), _C.ptr(label), _C.ptr(target),
                                        _C.ptr(npos), _C.stream()), "sassd_aux_prepare")
    return points, [k[:i.shape[0]] for k, i in zip(known, indices)], label, target, npos


def aux_head_fwd(feats, nn_idx, nn_d2, w1, w2, label, target, npos):
    """-> (loss_sums [2], wgt [N,9], h [N,64], out [N,4], gout [N,4])."""
    _chk_cuda(w1, w2, label, target, npos, *feats, *nn_idx, *nn_d2)
    n = label.shape[0]
    dev = label.device
    assert tuple(w1.shape) == (64, 160) and tuple(w2.shape) == (4, 64) and [f.shape[1] for f in feats] == [32, 64, 64]
    wgt = torch.empty(n, 9, dtype=torch.float32, device=dev)
    h = torch.empty(n, 64, dtype=torch.float32, device=dev)
    out = torch.empty(n, 4, dtype=torch.float32, device=dev)
    gout = torch.empty(n, 4, dtype=torch.float32, device=dev)
    sums = torch.empty(2, dtype=torch.float32, device=dev)
    L = _C.lib()
    wsb = L.sassd_aux_head_workspace_bytes(n)
    ws = workspace("aux_head", wsb, dev)
    _C.check(L.sassd_aux_head_fwd(n, _ptr3(feats), _ptr3(nn_idx), _ptr3(nn_d2), _C.ptr(w1), _C.ptr(w2), _C.ptr(label),
                                  _C.ptr(target), _C.ptr(npos), _C.ptr(wgt), _C.ptr(h), _C.ptr(out), _C.ptr(gout),
                                  _C.ptr(sums), _C.ptr(ws), wsb, _C.stream()), "sassd_aux_head_fwd")
    return sums, wgt, h, out, gout


def aux_head_bwd(feats, nn_idx, w1, w2, wgt, h, gout, grad_sums):
    """-> (grad_feats [3 x [M_s, C_s]], dw1 [64,160], dw2 [4,64])."""
    import ctypes as C
    _chk_cuda(w1, w2, wgt, h, gout, grad_sums, *feats, *nn_idx)
    n = h.shape[0]
    dev = h.device
    gf = [torch.empty_like(f) for f in feats]
    dw1 = torch.empty(64, 160, dtype=torch.float32, device=dev)
    dw2 = torch.empty(4, 64, dtype=torch.float32, device=dev)
    m = (C.c_int * 3)(*[int(f.shape[0]) for f in feats])
    L = _C.lib()
    wsb = L.sassd_aux_head_workspace_bytes(n)
    ws = workspace("aux_head", wsb, dev)
    _C.check(L.sassd_aux_head_bwd(n, _ptr3(feats), m, _ptr3(nn_idx), _C.ptr(w1), _C.ptr(w2), _C.ptr(wgt), _C.ptr(h),
                                  _C.ptr(gout), _C.ptr(grad_sums), _ptr3(gf), _C.ptr(dw1), _C.ptr(dw2), _C.ptr(ws), wsb,
                                  _C.stream()), "sassd_aux_head_bwd")
    return gf, dw1, dw2


# ---- guided-anchor / rescoring tail of the training step (train_heads.hip) ---------------------------------------------
def guided_decode_fwd(box_preds, dir_preds, anchors, sel, sel_count, gt_boxes, gt_off, gmax):
    """box_preds [B,A,7], dir_preds [B,A,2] or None, anchors [A,7] / [B,A,7], sel [B,cap] int64, sel_count [B] int32 ->
    (guided [B, gmax + cap, 7], counts [B] int32)."""
    _chk_cuda(box_preds, dir_preds, anchors, sel, sel_count, gt_boxes, gt_off)
    b, a, _ = box_preds.shape
    cap = sel.shape[1]
    guided = torch.empty(b, gmax + cap, 7, dtype=torch.float32, device=box_preds.device)
    counts = torch.empty(b, dtype=torch.int32, device=box_preds.device)
    _C.check(_C.lib().sassd_guided_decode_fwd(_C.ptr(box_preds), _C.ptr(dir_preds), _C.ptr(anchors),
                                              1 if anchors.dim() == 3 else 0, _C.ptr(sel), _C.ptr(sel_count),
                                              _C.ptr(gt_boxes), _C.ptr(gt_off), a, b, cap, int(gmax), _C.ptr(guided),
                                              _C.ptr(counts), _C.stream()), "sassd_guided_decode_fwd")
    return guided, counts


def guided_decode_bwd(box_preds, anchors, sel, sel_count, gt_off, gmax, dguided):
    _chk_cuda(box_preds, anchors, sel, sel_count, gt_off, dguided)
    b, a, _ = box_preds.shape
    dbox = torch.empty_like(box_preds)
    _C.check(_C.lib().sassd_guided_decode_bwd(_C.ptr(box_preds), _C.ptr(anchors), 1 if anchors.dim() == 3 else 0,
                                              _C.ptr(sel), _C.ptr(sel_count), _C.ptr(gt_off), a, b, sel.shape[1],
                                              int(gmax), _C.ptr(dguided), _C.ptr(dbox), _C.stream()),
             "sassd_guided_decode_bwd")
    return dbox


def boxes_iou3d_batch(boxes, counts, gt_boxes, gt_off, gmax, ov_off, total):
    """boxes [B,rows,7], counts [B] int32, gt_boxes [T,7], gt_off [B+1] int32, ov_off [B+1] int64 -> overlaps [total]."""
    _chk_cuda(boxes, counts, gt_boxes, gt_off, ov_off)
    b, rows, _ = boxes.shape
    ov = torch.empty(max(int(total), 1), dtype=torch.float32, device=boxes.device)
    _C.check(_C.lib().sassd_boxes_iou3d_batch(_C.ptr(boxes), _C.ptr(counts), b, rows, _C.ptr(gt_boxes), _C.ptr(gt_off),
                                              int(gmax), _C.ptr(ov_off), _C.ptr(ov), _C.stream()),
             "sassd_boxes_iou3d_batch")
    return ov


def focal_loss(logits, labels, num_pos):
    """logits [n] fp32, labels [n] int64, num_pos [nb] int32 -> (loss_sum [1], grad [n])."""
    _chk_cuda(logits, labels, num_pos)
    n = logits.numel()
    dev = logits.device
    out = torch.empty(1, dtype=torch.float32, device=dev)
    grad = torch.empty(n, dtype=torch.float32, device=dev)
    L = _C.lib()
    wsb = L.sassd_focal_loss_workspace_bytes(n)
    ws = workspace("focal_loss", wsb, dev)
    _C.check(L.sassd_focal_loss(_C.ptr(logits), _C.ptr(labels), n, _C.ptr(num_pos), num_pos.numel(), _C.ptr(out),
                                _C.ptr(grad), _C.ptr(ws), wsb, _C.stream()), "sassd_focal_loss")
    return out, grad
