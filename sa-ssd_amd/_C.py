"""ctypes binding of libsassd.so (C ABI declared in include/sassd.h).

There is NO fallback: if the HIP library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsassd.so")
_lib = None

OK, EINVAL, ENOSPC, EHIP = 0, -1, -2, -3
ST_VOXEL_OVERFLOW, ST_HASH_FULL, ST_BOX_OVERFLOW = 1, 2, 4


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into lib/libsassd.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if force:
        subprocess.check_call(cmd + ["clean"], stdout=subprocess.DEVNULL)
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
    if out.returncode != 0:
        raise RuntimeError("building libsassd.so failed")
    return LIB_PATH


_SZ = C.c_size_t
_P = C.c_void_p
_I = C.c_int
_F = C.c_float

_SIGS = {
    "sassd_version": (C.c_char_p, []),
    "sassd_last_hip_error": (_I, []),
    "sassd_last_hip_error_string": (C.c_char_p, []),
    "sassd_voxelize_workspace_bytes": (_SZ, [_I, _I]),
    "sassd_voxelize": (_I, [_P, _I, _I, _P, _P, _I, _I, _I, _P, _P, _I, _P, _P, _I, _P, _P, _I, _P, _P, _SZ, _P]),
    "sassd_voxelize_dev": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _P, _P, _I, _P, _P, _I, _P, _P, _I, _P, _P, _SZ, _P]),
    "sassd_voxel_mean": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "sassd_hash_bytes": (_SZ, [_I]),
    "sassd_hash_build": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _SZ, _P, _P]),
    "sassd_rulebook_subm": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _SZ, _P, _P]),
    "sassd_rulebook_conv_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "sassd_rulebook_conv": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _SZ, _P, _P, _I, _P, _P, _P, _SZ, _P]),
    "sassd_rulebook_pairs": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "sassd_rulebook_pyramid_workspace_bytes": (_SZ, [_I, _P, _I, _I, _I, _I]),
    "sassd_rulebook_pyramid": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P, _P, _SZ, _P]),
    "sassd_graph_begin": (_I, [_P]),
    "sassd_graph_end": (_I, [_P, _P]),
    "sassd_graph_launch": (_I, [_P, _P]),
    "sassd_graph_destroy": (_I, [_P]),
    "sassd_spconv_packed_floats": (_SZ, [_I, _I, _I]),
    "sassd_spconv_pack_weight": (_I, [_P, _I, _I, _I, _P, _P]),
    "sassd_spconv_fwd": (_I, [_P, _P, _P, _I, _P, _I, _I, _I, _P, _P, _I, _P, _I, _P]),
    "sassd_rulebook_transpose": (_I, [_P, _P, _I, _P, _I, _P]),
    "sassd_spconv_pack_weight_t": (_I, [_P, _I, _I, _I, _P, _P]),
    "sassd_spconv_bwd_data": (_I, [_P, _P, _P, _I, _P, _I, _I, _I, _P, _I, _P]),
    "sassd_spconv_bwd_weight_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "sassd_spconv_bwd_weight": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _I, _P, _SZ, _P]),
    "sassd_densify": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "sassd_conv2d_packed_floats": (_SZ, [_I, _I, _I]),
    "sassd_conv2d_pack_weight": (_I, [_P, _I, _I, _I, _P, _P]),
    "sassd_conv2d_fwd": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "sassd_conv2d_fwd_cfg": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sassd_anchor_mask_workspace_bytes": (_SZ, [_I, _I]),
    "sassd_anchor_mask": (_I, [_P, _P, _P, _I, _I, _P, _I, _P, _P, _F, _P, _P, _SZ, _P]),
    "sassd_anchor_mask_batch": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P, _F, _P, _P, _SZ, _P]),
    "sassd_decode_filter_workspace_bytes": (_SZ, [_I, _I]),
    "sassd_decode_filter": (_I, [_P, _P, _P, _SZ, _I, _I, _I, _I, _I, _P, _P, _F, _P, _P, _P, _P, _I, _P, _P, _SZ, _P]),
    "sassd_pswarp_sample": (_I, [_P, _I, _I, _I, _P, _P, _I, _F, _F, _F, _P, _P]),
    "sassd_pswarp_sample_bwd": (_I, [_P, _I, _I, _I, _P, _P, _I, _F, _F, _F, _P, _P, _P, _P]),
    "sassd_rescore_nms_workspace_bytes": (_SZ, [_I, _I]),
    "sassd_rescore_nms": (_I, [_P, _P, _P, _P, _I, _I, _F, _F, _P, _P, _P, _P, _I, _P, _P, _SZ, _P]),
    "sassd_boxes_overlap_bev": (_I, [_P, _I, _P, _I, _P, _P]),
    "sassd_boxes_iou_bev": (_I, [_P, _I, _P, _I, _P, _P]),
    "sassd_nms_workspace_bytes": (_SZ, [_I]),
    "sassd_nms_gpu": (_I, [_P, _I, _F, _P, _P, _P, _SZ, _P]),
    "sassd_nms_normal_gpu": (_I, [_P, _I, _F, _P, _P, _P, _SZ, _P]),
    "sassd_three_nn": (_I, [_I, _I, _P, _P, _P, _P, _P]),
    "sassd_three_interpolate": (_I, [_I, _I, _I, _P, _P, _P, _P, _P]),
    "sassd_three_interpolate_grad": (_I, [_I, _I, _I, _P, _P, _P, _P, _P]),
    "sassd_pts_in_boxes3d": (_I, [_P, _I, _P, _I, _P, _P, _P]),
    "sassd_three_nn_binned_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "sassd_three_nn_binned": (_I, [_I, _I, _P, _P, _F, _F, _F, _I, _I, _I, _P, _P, _P, _SZ, _P]),
    "sassd_conv2d_wino_supported": (_I, [_I, _I, _I, _I]),
    "sassd_conv2d_wino_packed_floats": (_SZ, [_I, _I]),
    "sassd_conv2d_wino_pack_weight": (_I, [_P, _I, _I, _P, _P]),
    "sassd_conv2d_wino_fwd": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "sassd_conv2d_wino4_supported": (_I, [_I, _I, _I, _I]),
    "sassd_conv2d_wino4_packed_floats": (_SZ, [_I, _I]),
    "sassd_conv2d_wino4_pack_weight": (_I, [_P, _I, _I, _P, _P]),
    "sassd_conv2d_wino4_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I]),
    "sassd_conv2d_wino4_fwd": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "sassd_conv1x1_gemm_supported": (_I, [_I, _I, _I, _I]),
    "sassd_conv1x1_gemm_pack_weight": (_I, [_P, _I, _I, _P, _P]),
    "sassd_conv1x1_narrow_supported": (_I, [_I, _I]),
    "sassd_conv1x1_narrow_pad": (_I, [_I]),
    "sassd_conv1x1_narrow_fwd": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "sassd_conv1x1_gemm_fwd": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "sassd_conv2d_bf16_supported": (_I, [_I, _I, _I, _I]),
    "sassd_conv2d_bf16_packed_elems": (_SZ, [_I, _I]),
    "sassd_conv2d_bf16_pack_weight": (_I, [_P, _I, _I, _P, _P]),
    "sassd_conv2d_bf16_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sassd_conv2d_bf16_fwd_cfg": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "sassd_conv2d_bf16_bnrelu_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sassd_conv1x1_bf16_supported": (_I, [_I, _I, _I]),
    "sassd_conv1x1_bf16_packed_elems": (_SZ, [_I, _I]),
    "sassd_conv1x1_bf16_pack_weight": (_I, [_P, _I, _I, _I, _P, _P]),
    "sassd_conv1x1_bf16_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "sassd_conv2d_wgrad_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I, _I]),
    "sassd_conv2d_bwd_weight": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "sassd_assign_targets_workspace_bytes": (_SZ, [_I, _I, _I]),
    "sassd_assign_targets": (_I, [_P, _I, _P, _I, _I, _P, _P, _P, _P, _I, _P, _P, _F, _F, _P, _P, _P, _SZ, _P, _I, _P,
                                  _SZ, _P]),
    "sassd_guided_select_workspace_bytes": (_SZ, [_I, _I]),
    "sassd_guided_select": (_I, [_P, _P, _I, _I, _I, _F, _I, _P, _P, _P, _P, _SZ, _P]),
    "sassd_rpn_loss_workspace_bytes": (_SZ, [_I, _I]),
    "sassd_rpn_loss": (_I, [_P, _P, _P, _I, _P, _P, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _SZ, _P]),
    "sassd_conv2d_bwd_weight_bf16": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "sassd_conv2d_bwd_weight_bf16_bnrelu": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "sassd_rotate_iou_eval": (_I, [_P, _I, _P, _I, _I, _P, _P]),
    "sassd_kitti_eval_statistics": (_I, [_P, C.c_int64, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, C.c_double, _P, _I, _I, _P,
                                         _P, _P]),
    "sassd_points_in_polytopes": (_I, [_P, _I, _I, _P, _I, _I, _P, _P]),
    "sassd_points_transform": (_I, [_P, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P]),
    "sassd_points_global_transform": (_I, [_P, _I, _I, _I, _F, _F, _F, _P]),
    "sassd_paste_objects": (_I, [_P, _P, _P, _I, C.c_int64, _P, _P, _P, _P]),
    "sassd_box_collision_test": (_I, [_P, _I, _P, _I, _I, _P]),
    "sassd_noise_per_box": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "sassd_bn_relu_workspace_bytes": (_SZ, [_I]),
    "sassd_bn_relu_fwd": (_I, [_P, _I, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _SZ, _P]),
    "sassd_bn_relu_bwd": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "sassd_bn2d_relu_workspace_bytes": (_SZ, [_I]),
    "sassd_bn2d_relu_fwd": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _SZ, _P]),
    "sassd_bn2d_stats": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _SZ, _P]),
    "sassd_bn2d_relu_bwd": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "sassd_aux_head_workspace_bytes": (_SZ, [_I]),
    "sassd_aux_prepare": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "sassd_aux_head_fwd": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "sassd_aux_head_bwd": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "sassd_guided_decode_fwd": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "sassd_guided_decode_bwd": (_I, [_P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "sassd_boxes_iou3d_batch": (_I, [_P, _P, _I, _I, _P, _P, _I, _P, _P, _P]),
    "sassd_focal_loss_workspace_bytes": (_SZ, [_I]),
    "sassd_focal_loss": (_I, [_P, _P, _I, _P, _I, _P, _P, _P, _SZ, _P]),
    "sassd_conv2d_wino4_chain_supported": (_I, [_I, _I, _I, _I]),
    "sassd_conv2d_wino4_chain_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "sassd_conv2d_wino4_chain": (_I, [_P, _I, _P, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _SZ, _P]),
    "sassd_conv2d_wino4_narrow_packed_floats": (_SZ, [_I]),
    "sassd_conv2d_wino4_pack_weight_narrow": (_I, [_P, _I, _I, _P, _P]),
    "sassd_conv2d_wino4_chain_tail": (_I, [_P, _P, _I, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _SZ, _P]),
    "sassd_wino4_tile_map_ints": (_SZ, [_I, _I, _I]),
    "sassd_wino4_tile_map": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "sassd_gather_pack": (_I, [_P, _P, _P, C.c_long, _I, _P]),
    "sassd_grad_sumsq": (_I, [_P, C.c_long, _P, _P]),
    "sassd_adam_step": (_I, [_P, _P, _P, _P, C.c_long, _P, _F, _F, _F, _F, _F, _I, _F, _F, _P]),
    "sassd_mfma_probe": (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
}

EXPORTS = tuple(_SIGS.keys())


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libsassd.so not found at %s -- run `python -c 'import __graft_entry__ as g; "
                               "g.build()'` (there is no CPU fallback)" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)            # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what):
    if rc != OK:
        l = lib()
        extra = ""
        if rc == EHIP:
            extra = ": " + l.sassd_last_hip_error_string().decode()
        raise RuntimeError("%s failed with code %d%s" % (what, rc, extra))


def ptr(t):
    """device/host pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream():
    """Raw hipStream_t of torch's current stream on the current device (one C call: this runs before every launch)."""
    import torch
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def csrc_hash():
    """Short hash over the kernel sources (csrc/*) next to this library: stamped into every PMC traffic record under
    profiles/ and compared by bench.py, so that counters measured on other kernels are never reported."""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "*"))):
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:12]
