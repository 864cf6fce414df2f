"""C-ABI error convention without a GPU (include/sassd.h: every entry point returns 0 or a negative code and validates
its arguments BEFORE touching the device): NULL pointers / unsupported shapes -> SASSD_EINVAL (-1), undersized
workspaces -> SASSD_ENOSPC (-2); workspace / packing queries are pure host arithmetic."""
import ctypes as C

import sassd  # noqa: F401
from sassd import _C

EINVAL, ENOSPC = -1, -2


def test_argument_validation_returns_error_codes():
    L = _C.lib()
    null = None
    assert L.sassd_voxelize(null, 10, 4, null, null, 5, 100, 0, null, null, 3, null, null, 4, null, null, 100, null,
                            null, 0, null) == EINVAL
    assert L.sassd_spconv_fwd(null, null, null, 0, null, 27, 16, 16, null, null, 0, null, 0, null) == EINVAL
    assert L.sassd_conv2d_fwd(null, null, null, null, 0, null, 1, 16, 16, 8, 8, 3, null) == EINVAL
    assert L.sassd_conv2d_fwd(C.c_void_p(16), C.c_void_p(16), null, null, 0, C.c_void_p(16), 1, 16, 16, 8, 8, 5,
                              null) == EINVAL                       # kernel size 5 does not exist on the path
    assert L.sassd_conv2d_wino_fwd(C.c_void_p(16), C.c_void_p(16), null, null, 0, C.c_void_p(16), 1, 28, 28, 200, 176,
                                   null) == EINVAL                  # channel counts the Winograd kernel rejects
    assert L.sassd_conv2d_bwd_weight(C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), 1, 16, 16, 8, 8, 3, 0,
                                     C.c_void_p(16), 4, null) == ENOSPC
    assert L.sassd_three_nn(-1, 5, null, null, null, null, null) == EINVAL
    assert L.sassd_three_nn_binned(4, 4, C.c_void_p(16), C.c_void_p(16), 0.0, 0.0, 1.6, 44, 50, 2, C.c_void_p(16),
                                   C.c_void_p(16), C.c_void_p(16), 8, null) == ENOSPC
    assert L.sassd_rotate_iou_eval(null, 3, null, 3, -1, null, null) == EINVAL
    assert L.sassd_adam_step(null, null, null, null, 10, null, 1e-3, 0.9, 0.99, 1e-8, 0.01, 1, 10.0, 1.0,
                             null) == EINVAL
    assert L.sassd_adam_step(C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), 10, null, 1e-3, 0.9, 0.99,
                             1e-8, 0.01, 0, 10.0, 1.0, null) == EINVAL      # step counts from 1
    assert L.sassd_nms_gpu(null, 4, 0.1, null, null, null, 0, null) == EINVAL
    # round-2 training entry points
    p16 = C.c_void_p(16)
    assert L.sassd_conv2d_bf16_fwd(null, null, null, null, 1, 32, 128, 8, 16, null) == EINVAL
    assert L.sassd_conv2d_bf16_fwd(p16, p16, null, p16, 1, 32, 100, 8, 16, null) == EINVAL        # Cout % 32
    assert L.sassd_conv2d_bf16_fwd(p16, p16, null, p16, 1, 32, 128, 8, 18, null) == EINVAL        # W % 4
    # round-6: 1x1 convolution on the bf16 MFMA
    assert L.sassd_conv1x1_bf16_fwd(null, null, null, null, 1, 256, 256, 64, null) == EINVAL
    assert L.sassd_conv1x1_bf16_fwd(p16, p16, null, p16, 1, 300, 256, 64, null) == EINVAL         # Cin > 256
    assert L.sassd_conv1x1_bf16_fwd(p16, p16, null, p16, 1, 256, 256, 66, null) == EINVAL         # HW % 4
    assert L.sassd_conv1x1_bf16_fwd(p16, p16, null, p16, 1, 40, 256, 64, null) == EINVAL          # two ragged k-steps
    assert L.sassd_conv1x1_bf16_fwd(C.c_void_p(20), p16, null, p16, 1, 256, 256, 64, null) == EINVAL     # x not 8-byte aligned
    assert L.sassd_conv1x1_bf16_pack_weight(null, 256, 256, 0, p16, null) == EINVAL
    assert L.sassd_conv1x1_bf16_pack_weight(p16, 256, 300, 0, p16, null) == EINVAL
    assert L.sassd_conv2d_bwd_weight_bf16(p16, p16, p16, 1, 16, 16, 8, 9, 3, 0, p16, 1 << 30, null) == EINVAL   # odd W
    assert L.sassd_conv2d_bwd_weight_bf16(p16, p16, p16, 1, 16, 16, 8, 8, 3, 0, p16, 4, null) == ENOSPC
    assert L.sassd_assign_targets(null, 0, null, 100, 2, null, null, null, null, 0, null, null, 0.6, 0.45, null, null,
                                  null, 100, null, 1, null, 0, null) == EINVAL
    assert L.sassd_assign_targets(p16, 0, null, 100, 2, null, null, null, p16, 0, null, null, 0.6, 0.45, p16, p16,
                                  null, 50, p16, 1, p16, 1 << 20, null) == EINVAL                  # output stride < anchors
    assert L.sassd_assign_targets(p16, 0, null, 100, 2, null, null, null, p16, 0, null, null, 0.6, 0.45, p16, p16,
                                  null, 100, p16, 1, p16, 8, null) == ENOSPC
    assert L.sassd_rpn_loss(null, null, null, 1, null, null, null, 0, null, 100, 2, null, null, null, null, null, 0,
                            null) == EINVAL
    assert L.sassd_guided_select(p16, null, 100, 2, 1, 0.1, 0, p16, p16, p16, p16, 1 << 20, null) == EINVAL     # cap 0
    assert L.sassd_guided_select(p16, null, 100, 2, 1, 0.1, 64, p16, p16, p16, p16, 1, null) == ENOSPC
    assert L.sassd_bn_relu_fwd(p16, 100, 48, p16, p16, null, null, 0.01, 1e-3, p16, p16, p16, p16, 1 << 30,
                               null) == EINVAL                                                     # 256 % C != 0
    assert L.sassd_bn_relu_fwd(p16, 100, 64, p16, p16, p16, null, 0.01, 1e-3, p16, p16, p16, p16, 1 << 30,
                               null) == EINVAL                                                     # one running stat only
    assert L.sassd_bn_relu_bwd(p16, p16, 100, 64, p16, p16, p16, p16, p16, p16, p16, p16, 8, null) == ENOSPC
    assert L.sassd_gather_pack(null, null, null, 10, 0, null) == EINVAL
    assert L.sassd_gather_pack(p16, p16, p16, 0, 1, null) == 0
    # empty problems are not errors
    assert L.sassd_three_nn(0, 0, null, null, C.c_void_p(16), C.c_void_p(16), null) == 0
    assert L.sassd_rotate_iou_eval(null, 0, null, 0, -1, C.c_void_p(16), null) == 0


def test_host_side_queries():
    L = _C.lib()
    assert L.sassd_conv2d_wino_supported(256, 256, 200, 176) == 1 and L.sassd_conv2d_wino_supported(320, 256, 200, 176) == 1
    assert L.sassd_conv2d_wino_supported(256, 28, 200, 176) == 0 and L.sassd_conv2d_wino_supported(256, 256, 200, 44) == 0
    assert L.sassd_conv2d_wino_packed_floats(256, 256) == 16 * 256 * 256
    assert L.sassd_conv2d_wgrad_workspace_bytes(2, 256, 256, 200, 176, 3) % (9 * 256 * 256 * 4) == 0
    assert L.sassd_conv2d_wgrad_workspace_bytes(2, 256, 256, 200, 176, 5) == 0
    assert L.sassd_three_nn_binned_workspace_bytes(1000, 44, 50, 2) > 1000 * 20
    assert L.sassd_three_nn_binned_workspace_bytes(1000, 4096, 4096, 2) == 0          # grid too large
    assert L.sassd_spconv_bwd_weight_workspace_bytes(16111, 27, 64, 64) >= 126 * 27 * 64 * 64 * 4
    assert L.sassd_hash_bytes(20000) >= 2 * 20000 * 8
    assert L.sassd_conv2d_bf16_supported(256, 256, 200, 176) == 1 and L.sassd_conv2d_bf16_supported(28, 256, 200, 176) == 1
    assert L.sassd_conv2d_bf16_supported(256, 28, 200, 176) == 0 and L.sassd_conv2d_bf16_supported(256, 256, 200, 182) == 0
    assert L.sassd_conv2d_bf16_supported(256, 256, 188, 188) == 1      # partial last tile column (W % 4 == 0)
    assert L.sassd_conv2d_bf16_packed_elems(28, 256) == 9 * 32 * 256          # Cin padded to 32 inside the pack
    assert L.sassd_conv1x1_bf16_supported(256, 256, 35200) == 1 and L.sassd_conv1x1_bf16_supported(20, 256, 35200) == 1
    assert L.sassd_conv1x1_bf16_supported(256, 20, 35200) == 1 and L.sassd_conv1x1_bf16_supported(28, 28, 35344) == 1
    assert L.sassd_conv1x1_bf16_supported(257, 256, 64) == 0 and L.sassd_conv1x1_bf16_supported(40, 256, 64) == 0
    assert L.sassd_conv1x1_bf16_supported(256, 256, 35201) == 0
    assert L.sassd_conv1x1_bf16_packed_elems(256, 256) == 256 * 256 and L.sassd_conv1x1_bf16_packed_elems(20, 256) == 256 * 32
    assert L.sassd_conv1x1_bf16_packed_elems(256, 20) == 64 * 256 and L.sassd_conv1x1_bf16_packed_elems(40, 256) == 0
    assert L.sassd_assign_targets_workspace_bytes(2, 70400, 16) >= 2 * 70400 * 8 + 16 * 4
    assert L.sassd_rpn_loss_workspace_bytes(2, 70400) == 2 * 275 * 3 * 4
    assert L.sassd_guided_select_workspace_bytes(2, 70400) == 2 * 275 * 4
    assert L.sassd_bn_relu_workspace_bytes(64) >= 256 * 2 * 64 * 8


def test_bench_refuses_a_mismatched_launch():
    """bench.py --gpus N under a launcher that started a different WORLD_SIZE must fail loudly (not run N' ranks and
    report N); with WORLD_SIZE unset and N > 1 it re-executes itself under torch.distributed.run (tests/test_gpu_multi.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                         env=env, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=2" in (out.stderr + out.stdout)
