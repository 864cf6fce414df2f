"""CPU oracle for the SA-SSD hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and only as
the checker / timed CPU port.  The product path (sa-ssd_amd/) never imports it and fails loudly when its
HIP library is missing.

Parity pinning status (see DESIGN.md "Oracle"):
  voxelizer        pinned against the reference's points_ops.py (identity-jit numba stub)  -> tests/golden
  rotated IoU/NMS  pinned against oracle/_ref (reference iou3d_kernel.cu device code built for the host)
  decode / warp    pinned against the reference's own torch functions (exec'd from source) -> tests/golden
  sparse conv      PARITY UNPINNED upstream (spconv v1.0 is not vendored, reference has no tests);
                   oracle-defined, cross-checked against torch.nn.functional.conv3d.
"""
from . import build as _build  # noqa: F401
