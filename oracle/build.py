"""Build recipe for the CPU oracle (TEST INFRASTRUCTURE, never imported by the product path).

  oracle/_build/liboracle.so   <- oracle/sassd_oracle.c         (our restatement, always buildable)
  oracle/_ref/libref_iou3d.so  <- /root/reference/mmdet/ops/iou3d/src/iou3d_kernel.cu lines 1-221
                                  (the reference's own __device__ functions compiled for the HOST;
                                  only when /root/reference exists; source is piped to g++, never copied)
  oracle/_ref/libref_points_op.so is NOT built: points_op.cpp needs torch headers (27 s build) and only
                                  serves the training-side pts_in_boxes3d row (later round).
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")
REF = os.path.join(HERE, "_ref")
REF_CU = "/root/reference/mmdet/ops/iou3d/src/iou3d_kernel.cu"

_PRELUDE = r"""
#include <cmath>
#include <cstdio>
#include <algorithm>
#define __device__
#define __global__
#define __shared__
using std::min; using std::max;
// CUDA resolves cos/sin/atan2/fabs on float arguments to the float overloads; do the same on the host
using std::cos; using std::sin; using std::atan2; using std::fabs;
"""
_WRAP = r"""
extern "C" float ref_box_overlap(const float* a, const float* b) { return box_overlap(a, b); }
extern "C" float ref_iou_bev(const float* a, const float* b) { return iou_bev(a, b); }
"""


def _newer(dst, srcs):
    if not os.path.exists(dst):
        return False
    t = os.path.getmtime(dst)
    return all(os.path.getmtime(s) <= t for s in srcs if os.path.exists(s))


def build_oracle(force=False):
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(HERE, "sassd_oracle.c")
    dst = os.path.join(BUILD, "liboracle.so")
    if force or not _newer(dst, [src]):
        subprocess.check_call(["gcc", "-O3", "-fno-fast-math", "-ffp-contract=off", "-shared", "-fPIC",
                               "-o", dst, src, "-lm"])
    return dst


def build_ref(force=False):
    """Compile the reference's own rotated-IoU device functions for the host (only where the reference
    tree is mounted).  Returns the .so path or None."""
    dst = os.path.join(REF, "libref_iou3d.so")
    if not os.path.exists(REF_CU):
        return dst if os.path.exists(dst) else None
    os.makedirs(REF, exist_ok=True)
    if force or not _newer(dst, [REF_CU, __file__]):
        with open(REF_CU) as f:
            body = "".join(f.readlines()[:221])          # device functions only (no <<<>>> launchers)
        tu = _PRELUDE + body + _WRAP
        subprocess.run(["g++", "-x", "c++", "-O2", "-fno-fast-math", "-ffp-contract=off", "-shared",
                        "-fPIC", "-w", "-o", dst, "-"], input=tu.encode(), check=True)
    return dst


if __name__ == "__main__":
    print(build_oracle(True))
    print(build_ref(True))
