"""Build recipe for the CPU oracle (TEST INFRASTRUCTURE, never imported by the product path).

  oracle/_build/liboracle.so   <- oracle/sassd_oracle.c         (our restatement, always buildable)
  oracle/_build/libharness.so  <- oracle/harness/augment_harness.hip: the product's own __host__ __device__ per-point
                                  functions (sa-ssd_amd/csrc/augment_core.h) looped on the CPU, hipcc --cuda-host-only
  oracle/_ref/libref_iou3d.so  <- /root/reference/mmdet/ops/iou3d/src/iou3d_kernel.cu lines 1-221
                                  (the reference's own __device__ functions compiled for the HOST;
                                  only when /root/reference exists; source is piped to g++, never copied)
  oracle/_ref/libref_interp.so <- /root/reference/mmdet/ops/pointnet2/src/interpolate_gpu.cu, the three __global__
                                  kernels (lines 9-56, 80-102, 124-146) compiled for the HOST: blockIdx / threadIdx are
                                  plain globals that a wrapper loops over, atomicAdd is a plain add
  oracle/_ref/points_op_cpu/   <- points_op.cpp JIT-built against torch by tests/golden/make_golden_points_op.py
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


REF_INTERP_CU = "/root/reference/mmdet/ops/pointnet2/src/interpolate_gpu.cu"
_INTERP_PRELUDE = r"""
#include <cmath>
#include <cstdio>
#define __device__
#define __global__
#define __restrict__
struct Idx3 { int x, y, z; };
static Idx3 blockIdx, threadIdx, blockDim = {1, 1, 1};
static inline void atomicAdd(float* p, float v) { *p += v; }
"""
_INTERP_WRAP = r"""
extern "C" void ref_three_nn(int n, int m, const float* unknown, const float* known, float* dist2, int* idx) {
    for (int p = 0; p < n; ++p) { blockIdx.x = p; threadIdx.x = 0; three_nn_kernel_fast(n, m, unknown, known, dist2, idx); }
}
extern "C" void ref_three_interpolate(int c, int m, int n, const float* points, const int* idx, const float* weight,
                                      float* out) {
    for (int p = 0; p < n; ++p) for (int ch = 0; ch < c; ++ch) {
        blockIdx.x = p; blockIdx.y = ch; threadIdx.x = 0;
        three_interpolate_kernel_fast(c, m, n, points, idx, weight, out);
    }
}
extern "C" void ref_three_interpolate_grad(int c, int n, int m, const float* grad_out, const int* idx,
                                           const float* weight, float* grad_points) {
    for (int p = 0; p < n; ++p) for (int ch = 0; ch < c; ++ch) {
        blockIdx.x = p; blockIdx.y = ch; threadIdx.x = 0;
        three_interpolate_grad_kernel_fast(c, n, m, grad_out, idx, weight, grad_points);
    }
}
"""


def build_ref_interp(force=False):
    """The reference's three_nn / three_interpolate(_grad) CUDA kernels compiled for the host (only where the
    reference tree is mounted).  Returns the .so path or None."""
    dst = os.path.join(REF, "libref_interp.so")
    if not os.path.exists(REF_INTERP_CU):
        return dst if os.path.exists(dst) else None
    os.makedirs(REF, exist_ok=True)
    if force or not _newer(dst, [REF_INTERP_CU, __file__]):
        with open(REF_INTERP_CU) as f:
            L = f.readlines()
        body = "".join(L[8:56]) + "".join(L[79:102]) + "".join(L[123:146])     # the three kernels, no launchers
        tu = _INTERP_PRELUDE + body + _INTERP_WRAP
        subprocess.run(["g++", "-x", "c++", "-O2", "-fno-fast-math", "-ffp-contract=off", "-shared", "-fPIC", "-w",
                        "-o", dst, "-"], input=tu.encode(), check=True)
    return dst


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


def build_harness(force=False):
    """oracle/harness/augment_harness.hip: the product's __host__ __device__ per-point functions looped on the CPU."""
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(HERE, "harness", "augment_harness.hip")
    core = os.path.join(os.path.dirname(HERE), "sa-ssd_amd", "csrc", "augment_core.h")
    dst = os.path.join(BUILD, "libharness.so")
    if force or not _newer(dst, [src, core]):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-x", "hip", "--cuda-host-only", "-O2", "-fno-fast-math",
                               "-ffp-contract=on", "-shared", "-fPIC", "-o", dst, src])
    return dst


if __name__ == "__main__":
    print(build_oracle(True))
    print(build_harness(True))
    print(build_ref(True))
    print(build_ref_interp(True))
