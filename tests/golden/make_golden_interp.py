"""Generate tests/golden/interp_ref.npz with the reference's own three_nn / three_interpolate / three_interpolate_grad
CUDA kernels (mmdet/ops/pointnet2/src/interpolate_gpu.cu:9-56,80-102,124-146) compiled for the HOST by
oracle/build.py::build_ref_interp (blockIdx / threadIdx become globals that a wrapper loops over; build container only).

    python tests/golden/make_golden_interp.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import build  # noqa: E402


def main():
    L = C.CDLL(build.build_ref_interp())
    r = np.random.default_rng(5)

    def cloud(n, nb):
        b = r.integers(0, nb, n).astype(np.float32)
        xyz = r.uniform([0, -40, -3], [70, 40, 1], (n, 3)).astype(np.float32)
        return np.concatenate([b[:, None], xyz], 1)
    u, k = cloud(1200, 3), cloud(1000, 3)
    k[:50] = u[:50]                                           # exact hits: distance 0 and ties
    P = lambda a: a.ctypes.data_as(C.c_void_p)                # noqa: E731
    d, i = np.empty((len(u), 3), np.float32), np.empty((len(u), 3), np.int32)
    L.ref_three_nn(len(u), len(k), P(u), P(k), P(d), P(i))
    feat = r.standard_normal((len(k), 12)).astype(np.float32)
    w = r.random((len(u), 3)).astype(np.float32)
    w /= w.sum(1, keepdims=True)
    out = np.zeros((len(u), 12), np.float32)
    L.ref_three_interpolate(12, len(k), len(u), P(feat), P(i), P(w), P(out))
    g = r.standard_normal((len(u), 12)).astype(np.float32)
    gp = np.zeros((len(k), 12), np.float32)
    L.ref_three_interpolate_grad(12, len(u), len(k), P(g), P(i), P(w), P(gp))
    np.savez_compressed(os.path.join(HERE, "interp_ref.npz"), unknown=u, known=k, dist2=d, idx=i, feat=feat, weight=w,
                        out=out, grad_out=g, grad_points=gp)
    print("interp_ref.npz", d.shape, out.shape, gp.shape)


if __name__ == "__main__":
    main()
