"""Generate tests/golden/eval_iou_ref.npz by running the reference's own numba device functions for the KITTI-eval
rotated IoU (mmdet/core/post_processing/rotate_nms_gpu.py:153-388,536-547) as PLAIN PYTHON: `numba` is replaced by a stub
whose `cuda.jit` is the identity and whose `cuda.local.array` returns numpy arrays, so `devRotateIoUEval(query, box,
criterion)` executes unchanged (build container only; nothing is copied into the repo).

    python tests/golden/make_golden_eval.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/mmdet/core/post_processing/rotate_nms_gpu.py"


def load_reference():
    nb = types.ModuleType("numba")
    cuda = types.ModuleType("numba.cuda")

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    cuda.jit = jit
    cuda.local = types.SimpleNamespace(array=lambda shape, dtype=np.float32: np.zeros(shape, np.float32))
    cuda.shared = types.SimpleNamespace(array=lambda shape, dtype=np.float32: np.zeros(shape, np.float32))
    nb.cuda, nb.float32, nb.jit = cuda, np.float32, jit
    sys.modules["numba"], sys.modules["numba.cuda"] = nb, cuda
    spec = importlib.util.spec_from_file_location("ref_rotate_nms_gpu", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def boxes5(r, n, spread):
    b = np.zeros((n, 5), np.float32)
    b[:, 0], b[:, 1] = r.uniform(0, spread, n), r.uniform(0, spread, n)
    b[:, 2], b[:, 3] = r.uniform(1.2, 2.2, n), r.uniform(3.0, 5.0, n)
    b[:, 4] = r.uniform(-3.3, 3.3, n)
    return b


def main():
    m = load_reference()
    r = np.random.default_rng(9)
    boxes, q = boxes5(r, 40, 12.0), boxes5(r, 36, 12.0)
    q[:4] = boxes[:4]                                     # identical boxes
    q[4:8, :4] = boxes[4:8, :4]; q[4:8, 4] = boxes[4:8, 4] + np.float32(np.pi / 2)
    boxes[8, 4] = 0; q[8] = boxes[8]; q[8, 0] += 0.5      # axis aligned, shifted
    q[9] = boxes[9]; q[9, 2:4] *= 0.5                     # contained
    q[10] = boxes[10]; q[10, 0] += 50                     # disjoint
    out = {"boxes": boxes, "query": q}
    for crit in (-1, 0, 1):
        iou = np.zeros((len(boxes), len(q)), np.float32)
        for i in range(len(boxes)):
            for j in range(len(q)):
                iou[i, j] = m.devRotateIoUEval(q[j].copy(), boxes[i].copy(), crit)    # kernel order: (query, box)
        out["iou_%d" % crit] = iou
    np.savez_compressed(os.path.join(HERE, "eval_iou_ref.npz"), **out)
    print("eval_iou_ref.npz", {k: v.shape for k, v in out.items()}, float(out["iou_-1"].max()))


if __name__ == "__main__":
    main()
