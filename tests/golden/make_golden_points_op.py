"""Generate tests/golden/pts_in_boxes_ref.npz with the reference's own C++ `pts_in_boxes3d_cpu`
(mmdet/ops/points_op/src/points_op.cpp:92-144), JIT-compiled from where it lies under /root/reference against this
container's torch (`-DAT_CHECK=TORCH_CHECK`; outputs only into oracle/_ref/, nothing copied into the repo).

    python tests/golden/make_golden_points_op.py
"""
import os

import numpy as np
import torch
from torch.utils.cpp_extension import load

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = "/root/reference/mmdet/ops/points_op/src/points_op.cpp"


def main():
    bd = os.path.join(ROOT, "oracle", "_ref", "points_op_cpu")
    os.makedirs(bd, exist_ok=True)
    mod = load(name="ref_points_op_cpu", sources=[SRC], extra_cflags=["-DAT_CHECK=TORCH_CHECK", "-O2", "-w"],
               build_directory=bd, verbose=False)
    r = np.random.default_rng(21)
    n, m = 4000, 14
    boxes = np.zeros((m, 7), np.float32)
    boxes[:, 0], boxes[:, 1], boxes[:, 2] = r.uniform(5, 60, m), r.uniform(-30, 30, m), r.uniform(-2, -1, m)
    boxes[:, 3], boxes[:, 4], boxes[:, 5] = r.uniform(1.4, 1.9, m), r.uniform(1.5, 2.0, m), r.uniform(3.4, 4.6, m)
    boxes[:, 6] = r.uniform(-3.2, 3.2, m)
    boxes[1] = boxes[0]; boxes[1, 0] += 0.8                      # overlapping boxes: the last matching box wins
    pts = np.zeros((n, 3), np.float32)
    k = r.integers(0, m, n)
    pts[:, 0] = boxes[k, 0] + r.normal(0, 1.6, n)
    pts[:, 1] = boxes[k, 1] + r.normal(0, 1.6, n)
    pts[:, 2] = boxes[k, 2] + r.uniform(-0.5, 2.2, n)
    pts[:50] = boxes[k[:50], :3]                                  # exactly on the bottom centre
    pts[50:60, 0] += 10.0                                         # on / beyond the 10 m coarse-reject distance
    flag = torch.zeros(m, n, dtype=torch.int32)
    reg = torch.zeros(n, 3, dtype=torch.float32)
    mod.pts_in_boxes3d(torch.from_numpy(pts), torch.from_numpy(boxes), flag, reg)
    np.savez_compressed(os.path.join(HERE, "pts_in_boxes_ref.npz"), pts=pts, boxes=boxes, flag=flag.numpy(),
                        reg=reg.numpy())
    print("pts_in_boxes_ref.npz: inside", int(flag.numpy().max(0).sum()), "of", n)


if __name__ == "__main__":
    main()
