"""Generate tests/golden/voxel_mean_ref.npz: the reference's own SimpleVoxel.forward
(mmdet/models/backbones/vxnet.py:110-116, extracted with ast and run on CPU torch) applied to the voxelizer goldens.

    python tests/golden/make_golden_voxel_mean.py
"""
import ast
import os
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/mmdet/models/backbones/vxnet.py"


def main():
    tree = ast.parse(open(REF).read())
    fwd = None
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == "SimpleVoxel":
            for f in node.body:
                if isinstance(f, ast.FunctionDef) and f.name == "forward":
                    fwd = f
    g = {"torch": torch}
    exec(compile(ast.Module([fwd], []), REF, "exec"), g)
    self = types.SimpleNamespace(num_input_features=4)
    out = {}
    for case in ("small", "oob", "dense_t3", "t8"):
        G = np.load(os.path.join(HERE, "voxelizer_%s.npz" % case))
        v, n = torch.from_numpy(G["voxels"]), torch.from_numpy(G["num_points"])
        out[case] = g["forward"](self, v, n).numpy()
    np.savez_compressed(os.path.join(HERE, "voxel_mean_ref.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
