"""Generate tests/golden/sampler_ref.npz with the reference's OWN samplers (mmdet/datasets/loader/sampler.py, imported
unchanged; it only needs torch + numpy): index sequences of DistributedGroupSampler for several dataset sizes, replica
counts, ranks and epochs, and of GroupSampler under a fixed numpy seed (build container only).

    python tests/golden/make_golden_sampler.py
"""
import importlib.util
import os
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [(7, 2, 2), (37, 2, 1), (37, 3, 4), (3712, 2, 8), (100, 1, 3)]        # (dataset size, samples_per_gpu, replicas)


def main():
    spec = importlib.util.spec_from_file_location("ref_sampler", "/root/reference/mmdet/datasets/loader/sampler.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    out = {}
    for n, spg, world in CASES:
        ds = types.SimpleNamespace(flag=np.ones(n, dtype=np.uint8))
        for epoch in (0, 1, 7):
            for rank in range(world):
                s = m.DistributedGroupSampler(ds, spg, world, rank)
                s.set_epoch(epoch)
                out["dist_%d_%d_%d_e%d_r%d" % (n, spg, world, epoch, rank)] = np.array(list(s), dtype=np.int64)
                out["dist_%d_%d_%d_len" % (n, spg, world)] = np.array(len(s))
        np.random.seed(n)
        g = m.GroupSampler(ds, spg)
        out["group_%d_%d" % (n, spg)] = np.array([int(i) for i in g], dtype=np.int64)
        out["group_%d_%d_again" % (n, spg)] = np.array([int(i) for i in g], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "sampler_ref.npz"), **out)
    print("sampler_ref.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
