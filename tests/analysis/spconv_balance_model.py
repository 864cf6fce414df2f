#!/usr/bin/env python
"""GPU-less critical-path model of a sparse-conv layer at KITTI scale (the reasoning behind csrc/spconv_gq.h).

At ~15 k rows a layer is ONE round of workgroups (one 64-row slice per CU), so it lasts as long as its heaviest workgroup,
and a workgroup as long as its slowest SIMD.  For every rulebook of a frame (CPU oracle: oracle/rulebook.py) this prints,
in units of one 16-pair MFMA tile (64 x v_mfma_f32_16x16x4_f32 = 2048 cycles on one SIMD at 64 -> 64 channels):

  * the round-2/3 work distribution -- consecutive 64-row slices, whole kernel offsets dealt to the 8 waves by the static
    table `c_assign8`, waves w and w+4 on one SIMD: mean and worst SIMD load over the workgroups;
  * the round-4 distribution -- interleaved slices of 8 XCD-local blocks (all 256 CUs), the workgroup's unit list cut into
    equal contiguous ranges per wave -- for 16-pair tiles and for 4-pair quads (x 1.25: the 4x4x1 MFMA form runs at 126 of
    the 151 TF of the 16x16x4 form, tools/probe_mfma4x4.hip);
  * the tile fill (pairs / issued MFMA rows) of both granularities and the perfectly balanced bound.

    python tests/analysis/spconv_balance_model.py [--batch 1] [--clock-ghz 2.07]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sassd  # noqa: E402,F401
from sassd import synth  # noqa: E402
from oracle import clib, rulebook as rb  # noqa: E402

ORDER = [13, 4, 10, 12, 14, 16, 22, 1, 3, 5, 7, 9, 11, 15, 17, 19, 21, 23, 25, 0, 2, 6, 8, 18, 20, 24, 26]
ASSIGN8 = [[0, 15, 23], [1, 9, 17], [2, 10, 18], [3, 11, 19, 25], [4, 12, 20, 26], [5, 13, 21], [6, 14, 22], [7, 8, 16, 24]]


def slice_counts(nbr, interleaved):
    """pairs per (workgroup, offset)"""
    n = len(nbr)
    if not interleaved:
        nwg = (n + 63) // 64
        pad = np.full((nwg * 64, 27), -1, np.int32)
        pad[:n] = nbr
        return (pad.reshape(nwg, 64, 27) >= 0).sum(1)
    nb8 = 1 if n <= 16384 else -(-n // 16384)
    bs = -(-n // (8 * nb8))
    out = []
    for j in range(8 * nb8):
        blk = nbr[j * bs:min(n, (j + 1) * bs)]
        out += [(blk[s::32] >= 0).sum(0) for s in range(32) if len(blk) > s]
    return np.array(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--clock-ghz", type=float, default=2.07)
    args = ap.parse_args()
    w = synth.workload("car")
    cal = w["cal"]
    idx = []
    for b in range(args.batch):
        _, c, _ = clib.points_to_voxel(w["frame"](b), cal["voxel_size"], cal["pc_range"], cal["max_points"], True,
                                       cal["max_voxels"])
        idx.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))
    cur, shape = np.concatenate(idx, 0), (40, 1600, 1408)
    books = {"subm0": rb.subm_rulebook(cur, shape)[1]}
    for lvl in range(1, 4):
        cur, nbr_d, shape = rb.conv_rulebook(cur, shape, args.batch)
        books["down%d" % (lvl - 1)] = nbr_d
        books["subm%d" % lvl] = rb.subm_rulebook(cur, shape)[1]
    us = 2048.0 / (args.clock_ghz * 1e3)
    print("MFMA critical path of a 64 -> 64 layer on each rulebook, microseconds at %.2f GHz (tiles of 16 pairs = %.2f us on "
          "one SIMD)" % (args.clock_ghz, us))
    print("%-6s %7s %8s | %-22s | %-22s | %-22s | %6s %6s | %s" % (
        "book", "rows", "pairs", "round 3: mean / worst", "round 4, 16-pair tiles", "round 4, 4-pair quads", "fill16", "fill4",
        "perfect balance, 100 % fill"))
    for name, nbr in books.items():
        c3 = slice_counts(nbr, False)
        t3 = -(-c3 // 16)
        wave = np.stack([sum(t3[:, ORDER[s]] for s in a) for a in ASSIGN8], 1)
        simd3 = wave[:, :4] + wave[:, 4:]
        c4 = slice_counts(nbr, True)
        u16, u4 = (-(-c4 // 16)).sum(1), (-(-c4 // 4)).sum(1)
        s16 = np.ceil(u16 / 8.0) * 2                         # eight equal ranges, two waves per SIMD
        s4 = np.ceil(u4 / 8.0) * 2 * 0.25 * 1.25
        ideal = c3.sum() / 16.0 / 1024.0
        print("%-6s %7d %8d | %8.1f / %-11.1f | %8.1f / %-11.1f | %8.1f / %-11.1f | %6.2f %6.2f | %.1f" % (
            name, len(nbr), c3.sum(), simd3.max(1).mean() * us, simd3.max() * us, s16.mean() * us, s16.max() * us,
            s4.mean() * us, s4.max() * us, c3.sum() / (t3.sum() * 16.0), c4.sum() / ((-(-c4 // 4)).sum() * 4.0), ideal * us))


if __name__ == "__main__":
    main()
