"""CPU numerics study: can the fp32 BEV convolutions run on the bf16 MFMA pipe (16x the fp32-MFMA rate on gfx950) by
splitting each fp32 operand into bf16 pieces, and still meet the 1e-4 box / score parity bar?

    a = a1 + a2 + a3   (a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2));   same for b
    a*b ~ sum of the partial products a_i*b_j, accumulated in fp32 (what v_mfma_f32_32x32x16_bf16 does)

Variants: 1 term (plain bf16), 3 terms (a1b1 + a1b2 + a2b1), 6 terms (all i+j <= 4).  The study pushes a random post-ReLU
feature map through a chain of 3x3 conv + folded-BN + ReLU layers (the BEVNet shape at reduced width / extent) and reports
the error of the last feature map against an fp64 evaluation, next to the plain fp32 error.

    python tools/split_bf16_study.py
"""
import numpy as np


def bf16(x):
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def split(x, k):
    parts, rest = [], np.asarray(x, dtype=np.float32)
    for _ in range(k):
        p = bf16(rest)
        parts.append(p)
        rest = rest - p
    return parts


def conv3x3(x, w, dt):
    """x [C,H,W], w [O,C,3,3] -> [O,H,W], accumulation dtype dt"""
    c, h, wd = x.shape
    xp = np.zeros((c, h + 2, wd + 2), dt)
    xp[:, 1:-1, 1:-1] = x
    out = np.zeros((w.shape[0], h, wd), dt)
    for a in range(3):
        for b in range(3):
            out += np.einsum('oc,chw->ohw', w[:, :, a, b].astype(dt), xp[:, a:a + h, b:b + wd], dtype=dt)
    return out


def conv_split(x, w, terms):
    xs, ws = split(x, 3), split(w, 3)
    pairs = {1: [(0, 0)], 3: [(0, 0), (0, 1), (1, 0)], 6: [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]}[terms]
    out = 0
    for i, j in sorted(pairs, key=lambda p: -(p[0] + p[1])):          # small terms first
        out = out + conv3x3(xs[i], ws[j], np.float32)
    return out


def main():
    r = np.random.default_rng(0)
    c, h, w, layers = 64, 48, 44, 7
    x0 = np.maximum(r.normal(size=(c, h, w)), 0).astype(np.float32)
    ws = [(r.normal(size=(c, c, 3, 3)) * (2.0 / (c * 9)) ** 0.5).astype(np.float32) for _ in range(layers)]
    shifts = [(r.normal(size=(c, 1, 1)) * 0.1).astype(np.float32) for _ in range(layers)]

    def run(fn):
        x = x0
        for wt, sh in zip(ws, shifts):
            x = np.maximum(fn(x, wt) + sh, 0).astype(np.float32 if fn is not ref else np.float64)
        return x

    ref = lambda x, wt: conv3x3(x.astype(np.float64), wt.astype(np.float64), np.float64)
    want = run(ref)
    scale = np.abs(want).max()
    rows = [("fp32 direct", lambda x, wt: conv3x3(x, wt, np.float32))]
    for t in (1, 3, 6):
        rows.append(("bf16 split, %d product%s" % (t, "s" if t > 1 else ""), lambda x, wt, t=t: conv_split(x, wt, t)))
    print("%d layers of 3x3 conv %d->%d on %dx%d, error of the last feature map vs fp64:" % (layers, c, c, h, w))
    for name, fn in rows:
        got = run(fn)
        print("  %-26s max |err| / max |x| = %.2e   rel L2 = %.2e" % (
            name, np.abs(got - want).max() / scale, np.linalg.norm(got - want) / np.linalg.norm(want)))


if __name__ == "__main__":
    main()
