"""CPU numerics study for a Winograd weight-gradient kernel (next-round candidate for the 67 %-of-peak direct wgrad):
F(3x3, 2x2) -- the 3x3 filter gradient from 4x4 input tiles and 2x2 output-gradient tiles, 16 transform-domain
multiplies per (tile, cin, cout) instead of 36, accumulated over all tiles in the transform domain and inverted once.

    dW[co,ci] = A^T [ sum_tiles (G dY_tile G^T) * (B^T X_tile B) ] A          (elementwise product inside the sum)

with the F(3,2) matrices (interpolation points 0, 1, -1, inf).  Prints the error of an fp32 evaluation against an fp64
direct wgrad next to the error of the direct fp32 wgrad, for a BEV-like layer at reduced width.

    python tools/wino_wgrad_study.py
"""
import numpy as np

# F(m=3, r=2): output (here: the filter gradient) 3, "filter" (here: the dY tile) 2, input tile 4
AT = np.array([[1, 1, 1, 0], [0, 1, -1, 0], [0, 1, 1, 1]], dtype=np.float64)             # 3x4
G = np.array([[1, 0], [0.5, 0.5], [0.5, -0.5], [0, 1]], dtype=np.float64)                # 4x2
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, -1, 0, 1]], dtype=np.float64)   # 4x4


def check_1d():
    r = np.random.default_rng(0)
    d, g = r.normal(size=4), r.normal(size=2)
    want = np.array([d[0] * g[0] + d[1] * g[1], d[1] * g[0] + d[2] * g[1], d[2] * g[0] + d[3] * g[1]])
    got = AT @ ((G @ g) * (BT @ d))
    assert np.abs(got - want).max() < 1e-12, (got, want)


def direct(x, dy, dt):
    """x [N,Ci,H,W] (pad 1 applied here), dy [N,Co,H,W] -> dW [Co,Ci,3,3] accumulated in dtype dt"""
    n, ci, h, w = x.shape
    xp = np.zeros((n, ci, h + 2, w + 2), dt)
    xp[:, :, 1:-1, 1:-1] = x
    dw = np.zeros((dy.shape[1], ci, 3, 3), dt)
    for a in range(3):
        for b in range(3):
            dw[:, :, a, b] = np.einsum('nohw,nihw->oi', dy.astype(dt), xp[:, :, a:a + h, b:b + w], dtype=dt)
    return dw


def winograd(x, dy, dt):
    n, ci, h, w = x.shape
    xp = np.zeros((n, ci, h + 2, w + 2), dt)
    xp[:, :, 1:-1, 1:-1] = x
    at, g, bt = AT.astype(dt), G.astype(dt), BT.astype(dt)
    acc = np.zeros((dy.shape[1], ci, 4, 4), dt)
    for ty in range(0, h, 2):
        for tx in range(0, w, 2):
            xt = xp[:, :, ty:ty + 4, tx:tx + 4]                               # [n,ci,4,4]
            yt = dy[:, :, ty:ty + 2, tx:tx + 2].astype(dt)                    # [n,co,2,2]
            u = np.einsum('ab,nibc,dc->niad', bt, xt, bt, dtype=dt)
            v = np.einsum('ab,nobc,dc->noad', g, yt, g, dtype=dt)
            acc += np.einsum('noad,niad->oiad', v, u, dtype=dt)               # fp32 accumulate over tiles, like the MFMA
    return np.einsum('ab,oibc,dc->oiad', at, acc, at, dtype=dt)


def main():
    check_1d()
    r = np.random.default_rng(1)
    n, ci, co, h, w = 2, 16, 16, 40, 44
    x = np.maximum(r.normal(size=(n, ci, h, w)), 0).astype(np.float32)        # post-ReLU activations
    dy = (r.normal(size=(n, co, h, w)) * 1e-3).astype(np.float32)
    ref = direct(x.astype(np.float64), dy.astype(np.float64), np.float64)
    scale = np.abs(ref).max()
    for name, fn in (("direct fp32", direct), ("winograd F(3,2) fp32", winograd)):
        got = fn(x, dy, np.float32)
        print("%-22s max |err| / max |dW| = %.2e   rel L2 = %.2e" % (
            name, np.abs(got - ref).max() / scale, np.linalg.norm(got - ref) / np.linalg.norm(ref)))
    w64 = winograd(x.astype(np.float64), dy.astype(np.float64), np.float64)
    print("winograd fp64 vs direct fp64: %.2e (algebra check)" % (np.abs(w64 - ref).max() / scale))


if __name__ == "__main__":
    main()
