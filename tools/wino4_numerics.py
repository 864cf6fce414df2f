#!/usr/bin/env python
"""CPU numerics study of Winograd F(4x4,3x3) -- and, round 4, F(6x6,3x3) -- in fp32 (torch emulation): seven chained 3x3
conv + shift + ReLU layers and a 256-channel single layer, direct vs F(2x2) vs F(4x4) vs F(6x6), error against fp64.
The F(6x6) result is recorded in DESIGN.md §8."""
import numpy as np, torch, torch.nn.functional as F
torch.manual_seed(0)
# F(4x4,3x3) matrices (Lavin & Gray)
Bt = np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],np.float64)
G = np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],np.float64)
At = np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]],np.float64)
Bt2 = np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],np.float64)
G2 = np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],np.float64)
At2 = np.array([[1,1,1,0],[0,1,-1,-1]],np.float64)

def cook_toom(points, m, r=3):
    """F(m, r) matrices from the interpolation points (the last point is infinity): A^T rows are powers of the points, G rows are
    [1, p, p^2] over the Lagrange denominators, and B^T is SOLVED from the identity A^T[(G g) * (B^T d)] = correlate(d, g) in
    fp64 -- nothing recalled from a table."""
    n = m + r - 1
    pts = np.array(points, np.float64)
    assert len(pts) == n - 1
    At = np.zeros((m, n)); G = np.zeros((n, r))
    for j, pj in enumerate(pts):
        At[:, j] = pj ** np.arange(m)
        G[j] = pj ** np.arange(r) / np.prod([pj - pk for k, pk in enumerate(pts) if k != j])
    At[m - 1, n - 1] = 1; G[n - 1, r - 1] = 1
    M = np.stack([At[i] * G[:, k] for i in range(m) for k in range(r)])            # (m r, n)
    Bt = np.zeros((n, n))
    for l in range(n):
        rhs = np.array([1.0 if l == i + k else 0.0 for i in range(m) for k in range(r)])
        Bt[:, l] = np.linalg.lstsq(M, rhs, rcond=None)[0]
    return Bt, G, At


Bt6, G6, At6 = cook_toom([0, 1, -1, 2, -2, 0.5, -0.5], 6)


def wino_conv(x, w, Bt, G, At, m, dt):
    # x [C,H,W], w [K,C,3,3]; H,W multiples of m; pad 1
    C,H,W = x.shape; K = w.shape[0]; a = m+2
    Bt_, G_, At_ = (torch.tensor(M, dtype=dt) for M in (Bt,G,At))
    xp = F.pad(x, (1,1,1,1))
    th, tw = H//m, W//m
    # tiles [C, th, tw, a, a]
    t = xp.unfold(1, a, m).unfold(2, a, m)          # C, th, tw, a, a
    V = torch.einsum('ij,cthjk,lk->cthil', Bt_, t, Bt_)   # B^T d B
    U = torch.einsum('ij,kcjl,ml->kcim', G_, w, G_)       # G g G^T  [K,C,a,a]
    # per position GEMM in dtype dt (sum over C sequentially emulated by matmul in dt)
    Mm = torch.einsum('kcil,cthil->kthil', U, V)
    Y = torch.einsum('ij,kthjl,ml->kthim', At_, Mm, At_)  # [K,th,tw,m,m]
    return Y.permute(0,1,3,2,4).reshape(K, H, W)

def chain(dt, mode, L=7, C=64, H=48, W=48):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(C,H,W, generator=g, dtype=torch.float64).clamp(min=0)
    ws = [torch.randn(C,C,3,3, generator=g, dtype=torch.float64)*(2.0/(C*9))**0.5 for _ in range(L)]
    sh = [torch.randn(C, generator=g, dtype=torch.float64)*0.1 for _ in range(L)]
    x = x.to(dt)
    for w, s in zip(ws, sh):
        w = w.to(dt); s = s.to(dt)
        if mode == 'direct': y = F.conv2d(x[None], w, None, 1, 1)[0]
        elif mode == 'f2': y = wino_conv(x, w, Bt2, G2, At2, 2, dt)
        elif mode == 'f6': y = wino_conv(x, w, Bt6, G6, At6, 6, dt)
        else: y = wino_conv(x, w, Bt, G, At, 4, dt)
        x = torch.relu(y + s[:,None,None])
    return x.double()
ref = chain(torch.float64, 'direct')
print('F(6x6) matrices in fp64 vs direct (a correctness check of the matrices): max abs %.1e' %
      (chain(torch.float64, 'f6') - ref).abs().max())
for mode in ('direct','f2','f4','f6'):
    y = chain(torch.float32, mode)
    e = (y-ref)
    print(mode, 'rel L2 %.2e  max abs %.2e  (max ref %.2f)' % (e.norm()/ref.norm(), e.abs().max(), ref.abs().max()))
# C=256 single layer error growth
for C in (256,):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(C,24,24, generator=g, dtype=torch.float64).clamp(min=0); w = torch.randn(64,C,3,3,generator=g,dtype=torch.float64)*(2.0/(C*9))**0.5
    r = F.conv2d(x[None], w, None,1,1)[0]
    for mode,(B_,G_,A_,m) in dict(f2=(Bt2,G2,At2,2), f4=(Bt,G,At,4), f6=(Bt6,G6,At6,6)).items():
        y = wino_conv(x.float(), w.float(), B_,G_,A_,m, torch.float32).double()
        print('C=256 single', mode, 'rel L2 %.2e max abs %.2e' % ((y-r).norm()/r.norm(), (y-r).abs().max()))
    y = F.conv2d(x[None].float(), w.float(), None,1,1)[0].double()
    print('C=256 single direct rel L2 %.2e max abs %.2e' % ((y-r).norm()/r.norm(), (y-r).abs().max()))
