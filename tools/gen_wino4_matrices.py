#!/usr/bin/env python
"""Winograd F(4,3) transform matrices for the interpolation points {0, +-a, +-b, inf} (Toom-Cook construction: A^T and G
are Vandermonde matrices of the points, B^T the transposed inverse of the 6x6 Vandermonde matrix) and their C
expressions for sa-ssd_amd/csrc/conv2d_wino4.hip.  The points (a, b) = (5/8, 3/2) were chosen by a search over
{k/16}: in fp32 they give 4.7x less max-abs error (2.3x less L2 error) than the usual {0, +-1, +-2} on a 256-channel
3x3 layer (7.4e-6 vs 3.5e-5 at |y| ~ 3; direct convolution 9e-7).   python tools/gen_wino4_matrices.py [a b]"""
import sys
from fractions import Fraction as Fr


def matrices(points, m=4, r=3):
    n = m + r - 1
    pts = [Fr(p) for p in points]

    def E(cols):
        M = [[p ** i for i in range(cols)] for p in pts]
        M.append([Fr(0)] * (cols - 1) + [Fr(1)])
        return M
    En = E(n)
    A = [row[:] + [Fr(int(i == j)) for j in range(n)] for i, row in enumerate(En)]
    for c in range(n):
        piv = next(i for i in range(c, n) if A[i][c] != 0)
        A[c], A[piv] = A[piv], A[c]
        pv = A[c][c]
        A[c] = [v / pv for v in A[c]]
        for i in range(n):
            if i != c and A[i][c] != 0:
                f = A[i][c]
                A[i] = [x - f * y for x, y in zip(A[i], A[c])]
    inv = [row[n:] for row in A]
    Bt = [[inv[j][i] for j in range(n)] for i in range(n)]
    G = E(r)
    At = [[E(m)[j][i] for j in range(n)] for i in range(m)]
    return Bt, G, At


def expr(row, var):
    terms = []
    for j, c in enumerate(row):
        if c == 0:
            continue
        f = float(c)
        if c == 1:
            terms.append("+ %s[%d]" % (var, j))
        elif c == -1:
            terms.append("- %s[%d]" % (var, j))
        else:
            terms.append("%s %.9gf * %s[%d]" % ("+" if f > 0 else "-", abs(f), var, j))
    s = " ".join(terms).lstrip("+ ").strip()
    return s if s else "0.f"


def main():
    a, b = (Fr(sys.argv[1]), Fr(sys.argv[2])) if len(sys.argv) > 2 else (Fr(5, 8), Fr(3, 2))
    Bt, G, At = matrices([0, a, -a, b, -b])
    for name, M, var in (("B^T", Bt, "d"), ("G", G, "g"), ("A^T", At, "m")):
        print("// %s (points 0, +-%s, +-%s, inf)" % (name, a, b))
        for i, row in enumerate(M):
            print("//   [%s]" % ", ".join(str(v) for v in row))
        for i, row in enumerate(M):
            print("    o[%d] = %s;" % (i, expr(row, var)))


if __name__ == "__main__":
    main()
