"""Derivation of the Winograd F(2, 4) matrices used by csrc/igemm.hip (w24_bt / w24_g / w24_at and their adjoints) and a small
fp32 error study of the candidate interpolation point sets.

    y = A^T [(G g) (.) (B^T d)]        g: 4 filter taps, d: 5 inputs, y: 2 outputs of the correlation y[i] = sum_k g[k] d[i + k]

A^T and G follow from evaluating the polynomials at the points (the last point is infinity); B^T is the unique solution of the
exactness conditions.  Run: python tools/winograd_f24.py   (needs sympy; nothing here runs on the training path)."""
import numpy as np
import sympy as sp


def matrices(m, r, pts):
    n = m + r - 1
    a = [sp.Rational(p) for p in pts]
    AT, G = sp.zeros(m, n), sp.zeros(n, r)
    for t in range(n - 1):
        F = sp.prod([a[t] - a[j] for j in range(n - 1) if j != t])
        for i in range(m):
            AT[i, t] = a[t] ** i
        for k in range(r):
            G[t, k] = a[t] ** k / F
    AT[m - 1, n - 1] = 1
    G[n - 1, r - 1] = 1
    X = sp.symbols('b0:%d' % (n * n))
    eqs = [sum(AT[i, t] * G[t, k] * X[t * n + j] for t in range(n)) - (1 if j == i + k else 0)
           for i in range(m) for k in range(r) for j in range(n)]
    sol = sp.solve(eqs, X, dict=True)[0]
    BT = sp.Matrix(n, n, lambda i, j: sol[X[i * n + j]])
    return AT, G, BT


def fp32_error(AT, G, BT, C=256, seed=0):
    """max |error| of one 2x2 output tile summed over C channels in fp32 (2-D form), Winograd vs direct, against fp64"""
    f32 = np.float32
    ATn, Gn, BTn = [np.array(M.tolist(), dtype=np.float64) for M in (AT, G, BT)]
    rng = np.random.RandomState(seed)
    g2, d2 = rng.randn(C, 4, 4) * 0.02, rng.randn(C, 5, 5)
    ref = np.array([[(g2 * d2[:, i:i + 4, j:j + 4]).sum() for j in range(2)] for i in range(2)])
    U = np.einsum('ik,ckl,jl->cij', Gn.astype(f32), g2.astype(f32), Gn.astype(f32)).astype(f32)
    V = np.einsum('ik,ckl,jl->cij', BTn.astype(f32), d2.astype(f32), BTn.astype(f32)).astype(f32)
    M = np.zeros((5, 5), f32)
    for c in range(C):
        M = (M + U[c] * V[c]).astype(f32)
    Y = (ATn.astype(f32) @ M @ ATn.astype(f32).T).astype(f32)
    direct = np.zeros((2, 2), f32)
    for i in range(2):
        for j in range(2):
            acc = f32(0)
            for c in range(C):
                acc = f32(acc + (g2[c].astype(f32) * d2[c, i:i + 4, j:j + 4].astype(f32)).sum(dtype=f32))
            direct[i, j] = acc
    return float(np.abs(Y - ref).max()), float(np.abs(direct - ref).max()), float(np.abs(ref).max())


if __name__ == '__main__':
    for pts in ([0, 1, -1, -2], [0, 1, -1, 2], [0, 1, -1, sp.Rational(1, 2)], [0, sp.Rational(1, 2), -sp.Rational(1, 2), 1]):
        AT, G, BT = matrices(2, 4, pts)
        errs = [fp32_error(AT, G, BT, seed=s) for s in range(8)]
        print('points %s + inf: fp32 error winograd %.2e  direct %.2e  (|y| ~ %.2f)' % (
            pts, np.mean([e[0] for e in errs]), np.mean([e[1] for e in errs]), np.mean([e[2] for e in errs])))
        if pts == [0, 1, -1, -2]:
            print('A^T =', AT.tolist(), '\nG   =', G.tolist(), '\nB^T =', BT.tolist())
