"""fp32 error study of Winograd F(4x4,3x3) for the 1024-channel 8x8 trunk convs, next to the F(2x2,3x3)
the step runs and the direct form.  Same method as tools/winograd_f24.py (whose ``matrices`` builds A^T, G, B^T for a point
set): per output element, the transformed products are accumulated over C channels in fp32 in channel order -- what the
batched MFMA GEMM does -- and compared with the fp64 correlation.  Three uses of a form are measured:

  forward / data gradient   y  = A^T [sum_c (G g_c G^T) (.) (B^T d_c B)] A              reduction over C = 1024 channels
  weight gradient           gw = G^T [sum_p (A dy_p A^T) (.) (B^T d_p B)] G             reduction over P tiles (128 at F(4x4), 512 at F(2x2))

Run: python tools/winograd_f43.py   (sympy + numpy; nothing here runs on the training path).  Results: profiles/r05_f43_study.md."""
import numpy as np
import sympy as sp

from winograd_f24 import matrices

f32 = np.float32


def _np(M):
    return np.array(M.tolist(), dtype=np.float64).astype(f32)


def fwd_error(AT, G, BT, m, C=1024, seed=0):
    n = m + 2
    rng = np.random.RandomState(seed)
    g, d = (rng.randn(C, 3, 3) * (2.0 / (9 * C)) ** 0.5), rng.randn(C, n, n)        # He-scaled filter, unit-variance input
    ref = np.array([[(g * d[:, i:i + 3, j:j + 3]).sum() for j in range(m)] for i in range(m)])
    A, Gm, B = _np(AT), _np(G), _np(BT)
    U = np.einsum('ik,ckl,jl->cij', Gm, g.astype(f32), Gm).astype(f32)
    V = np.einsum('ik,ckl,jl->cij', B, d.astype(f32), B).astype(f32)
    M = np.zeros((n, n), f32)
    for c in range(C):
        M = (M + U[c] * V[c]).astype(f32)
    Y = (A @ M @ A.T).astype(f32)
    direct = np.zeros((m, m), f32)
    for c in range(C):
        for i in range(m):
            for j in range(m):
                direct[i, j] = f32(direct[i, j] + (g[c].astype(f32) * d[c, i:i + 3, j:j + 3].astype(f32)).sum(dtype=f32))
    s = np.abs(ref).max()
    return float(np.abs(Y - ref).max() / s), float(np.abs(direct - ref).max() / s)


def wgrad_error(AT, G, BT, m, P, seed=0):
    n = m + 2
    rng = np.random.RandomState(seed)
    dy, d = rng.randn(P, m, m) * 0.1, rng.randn(P, n, n)
    ref = np.array([[(dy * d[:, a:a + m, b:b + m]).sum() for b in range(3)] for a in range(3)])
    A, Gm, B = _np(AT), _np(G), _np(BT)
    Yt = np.einsum('ki,pkl,lj->pij', A, dy.astype(f32), A).astype(f32)           # A dy A^T with A = (A^T)^T : (n x m)(m x m)(m x n)
    V = np.einsum('ik,pkl,jl->pij', B, d.astype(f32), B).astype(f32)
    T = np.zeros((n, n), f32)
    for p in range(P):
        T = (T + Yt[p] * V[p]).astype(f32)
    gw = (Gm.T @ T @ Gm).astype(f32)
    direct = np.zeros((3, 3), f32)
    for p in range(P):
        for a in range(3):
            for b in range(3):
                direct[a, b] = f32(direct[a, b] + (dy[p].astype(f32) * d[p, a:a + m, b:b + m].astype(f32)).sum(dtype=f32))
    s = np.abs(ref).max()
    return float(np.abs(gw - ref).max() / s), float(np.abs(direct - ref).max() / s)


if __name__ == '__main__':
    H = sp.Rational(1, 2)
    forms = [('F(2x2,3x3) points 0, 1, -1', 2, [0, 1, -1]),
             ('F(4x4,3x3) points 0, 1, -1, 2, -2', 4, [0, 1, -1, 2, -2]),
             ('F(4x4,3x3) points 0, 1, -1, 1/2, -1/2', 4, [0, 1, -1, H, -H]),
             ('F(4x4,3x3) points 0, 1, -1, 1/2, -2', 4, [0, 1, -1, H, -2])]
    print('relative max error (max |err| / max |exact|), mean over 8 seeds; reduction in fp32, channel / tile order')
    for name, m, pts in forms:
        AT, G, BT = matrices(m, 3, pts)
        fe = np.array([fwd_error(AT, G, BT, m, seed=s) for s in range(8)]).mean(0)
        P = 32 * (8 // m) ** 2
        we = np.array([wgrad_error(AT, G, BT, m, P, seed=s) for s in range(8)]).mean(0)
        print('%-42s forward (C = 1024): winograd %.2e  direct %.2e  (x%.1f) | weight gradient (P = %d): winograd %.2e  direct %.2e  (x%.1f)'
              % (name, fe[0], fe[1], fe[0] / fe[1], P, we[0], we[1], we[0] / we[1]))
