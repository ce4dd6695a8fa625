"""numpy prototype of the constructor's two-stage eigen-solver (cellregmap_amd/csrc/eigh2_*.hip follow it statement
for statement; tests/test_eigh2_prototype_cpu.py keeps it honest against LAPACK).

The background's grid points are ONE family  A(rho) = D(rho) C D(rho),  D = diag(sqrt(rho) I_k1, sqrt(1 - rho) I)
(cellregmap/_cellregmap.py:101-131: hS(rho) = [sqrt(rho) E1, sqrt(1 - rho) B]; C its Gram matrix at unit weights).

  stage 1   C -> band of half-width w, ONCE for all grid points: the first panel is exactly the k1 leading columns, so
            every reflector acts on rows >= k1 only, Q1 = diag(I_k1, Q2) commutes with D(rho) and
            Q1' A(rho) Q1 = D(rho) (Q1' C Q1) D(rho) is the same band, rescaled.  Panel QR + two-sided compact-WY
            update: the n^3 work is three products per panel (matrix pipe).
  stage 2   band -> tridiagonal per grid point by bulge chasing (column-wise elimination: sweep s annihilates column s
            below its sub-diagonal with a reflector of length <= w and chases the fill down the band, one reflector
            per w rows).  O(n^2 w) flops, no n^3 term.
  D&C       on the tridiagonals (eigh_dc.hip, unchanged).
  back 2    Z <- Q2(rho) Z: the sweeps' reflectors grouped g sweeps at a time per chain position into compact-WY
            blocks over windows of w + g - 1 rows; order: sweep blocks last to first, chain positions ascending.
  back 1    Z <- Q1 Z with the panels' compact-WY blocks (shared by all grid points).
"""
import numpy as np


def house(x):
    """LAPACK dlarfg: (v, tau, beta) with v[0] = 1, (I - tau v v') x = beta e1."""
    alpha = x[0]
    xnorm2 = float(np.dot(x[1:], x[1:]))
    v = np.zeros_like(x)
    v[0] = 1.0
    if xnorm2 == 0.0:
        return v, 0.0, alpha
    nrm = np.sqrt(alpha * alpha + xnorm2)
    beta = -nrm if alpha >= 0.0 else nrm
    tau = (beta - alpha) / beta
    v[1:] = x[1:] / (alpha - beta)
    return v, tau, beta


def larft(V, tau):
    """Forward column-wise T with H_0 H_1 ... H_{k-1} = I - V T V'."""
    k = V.shape[1]
    T = np.zeros((k, k))
    for i in range(k):
        T[i, i] = tau[i]
        if i > 0:
            T[:i, i] = -tau[i] * (T[:i, :i] @ (V[:, :i].T @ V[:, i]))
    return T


def panel_qr(P):
    """Householder QR of a tall panel; returns V (unit lower trapezoidal), tau, R (upper triangular, k x k)."""
    P = P.copy()
    m, k = P.shape
    V = np.zeros((m, k))
    tau = np.zeros(k)
    for j in range(min(m, k)):
        v, t, beta = house(P[j:, j])
        V[j:, j] = v
        tau[j] = t
        P[j, j] = beta
        P[j + 1:, j] = 0.0
        if j + 1 < k:
            P[j:, j + 1:] -= t * np.outer(v, v @ P[j:, j + 1:])
    return V, tau, np.triu(P[:k])


def stage1(C, k1, w):
    """Dense symmetric C -> band (half-width w >= k1 columns of the first panel).  Returns (band as a dense matrix,
    [(row0, V, T)] -- reflector blocks acting on rows row0 ...)."""
    A = C.copy()
    n = A.shape[0]
    blocks = []
    c0, width = 0, k1 if k1 > 0 else w
    while c0 + width < n - 1:
        r0 = c0 + width            # reflectors act on rows r0 ..
        V, tau, R = panel_qr(A[r0:, c0:c0 + width])
        T = larft(V, tau)
        # panel columns: R on top, zeros below (and the mirror image)
        A[r0:, c0:c0 + width] = 0.0
        A[r0:r0 + R.shape[0], c0:c0 + width] = R
        A[c0:c0 + width, r0:] = A[r0:, c0:c0 + width].T
        # two-sided update of the trailing matrix: A22 <- (I - V T' V') A22 (I - V T V')
        A22 = A[r0:, r0:]
        Y = A22 @ V @ T                        # product 1 (n x n x width)
        Z = Y - 0.5 * V @ (T.T @ (V.T @ Y))
        A22 -= V @ Z.T + Z @ V.T               # products 2, 3 (rank-2 width update)
        blocks.append((r0, V, T))
        c0, width = r0, w
    return A, blocks


def chase(A, w):
    """Symmetric band matrix (dense storage, half-width w) -> tridiagonal.  Returns d, e and the reflectors
    {(s, k): (r0, v, tau)} (sweep s, chain position k; rows r0 = s + 1 + k w ...).  The device keeps the band in lower
    band storage AB[c][row - c] with room for the fill (2 w entries per column); the arithmetic is the same."""
    A = A.copy()
    n = A.shape[0]
    refl = {}
    for s in range(n - 2):
        # reflector of the sweep's own column
        r0 = s + 1
        L = min(w, n - r0)
        v, tau, beta = house(A[r0:r0 + L, s].copy())
        A[r0:r0 + L, s] = 0.0
        A[r0, s] = beta
        A[s, r0:r0 + L] = A[r0:r0 + L, s]
        k = 0
        while True:
            refl[(s, k)] = (r0, v, tau)
            # (a) two-sided update of the diagonal block
            D = A[r0:r0 + L, r0:r0 + L]
            p = tau * (D @ v)
            q = p - 0.5 * tau * np.dot(p, v) * v
            D -= np.outer(v, q) + np.outer(q, v)
            # (b) the block below: right-multiplied, first column annihilated, the rest left-multiplied
            r1 = r0 + L
            L1 = min(w, n - r1)
            if L1 <= 0:
                break
            B = A[r1:r1 + L1, r0:r0 + L]
            B -= tau * np.outer(B @ v, v)
            v1, tau1, beta1 = house(B[:, 0].copy())
            B[0, 0] = beta1
            B[1:, 0] = 0.0
            if L > 1:
                B[:, 1:] -= tau1 * np.outer(v1, v1 @ B[:, 1:])
            A[r0:r0 + L, r1:r1 + L1] = B.T
            r0, L, v, tau = r1, L1, v1, tau1
            k += 1
    return np.diag(A).copy(), np.diag(A, -1).copy(), refl


def back2(Z, refl, n, w, g):
    """Z <- Q2 Z with Q2 = prod over sweeps (ascending), chain positions of H(s, k), regrouped: sweep blocks of g,
    inside a block chain positions DESCENDING (as factors of Q2), sweeps ascending inside a group."""
    Z = Z.copy()
    nsweeps = n - 2
    kmax = max(k for _, k in refl) if refl else -1
    for s0 in range((nsweeps - 1) // g * g, -1, -g):          # rightmost factors first: last sweep block first
        for k in range(0, kmax + 1):                           # ... and inside it ascending chain position
            members = [s for s in range(s0, min(s0 + g, nsweeps)) if (s, k) in refl]
            if not members:
                continue
            lo = members[0] + 1 + k * w
            hi = max(refl[(s, k)][0] + len(refl[(s, k)][1]) for s in members)
            V = np.zeros((hi - lo, len(members)))
            tau = np.zeros(len(members))
            for j, s in enumerate(members):
                r0, v, t = refl[(s, k)]
                V[r0 - lo:r0 - lo + len(v), j] = v
                tau[j] = t
            T = larft(V, tau)
            Z[lo:hi] -= V @ (T @ (V.T @ Z[lo:hi]))
    return Z


def back1(Z, blocks):
    Z = Z.copy()
    for r0, V, T in reversed(blocks):
        Z[r0:] -= V @ (T @ (V.T @ Z[r0:]))
    return Z


def eigh_family(C, k1, rhos, w=8, g=None):
    """Eigen-decompositions of D(rho) C D(rho) for every rho; returns [(lam ascending, Z columns)]."""
    from scipy.linalg import eigh_tridiagonal

    n = C.shape[0]
    g = w if g is None else g
    Band, blocks = stage1(C, k1, w)
    out = []
    for rho in rhos:
        dscale = np.r_[np.full(k1, np.sqrt(rho)), np.full(n - k1, np.sqrt(1.0 - rho))]
        d, e, refl = chase(Band * np.outer(dscale, dscale), w)
        lam, Zt = eigh_tridiagonal(d, e)
        Z = back1(back2(Zt, refl, n, w, g), blocks)
        out.append((lam, Z))
    return out


if __name__ == "__main__":
    rng = np.random.default_rng(3)
    for n, k1, w, g in ((61, 5, 8, 8), (97, 7, 8, 5), (130, 16, 16, 16), (50, 3, 4, 4), (40, 0, 8, 8)):
        H = rng.normal(size=(n + 30, n))
        C = H.T @ H
        rhos = [0.0, 0.3, 0.9]
        for rho, (lam, Z) in zip(rhos, eigh_family(C, k1, rhos, w=w, g=g)):
            dscale = np.r_[np.full(k1, np.sqrt(rho)), np.full(n - k1, np.sqrt(1.0 - rho))]
            A = C * np.outer(dscale, dscale)
            ref = np.linalg.eigvalsh(A)
            print(n, k1, w, g, rho, "eig %.1e" % (np.abs(lam - ref).max() / np.abs(ref).max()),
                  "orth %.1e" % np.abs(Z.T @ Z - np.eye(n)).max(),
                  "resid %.1e" % (np.abs(A @ Z - Z * lam).max() / np.abs(ref).max()))
