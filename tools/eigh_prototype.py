"""CPU prototype (numpy) of the device eigen-solver of cellregmap_amd/csrc/eigh*.hip: Householder
tridiagonalisation -> divide & conquer on the tridiagonal (host-side deflation, secular equation and
Loewner re-computation of the update vector as the device kernels do them) -> back-transformation.
Design aid and cross-check for the HIP port; not part of the product, not the oracle.

    python tools/eigh_prototype.py          # self-test on random / clustered / rank-deficient matrices
"""
import numpy as np

EPS = np.finfo(float).eps
LEAF = 32


def tridiagonalise(A):
    """Unblocked Householder reduction (lower): returns d, e, V (column j = v_j, unit at j+1), tau."""
    A = np.array(A, float)
    n = A.shape[0]
    d = np.zeros(n)
    e = np.zeros(max(n - 1, 0))
    V = np.zeros((n, max(n - 1, 0)))
    tau = np.zeros(max(n - 1, 0))
    for j in range(n - 1):
        x = A[j + 1:, j].copy()
        alpha = x[0]
        xnorm = np.linalg.norm(x[1:])
        if xnorm == 0.0:
            t, beta = 0.0, alpha
            v = np.zeros_like(x)
            v[0] = 1.0
        else:
            beta = -np.copysign(np.hypot(alpha, xnorm), alpha)
            t = (beta - alpha) / beta
            v = x / (alpha - beta)
            v[0] = 1.0
        d[j] = A[j, j]
        e[j] = beta
        tau[j] = t
        V[j + 1:, j] = v
        if t != 0.0:
            T = A[j + 1:, j + 1:]
            w = t * (T @ v)
            w += (-0.5 * t * (w @ v)) * v
            T -= np.outer(v, w) + np.outer(w, v)
    d[n - 1] = A[n - 1, n - 1]
    return d, e, V, tau


def back_transform(V, tau, Z):
    """Z <- H_0 H_1 ... H_{n-2} Z."""
    Z = np.array(Z, float)
    for j in range(V.shape[1] - 1, -1, -1):
        if tau[j] != 0.0:
            v = V[:, j]
            Z -= tau[j] * np.outer(v, v @ Z)
    return Z


def jacobi_leaf(T):
    """Cyclic Jacobi on a small symmetric matrix; returns (lam, Qt) with rows = eigenvectors."""
    A = np.array(T, float)
    n = A.shape[0]
    Vt = np.eye(n)
    for sweep in range(30):
        off = np.sqrt(np.sum(np.tril(A, -1) ** 2))
        if off <= 1e-300 or off <= EPS * np.sqrt(np.sum(np.diag(A) ** 2)) * 1e-3:
            break
        for p in range(n - 1):
            for q in range(p + 1, n):
                apq = A[p, q]
                if abs(apq) <= 1e-300:
                    continue
                theta = (A[q, q] - A[p, p]) / (2.0 * apq)
                t = np.sign(theta) / (abs(theta) + np.sqrt(1.0 + theta * theta)) if theta != 0 else 1.0
                c = 1.0 / np.sqrt(1.0 + t * t)
                s = t * c
                J = np.array([[c, s], [-s, c]])
                A[[p, q], :] = J.T @ A[[p, q], :]
                A[:, [p, q]] = A[:, [p, q]] @ J
                Vt[[p, q], :] = J.T @ Vt[[p, q], :]
    return np.diag(A).copy(), Vt


def secular_root(j, dl, w, rho):
    """Root j of 1 + rho * sum w_i^2 / (dl_i - lam) = 0 as (origin index, tau): lam = dl[origin] + tau.
    Safeguarded 'middle way' iteration (two-pole rational model), bisection when the model steps out."""
    k = dl.shape[0]
    w2 = rho * w * w
    last = j == k - 1

    def setup(o):
        return dl - dl[o]

    if last:
        o = k - 1
        delta = setup(o)
        lo, hi = 0.0, rho * float(np.sum(w * w))
    else:
        gap = dl[j + 1] - dl[j]
        delta = setup(j)
        mid = 0.5 * gap
        gmid = 1.0 + float(np.sum(w2 / (delta - mid)))
        if gmid > 0.0:
            o, lo, hi = j, 0.0, mid
        else:
            o, lo, hi = j + 1, -mid, 0.0
            delta = setup(o)
    # both bracket ends: g(lo) < 0 < g(hi) in exact arithmetic (g increases between the poles)
    tau = 0.5 * (lo + hi)
    for it in range(200):
        den = delta - tau
        terms = w2 / den
        lower = terms[: j + 1]
        upper = terms[j + 1:]
        psi, phi = float(np.sum(lower)), float(np.sum(upper))
        dpsi = float(np.sum(lower / den[: j + 1]))
        dphi = float(np.sum(upper / den[j + 1:]))
        g = 1.0 + psi + phi
        err = 8.0 * EPS * (1.0 + abs(psi) + abs(phi)) + EPS * abs(tau) * (dpsi + dphi)
        if abs(g) <= err:
            break
        if g < 0.0:
            lo = tau
        else:
            hi = tau
        if hi - lo <= 2.0 * EPS * max(abs(lo), abs(hi)):
            tau = lo if abs(lo) > 0 else hi
            if tau == 0.0:
                tau = 0.5 * (lo + hi)
            break
        # model: c + a / (dj - x) + b / (dj1 - x) = 0 matching psi, psi', phi, phi' at tau
        dj = delta[j] - tau
        a = dpsi * dj * dj
        sp = psi - dpsi * dj
        if last:
            c = 1.0 + sp
            x = tau + dj + a / c if c > 0.0 else np.inf     # c + a/(dj - dx) = 0, dx relative to tau
            new = x
        else:
            dj1 = delta[j + 1] - tau
            b = dphi * dj1 * dj1
            sf = phi - dphi * dj1
            c = 1.0 + sp + sf
            # c (dj - s)(dj1 - s) + a (dj1 - s) + b (dj - s) = 0, s = step from tau
            A2 = c
            B2 = -(c * (dj + dj1) + a + b)
            C2 = c * dj * dj1 + a * dj1 + b * dj
            disc = B2 * B2 - 4.0 * A2 * C2
            if A2 == 0.0:
                s = -C2 / B2 if B2 != 0 else np.nan
            elif disc < 0.0:
                s = np.nan
            else:
                q = -0.5 * (B2 + np.copysign(np.sqrt(disc), B2))
                r1 = q / A2
                r2 = C2 / q if q != 0 else np.nan
                cand = [r for r in (r1, r2) if np.isfinite(r) and dj < r < dj1]
                s = cand[0] if cand else np.nan
            new = tau + s
        if not np.isfinite(new) or not (lo < new < hi):
            new = 0.5 * (lo + hi)
        tau = new
    return o, tau


def merge(lamL, lamR, zL, zR, beta):
    """One rank-one merge.  Returns dict with everything the device needs:
    order (sorted positions), rotations [(row_a, row_b, c, s)], nondefl (rows in pole order), defl rows,
    dl, w, rho, roots (origin, tau), U (k x k), lam_new (k roots then deflated values)."""
    n1 = lamL.shape[0]
    D = np.concatenate([lamL, lamR])
    z = np.concatenate([zL, np.sign(beta) * zR if beta != 0 else zR]) / np.sqrt(2.0)
    rho = 2.0 * abs(beta)
    n = D.shape[0]
    order = np.argsort(D, kind="stable")
    Ds = D[order].copy()
    zs = z[order].copy()
    tol = 8.0 * EPS * max(np.abs(Ds).max(), np.abs(zs).max())
    rotations = []
    nondefl, defl = [], []
    if rho * np.abs(zs).max() <= tol:
        defl = list(range(n))
    else:
        pj = -1
        for jj in range(n):
            if rho * abs(zs[jj]) <= tol:
                defl.append(jj)
                continue
            if pj < 0:
                pj = jj
                continue
            s, c = zs[pj], zs[jj]
            tau = np.hypot(c, s)
            t = Ds[jj] - Ds[pj]
            c /= tau
            s = -s / tau
            if abs(t * c * s) <= tol:
                zs[jj] = tau
                zs[pj] = 0.0
                rotations.append((int(order[pj]), int(order[jj]), c, s))
                tnew = Ds[pj] * c * c + Ds[jj] * s * s
                Ds[jj] = Ds[pj] * s * s + Ds[jj] * c * c
                Ds[pj] = tnew
                defl.append(pj)
                pj = jj
            else:
                nondefl.append(pj)
                pj = jj
        if pj >= 0:
            nondefl.append(pj)
    k = len(nondefl)
    out = {"order": order, "rotations": rotations, "nondefl": [int(order[i]) for i in nondefl],
           "defl": [int(order[i]) for i in defl], "k": k, "rho": rho}
    lam_defl = Ds[defl] if defl else np.zeros(0)
    if k == 0:
        out.update(U=np.zeros((0, 0)), lam_new=lam_defl)
        return out
    dl = Ds[nondefl]
    w = zs[nondefl]
    # the deflation may leave dl unsorted by a hair (rotated values); keep the secular solver's order assumption
    assert np.all(np.diff(dl) > 0), "poles must be strictly increasing"
    nrm = np.linalg.norm(w)
    w = w / nrm
    rho_eff = rho * nrm * nrm
    roots = [secular_root(j, dl, w, rho_eff) for j in range(k)]
    org = np.array([r[0] for r in roots])
    tau = np.array([r[1] for r in roots])
    # differences dl_i - lam_j = (dl_i - dl_org_j) - tau_j, to full relative accuracy
    diff = (dl[:, None] - dl[org][None, :]) - tau[None, :]
    # Loewner / Gu-Eisenstat: zhat_i^2 = prod_j (lam_j - dl_i) / (rho prod_{j != i} (dl_j - dl_i))
    zhat = np.empty(k)
    for i in range(k):
        p = -diff[i, i]           # lam_i - dl_i  (j = i factor of the numerator)
        for jx in range(k):
            if jx != i:
                p *= (-diff[i, jx]) / (dl[jx] - dl[i])
        zhat[i] = np.copysign(np.sqrt(abs(p) / rho_eff), w[i])
    U = zhat[:, None] / diff
    U /= np.linalg.norm(U, axis=0, keepdims=True)
    lam_new = dl[org] + tau
    out.update(U=U, lam_new=np.concatenate([lam_new, lam_defl]), dl=dl, w=w, roots=roots)
    return out


def dc_tridiagonal(d, e):
    """Eigen-decomposition of the symmetric tridiagonal (d, e): (lam ascending, Qt rows = eigenvectors)."""
    d = np.array(d, float)
    e = np.array(e, float)
    n = d.shape[0]
    # block tree: split until <= LEAF
    blocks = [(0, n)]
    while any(t - s > LEAF for s, t in blocks):
        nxt = []
        for s, t in blocks:
            if t - s > LEAF:
                m = (s + t) // 2
                nxt += [(s, m), (m, t)]
            else:
                nxt.append((s, t))
        blocks = nxt
    cuts = [t for s, t in blocks[:-1]]
    for c in cuts:
        b = abs(e[c - 1])
        d[c - 1] -= b
        d[c] -= b
    lam = np.zeros(n)
    Qt = np.zeros((n, n))
    for s, t in blocks:
        T = np.diag(d[s:t]) + np.diag(e[s:t - 1], 1) + np.diag(e[s:t - 1], -1)
        lam[s:t], Qt[s:t, s:t] = jacobi_leaf(T)
    # merge adjacent pairs level by level (the pairing follows the splitting tree bottom-up)
    while len(blocks) > 1:
        # merge blocks that were split last: pair up neighbours (s, m), (m, t) of (nearly) equal size
        sizes = [t - s for s, t in blocks]
        smallest = min(sizes)
        nxt = []
        i = 0
        merged_any = False
        while i < len(blocks):
            if i + 1 < len(blocks) and sizes[i] <= smallest + 1 and sizes[i + 1] <= smallest + 1 and not (
                    merged_any and False):
                (s, m), (_, t) = blocks[i], blocks[i + 1]
                beta = e[m - 1]
                r = merge(lam[s:m], lam[m:t], Qt[s:m, m - 1].copy(), Qt[m:t, m].copy(), beta)
                Q = Qt[s:t, s:t].copy()
                for a, b, c, sn in r["rotations"]:
                    ra, rb = Q[a].copy(), Q[b].copy()
                    Q[a] = c * ra + sn * rb      # drot(x = row a, y = row b, c, s): x' = c x + s y ; y' = c y - s x
                    Q[b] = c * rb - sn * ra
                k = r["k"]
                new = np.zeros_like(Q)
                if k:
                    new[:k] = r["U"].T @ Q[r["nondefl"]]
                new[k:] = Q[r["defl"]]
                Qt[s:t, s:t] = new
                lam[s:t] = r["lam_new"]
                nxt.append((s, t))
                i += 2
                merged_any = True
            else:
                nxt.append(blocks[i])
                i += 1
        blocks = nxt
    o = np.argsort(lam, kind="stable")
    return lam[o], Qt[o]


def eigh(A):
    d, e, V, tau = tridiagonalise(A)
    lam, Qt = dc_tridiagonal(d, e)
    Z = back_transform(V, tau, Qt.T)
    return lam, Z


def _check(name, A):
    lam, Z = eigh(A)
    n = A.shape[0]
    ref = np.linalg.eigvalsh(A)
    scale = max(np.abs(ref).max(), 1e-300)
    res = np.abs(A @ Z - Z * lam).max() / scale
    orth = np.abs(Z.T @ Z - np.eye(n)).max()
    ev = np.abs(lam - ref).max() / scale
    print(f"{name:34s} n={n:4d} residual {res:.2e} orthogonality {orth:.2e} eigenvalues {ev:.2e}")
    assert res < 1e-12 * n and orth < 1e-12 * n and ev < 1e-13 * n, name


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 17, 33, 64, 65, 130, 257):
        X = rng.normal(size=(n, n))
        _check("random symmetric", X + X.T)
    X = rng.normal(size=(200, 40))
    _check("rank-deficient Gram (40 of 200)", X @ X.T)
    Q, _ = np.linalg.qr(rng.normal(size=(150, 150)))
    _check("clustered (3 distinct values)", (Q * np.repeat([1.0, 2.0, 3.0], 50)) @ Q.T)
    _check("graded 1e0..1e-12", (Q * np.logspace(0, -12, 150)) @ Q.T)
    _check("identity", np.eye(70))
    _check("diagonal", np.diag(rng.normal(size=90)))
    H = rng.normal(size=(300, 20))
    C = H.T @ H
    Dm = np.r_[np.zeros(5), np.ones(15)]
    _check("block with zero rows/cols (rho = 0)", np.pad(Dm[:, None] * C * Dm[None, :], ((0, 100), (0, 100))))
    T = np.diag(np.full(100, 2.0)) + np.diag(np.full(99, -1.0), 1) + np.diag(np.full(99, -1.0), -1)
    _check("1-2-1 Toeplitz (tridiagonal already)", T)
    W = np.diag(np.abs(np.arange(-10, 11)).astype(float)) + np.diag(np.ones(20), 1) + np.diag(np.ones(20), -1)
    _check("Wilkinson W21+", W)
    print("ok")
