"""chiscore ``davies_pvalue`` / ``liu_sf`` restated (oracle; test infrastructure only).

Reference call sites: ``davies_pvalue(Q, F, True)`` at
cellregmap/_cellregmap.py:333,435; ``liu_sf`` at cellregmap/_math.py:169,179.
chiscore (>= 0.2.3) and chi2comb are absent from this image; chiscore is a
port of SKAT's ``Get_Lambda`` / ``Get_PValue.Lambda`` and that published
procedure is what is restated here:

    lam  = eigvalsh(F)                               (lower triangle)
    lam  = lam[lam > mean(lam[lam >= 0]) / 1e5]
    p    = 1 - qfc(Q; lam, dof 1, nc 0, sigma 0, lim 10000, acc 1e-6)
    p    = liu_mod(Q, lam)   if len(lam) == 1 or p > 1 or p <= 0

A non-zero ``ifault`` only clears ``Is_Converged`` (as in SKAT); the p-value
is replaced by the modified-Liu value only when it is outside (0, 1] or when a
single eigenvalue survives the filter.  **Parity unpinned** for the Davies
branch; the Liu branch is pinned by cellregmap/test/test_math.py:76-83.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

DAVIES_LIM = 10000
DAVIES_ACC = 1e-6


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libcrm_oracle.so")
        if not os.path.exists(path):
            import subprocess

            subprocess.check_call(["make", "-C", _HERE, "libcrm_oracle.so"])
        lib = ctypes.CDLL(path)
        lib.crm_oracle_qfc.restype = ctypes.c_int
        lib.crm_oracle_qfc.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
            ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ]
        _LIB = lib
    return _LIB


def qfc(lam, q, dof=None, nc=None, sigma=0.0, lim=DAVIES_LIM, acc=DAVIES_ACC):
    """P[sum lam_j chi2(dof_j, nc_j) + sigma N(0,1) < q] by Davies' method.

    Returns (cdf, ifault, trace[7])."""
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    r = lam.shape[0]
    dof = np.ones(r, np.int32) if dof is None else np.ascontiguousarray(dof, np.int32)
    nc = np.zeros(r) if nc is None else np.ascontiguousarray(nc, np.float64)
    trace = np.zeros(7)
    ifault = ctypes.c_int(0)
    res = ctypes.c_double(0.0)
    _lib().crm_oracle_qfc(
        lam.ctypes.data, nc.ctypes.data, dof.ctypes.data, r, float(sigma), float(q),
        int(lim), float(acc), trace.ctypes.data, ctypes.byref(ifault), ctypes.byref(res),
    )
    return res.value, ifault.value, trace


def liu_sf(t, lambs, dofs, deltas, kurtosis=False):
    """Liu-Tang-Zhang (2009) survival function; ``kurtosis=True`` is the
    Lee-Wu-Lin (2012) modification.  Returns (sf, dof_x, delta_x, info)."""
    from scipy.stats import ncx2

    t = np.asarray(t, float)
    lambs = np.asarray(lambs, float)
    dofs = np.asarray(dofs, float)
    deltas = np.asarray(deltas, float)
    c = {}
    for i in range(1, 5):
        li = lambs ** i
        c[i] = np.sum(li * dofs) + i * np.sum(li * deltas)
    s1 = c[3] / np.sqrt(c[2]) ** 3
    s2 = c[4] / c[2] ** 2
    s12 = s1 ** 2
    if s12 > s2:
        a = 1.0 / (s1 - np.sqrt(s12 - s2))
        delta_x = s1 * a ** 3 - a ** 2
        dof_x = a ** 2 - 2.0 * delta_x
    else:
        delta_x = 0.0
        if kurtosis:
            a = 1.0 / np.sqrt(s2)
            dof_x = 1.0 / s2
        else:
            a = 1.0 / s1
            dof_x = 1.0 / s12
    mu_q = c[1]
    sigma_q = np.sqrt(2.0 * c[2])
    mu_x = dof_x + delta_x
    sigma_x = np.sqrt(2.0 * (dof_x + 2.0 * delta_x))
    t_star = (t - mu_q) / sigma_q
    tfinal = t_star * sigma_x + mu_x
    sf = ncx2.sf(tfinal, dof_x, np.maximum(delta_x, 1e-9))
    return sf, dof_x, delta_x, {"mu_q": mu_q, "sigma_q": sigma_q}


def filter_weights(F):
    """SKAT ``Get_Lambda``: eigenvalues of F above mean(non-negative)/1e5."""
    lam = np.linalg.eigvalsh(np.asarray(F, float))
    nonneg = lam[lam >= 0]
    keep = lam > nonneg.mean() / 100000.0 if nonneg.size else np.zeros(lam.shape, bool)
    if not keep.any():
        raise RuntimeError("No eigenvalue is bigger than 0.")
    return lam[keep]


def pvalue_from_weights(q, lam):
    """SKAT ``Get_PValue.Lambda`` for one statistic. Returns (p, info)."""
    lam = np.asarray(lam, float)
    p_liu = float(liu_sf(q, lam, np.ones(lam.size), np.zeros(lam.size), True)[0])
    cdf, ifault, trace = qfc(lam, q)
    p = 1.0 - cdf
    converged = 1
    if lam.size == 1:
        p = p_liu
    elif ifault != 0:
        converged = 0
    if p > 1.0 or p <= 0.0:
        converged = 0
        p = p_liu
    return p, {"liu_pval": p_liu, "Is_Converged": converged, "ifault": ifault, "trace": trace}


def davies_pvalue(q, w, return_info=False):
    """chiscore.davies_pvalue(q, w, return_info)."""
    lam = filter_weights(w)
    p, info = pvalue_from_weights(float(np.atleast_1d(q)[0]), lam)
    if return_info:
        return p, info
    return p
