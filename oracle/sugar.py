"""numpy-sugar pieces the reference path calls (oracle; test infrastructure only).

Call sites in the reference: cellregmap/_cellregmap.py:16-17,106,114,129,415,540,544
and cellregmap/_math.py:29,54,73.  The algorithm of the two decompositions is
documented by the reference's in-tree twins, cellregmap/_math.py:204-235
(``economic_qs``) and :238-256 (``economic_qs_linear``).
"""
import numpy as np

_FINFO = np.finfo(float)


class _Epsilon:
    """numpy_sugar.epsilon: three machine-precision landmarks.

    ``super_tiny`` is pinned by cellregmap/test/test_fixed_gxe.py:108
    (2.2250738585072014e-308 == finfo.tiny).
    """

    super_tiny = float(_FINFO.tiny)
    tiny = float(_FINFO.eps)
    small = float(np.sqrt(_FINFO.eps))


epsilon = _Epsilon()

#: optimix/glimix-core bound on the logistic variable: log(finfo.max)
LOGMAX = float(np.log(_FINFO.max))


def ddot(L, R, left=None):
    """diag(L) @ R if L is a vector, else L @ diag(R)  (numpy_sugar.ddot)."""
    L = np.asarray(L, float)
    R = np.asarray(R, float)
    if left is None:
        left = L.ndim == 1
    if left:
        return (L[:, None] * R) if R.ndim == 2 else L * R
    return L * R[None, :]


def economic_svd(G, eps=epsilon.small):
    """Thin SVD with singular values below sqrt(machine eps) dropped.

    numpy_sugar.linalg.economic_svd; used at _cellregmap.py:540 and inside
    glimix-core's LMM to reduce the covariates matrix.
    """
    from scipy.linalg import svd

    G = np.asarray(G, float)
    U, s, Vt = svd(G, full_matrices=False, check_finite=False)
    keep = s >= eps
    return U[:, keep], s[keep], Vt[keep, :]


def economic_qs(K, eps=epsilon.small):
    """Eigendecomposition of a symmetric PSD matrix split at ``eps``.

    Follows cellregmap/_math.py:204-235 including the scipy retry heuristic
    (:223-228).  Returns ``((Q0, Q1), S0)``.
    """
    K = np.asarray(K, float)
    S, Q = np.linalg.eigh(K)
    first_row_extreme = abs(max(Q[0].min(), Q[0].max(), key=abs))
    k_extreme = abs(max(K.min(), K.max(), key=abs))
    if first_row_extreme < eps and k_extreme >= eps:
        from scipy.linalg import eigh as sp_eigh

        S, Q = sp_eigh(K)
    ok = S >= eps
    return (Q[:, ok], Q[:, ~ok]), S[ok]


def economic_qs_linear(G, return_q1=True):
    """Economic eigendecomposition of ``G @ G.T`` from the half matrix ``G``.

    Follows cellregmap/_math.py:238-256: thin SVD when rows > cols (no
    thresholding of the squared singular values), otherwise ``economic_qs`` of
    the n x n product.  ``return_q1=False`` (the only form the reference path
    uses, _cellregmap.py:106,114,129) yields ``((Q0,), S0)``.
    """
    G = np.asarray(G, float)
    if G.shape[0] > G.shape[1]:
        Q, s, _ = np.linalg.svd(G, full_matrices=return_q1)
        S0 = s ** 2
        if not return_q1:
            return (Q,), S0
        r = S0.shape[0]
        return (Q[:, :r], Q[:, r:]), S0
    (Q0, Q1), S0 = economic_qs(G @ G.T)
    if not return_q1:
        return (Q0,), S0
    return (Q0, Q1), S0
