"""Score-test algebra of cellregmap/_math.py restated (oracle; test infrastructure).

Two families, as in the reference:

* the *implicit* forms the scan uses -- K = a*Q0 diag(S0) Q0' + b*I is never
  formed (``QSCov`` _math.py:40-76, ``PMat`` :79-93, ``ScoreStatistic`` :102-128);
  here they are plain functions over a small record, evaluated with exactly the
  reference's sequence of products (three products with Q0 per solve);
* the dense textbook definitions (``P_matrix`` :96-99, ``score_statistic``
  :131-138, ``score_statistic_distr_weights`` :150-160), kept as an independent
  O(n^3) check.

PINNED by the reference's known-answer values (test_math.py:55-83) and by golden
vectors generated from the reference module itself (tests/golden/).
"""
from collections import namedtuple

import numpy as np

LowRankCov = namedtuple("LowRankCov", "Q0 S0 a b")


def lstsq_solve(A, B):
    """The reference's ``rsolve`` (_math.py:33-37): minimum-norm least squares."""
    return np.linalg.lstsq(A, B, rcond=None)[0]


# --- implicit covariance a*Q0 S0 Q0' + b*I ---------------------------------------
def cov_apply(K: LowRankCov, v):
    """K @ v  (_math.py:53-56)."""
    t = K.Q0.T @ v
    t = (K.S0 * t.T).T
    return K.a * (K.Q0 @ t) + K.b * v


def cov_solve(K: LowRankCov, v):
    """K^-1 @ v  (_math.py:58-73): (Q0 R0 Q0'v + v - Q0 Q0'v) / b."""
    shrink = 1.0 / (1.0 + (K.a / K.b) * K.S0)
    t = K.Q0.T @ v
    shrunk = (shrink * t.T).T
    return (K.Q0 @ shrunk + v - K.Q0 @ t) / K.b


# --- implicit projection P = K^-1 - K^-1 X (X'K^-1 X)^-1 X'K^-1 ---------------------
class Projection:
    """State of the reference's ``PMat`` (_math.py:79-93): K^-1 X is cached."""

    def __init__(self, K: LowRankCov, X):
        self.K = K
        self.X = X
        self.KiX = cov_solve(K, X)

    def apply(self, v):
        Kiv = cov_solve(self.K, v)
        coef = lstsq_solve(self.X.T @ self.KiX, self.KiX.T @ v)
        return Kiv - self.KiX @ coef


def score_Q(P: Projection, half_dK, y):
    """Q = 1/2 y'P dK P y with dK = half_dK half_dK'  (_math.py:114-117,
    evaluated left to right like the reference)."""
    Py = P.apply(y)
    return Py.T @ half_dK @ half_dK.T @ Py / 2


def score_F(P: Projection, half_dK):
    """F = 1/2 half_dK' P half_dK  (_math.py:119-124); its eigenvalues are the
    chi-square mixture weights of Q."""
    return half_dK.T @ P.apply(half_dK) / 2


# --- dense definitions -----------------------------------------------------------------
def dense_P(X, K):
    """_math.py:96-99."""
    KiX = np.linalg.solve(K, X)
    return np.linalg.inv(K) - KiX @ np.linalg.solve(X.T @ KiX, KiX.T)


def dense_Q(y, X, K, dK):
    """_math.py:131-138."""
    P = dense_P(X, K)
    return y.T @ P @ dK @ P @ y / 2


def dense_weights(X, K, dK):
    """_math.py:150-160: non-zero eigenvalues of 1/2 sqrt(P) dK sqrt(P)."""
    from scipy.linalg import sqrtm

    P = dense_P(X, K)
    rP = sqrtm(P)
    w = np.linalg.eigvalsh(rP @ dK @ rP) / 2
    return w[w > 1e-16]


def liu_params(q, weights):
    """_math.py:163-180 (modified Liu parameters)."""
    from .davies import liu_sf

    k = len(weights)
    pv, dof_x, _, info = liu_sf(q, weights, [1] * k, [0] * k, True)
    return {"pv": pv, "mu_q": info["mu_q"], "sigma_q": info["sigma_q"], "dof_x": dof_x}
