"""glimix-core ``LMM`` / ``FastScanner`` restated (oracle; test infrastructure only).

Reference call sites (cellregmap/_cellregmap.py): REML null fits inside the
interaction scan :351-357 (``LMM(y, X, QS, restricted=True).fit(verbose=False)``,
``.lml()``, ``.v0``, ``.v1`` :367-369,:382-383); ML fits for the association LRT
:254-255, :274-275, :292-293; ``get_fast_scanner().fast_scan(G)['lml']`` :308-309.

Model:  y ~ N(X b, s * [(1-d) * Q0 diag(S0) Q0' + d * I]),  d = logistic(x).

glimix-core (>= 3.1.12, the version that accepts ``QS = ((Q0,), S0)`` without
the complement Q1) is absent from this image; this restates its published
model: the fixed effects and the scale are profiled out in closed form, the
remaining scalar x = logit(d) is maximised by brent-search with
rtol = atol = 1e-6 over (-log(max float), +log(max float)), d clipped to
[eps, 1-eps].  The complement of span(Q0) is handled by differences:

    u' Kt^-1 v = sum_j tu_j tv_j / ((1-d) S0_j + d) + (u'v - tu'tv) / d
    log|Kt|    = sum_j log((1-d) S0_j + d) + (n - r) log d

with tu = Q0'u.  **Parity unpinned** (no golden vector in the reference for any
LMM output); ``tests/test_oracle_lmm.py`` checks the likelihood against a dense
brute-force evaluation.
"""
import numpy as np

from . import brent
from .sugar import LOGMAX, economic_svd, epsilon

LOG2PI = float(np.log(2.0 * np.pi))


def _logistic(x):
    if x > 0.0:
        v = 1.0 / (1.0 + np.exp(-x))
    else:
        v = np.exp(x)
        v = v / (v + 1.0)
    return min(max(v, epsilon.tiny), 1.0 - epsilon.tiny)


def _rsolve(A, b):
    """The fixed effects' solve.  glimix-core routes it through numpy_sugar.linalg.rsolve, a truncated ``lstsq``; which
    ``rcond`` that wrapper passes is NOT recoverable here (recalled as sqrt(eps), i.e. singular values of X'K^-1X below
    1.5e-8 of the largest are cut; the reference's own in-tree twin, cellregmap/_math.py:33-37, uses ``rcond=None`` = eps
    times the dimension).  The two only differ for a direction of [W, g] whose singular value is below ~1e-4 of the largest
    -- covariates on wildly different scales, or a variant collinear with W to seven digits -- and nothing in the
    reference's tests reaches there; this restatement takes ``rcond=None`` like the in-tree twin.  Consequence worth
    knowing: a direction between sqrt(eps) (economic_svd's absolute rule, which decides the rank and
    the degrees of freedom) and that relative cut-off is counted in df while its beta is zeroed."""
    return np.linalg.lstsq(A, b, rcond=None)[0]


class LMM:
    """Profile-likelihood LMM over a fixed economic eigendecomposition.

    Every construction redoes the rotations Q0'y and Q0'X, exactly as the
    reference's per-(variant, rho) ``LMM(...)`` objects do; that is the cost the
    reference pays and what ``bench.py``'s cpu_baseline times.
    """

    def __init__(self, y, X, QS, restricted=False):
        y = np.asarray(y, float).ravel()
        if not np.all(np.isfinite(y)):
            raise ValueError("There are non-finite values in the outcome.")
        if y.size == 0:
            raise ValueError("The outcome array is empty.")
        X = np.atleast_2d(np.asarray(X, float).T).T
        if not np.all(np.isfinite(X)):
            raise ValueError("There are non-finite values in the covariates matrix.")
        Q0 = QS[0][0]
        S0 = np.asarray(QS[1], float)
        if Q0.shape[0] != y.shape[0]:
            raise ValueError("Sample size differs between outcome and covariance decomposition.")
        if X.shape[0] != y.shape[0]:
            raise ValueError("Sample size differs between outcome and covariates.")
        self._y = y
        self._X = X
        self._n = y.shape[0]
        self._Q0 = Q0
        self._S0 = S0
        self._restricted = bool(restricted)
        U, s, Vt = economic_svd(X)
        tX = U * s  # n x rank basis of span(X); beta lives in this basis
        self._tX = tX
        self._Vt = Vt
        self._s = s
        # plain inner products
        self._yy = float(y @ y)
        self._Xy = tX.T @ y
        self._XX = tX.T @ tX
        # rotations (the O(n r) part)
        self._ty = Q0.T @ y
        self._tXr = Q0.T @ tX
        self._tyty = float(self._ty @ self._ty)
        self._tXty = self._tXr.T @ self._ty
        self._tXtX = self._tXr.T @ self._tXr
        self._x = 0.0  # logit(0.5)
        self._tbeta = np.zeros(tX.shape[1])
        self._scale = 1.0
        self._update()

    # -- geometry -----------------------------------------------------------
    @property
    def _df(self):
        return self._n - self._tX.shape[1] if self._restricted else self._n

    @property
    def delta(self):
        return _logistic(self._x)

    def _terms(self, delta):
        D = (1.0 - delta) * self._S0 + delta
        w = 1.0 / D
        ty, tX = self._ty, self._tXr
        yKy = float((ty * w) @ ty) + (self._yy - self._tyty) / delta
        XKy = tX.T @ (w * ty) + (self._Xy - self._tXty) / delta
        XKX = (tX.T * w) @ tX + (self._XX - self._tXtX) / delta
        logdet = float(np.log(D).sum()) + (self._n - self._S0.shape[0]) * np.log(delta)
        return yKy, XKy, XKX, logdet

    def _update(self):
        delta = self.delta
        yKy, XKy, XKX, logdet = self._terms(delta)
        self._tbeta = _rsolve(XKX, XKy)
        self._scale = max((yKy - float(XKy @ self._tbeta)) / self._df, epsilon.small)
        self._cache = (XKX, logdet)

    # -- likelihood -----------------------------------------------------------
    def lml(self):
        XKX, logdet = self._cache
        s = self._scale
        df = self._df
        val = -0.5 * (df * LOG2PI + df + self._n * np.log(s) + logdet)
        if self._restricted:
            sgn0, ld0 = np.linalg.slogdet(self._XX)
            sgn1, ld1 = np.linalg.slogdet(XKX / s)
            if sgn0 != 1.0 or sgn1 != 1.0:
                raise ValueError("The determinant of X'X / H should be positive.")
            val += 0.5 * (ld0 - ld1)
        return float(val)

    def _neg_lml_at(self, x):
        self._x = float(x)
        self._update()
        return -self.lml()

    # -- derivative of the profiled objective (for the polish) ---------------------------------
    def _neg_lml_grad_at(self, x):
        """d(-lml)/dx at x = logit(delta), beta and scale profiled out (envelope theorem)."""
        self._x = float(x)
        delta = self.delta
        if delta <= epsilon.tiny or delta >= 1.0 - epsilon.tiny:
            return 0.0
        S0, ty, tX = self._S0, self._ty, self._tXr
        D = (1.0 - delta) * S0 + delta
        w = 1.0 / D
        w2 = (1.0 - S0) * w * w
        yKy, XKy, XKX, _ = self._terms(delta)
        i2 = 1.0 / (delta * delta)
        dyKy = -float((ty * w2) @ ty) - (self._yy - self._tyty) * i2
        dXKy = -(tX.T @ (w2 * ty)) - (self._Xy - self._tXty) * i2
        dXKX = -((tX.T * w2) @ tX) - (self._XX - self._tXtX) * i2
        dlogdet = float(((1.0 - S0) * w).sum()) + (self._n - S0.shape[0]) / delta
        beta = _rsolve(XKX, XKy)
        R = yKy - float(XKy @ beta)
        dR = dyKy - 2.0 * float(dXKy @ beta) + float(beta @ dXKX @ beta)
        m = self._df  # n (ML) or n - rank(X) (REML): net power of the scale in the objective
        d = m * dR / R + dlogdet
        if self._restricted:
            d += float(np.trace(_rsolve(XKX, dXKX)))
        return 0.5 * d * delta * (1.0 - delta)

    def fit(self, verbose=False, polish=False):
        """``polish=False``: the reference procedure (brent-search, rtol = atol = 1e-6).
        ``polish=True``: followed by secant steps on the analytic derivative, which pins the
        optimum to ~1e-12 instead of the ~1e-6 a function-value search can guarantee; the HIP
        engine offers the same option (cellregmap_amd/csrc/nullfit.hip, same statements)."""
        x, fx, _ = brent.minimize(self._neg_lml_at, a=-LOGMAX, b=LOGMAX, rtol=1e-6, atol=1e-6)
        if polish:
            x = self._polish(float(x), float(fx))
        self._x = float(x)
        self._update()

    def _polish(self, x0, f0):
        xa = x0
        ga = self._neg_lml_grad_at(xa)
        if not np.isfinite(ga) or ga == 0.0:
            return x0
        xb = xa - 1e-4 if ga > 0.0 else xa + 1e-4
        gb = self._neg_lml_grad_at(xb)
        for _ in range(8):
            if not np.isfinite(gb) or gb == ga:
                break
            xn = xb - gb * (xb - xa) / (gb - ga)
            if not np.isfinite(xn) or abs(xn - x0) > 1e-2:
                return x0
            step = abs(xn - xb)
            xa, ga = xb, gb
            xb = xn
            gb = self._neg_lml_grad_at(xb)
            if gb == 0.0 or step <= 1e-12 * (1.0 + abs(xn)):
                break
        if not np.isfinite(gb):
            return x0
        fb = self._neg_lml_at(xb)
        if not (fb <= f0 + 1e-9 * abs(f0)):
            return x0
        return xb

    # -- fitted quantities ------------------------------------------------------
    @property
    def scale(self):
        return self._scale

    @property
    def v0(self):
        return self._scale * (1.0 - self.delta)

    @property
    def v1(self):
        return self._scale * self.delta

    @property
    def beta(self):
        # back from the SVD basis: X b = tX tb, tX = X Vt' => b = Vt' tb
        return self._Vt.T @ self._tbeta

    def mean(self):
        return self._tX @ self._tbeta

    def get_fast_scanner(self):
        return FastScanner(self)


class FastScanner:
    """Per-candidate ML refit with the covariance ratio frozen at the null.

    glimix-core ``FastScanner.fast_scan(G)['lml']`` (_cellregmap.py:308-309):
    K = v0 * Sigma + v1 * I is fixed at the null fit; for each candidate column
    g the fixed effects of [X, g] and one overall scale multiplier are
    re-estimated in closed form and the ML log-likelihood is returned.
    """

    def __init__(self, lmm: LMM):
        self._lmm = lmm

    def fast_scan(self, G, verbose=False):
        lm = self._lmm
        G = np.asarray(G, float)
        n = lm._n
        delta = lm.delta
        yKy, XKy, XKX, logdet = lm._terms(delta)
        w = 1.0 / ((1.0 - delta) * lm._S0 + delta)
        tG = lm._Q0.T @ G  # r x p, the one large product
        lmls = np.empty(G.shape[1])
        for i in range(G.shape[1]):
            g = G[:, i]
            tg = tG[:, i]
            gKg = float((tg * w) @ tg) + (float(g @ g) - float(tg @ tg)) / delta
            gKy = float((tg * w) @ lm._ty) + (float(g @ lm._y) - float(tg @ lm._ty)) / delta
            gKX = lm._tXr.T @ (w * tg) + (lm._tX.T @ g - lm._tXr.T @ tg) / delta
            A = np.block([[XKX, gKX[:, None]], [gKX[None, :], np.array([[gKg]])]])
            b = np.concatenate([XKy, [gKy]])
            beta = _rsolve(A, b)
            s = max((yKy - float(b @ beta)) / n, epsilon.small)
            lmls[i] = -0.5 * (n * LOG2PI + n + n * np.log(s) + logdet)
        return {"lml": lmls}
