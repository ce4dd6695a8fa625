"""oracle.lmm / oracle.brent against independent mathematics (parity unpinned
by the reference: glimix-core is not available; see oracle/__init__.py)."""
import numpy as np
from numpy.testing import assert_allclose
from scipy.optimize import minimize_scalar

from oracle import brent
from oracle.lmm import LMM, LOG2PI
from oracle.sugar import economic_qs_linear


def _dense_lml(y, X, Sigma, delta, restricted):
    n, p = X.shape
    Kt = (1 - delta) * Sigma + delta * np.eye(n)
    Ki = np.linalg.inv(Kt)
    XKX = X.T @ Ki @ X
    beta = np.linalg.solve(XKX, X.T @ Ki @ y)
    r = y - X @ beta
    df = n - p if restricted else n
    s = (r @ Ki @ r) / df
    _, ld = np.linalg.slogdet(Kt)
    val = -0.5 * (df * LOG2PI + df + n * np.log(s) + ld)
    if restricted:
        val += 0.5 * (np.linalg.slogdet(X.T @ X)[1] - np.linalg.slogdet(XKX / s)[1])
    return val, s, beta


def _problem(seed, n=60, r=9, c=2):
    rng = np.random.default_rng(seed)
    H = rng.normal(size=(n, r))
    X = np.concatenate([np.ones((n, 1)), rng.normal(size=(n, c))], axis=1)
    y = X @ rng.normal(size=c + 1) + H @ rng.normal(size=r) * 0.5 + rng.normal(size=n)
    return y, X, H


def test_profile_likelihood_matches_dense():
    for seed in range(3):
        y, X, H = _problem(seed)
        QS = economic_qs_linear(H, return_q1=False)
        for restricted in (False, True):
            lmm = LMM(y, X, QS, restricted=restricted)
            for x in (-3.0, -0.4, 0.0, 1.7):
                val = -lmm._neg_lml_at(x)
                ref, s, beta = _dense_lml(y, X, H @ H.T, lmm.delta, restricted)
                assert_allclose(val, ref, rtol=1e-10)
                assert_allclose(lmm.scale, s, rtol=1e-9)
                assert_allclose(lmm.beta, beta, rtol=1e-8)


def test_fit_finds_the_dense_optimum():
    y, X, H = _problem(7)
    QS = economic_qs_linear(H, return_q1=False)
    lmm = LMM(y, X, QS, restricted=True)
    lmm.fit()
    f = lambda x: -_dense_lml(y, X, H @ H.T, 1 / (1 + np.exp(-x)), True)[0]
    ref = minimize_scalar(f, bounds=(-30, 30), method='bounded', options={'xatol': 1e-9})
    assert abs(lmm._x - ref.x) < 5e-6 * (1 + abs(ref.x))
    assert_allclose(lmm.lml(), -ref.fun, rtol=1e-10)
    assert_allclose(lmm.v0 + lmm.v1, lmm.scale)


def test_rank_deficient_covariates_are_tolerated():
    y, X, H = _problem(3)
    X2 = np.concatenate([X, X[:, [1]] * 2.0], axis=1)  # duplicate direction
    QS = economic_qs_linear(H, return_q1=False)
    a = LMM(y, X, QS, restricted=True)
    b = LMM(y, X2, QS, restricted=True)
    a.fit(); b.fit()
    assert_allclose(a.delta, b.delta, rtol=1e-5)
    assert_allclose(a.mean(), b.mean(), atol=1e-6)


def test_brent_on_known_functions():
    x, fx, nfev = brent.minimize(lambda x: (x - 2.5) ** 2 + 1.0, -700, 700, 1e-6, 1e-6)
    assert abs(x - 2.5) < 1e-5 and nfev < 40
    x, fx, _ = brent.minimize(lambda x: np.cosh(x + 4.0), -700, 700, 1e-6, 1e-6)
    assert abs(x + 4.0) < 1e-5
    # monotone: runs into the bound
    x, fx, _ = brent.minimize(lambda x: -x, -10, 10, 1e-6, 1e-6)
    assert abs(x - 10) < 1e-4


def test_fast_scanner_equals_refit_with_frozen_delta():
    y, X, H = _problem(11)
    rng = np.random.default_rng(5)
    G = rng.normal(size=(y.size, 4))
    QS = economic_qs_linear(H, return_q1=False)
    null = LMM(y, X, QS, restricted=False)
    null.fit()
    lmls = null.get_fast_scanner().fast_scan(G)["lml"]
    for i in range(G.shape[1]):
        Xa = np.concatenate([X, G[:, [i]]], axis=1)
        ref, _, _ = _dense_lml(y, Xa, H @ H.T, null.delta, False)
        assert_allclose(lmls[i], ref, rtol=1e-10)


def test_polish_stays_within_the_reference_tolerance():
    """The derivative-based polish moves the Brent(1e-6) optimum by less than Brent's own
    tolerance and lands on a stationary point of the profiled likelihood."""
    for seed, restricted in ((3, True), (4, False), (5, True)):
        y, X, H = _problem(seed)
        QS = economic_qs_linear(H, return_q1=False)
        a = LMM(y, X, QS, restricted=restricted)
        b = LMM(y, X, QS, restricted=restricted)
        a.fit(polish=False)
        b.fit(polish=True)
        assert abs(a._x - b._x) <= 2e-6 * (1 + abs(a._x))
        assert b.lml() >= a.lml() - 1e-12 * abs(a.lml())
        assert abs(b._neg_lml_grad_at(b._x)) < 1e-9
        # analytic derivative vs central difference
        for x in (-1.0, 0.4, 2.0):
            h = 1e-5
            num = (b._neg_lml_at(x + h) - b._neg_lml_at(x - h)) / (2 * h)
            assert abs(num - b._neg_lml_grad_at(x)) < 1e-6 * (1 + abs(num))


def test_documented_examples_of_glimix_core():
    """Worked examples printed in glimix-core's own documentation (LMM class docstring / user guide,
    version 3.1.x; quoted from memory -- the package is not installable here): ML fits on tiny data.
    The log-likelihoods pin the model algebra to 13 digits; v0 / v1 carry the 1e-6 tolerance of the
    Brent search on logit(delta)."""
    X = np.array([[1, 2], [3, -1]], float)
    lmm = LMM(np.array([-1, 2], float), np.ones((2, 1)), economic_qs_linear(X))
    lmm.fit(verbose=False)
    assert "%.3f" % lmm.lml() == "-3.649"

    G = np.array([[1, 2], [3, -1], [1.1, 0.5], [0.5, -0.4]], float)
    lmm = LMM(np.array([-1, 2, 0.3, 0.5]), np.ones((4, 1)), economic_qs_linear(G))
    lmm.fit(verbose=False)
    assert abs(lmm.lml() - (-2.2726234086180557)) < 1e-10
    assert_allclose(lmm.v0, 0.33736446158226896, rtol=1e-6)
    assert_allclose(lmm.v1, 0.012503600451739165, rtol=1e-6)
