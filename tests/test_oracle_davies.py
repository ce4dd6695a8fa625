"""oracle.davies (qfc.c) against independent mathematics: closed forms and a
numerical Imhof (1961) integral.  Davies' accuracy target is absolute (1e-6)."""
import numpy as np
from scipy import integrate
from scipy.stats import chi2

from oracle.davies import davies_pvalue, filter_weights, pvalue_from_weights, qfc


def imhof_sf(q, lam):
    lam = np.asarray(lam, float)

    def integrand(u):
        theta = 0.5 * np.sum(np.arctan(lam * u)) - 0.5 * q * u
        rho = np.prod((1 + (lam * u) ** 2) ** 0.25)
        return np.sin(theta) / (u * rho)

    val, _ = integrate.quad(integrand, 0, np.inf, limit=2000, epsabs=1e-11, epsrel=1e-11)
    return 0.5 + val / np.pi


def test_equal_weights_closed_form():
    for k, w in ((3, 2.0), (6, 0.5), (10, 1.3)):
        for q in (0.3 * k * w, k * w, 3.0 * k * w):
            cdf, ifault, _ = qfc([w] * k, q)
            assert ifault == 0
            assert abs(cdf - chi2(k).cdf(q / w)) < 2e-6


def test_against_imhof():
    rng = np.random.default_rng(0)
    for _ in range(12):
        k = rng.integers(2, 12)
        lam = np.sort(rng.gamma(1.0, 1.0, size=k))
        for frac in (0.5, 1.0, 2.5, 5.0):
            q = frac * lam.sum()
            cdf, ifault, _ = qfc(lam, q)
            assert ifault == 0
            assert abs((1 - cdf) - imhof_sf(q, lam)) < 2e-6


def test_filter_and_fallbacks():
    rng = np.random.default_rng(1)
    A = rng.normal(size=(8, 3))
    F = A @ A.T  # rank 3: five ~0 eigenvalues (some slightly negative)
    lam = filter_weights(F)
    assert lam.size == 3
    # far tail: Davies returns <= 0 -> modified Liu takes over, result in (0, 1)
    p, info = pvalue_from_weights(200.0 * lam.sum(), lam)
    assert 0.0 < p < 1e-20 and info["Is_Converged"] == 0 and p == info["liu_pval"]
    # single surviving eigenvalue -> Liu
    F1 = np.outer(A[:, 0], A[:, 0])
    p1, info1 = davies_pvalue(1.3, F1, True)
    assert p1 == info1["liu_pval"]
    assert abs(p1 - chi2(1).sf(1.3 / (A[:, 0] @ A[:, 0]))) < 1e-6
