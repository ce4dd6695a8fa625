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


# Published check values of the algorithm: Imhof (1961) Table 1 as reproduced by Davies (1980,
# AS 155, Table 1), four decimals.  Q = sum_j lam_j chi2(h_j, delta_j^2); entries (x, P[Q < x]).
# The paper ran with lim = 1000, acc = 1e-4; the tolerance covers the printed rounding.
_AS155_TABLE = [
    ((6, 3, 1), (1, 1, 1), (0, 0, 0), ((1, 0.0542), (7, 0.4936), (20, 0.8760))),
    ((6, 3, 1), (2, 2, 2), (0, 0, 0), ((2, 0.0065), (20, 0.6002), (60, 0.9839))),
    ((6, 3, 1), (6, 4, 2), (0, 0, 0), ((10, 0.0027), (50, 0.5648), (120, 0.9912))),
    ((6, 3, 1), (2, 4, 6), (0, 0, 0), ((10, 0.0334), (30, 0.5803), (80, 0.9913))),
    ((7, 3), (6, 2), (6, 2), ((20, 0.0061), (100, 0.5913), (200, 0.9779))),
    ((7, 3), (1, 1), (6, 2), ((10, 0.0451), (60, 0.5924), (150, 0.9777))),
    ((7, 3, 7, 3), (6, 2, 1, 1), (6, 2, 6, 2), ((70, 0.0437), (160, 0.5848), (260, 0.9538))),
    ((7, 3, -7, -3), (6, 2, 1, 1), (6, 2, 6, 2), ((-40, 0.0782), (40, 0.5221), (140, 0.9604))),
]


def test_published_table_of_as155():
    for lam, dof, nc, points in _AS155_TABLE:
        for x, expected in points:
            got, ifault, _ = qfc(np.asarray(lam, float), float(x), np.asarray(dof), np.asarray(nc, float),
                                 lim=1000, acc=1e-4)
            assert ifault == 0
            assert abs(got - expected) < 2e-4, (lam, dof, nc, x, got, expected)


def test_term_limit_is_reported_like_the_reference_settings_would():
    # chiscore calls with lim = 10000, acc = 1e-6: the first table entry then needs more terms than
    # allowed -> ifault 1 and a value outside (0, 1], which davies_pvalue answers with modified Liu
    _, ifault, _ = qfc(np.array([6.0, 3.0, 1.0]), 1.0, np.array([1, 1, 1]), np.zeros(3), lim=10000, acc=1e-6)
    assert ifault == 1
