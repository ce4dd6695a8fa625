"""Effect sizes on the device against the oracle's restatement of
cellregmap/_cellregmap.py:137-244 and :640-682."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from cellregmap_amd.synth import make_cohort  # noqa: E402


def _cohort(seed=3, donors=8, cells=15, k0=3, p=4):
    c = make_cohort(donors, cells, k0, p, seed=seed)
    maf = np.clip(np.minimum(c.G.mean(0) / 2, 1 - c.G.mean(0) / 2), 0.05, 0.5)
    return c, maf


def _close(a, b, rtol):
    scale = np.max(np.abs(b))
    assert a.shape == b.shape
    assert np.max(np.abs(a - b)) <= rtol * scale, (np.max(np.abs(a - b)), scale)


class _polished:
    """Both procedures side by side.  The reference stops every fit of the effect-size path with Brent's search at 1e-6 on
    logit(delta) (_cellregmap.py:175-176, 223-224), and the fixed effects move with delta: two faithful runs of the verbatim
    procedure are only as close as each of them is to the TRUE optimum.  The polished procedure (secant steps on the analytic
    derivative, device and oracle alike) pins that optimum to ~1e-12, so the verbatim results are held to
        |device - oracle|  <=  |device - device polished| + |oracle - oracle polished| + 1e-7 scale
    (the triangle through the common optimum; the last term is what the polished sides differ by, asserted as well) --
    a bound each test measures on its own problem instead of a fixed envelope."""

    def __init__(self):
        from cellregmap_amd import _engine, _lib

        self.lib, self.ctx, self.check = _lib.load(), _engine._context(0), _lib.check

    def __enter__(self):
        self.check(self.lib.crm_set_null_fit_polish(self.ctx, 1))

    def __exit__(self, *exc):
        self.check(self.lib.crm_set_null_fit_polish(self.ctx, 0))


def _close_through_the_optimum(dev, ora, dev_polished, ora_polished):
    for a, b, ap, bp in zip(dev, ora, dev_polished, ora_polished):
        scale = np.max(np.abs(bp))
        assert a.shape == b.shape == ap.shape == bp.shape
        assert np.max(np.abs(ap - bp)) <= 1e-7 * scale, (np.max(np.abs(ap - bp)), scale)
        own = np.max(np.abs(a - ap)) + np.max(np.abs(b - bp))
        assert np.max(np.abs(a - b)) <= own + 1e-7 * scale, (np.max(np.abs(a - b)), own, scale)
        # (and the search's tolerance is what it is: neither side strays further from the optimum than a few of them)
        assert own <= 40 * 1e-6 * scale, (own, scale)


@pytest.mark.parametrize("with_kinship", [True, False])
def test_estimate_betas_matches_the_oracle(with_kinship):
    import cellregmap_amd as crm
    from oracle import crm as ocrm

    c, maf = _cohort()
    hK = c.hK if with_kinship else None
    bg, bgxe = crm.estimate_betas(c.y, c.W, c.E, c.G, maf=maf, hK=hK)
    obg, obgxe = ocrm.estimate_betas(c.y, c.W, c.E, c.G, maf=maf, hK=hK)
    assert bgxe.shape == (1, c.y.size, c.G.shape[1])  # the reference's stack(...).T of (n, 1) columns
    with _polished():
        pol = crm.estimate_betas(c.y, c.W, c.E, c.G, maf=maf, hK=hK)
    opol = ocrm.estimate_betas(c.y, c.W, c.E, c.G, maf=maf, hK=hK, polish=True)
    _close_through_the_optimum((bg, bgxe), (obg, obgxe), pol, opol)


def test_estimate_betas_at_two_thousand_cells():
    """The same at a size where the per-SNP backgrounds take the constructor's thin branch (10 + 10 x 40 = 410 columns
    against 2 000 cells: Gram matrix -> eigen-solver -> mixing matrices, the rho = 1 grid point on its leading block) and
    the fits its LDS kernel (1 + 1 + 10 = 12 fixed-effect columns): _cellregmap.py:137-205, :640-682."""
    import cellregmap_amd as crm
    from oracle import crm as ocrm

    c, maf = _cohort(seed=29, donors=40, cells=50, k0=10, p=3)
    bg, bgxe = crm.estimate_betas(c.y, c.W, c.E, c.G, maf=maf, hK=c.hK)
    obg, obgxe = ocrm.estimate_betas(c.y, c.W, c.E, c.G, maf=maf, hK=c.hK)
    assert bgxe.shape == (1, 2000, 3)
    # (12 fixed-effect columns: the derivative polish is built for up to 8, so this size is held to the oracle through the
    # oracle's own optimum -- how far the oracle's verbatim stop is from it bounds what a second faithful run may differ by,
    # doubled for the two sides)
    opol = ocrm.estimate_betas(c.y, c.W, c.E, c.G, maf=maf, hK=c.hK, polish=True)
    for a, b, bp in ((bg, obg, opol[0]), (bgxe, obgxe, opol[1])):
        scale = np.max(np.abs(bp))
        assert np.max(np.abs(a - b)) <= 2.0 * np.max(np.abs(b - bp)) + 3e-6 * scale, (np.max(np.abs(a - b)), np.max(np.abs(b - bp)), scale)


def test_estimate_betas_with_polished_fits_is_tight():
    import cellregmap_amd as crm
    from cellregmap_amd import _engine, _lib
    from oracle import crm as ocrm

    c, maf = _cohort(seed=5)
    ctx = _engine._context(0)
    _lib.check(_lib.load().crm_set_null_fit_polish(ctx, 1))
    try:
        bg, bgxe = crm.estimate_betas(c.y, c.W, c.E, c.G, maf=maf, hK=c.hK)
    finally:
        _lib.check(_lib.load().crm_set_null_fit_polish(ctx, 0))
    obg, obgxe = ocrm.estimate_betas(c.y, c.W, c.E, c.G, maf=maf, hK=c.hK, polish=True)
    _close(bg, obg, 1e-7)
    _close(bgxe, obgxe, 1e-7)


def test_maf_default_and_compute_maf():
    import cellregmap_amd as crm
    from oracle import crm as ocrm

    X = np.random.default_rng(0).integers(0, 3, size=(100, 10)).astype(float)
    X[3, 4] = np.nan
    np.testing.assert_allclose(crm.compute_maf(X), ocrm.compute_maf(X), rtol=0, atol=0)
    assert np.all(crm.compute_maf(X) <= 0.5)
    c, _ = _cohort(seed=7, p=2)
    G = np.random.default_rng(1).integers(0, 3, size=(8, 2)).astype(float)[c.donor_of_cell]
    bg, bgxe = crm.estimate_betas(c.y, c.W, c.E, G, hK=c.hK)
    obg, obgxe = ocrm.estimate_betas(c.y, c.W, c.E, G, hK=c.hK)
    with _polished():
        pol = crm.estimate_betas(c.y, c.W, c.E, G, hK=c.hK)
    _close_through_the_optimum((bg, bgxe), (obg, obgxe), pol, ocrm.estimate_betas(c.y, c.W, c.E, G, hK=c.hK, polish=True))


def test_estimate_aggregate_environment_matches_the_oracle():
    import cellregmap_amd as crm
    from oracle import crm as ocrm

    # With E1 = E0 the restricted likelihood is flat in rho (E0 is also a fixed effect of this fit,
    # so the E1E1' component is not identifiable) and the reference's pick of rho -- hence its
    # result -- hangs on rounding noise of ~1e-13 in the lml.  A distinct E1 makes the case well posed.
    c, _ = _cohort(seed=11)
    E1 = np.random.default_rng(2).standard_normal((c.y.size, 4))
    Ls = crm.get_L_values(c.hK, c.E)
    dev = crm.CellRegMap(c.y, c.E, W=c.W, Ls=Ls, E1=E1)
    ora = ocrm.OracleCellRegMap(c.y, c.E, W=c.W, Ls=ocrm.khatri_rao_halves(c.hK, c.E), E1=E1)
    ora_pol = ocrm.OracleCellRegMap(c.y, c.E, W=c.W, Ls=ocrm.khatri_rao_halves(c.hK, c.E), E1=E1, polish=True)
    for j in (0, 2):
        a = dev.estimate_aggregate_environment(c.G[:, j])
        b = ora.estimate_aggregate_environment(c.G[:, j])
        with _polished():
            ap = dev.estimate_aggregate_environment(c.G[:, j])
        _close_through_the_optimum((a,), (b,), (ap,), (ora_pol.estimate_aggregate_environment(c.G[:, j]),))


def test_collinear_contexts_go_through_the_svd_basis():
    """E0 with a constant column makes M = [W, g, E0] rank deficient: glimix-core's LMM (and the
    oracle) fit in the SVD basis and report the minimum-norm beta."""
    import cellregmap_amd as crm
    from oracle import crm as ocrm

    c, maf = _cohort(seed=13, p=2)
    E = np.concatenate([c.E, np.ones((c.y.size, 1))], axis=1)
    bg, bgxe = crm.estimate_betas(c.y, c.W, E, c.G, maf=maf, hK=c.hK)
    obg, obgxe = ocrm.estimate_betas(c.y, c.W, E, c.G, maf=maf, hK=c.hK)
    with _polished():
        pol = crm.estimate_betas(c.y, c.W, E, c.G, maf=maf, hK=c.hK)
    _close_through_the_optimum((bg, bgxe), (obg, obgxe), pol, ocrm.estimate_betas(c.y, c.W, E, c.G, maf=maf, hK=c.hK, polish=True))


def test_cov_solve_is_the_inverse_of_the_covariance():
    import cellregmap_amd as crm

    c, _ = _cohort(seed=17)
    dev = crm.CellRegMap(c.y, c.E, W=c.W, Ls=crm.get_L_values(c.hK, c.E))
    bg = dev._bg
    rng = np.random.default_rng(0)
    rhs = rng.standard_normal((c.y.size, 3))
    v0, v1 = 0.7, 0.4
    for ri in (0, 5, 10):
        Q0, S0 = bg.read(ri, c.y.size)
        K = v0 * (Q0 * S0) @ Q0.T + v1 * np.eye(c.y.size)
        out = dev._cov_solve(bg, ri, v0, v1, rhs)
        np.testing.assert_allclose(K @ out, rhs, rtol=0, atol=1e-10)


def test_lmm_fit_reproduces_the_documented_glimix_core_example():
    """The ML fit of glimix-core's documentation example (see tests/test_oracle_lmm.py) through
    crm_lmm_fit: lml to 1e-10, variances within the Brent tolerance, beta against the oracle."""
    import ctypes

    from cellregmap_amd import _engine, _lib
    from oracle.lmm import LMM
    from oracle.sugar import economic_qs_linear

    G = np.array([[1, 2], [3, -1], [1.1, 0.5], [0.5, -0.4]], float)
    y = np.array([-1, 2, 0.3, 0.5])
    X = np.ones((4, 1))
    lib = _lib.load()
    bg = _engine._make_background(G, None, [1.0], 0, cache=False)
    h = ctypes.c_void_p()
    E0 = np.zeros((4, 1))
    _lib.check(lib.crm_gene_create(bg.handle, _lib.ptr(y), _lib.ptr(X), 1, _lib.ptr(E0), 1, ctypes.byref(h)))
    try:
        fit = np.empty(6)
        beta = np.empty(1)
        _lib.check(lib.crm_lmm_fit(h, 0, _lib.ptr(fit), _lib.ptr(beta)))
    finally:
        lib.crm_gene_destroy(h)
    assert abs(fit[3] - (-2.2726234086180557)) < 1e-10
    np.testing.assert_allclose(fit[1], 0.33736446158226896, rtol=1e-6)
    np.testing.assert_allclose(fit[2], 0.012503600451739165, rtol=1e-6)
    ref = LMM(y, X, economic_qs_linear(G))
    ref.fit(verbose=False)
    np.testing.assert_allclose(beta, ref.beta, rtol=1e-6)


def test_lmm_fit_returns_beta_in_the_callers_basis_for_correlated_covariates():
    """The C-ABI on its own (no Python host in between): crm_gene_create brings correlated covariate columns to mutually
    orthogonal ones (W V, cyclic Jacobi); crm_lmm_fit must hand back the coefficients of the columns the CALLER passed --
    glimix-core's LMM.beta (_cellregmap.py:186,232) -- not those of the rotated basis.  Both sides with the polished
    optimum: the comparison is on the algebra, not on where a 1e-6 search stops."""
    import ctypes

    from cellregmap_amd import _engine, _lib
    from oracle.lmm import LMM

    c = make_cohort(9, 14, 3, 2, seed=41)
    rng = np.random.default_rng(2)
    n = c.y.size
    base = rng.normal(size=(n, 1))
    W = np.concatenate([np.ones((n, 1)), base + 0.05 * rng.normal(size=(n, 1)), 2.0 * base + 0.3 * rng.normal(size=(n, 1)),
                        rng.normal(size=(n, 1)) + 3.0], axis=1)           # columns 1 and 2 correlate at 0.99
    obj = _engine.CellRegMap(c.y, c.E, W=c.W, hK=c.hK)
    lib = _lib.load()
    ctx = _engine._context(0)
    _lib.check(lib.crm_set_null_fit_polish(ctx, 1))
    try:
        h = ctypes.c_void_p()
        y, Wc, E0 = _lib.f64(c.y), _lib.f64(W), _lib.f64(c.E)
        _lib.check(lib.crm_gene_create(obj._bg.handle, _lib.ptr(y), _lib.ptr(Wc), Wc.shape[1], _lib.ptr(E0), E0.shape[1],
                                       ctypes.byref(h)))
        try:
            fit, beta = np.empty(6), np.empty(W.shape[1])
            _lib.check(lib.crm_lmm_fit(h, 1, _lib.ptr(fit), _lib.ptr(beta)))
        finally:
            lib.crm_gene_destroy(h)
    finally:
        _lib.check(lib.crm_set_null_fit_polish(ctx, 0))
    ri = int(fit[5])
    Q0, S0 = obj._bg.read(ri, n)
    o = LMM(c.y, W, ((Q0,), S0), restricted=True)
    o.fit(verbose=False, polish=True)
    assert abs(fit[3] - o.lml()) <= 1e-10 * abs(o.lml())
    np.testing.assert_allclose(beta, o.beta, rtol=1e-7, atol=1e-9 * np.abs(o.beta).max())
    # (and it is not the rotated basis' vector: with these columns the two differ by far more than the tolerance)
    assert np.abs(W @ beta - o.mean()).max() <= 1e-7 * np.abs(o.mean()).max()
