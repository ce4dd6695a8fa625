"""Edge cases of the interaction scan: empty and single-variant panels, one context (single
eigenvalue -> Liu branch), genotypes inside span(W), rank-deficient covariates, non-finite input."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import parity_bounds

pytestmark = pytest.mark.gpu

P_RTOL, P_ATOL = 1e-5, 1e-13


def _cohort(donors, cells, k, p, seed):
    from cellregmap_amd.synth import make_cohort

    return make_cohort(donors, cells, k, p, seed=seed)


def test_empty_and_single_variant():
    from cellregmap_amd import CellRegMap
    from oracle.crm import OracleCellRegMap

    c = _cohort(6, 10, 3, 4, seed=31)
    crm = CellRegMap(c.y, c.E, W=c.W, hK=c.hK)
    pv, info = crm.scan_interaction(c.G[:, :0])
    assert pv.shape == (0,) and all(v.shape == (0,) for v in info.values())
    pv, info = crm.scan_interaction(c.G[:, [2]])
    opv, oinfo = OracleCellRegMap(c.y, c.E, W=c.W, hK=c.hK).scan_interaction(c.G[:, [2]])
    assert np.all(np.abs(pv - opv) <= P_RTOL * opv + P_ATOL)


def test_single_context_uses_the_liu_branch():
    from cellregmap_amd import CellRegMap
    from oracle.crm import OracleCellRegMap

    c = _cohort(8, 12, 1, 12, seed=32)
    pv, info = CellRegMap(c.y, c.E, W=c.W, hK=c.hK).scan_interaction(c.G)
    opv, oinfo = OracleCellRegMap(c.y, c.E, W=c.W, hK=c.hK).scan_interaction(c.G)
    assert_allclose(info["rho1"], oinfo["rho1"], atol=1e-12)
    assert np.all(np.abs(pv - opv) <= P_RTOL * opv + P_ATOL), np.c_[pv, opv]


def test_genotype_inside_span_of_covariates_and_rank_deficient_W():
    """A constant genotype column is collinear with the intercept; W with a duplicated column
    is rank deficient.  The reference tolerates both (SVD-reduced covariates in glimix-core's
    LMM, lstsq in PMat, _math.py:33-37)."""
    from cellregmap_amd import CellRegMap, GenotypePanel
    from oracle.crm import OracleCellRegMap

    c = _cohort(8, 12, 3, 6, seed=33)
    G = c.G.copy()
    G[:, 1] = 1.0            # collinear with W = 1
    rng = np.random.default_rng(0)
    x = rng.normal(size=(c.y.size, 1))
    W = np.concatenate([c.W, x, 2.0 * x], axis=1)  # rank 2, three columns
    for Wc in (c.W, W):
        crm = CellRegMap(c.y, c.E, W=Wc, hK=c.hK)
        ocrm = OracleCellRegMap(c.y, c.E, W=Wc, hK=c.hK)
        for groups in (None, "auto"):
            pv, info = crm.scan_interaction(GenotypePanel(G, groups=groups))
            opv, oinfo = ocrm.scan_interaction(G)
            assert_allclose(info["rho1"], oinfo["rho1"], atol=1e-12)
            assert np.all(np.abs(pv - opv) <= P_RTOL * opv + P_ATOL), np.c_[pv, opv]


def _variants_at_the_rank_rule(W, n, targets, seed=3):
    """Columns base + e u (base in span(W), u a unit vector orthogonal to it) whose smallest singular value in [W, g]
    is targets[col] * sqrt(eps); returns {col: vector} and whether numpy_sugar.economic_svd keeps three directions."""
    from oracle.sugar import economic_svd, epsilon

    rng = np.random.default_rng(seed)
    Qw, _ = np.linalg.qr(W)
    u = rng.normal(size=n)
    u -= Qw @ (Qw.T @ u)
    u /= np.linalg.norm(u)
    base = W @ np.array([0.7, -0.4])
    cols, kept = {}, {}
    for col, target in targets.items():
        lo, hi = 1e-14, 1e-4
        for _ in range(200):
            mid = np.sqrt(lo * hi)
            small = np.linalg.svd(np.c_[W, base + mid * u], compute_uv=False)[-1]
            lo, hi = (mid, hi) if small < target * epsilon.small else (lo, mid)
        cols[col] = base + hi * u
        kept[col] = economic_svd(np.c_[W, cols[col]])[1].shape[0] == 3
    return cols, kept


@pytest.mark.parametrize("scale", [1.0, 0.003])
def test_variant_on_either_side_of_the_references_rank_rule(scale):
    """glimix-core's LMM reduces X = [W, g] by numpy_sugar.economic_svd: a direction whose singular value lies below
    sqrt(eps) = 1.49e-8 (absolute) is dropped.  The engine applies the same rule to the smallest singular value of
    [W, g] (csrc/blockops.hip: ortho_coef_kernel -- the secular equation of [W, g]'[W, g] at eps): variants built to sit
    at 0.5 and at 2 times the threshold come out on the oracle's side of it (MODEL_G_IN_SPAN_W, the null fit's degrees
    of freedom).  The projection of the score test follows ANOTHER rule of the reference -- PMat's lstsq(rcond=None) on
    X'K^-1X, relative eps (c + 1) (_math.py:33-37) -- which the engine decides separately (csrc/assemble.hip:
    finalize_kernel).  scale = 1 (columns of norm ~ sqrt(n)): lstsq cuts both built variants; the one below the LMM's
    threshold is compared (the one above it sits where the oracle's own LMM solves its fixed effects by a truncated
    lstsq while counting the direction in its degrees of freedom -- a corner nothing pins).
    scale = 0.003 (tiny columns): lstsq keeps both, the LMM drops one -- the two rules part, and the engine parts with
    them."""
    from cellregmap_amd import CellRegMap, GenotypePanel
    from fuzz_cases import random_problem
    from oracle.crm import OracleCellRegMap

    y, E, W, G, kw = random_problem(120, 3, 2, 4, 6, seed=11, mode="B")
    W = scale * W
    cols, kept = _variants_at_the_rank_rule(W, y.size, {1: 0.5, 3: 2.0})
    assert kept == {1: False, 3: True}
    G = G.copy()
    for col, g in cols.items():
        G[:, col] = g
    crm = CellRegMap(y, E, W=W, **kw)
    opv, oinfo, ost = OracleCellRegMap(y, E, W=W, **kw).scan_interaction(G, return_stats=True)
    panel = GenotypePanel(G, groups=None)
    pv, info = crm.scan_interaction_info(panel)
    assert list((info["model_flags"] & 4) != 0) == [False, True, False, False], info["model_flags"]
    pv2, info2, st = crm.scan_interaction(panel, return_stats=True)
    assert_allclose(info2["rho1"], oinfo["rho1"], atol=1e-12)
    for j in (0, 2):      # untouched variants at the usual bar
        assert abs(st["Q"][j] - ost["Q"][j]) <= 1e-6 * ost["Q"][j] and abs(pv2[j] - opv[j]) <= P_RTOL * opv[j]
    # The built ones.  Where PMat keeps the variant (scale 0.003) the reference's own projection is solved by lstsq on a
    # 3 x 3 matrix of condition (0.04 / 7e-9)^2 ~ 3e13 in the raw basis: its Q carries eps * cond ~ 1e-3 of noise, which
    # the orthogonalised basis here does not (measured difference 1e-4)
    tol = 1e-5 if scale == 1.0 else 2e-3
    for j in ((1,) if scale == 1.0 else (1, 3)):
        assert abs(st["lml"][j] - ost["lml"][j]) <= 1e-8 * abs(ost["lml"][j]), (j, st["lml"][j], ost["lml"][j])
        assert abs(st["Q"][j] - ost["Q"][j]) <= tol * ost["Q"][j], (j, st["Q"][j], ost["Q"][j])
        assert abs(pv2[j] - opv[j]) <= 10 * tol * opv[j], (j, pv2[j], opv[j])


@pytest.mark.parametrize("problem", [238, 344])
def test_nearly_collinear_variants_in_the_fixed_effects_own_basis(problem):
    """Two-donor problems of the fuzz stream (tools/fuzz_scan.py verbatim 400 2026, problems 238 and 344: the two
    donors' dosages nearly coincide on many variants, which then keep 1e-4 ... 1e-2 of their squared norm outside
    span(W)).  Round 3 solved the fixed effects by Cholesky in the raw [W, g] basis and its likelihood was 1e-13 ... 2e-11
    off the oracle's there; with the variants orthogonalised against W in the cell axis (the reference's economic_svd
    basis; csrc/blockops.hip) the log-likelihoods agree to 1e-13 on the dense path and on the collapsed path, which
    hands the nearly collinear variants to the dense one.  Under the verbatim procedure Q still differs by one stopping
    tolerance of Brent's search (1.8e-6 here) on a quarter of these variants: their likelihood is flat enough for the
    last comparison of the search to be decided by the last bits, with the objective equal to 2e-15 at fixed points
    (profiles/r04_collinear_diag.json) -- the library says so itself: these variants carry bounds beyond the tolerances
    (scan_interaction_info: bound_Q, bound_p), and every variant is held to the tolerance or to its own bound."""
    from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib
    from fuzz_cases import build_case, fuzz_cases
    from oracle.crm import OracleCellRegMap

    case = [c for c in fuzz_cases(400, seed=2026) if c[0] == problem][0]
    y, E, W, G, kw, hooks = build_case(case)
    crm = CellRegMap(y, E, W=W, **kw)
    opv, oinfo, ost = OracleCellRegMap(y, E, W=W, **kw).scan_interaction(G, return_stats=True, **hooks)
    lib, ctx = _lib.load(), _engine._context(0)
    before = lib.crm_test_dense_repeats(ctx)
    bq, bp, _ = parity_bounds.bounds(crm, GenotypePanel(G, groups=None), **hooks)
    parity_bounds.assert_bounds_are_informative(bq, bp, opv)
    for groups in (None, "auto"):
        pv, info, st = crm.scan_interaction(GenotypePanel(G, groups=groups), return_stats=True, **hooks)
        assert_allclose(info["rho1"], oinfo["rho1"], atol=1e-12)
        assert np.max(np.abs(st["lml"] - ost["lml"]) / np.abs(ost["lml"])) < 1e-13
        qscale = np.maximum(np.abs(ost["Q"]), [np.trace(F) for F in ost["F"]])
        parity_bounds.assert_Q_within(st["Q"], ost["Q"], bq, qscale, groups)
        parity_bounds.assert_p_within(pv, opv, bp, groups)
    assert lib.crm_test_dense_repeats(ctx) > before      # the collapsed scan did hand variants to the dense path


def test_zero_genotype_gives_nan_not_a_crash():
    """g = 0 makes dK = 0: chiscore raises "No eigenvalue is bigger than 0" and the reference's whole
    scan dies; the engine flags that variant with NaN and carries on."""
    from cellregmap_amd import CellRegMap

    c = _cohort(8, 12, 3, 6, seed=36)
    G = c.G.copy()
    G[:, 4] = 0.0
    pv, info = CellRegMap(c.y, c.E, W=c.W, hK=c.hK).scan_interaction(G)
    assert np.isnan(pv[4]) and np.all(np.isfinite(np.delete(pv, 4)))


def test_non_finite_genotypes_raise():
    from cellregmap_amd import CellRegMap

    c = _cohort(6, 10, 3, 4, seed=34)
    G = c.G.copy()
    G[3, 1] = np.inf
    with pytest.raises(ValueError):
        CellRegMap(c.y, c.E, W=c.W).scan_interaction(G)


def test_many_covariates_in_the_interaction_scan():
    """More than 8 fixed-effect columns (intercept + covariates + PCs): the LDS null-fit kernel and the
    dynamically sized assembly take over."""
    from cellregmap_amd import CellRegMap, GenotypePanel
    from oracle.crm import OracleCellRegMap

    c = _cohort(10, 30, 4, 10, seed=38)
    rng = np.random.default_rng(2)
    W = np.concatenate([c.W, rng.normal(size=(c.y.size, 13))], axis=1)   # 14 columns
    crm = CellRegMap(c.y, c.E, W=W, hK=c.hK)
    opv, oinfo, ost = OracleCellRegMap(c.y, c.E, W=W, hK=c.hK).scan_interaction(c.G, return_stats=True)
    for groups in (None, "auto"):
        pv, info, st = crm.scan_interaction(GenotypePanel(c.G, groups=groups), return_stats=True)
        assert_allclose(info["rho1"], oinfo["rho1"], atol=1e-12)
        assert_allclose(st["Q"], ost["Q"], rtol=1e-6)
        assert np.all(np.abs(pv - opv) <= P_RTOL * opv + P_ATOL), np.c_[pv, opv]


def test_covariates_as_given_through_the_c_abi(monkeypatch):
    """The Python host hands W to the library as U diag(s) of its thin SVD (mutually orthogonal columns).  A C caller may
    pass W as it is: crm_gene_create then brings it to orthogonal columns itself (W V, V from repeated Jacobi passes on the
    c x c Gram matrix).  Same span, same statistics -- with the optimum pinned, to 1e-9, also for columns that are
    correlated to one part in 1e7; a rank-deficient W passed raw is refused with the remedy in the message."""
    from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib

    c = _cohort(9, 25, 4, 30, seed=40)
    rng = np.random.default_rng(6)
    W = np.concatenate([c.W, rng.normal(size=(c.y.size, 3)) + 0.5], axis=1)     # four correlated, non-orthogonal columns
    G = c.G + 0.1 * rng.normal(size=c.G.shape)
    G[:, 5] = W @ np.array([1.0, -0.5, 0.25, 2.0])                              # one variant inside span(W)
    panel = GenotypePanel(G, groups=None)
    lib, ctx = _lib.load(), _engine._context(0)
    _lib.check(lib.crm_set_null_fit_polish(ctx, 1))
    try:
        pv, info, st = CellRegMap(c.y, c.E, W=W, hK=c.hK).scan_interaction(panel, return_stats=True)
        pvi, infoi = CellRegMap(c.y, c.E, W=W, hK=c.hK).scan_interaction_info(panel)
        monkeypatch.setattr(CellRegMap, "_fixed_effect_basis", lambda self: self._W)
        raw = CellRegMap(c.y, c.E, W=W, hK=c.hK)
        pv2, info2, st2 = raw.scan_interaction(panel, return_stats=True)
        pvi2, infoi2 = raw.scan_interaction_info(panel)
    finally:
        _lib.check(lib.crm_set_null_fit_polish(ctx, 0))
    assert np.array_equal(info["rho1"], info2["rho1"])
    assert np.array_equal(infoi["model_flags"], infoi2["model_flags"]) and infoi["model_flags"][5] & 4
    assert_allclose(st2["lml"] - st["lml"], (st2["lml"] - st["lml"])[0], atol=1e-8)   # (log|X'X| differs by the basis: a constant)
    assert_allclose(st2["Q"], st["Q"], rtol=1e-9)
    assert np.all(np.abs(pv2 - pv) <= 2e-6 * pv + P_ATOL)
    # nearly collinear columns (cond(W) ~ 1e7): accepted raw since 0.5.0, the same answers as through the host's SVD basis
    Will = np.concatenate([W, W[:, [1]] + 1e-7 * rng.normal(size=(c.y.size, 1))], axis=1)
    assert np.linalg.cond(Will) > 1e6
    _lib.check(lib.crm_set_null_fit_polish(ctx, 1))
    try:
        monkeypatch.undo()
        pv3, info3, st3 = CellRegMap(c.y, c.E, W=Will, hK=c.hK).scan_interaction(panel, return_stats=True)
        monkeypatch.setattr(CellRegMap, "_fixed_effect_basis", lambda self: self._W)
        pv4, info4, st4 = CellRegMap(c.y, c.E, W=Will, hK=c.hK).scan_interaction(panel, return_stats=True)
    finally:
        _lib.check(lib.crm_set_null_fit_polish(ctx, 0))
    assert np.array_equal(info3["rho1"], info4["rho1"])
    assert_allclose(st4["Q"], st3["Q"], rtol=1e-7)
    assert np.all(np.abs(pv4 - pv3) <= 2e-6 * pv3 + P_ATOL)
    Wdef = np.concatenate([W, W[:, [1]] - W[:, [2]]], axis=1)                           # rank 4, five columns, passed raw
    with pytest.raises(_lib.CrmError, match="basis of span"):
        CellRegMap(c.y, c.E, W=Wdef, hK=c.hK).scan_interaction(panel)


@pytest.mark.parametrize("genotypes", ["dense", "donor-level"])
def test_interaction_scan_with_seventy_covariate_columns(genotypes):
    """63 .. 128 fixed-effect columns (as long as contexts + covariates + 2 <= 144): the slower null-fit kernel
    (nullfit_xwide.hip) under the interaction scan, against the oracle -- on the dense path and on the donor-collapsed
    one, whose per-donor sums table holds a column per covariate (it was 64 columns wide until round 5: rows ran into
    each other from 63 covariates on)."""
    from cellregmap_amd import CellRegMap, GenotypePanel
    from oracle.crm import OracleCellRegMap

    c = _cohort(10, 40, 4, 6, seed=39)                                   # 400 cells
    rng = np.random.default_rng(3)
    W = np.concatenate([c.W, rng.normal(size=(c.y.size, 69))], axis=1)   # 70 columns
    crm = CellRegMap(c.y, c.E, W=W, hK=c.hK)
    opv, oinfo, ost = OracleCellRegMap(c.y, c.E, W=W, hK=c.hK).scan_interaction(c.G, return_stats=True)
    panel = GenotypePanel(c.G, groups=None if genotypes == "dense" else "auto")
    assert (panel.n_groups is not None) == (genotypes != "dense")
    pv, info, st = crm.scan_interaction(panel, return_stats=True)
    assert_allclose(info["rho1"], oinfo["rho1"], atol=1e-12)
    # verbatim procedure: every variant to the north-star tolerance or, where the library says two faithful runs may differ
    # by more (its null-fit kernel for 63 .. 128 columns leaves the same trace as the others), to its own bound
    bq, bp, _ = parity_bounds.bounds(crm, GenotypePanel(c.G, groups=None))
    parity_bounds.assert_bounds_are_informative(bq, bp, opv)
    qscale = np.maximum(np.abs(ost["Q"]), [np.trace(F) for F in ost["F"]])
    parity_bounds.assert_Q_within(st["Q"], ost["Q"], bq, qscale)
    parity_bounds.assert_p_within(pv, opv, bp)


@pytest.mark.parametrize("k0,c,mode,route", [
    (160, 70, "A", None),        # the review's case: Gram over 232 rows (4 workgroups per variant), finalisation rows and the
                                 # eigenvalue kernel's working copy in global memory, Khatri-Rao with 256-wide context tiles
    (100, 60, "A", None),        # 162 Gram rows with contexts within the fast kernels' tiles
    (250, 1, "A", None),         # 253 rows: the 288-row Gram image
    (136, 1, "C", None),         # kinship factor with donor structure: the folded route's per-donor launch, transposed store
    (136, 1, "C", "1"),          # ... the unfolded kinship-structure route
    (136, 1, "C", "0"),          # ... the direct contraction against H
    (150, 3, "B", None),         # mode B (hK alone)
])
def test_interaction_scan_with_many_contexts(k0, c, mode, route, monkeypatch, kernel_form):
    """More than 128 contexts / more than 144 rows of contexts + covariates + 2: the slower forms of the
    Khatri-Rao, Gram, finalisation and eigenvalue kernels under the whole scan, against the oracle."""
    from cellregmap_amd import CellRegMap, GenotypePanel
    from oracle.crm import OracleCellRegMap

    from cellregmap_amd import _engine, _lib

    lib, ctx = _lib.load(), _engine._context(0)
    if route == "1":     # (read when the structure is announced; the seed below keeps this background out of the cache)
        kernel_form("kin_fold", 0)
    donors, cells = (3, 400) if mode == "C" else (8, 100)      # mode C: 136 + 3 x 136 = 544 columns for 1200 cells
    co = _cohort(donors, cells, k0, 5, seed=41 + k0 + (1000 if route == "1" else 0))
    rng = np.random.default_rng(k0 + c)
    W = np.concatenate([co.W, rng.normal(size=(co.y.size, c - 1))], axis=1) if c > 1 else co.W
    from cellregmap_amd import get_L_values
    from oracle.crm import khatri_rao_halves
    kw, okw = {}, {}
    if mode == "C":      # K o EE' through its factored halves (run_interaction's E2 = E)
        kw, okw = dict(Ls=get_L_values(co.hK, co.E)), dict(Ls=khatri_rao_halves(co.hK, co.E))
    elif mode == "B":    # hS = [sqrt(rho) E1, sqrt(1 - rho) hK]
        kw = okw = dict(hK=co.hK)
    opv, oinfo, ost = OracleCellRegMap(co.y, co.E, W=W, **okw).scan_interaction(co.G, return_stats=True)
    crm = CellRegMap(co.y, co.E, W=W, **kw)
    if mode != "A":      # the donor structure was found and announced; folded unless told otherwise
        assert lib.crm_background_kinship_groups(crm._bg.handle) == donors
        assert (lib.crm_background_kinship_folded(crm._bg.handle) > 0) == (route != "1")
    _lib.check(lib.crm_test_set_kinship_route(ctx, {None: 2, "1": 2, "0": 0}[route]))
    try:
        pv, info, st = crm.scan_interaction(GenotypePanel(co.G, groups=None), return_stats=True)
        bq, bp, _ = parity_bounds.bounds(crm, GenotypePanel(co.G, groups=None))
    finally:
        _lib.check(lib.crm_test_set_kinship_route(ctx, 1))
    assert_allclose(info["rho1"], oinfo["rho1"], atol=1e-12)
    # (verbatim procedure: the north-star tolerances, or the variant's own bound where the library reports a wider one)
    parity_bounds.assert_bounds_are_informative(bq, bp, opv)
    oF = np.asarray(ost["F"])
    qscale = np.maximum(np.abs(ost["Q"]), np.trace(oF, axis1=1, axis2=2))
    parity_bounds.assert_Q_within(st["Q"], ost["Q"], bq, qscale)
    parity_bounds.assert_p_within(pv, opv, bp)
    for j in range(oF.shape[0]):                            # F moves with delta as Q does
        assert np.abs(st["F"][j] - oF[j]).max() <= max(1e-6, 1.001 * bq[j]) * np.abs(oF[j]).max(), j
    for lam, F in zip(st["lambda"], st["F"]):               # the eigenvalue kernel on the device's own F
        ref = np.linalg.eigvalsh(F)
        assert_allclose(lam, ref, rtol=0, atol=1e-12 * np.abs(ref).max())


def test_many_contexts_on_the_collapsed_and_the_shared_passes():
    """136 contexts: the donor-collapsed path equals the dense one (donor-constant genotypes), and several phenotypes in
    one pass equal their single scans."""
    from cellregmap_amd import CellRegMap, GenotypePanel, get_L_values, scan_interaction_many

    co = _cohort(3, 400, 136, 6, seed=77)
    rng = np.random.default_rng(5)
    Ls = get_L_values(co.hK, co.E)
    first = CellRegMap(co.y, co.E, W=co.W, Ls=Ls)
    crms = [first] + [CellRegMap(y, co.E, W=co.W, Ls=Ls, background=first._bg)
                      for y in (co.y[rng.permutation(co.y.size)], rng.normal(size=co.y.size))]
    dense, auto = GenotypePanel(co.G, groups=None), GenotypePanel(co.G)
    assert auto.n_groups == 3
    pv_dense, _ = first.scan_interaction(dense)
    pv_auto, _ = first.scan_interaction(auto)
    # (two summation orders through Brent's search are two faithful runs: the p-value tolerance, or the variant's own bound)
    parity_bounds.assert_p_within(pv_auto, pv_dense, parity_bounds.bounds(first, dense)[1], "collapsed")
    pv, info = scan_interaction_many(crms, dense)
    for i, crm in enumerate(crms):   # (the shared pass orders its sums by pair)
        parity_bounds.assert_p_within(pv[i], crm.scan_interaction(dense)[0], parity_bounds.bounds(crm, dense)[1], i)


def test_unsupported_sizes_fail_loudly():
    from cellregmap_amd import CellRegMap, _lib

    c = _cohort(8, 20, 3, 4, seed=35)
    rng = np.random.default_rng(1)
    W = np.concatenate([c.W, rng.normal(size=(c.y.size, 128))], axis=1)  # 129 covariate columns > 128
    with pytest.raises(_lib.CrmError, match="covariate columns"):
        CellRegMap(c.y, c.E, W=W).scan_interaction(c.G)
    c2 = _cohort(8, 40, 170, 4, seed=35)
    W = np.concatenate([c2.W, rng.normal(size=(c2.y.size, 119))], axis=1)  # 170 contexts + 120 covariates + 2 > 288
    crm = CellRegMap(c2.y, c2.E, W=W)
    with pytest.raises(_lib.CrmError, match="contexts \\+ covariates \\+ 2 <= 288"):
        crm.scan_interaction(c2.G)
    pv, info = crm.scan_association(c2.G, progress=False)      # ... which the association scans take (up to 128 columns)
    assert np.all(np.isfinite(pv))
    E = rng.normal(size=(c.y.size, 257))                               # 257 contexts > 256
    with pytest.raises(_lib.CrmError, match="contexts"):
        CellRegMap(c.y, E, W=c.W).scan_interaction(c.G)


def test_device_side_group_verification():
    """A cell that deviates from its candidate group in a column the host sample did not look at must
    keep the panel dense (the verification runs on the device over every column)."""
    from cellregmap_amd import CellRegMap, GenotypePanel

    c = _cohort(7, 20, 3, 400, seed=37)
    assert GenotypePanel(c.G).n_groups == 7
    G2 = c.G.copy()
    G2[33, 123] += 0.5                      # not among the 64 sampled columns
    panel = GenotypePanel(G2)
    assert panel.n_groups is None
    crm = CellRegMap(c.y, c.E, W=c.W, hK=c.hK)
    pv_auto, _ = crm.scan_interaction(panel)
    pv_dense, _ = crm.scan_interaction(GenotypePanel(G2, groups=None))
    assert np.array_equal(pv_auto, pv_dense)


def test_int8_dosage_ingest_standardised_on_the_device():
    """Donor-level allele counts as int8 + donor index (compact ingest): the device standardises
    every variant by the moments of its expanded column; results equal the float64 route where the host does that."""
    from cellregmap_amd import CellRegMap, GenotypePanel
    from cellregmap_amd.synth import donor_genotypes

    rng = np.random.default_rng(12)
    donors = 9
    cells_of = rng.integers(5, 30, size=donors)                # ragged donors
    donor_of_cell = np.repeat(np.arange(donors), cells_of)
    n = donor_of_cell.size
    D, _ = donor_genotypes(donors, 40, rng)                    # int8 allele counts, no monomorphic column
    E = rng.normal(size=(n, 4))
    hK = np.zeros((n, donors)); hK[np.arange(n), donor_of_cell] = 1.0
    Gx = D[donor_of_cell].astype(float)
    y = 0.5 * Gx[:, 3] * E[:, 0] + E @ rng.normal(size=4) * 0.3 + rng.normal(size=n)
    Gs = (Gx - Gx.mean(0)) / Gx.std(0)                         # the host route: standardise the expanded matrix
    crm = CellRegMap(y, E, hK=hK)
    ref_pv, ref_info = crm.scan_interaction(GenotypePanel(Gs, groups=None))
    panel = GenotypePanel.from_dosages(D, donor_of_cell)
    assert panel.n_groups == donors and panel.shape == (n, 40)
    pv, info = crm.scan_interaction(panel)
    assert np.array_equal(info["rho1"], ref_info["rho1"])
    assert np.all(np.abs(pv - ref_pv) <= 1e-5 * ref_pv + 1e-13)
    # against the float64 donor-level route the difference is rounding only
    pv2, _ = crm.scan_interaction(GenotypePanel.from_donors(Gs[np.r_[0, np.cumsum(cells_of)[:-1]]], donor_of_cell))
    assert np.all(np.abs(pv - pv2) <= 1e-7 * pv2 + 1e-13)
    # raw counts on request; a monomorphic variant cannot be standardised
    raw, _ = crm.scan_interaction(GenotypePanel.from_dosages(D, donor_of_cell, standardize=False))
    assert np.all(np.isfinite(raw))
    D2 = D.copy(); D2[:, 5] = 1
    with pytest.raises(ValueError, match="monomorphic"):
        GenotypePanel.from_dosages(D2, donor_of_cell)


def test_progress_callback_reports_every_block():
    from cellregmap_amd import CellRegMap, _engine, _lib
    from cellregmap_amd.synth import make_cohort

    c = make_cohort(6, 20, 3, 300, seed=8)
    crm = CellRegMap(c.y, c.E, W=c.W, hK=c.hK)
    lib = _lib.load()
    _lib.check(lib.crm_set_block_variants(_engine._context(0), 128))
    seen = []
    try:
        pv, _ = crm.scan_interaction(c.G, progress=lambda done, total: seen.append((done, total)))
    finally:
        _lib.check(lib.crm_set_block_variants(_engine._context(0), 0))
    assert seen == [(128, 300), (256, 300), (300, 300)]
    pv2, _ = crm.scan_interaction(c.G, progress=True)          # tqdm bar on stderr
    assert np.array_equal(pv, pv2)


def test_davies_info_is_surfaced():
    """chiscore's (liu_pval, Is_Converged) beside the p-value: same scan, same p-values, the oracle's info."""
    from cellregmap_amd import CellRegMap
    from cellregmap_amd.synth import make_cohort
    from oracle.crm import OracleCellRegMap
    from oracle.davies import davies_pvalue

    c = make_cohort(8, 20, 4, 20, seed=31)
    crm = CellRegMap(c.y, c.E, W=c.W, hK=c.hK)
    pv, _ = crm.scan_interaction(c.G)
    pv2, info = crm.scan_interaction_info(c.G)
    assert np.array_equal(pv, pv2)
    assert set(info) == {"liu_pval", "Is_Converged", "ifault", "model_flags", "degenerate", "flat_optimum", "statistic_at_tolerance",
                         "rho_tie", "bound_Q", "bound_p"} and info["ifault"].dtype == np.int32
    # a well-posed problem: nothing degenerate, no p-value at risk; the statistic may sit at the search's tolerance (bit 32)
    assert not info["degenerate"].any() and not (info["model_flags"] & ~32).any()
    assert np.all(info["bound_p"] <= 1e-5) and np.all(info["bound_Q"] >= 0) and np.all(np.isfinite(info["bound_Q"]))
    _, _, ost = OracleCellRegMap(c.y, c.E, W=c.W, hK=c.hK).scan_interaction(c.G, return_stats=True)
    for j in range(20):
        _, oinfo = davies_pvalue(ost["Q"][j], ost["F"][j], True)
        assert info["Is_Converged"][j] == oinfo["Is_Converged"]
        assert abs(info["liu_pval"][j] - oinfo["liu_pval"]) <= 1e-6 * oinfo["liu_pval"] + 1e-12


def test_null_p_values_are_not_inflated():
    """Statistical acceptance in the spirit of cellregmap/test/test_struct_lmm2.py:210-211, 278-279 (under the null the
    median p-value stays above 0.3 and the smallest above 0.04 over a handful of variants): here 1 500 independent
    variants on a phenotype with context and kinship structure but no genetic effect.  The score test with estimated
    variance components is mildly conservative on 60 donors (the p-values lean towards 1), so the check is one-sided:
    no inflation of small p-values."""
    from cellregmap_amd import CellRegMap, GenotypePanel

    rng = np.random.default_rng(2027)
    donors, cells, k = 60, 12, 5
    n = donors * cells
    donor = np.repeat(np.arange(donors), cells)
    E = rng.normal(size=(n, k)); E = (E - E.mean(0)) / E.std(0)
    hK = np.zeros((n, donors)); hK[np.arange(n), donor] = 1.0
    y = E @ rng.normal(size=k) * 0.4 + hK @ rng.normal(size=donors) * 0.6 + rng.normal(size=n)
    maf = rng.uniform(0.1, 0.5, size=1500)
    Gd = rng.binomial(2, maf, size=(donors, 1500)).astype(float)
    Gd = Gd[:, Gd.std(0) > 0]
    G = ((Gd - Gd.mean(0)) / Gd.std(0))[donor]
    pv, info = CellRegMap(y, E, hK=hK).scan_interaction(GenotypePanel(G))
    assert np.all((pv > 0) & (pv <= 1))
    assert np.mean(pv < 0.05) < 0.075 and np.mean(pv < 0.01) < 0.02
    assert np.median(pv) > 0.3
    assert pv.min() > 0.01 / pv.size      # nothing a Bonferroni threshold at 1 % would call


def test_cached_eigen_workspace_is_reused_and_can_be_released():
    """The constructor keeps its eigen-solver workspace in the context between calls (crm_ctx_trim releases it):
    a larger problem, a smaller one on the cached buffers, a release, and the first again -- bit-identical spectra."""
    import cellregmap_amd as crm
    from cellregmap_amd import _engine
    from cellregmap_amd.synth import make_cohort

    def spectra(c):
        _engine._bg_cache.clear()
        obj = crm.CellRegMap(c.y, c.E, W=c.W, Ls=crm.get_L_values(c.hK, c.E))
        return [obj._bg.read(i, c.y.size)[1] for i in range(11)]

    big, small = make_cohort(10, 30, 6, 8, seed=2), make_cohort(5, 12, 3, 8, seed=3)
    first = spectra(big)
    on_cached = spectra(small)
    crm.release_workspaces()
    crm.release_workspaces()                      # idempotent
    for a, b in zip(spectra(small), on_cached):
        assert np.array_equal(a, b)
    for a, b in zip(spectra(big), first):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("shape", [(8, 30, 5, 40), (6, 40, 37, 24), (10, 20, 60, 12)])
def test_gram_kernel_forms_agree(shape, kernel_form):
    """The score-statistic Gram through direct-to-LDS loads (default for k0 + c + 2 <= 64) against the register-staged
    kernel (crm_test_set_form("gram_staged"); also what wider problems and unaligned rows use): same Q and F to rounding.  The shapes
    cover 2 and 4 row blocks, spectra that are not multiples of the 64-column chunk, and one (k0 = 60) that only the
    staged kernel serves."""
    import cellregmap_amd as crm
    from cellregmap_amd.synth import make_cohort

    donors, cells, k, p = shape
    c = make_cohort(donors, cells, k, p, seed=17)
    obj = crm.CellRegMap(c.y, c.E, W=c.W, hK=c.hK)
    panel = crm.GenotypePanel(c.G, groups=None)
    pv, _, st = obj.scan_interaction(panel, return_stats=True)
    kernel_form("gram_staged", 1)
    pv2, _, st2 = obj.scan_interaction(panel, return_stats=True)
    scale = np.maximum(np.abs(st2["Q"]), np.trace(st2["F"], axis1=1, axis2=2))
    assert np.all(np.abs(st["Q"] - st2["Q"]) <= 1e-11 * scale)
    fs = np.abs(st2["F"]).max(axis=(1, 2), keepdims=True)
    assert np.all(np.abs(st["F"] - st2["F"]) <= 1e-11 * fs)
    assert np.all(np.abs(pv - pv2) <= 1e-6 * pv2 + 1e-13)


@pytest.mark.parametrize("route", [0, 2])
def test_spectrum_a_little_longer_than_a_multiple_of_the_tile(route, kernel_form):
    """A spectrum of 128 q + 16 entries.  Direct route (0, r = 2064): the last 144 columns of the Khatri-Rao contraction go
    through a launch of 160-column tiles; kinship-structure route (2, r = 1040): the last 16 columns of the mixing-matrix
    product through one pass over the operand (gemm_tn.hip: skinny_tn_kernel).  The form "kr_no_tail" keeps the single launch over nine columns of
    128-column tiles.  Both forms must give the same statistics to rounding, and the oracle's on a few variants.
    (Until round 6 this test ran on a cohort of 17 cells per donor against 16 contexts, whose spectrum is too ill-conditioned
    for the library to keep its half factor: the kinship-structure route never ran, and one shared launch counter hid it.)"""
    import cellregmap_amd as crm
    from cellregmap_amd.synth import make_cohort
    from oracle import crm as ocrm

    from cellregmap_amd import _engine, _lib

    variants = 2048
    if route == 0:       # rank 16 * 129 = 2064 = 16 x 128 + 16 (cols 2080) < n = 2193: the direct route's launch shape of rounds 2-5
        c = make_cohort(129, 17, 16, variants, seed=23)
        want_rank, want_groups = 2064, None     # (17 cells against 16 contexts per donor: too ill-conditioned for the half factor
                                                # to be kept -- no kinship structure, which the direct route does not need)
    else:                # rank 16 * 65 = 1040 = 8 x 128 + 16 (cols 1056) < n = 2600, forty cells per donor: structure kept
        c = make_cohort(65, 40, 16, variants, seed=23)
        want_rank, want_groups = 1040, 65
    rng = np.random.default_rng(1)
    G = c.G + 0.05 * rng.normal(size=c.G.shape)          # general genotypes: the dense path
    Ls = crm.get_L_values(c.hK, c.E)
    obj = crm.CellRegMap(c.y, c.E, W=c.W, Ls=Ls)
    assert max(obj._bg.rank(i) for i in range(11)) == want_rank
    if want_groups is not None:
        assert _lib.load().crm_background_kinship_groups(obj._bg.handle) == want_groups      # the structure is there to be used
    pv0, _ = obj.scan_interaction(crm.GenotypePanel(G, groups=None))
    lib, ctx = _lib.load(), _engine._context(0)
    panel = crm.GenotypePanel(G, groups=None)
    _lib.check(lib.crm_test_set_kinship_route(ctx, route))
    # (the direct route's 160-column-tile launch and the kinship-structure route's one-pass kernel count separately)
    tails = lib.crm_test_tail_launches if route == 0 else lib.crm_test_spectrum_tail_launches
    other = lib.crm_test_spectrum_tail_launches if route == 0 else lib.crm_test_tail_launches
    try:
        before, other_before = tails(ctx), other(ctx)
        pv, info, st = obj.scan_interaction(panel, return_stats=True)
        used = tails(ctx)
        assert used > before and other(ctx) == other_before   # the form under test ran (and only it) ...
        kernel_form("kr_no_tail", 1)
        pv1, info1, st1 = obj.scan_interaction(panel, return_stats=True)
        kernel_form("kr_no_tail", 0, reset=True)
        assert tails(ctx) == used                             # ... and the knob really switches it off
    finally:
        _lib.check(lib.crm_test_set_kinship_route(ctx, 1))
    assert np.array_equal(info["rho1"], info1["rho1"])
    scale = np.maximum(np.abs(st1["Q"]), np.trace(st1["F"], axis1=1, axis2=2))
    assert np.all(np.abs(st["Q"] - st1["Q"]) <= 1e-11 * scale)
    assert np.all(np.abs(st["F"] - st1["F"]) <= 1e-11 * np.abs(st1["F"]).max(axis=(1, 2), keepdims=True))
    assert np.all(np.abs(pv - pv1) <= 1e-6 * pv1 + 1e-13)
    pick = np.array([0, 5, 10, 777, 2047])
    opv, oinfo = ocrm.OracleCellRegMap(c.y, c.E, W=c.W, Ls=ocrm.khatri_rao_halves(c.hK, c.E)).scan_interaction(G[:, pick])
    parity_bounds.assert_p_within(pv[pick], opv, parity_bounds.bounds(obj, panel, pick)[1])


def test_null_fit_forms_are_bit_identical(kernel_form):
    """Null fits by LDS-sharing workgroups that draw variants from a queue (default from 1024 variants on, one covariate
    column) against one independent wavefront per (variant, grid point) (the form "nullfit_per_wave"): the arithmetic and its
    order are the same, so every output must agree bit for bit -- including which variants land where in the queue."""
    import cellregmap_amd as crm
    from cellregmap_amd.synth import make_cohort

    c = make_cohort(8, 30, 4, 1536, seed=3)
    for kw in ({"hK": c.hK}, {"Ls": crm.get_L_values(c.hK, c.E)}):
        obj = crm.CellRegMap(c.y, c.E, W=c.W, **kw)
        rng = np.random.default_rng(2)
        panel = crm.GenotypePanel(c.G + 0.05 * rng.normal(size=c.G.shape), groups=None)
        pv, info, st = obj.scan_interaction(panel, return_stats=True)
        kernel_form("nullfit_per_wave", 1)
        pv1, info1, st1 = obj.scan_interaction(panel, return_stats=True)
        kernel_form("nullfit_per_wave", 0, reset=True)
        assert np.array_equal(pv, pv1, equal_nan=True)
        for k in info:
            assert np.array_equal(info[k], info1[k], equal_nan=True), k
        for k in ("delta", "lml", "scale", "Q"):
            assert np.array_equal(st[k], st1[k], equal_nan=True), k


def test_wide_panel_on_a_small_cohort_against_the_oracle():
    """1 200 variants on 120 cells: enough variants for the LDS-sharing null-fit workgroups and their queue (from 1 024
    on), few enough cells for the oracle to scan all of them -- every p-value at the north-star tolerance."""
    import cellregmap_amd as crm
    from cellregmap_amd.synth import make_cohort
    from oracle import crm as ocrm

    c = make_cohort(6, 20, 3, 1200, seed=41)
    rng = np.random.default_rng(4)
    G = c.G + 0.05 * rng.normal(size=c.G.shape)
    obj = crm.CellRegMap(c.y, c.E, W=c.W, hK=c.hK)
    pv, info = obj.scan_interaction(crm.GenotypePanel(G, groups=None))
    opv, oinfo = ocrm.OracleCellRegMap(c.y, c.E, W=c.W, hK=c.hK).scan_interaction(G)
    assert np.all(np.abs(pv - opv) <= 1e-5 * opv + 1e-13), np.c_[pv, opv][np.abs(pv - opv) > 1e-5 * opv + 1e-13][:5]
    same = info["rho1"] == oinfo["rho1"]
    total = oinfo["e2"] + oinfo["g2"] + oinfo["eps2"]
    flat = (oinfo["e2"] + oinfo["g2"] <= 1e-6 * total) & (info["e2"] + info["g2"] <= 1e-6 * total)
    assert np.all(same | flat)


@pytest.mark.parametrize("variants", [1024, 1025, 4096])
def test_queue_drawn_null_fits_cover_every_variant(kernel_form, variants):
    """The LDS-sharing null-fit workgroups draw (variant, grid point) work from a queue with one ticket per wavefront
    (_cellregmap.py:345-357 fits all 11 grid points of every variant).  At the first size that uses the queue, one past it
    and a full 4096-variant block: the scan must complete -- launch_nullfit poisons the trial records and
    select_rho_kernel turns any slot no wavefront fitted into CRM_ERR_NUMERIC -- and agree bit for bit with one
    independent wavefront per (variant, grid point), i.e. no variant fitted twice in place of another."""
    import cellregmap_amd as crm
    from cellregmap_amd.synth import make_cohort

    c = make_cohort(6, 25, 3, variants, seed=11)
    obj = crm.CellRegMap(c.y, c.E, W=c.W, Ls=crm.get_L_values(c.hK, c.E))
    rng = np.random.default_rng(variants)
    panel = crm.GenotypePanel(c.G + 0.05 * rng.normal(size=c.G.shape), groups=None)
    pv, info, st = obj.scan_interaction(panel, return_stats=True)
    assert np.all(np.isfinite(st["lml"])) and np.all(np.isfinite(pv))
    kernel_form("nullfit_per_wave", 1)
    pv1, info1, st1 = obj.scan_interaction(panel, return_stats=True)
    kernel_form("nullfit_per_wave", 0, reset=True)
    assert np.array_equal(pv, pv1)
    for k in ("delta", "lml", "scale", "Q"):
        assert np.array_equal(st[k], st1[k]), k
    assert np.array_equal(info["rho1"], info1["rho1"])


def test_progress_is_reported_and_a_failing_callback_is_not_swallowed():
    """The reference shows a tqdm bar over variants (_cellregmap.py:340); here a callable sees (done, total) block by
    block, a scan started from inside a callback is refused with a status code, and an exception raised by the user's
    callback surfaces after the scan (ctypes would drop it otherwise)."""
    import cellregmap_amd as crm
    from cellregmap_amd import _engine, _lib
    from cellregmap_amd.synth import make_cohort

    c = make_cohort(6, 20, 3, 300, seed=5)
    obj = crm.CellRegMap(c.y, c.E, W=c.W, hK=c.hK)
    lib, ctx = _lib.load(), _engine._context(0)
    _lib.check(lib.crm_set_block_variants(ctx, 128))
    try:
        seen = []
        pv, _ = obj.scan_interaction(c.G, progress=lambda d, t: seen.append((d, t)))
        assert seen == [(128, 300), (256, 300), (300, 300)]
        pv0, _ = obj.scan_interaction(c.G, progress=False)
        assert np.array_equal(pv, pv0)

        def nested(done, total):   # a scan from inside a callback would reuse the running scan's work buffers: refused
            obj.scan_interaction(c.G[:, :130], progress=False)

        with pytest.raises(_lib.CrmError, match="another scan is running"):
            obj.scan_interaction(c.G, progress=nested)
        pv1, _ = obj.scan_interaction(c.G, progress=False)     # ... and the context is usable afterwards
        assert np.array_equal(pv, pv1)

        def bad(done, total):
            raise KeyError("user callback failed")

        with pytest.raises(KeyError, match="user callback failed"):
            obj.scan_interaction(c.G, progress=bad)
        assert not any(_engine._progress_stack.values())      # (keyed by device and calling thread)
        # association scans report too (the reference's tqdm at :270)
        seen.clear()
        obj.scan_association(c.G, progress=lambda d, t: seen.append((d, t)))
        assert seen and seen[-1] == (300, 300)
    finally:
        _lib.check(lib.crm_set_block_variants(ctx, 0))


def test_degenerate_models_are_flagged():
    """The problem families the fuzz generator leaves out (tests/fuzz_cases.py) because the reference's own answer is
    decided by rounding noise there are reported, not silently scanned: a saturated model (background columns + fixed
    effects span all cells; mode A with as many contexts as cells is the extreme case) warns at bind time and sets
    MODEL_SATURATED on every variant; a null fit that ends at delta = 0 (here: a phenotype without residual) sets
    MODEL_DELTA_AT_ZERO; a variant inside span(W) sets MODEL_G_IN_SPAN_W."""
    import warnings

    from cellregmap_amd import CellRegMap
    from fuzz_cases import random_problem

    # saturated: 40 cells, 45 contexts (mode A)
    y, E, W, G, kw = random_problem(40, 45, 1, 6, 5, seed=1, mode="A")
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        crm = CellRegMap(y, E, W=W, **kw)
        pv, info = crm.scan_interaction_info(G)
    assert any("saturated model" in str(w.message) for w in caught)
    assert np.all(info["model_flags"] & 1) and info["degenerate"].all()
    # a phenotype that the random effect explains without residual (y in span(E) + intercept, Sigma = EE'): the
    # likelihood rises all the way to delta = 0 and the fit ends at the clamp (the oracle reports delta = 2.2e-16)
    y, E, W, G, kw = random_problem(100, 3, 1, 5, 6, seed=3, mode="A")
    y = E @ np.array([1.0, -2.0, 0.5]) + 0.3
    with warnings.catch_warnings():
        warnings.simplefilter("error")          # not saturated: no warning
        pv, info = CellRegMap(y, E, W=W, **kw).scan_interaction_info(G)
    assert np.all((info["model_flags"] & 1) == 0)
    assert np.all(info["model_flags"] & 2) and info["degenerate"].all(), info["model_flags"]
    # a variant inside span(W)
    y, E, W, G, kw = random_problem(150, 3, 3, 5, 6, seed=3, mode="B")
    G = G.copy()
    G[:, 2] = W @ np.array([0.5, -1.0, 2.0])
    pv, info = CellRegMap(y, E, W=W, **kw).scan_interaction_info(G)
    assert info["model_flags"][2] & 4 and not np.any(np.delete(info["model_flags"], 2) & 4)


def test_scans_from_several_threads_on_one_device_are_serialised_by_the_library():
    """ctypes releases the GIL during a call, so two Python threads can enter the library at the same time on the one
    context of a device, whose work buffers a scan owns while it runs.  Every entry point takes the context's lock
    (crm::guarded_on): the calls queue up instead of corrupting each other, and every thread gets the result of a
    serial run, bit for bit."""
    import threading

    import cellregmap_amd as crm
    from cellregmap_amd.synth import make_cohort

    c = make_cohort(8, 40, 4, 700, seed=17)
    rng = np.random.default_rng(5)
    ys = [c.y, c.y[rng.permutation(c.y.size)], rng.normal(size=c.y.size), c.y + rng.normal(size=c.y.size)]
    first = crm.CellRegMap(ys[0], c.E, W=c.W, Ls=crm.get_L_values(c.hK, c.E))
    objs = [first] + [crm.CellRegMap(y, c.E, W=c.W, Ls=crm.get_L_values(c.hK, c.E), background=first._bg) for y in ys[1:]]
    panel = crm.GenotypePanel(c.G, groups=None)
    serial = [o.scan_interaction(panel, progress=False) for o in objs]
    got, errors = [None] * len(objs), []

    def work(i):
        try:
            for _ in range(3):
                got[i] = objs[i].scan_interaction(panel, progress=False)
        except Exception as exc:  # noqa: BLE001
            errors.append((i, exc))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(objs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for (pv, info), (spv, sinfo) in zip(got, serial):
        assert np.array_equal(pv, spv) and np.array_equal(info["rho1"], sinfo["rho1"])


@pytest.mark.parametrize("fold", [2, 0], ids=["folded", "unfolded"])
@pytest.mark.parametrize("donors,cells,k0,variants", [(6, 120, 7, 37), (12, 90, 20, 130), (5, 300, 50, 64)])
def test_per_donor_sums_from_the_symmetric_pair_features(donors, cells, k0, variants, fold, kernel_form):
    """The kinship term's contexts are the scan's own (run_interaction's default E2 = E): the kinship-structure route
    takes the per-donor sums S_d = sum_c g_c e_c e_c' from one batched product against E (x) E in donor order and the E1
    rows as their sum over the donors (scan.hip: donor pairs; blockops.hip: donor_pairs_expand_kernel) instead of the
    Khatri-Rao launch per donor plus the E1 rows' own product -- in its folded form (the rows of S from the products) and,
    round 6, in its unfolded one (the contraction over the donors with the kinship factor applied to the pair products,
    the rows of H'(g o E0) from the result: BASELINE config 2's form).  Both forms must give the same statistics to
    rounding -- odd context counts, variant counts that are no multiple of the four a workgroup takes -- and the oracle's."""
    import cellregmap_amd as crm
    from cellregmap_amd import _engine, _lib
    from cellregmap_amd.synth import make_cohort
    from oracle import crm as ocrm

    c = make_cohort(donors, cells, k0, variants, seed=300 + k0)
    rng = np.random.default_rng(k0)
    G = c.G + 0.05 * rng.normal(size=c.G.shape)          # general genotypes: the dense path
    kernel_form("kin_fold", fold)                         # (read when the structure is announced: folded with few contexts too / never)
    _engine._bg_cache.clear()
    Ls = crm.get_L_values(c.hK, c.E)
    assert Ls.device_us is not Ls.us and np.array_equal(Ls.device_us, c.E)   # the device gets the contexts' own basis
    obj = crm.CellRegMap(c.y, c.E, W=c.W, Ls=Ls)
    lib, ctx = _lib.load(), _engine._context(0)
    assert (lib.crm_background_kinship_folded(obj._bg.handle) > 0) == (fold == 2)
    assert lib.crm_background_kinship_groups(obj._bg.handle) == donors
    panel = crm.GenotypePanel(G, groups=None)
    _lib.check(lib.crm_test_set_kinship_route(ctx, 2))
    try:
        kernel_form("donor_pairs", 2)                     # always (the cost model would leave small cohorts to the old form)
        before = lib.crm_test_donor_pair_blocks(ctx)
        pv, info, st = obj.scan_interaction(panel, return_stats=True)
        used = lib.crm_test_donor_pair_blocks(ctx)
        assert used > before                              # the form under test ran ...
        kernel_form("donor_pairs", 0)
        pv0, info0, st0 = obj.scan_interaction(panel, return_stats=True)
        assert lib.crm_test_donor_pair_blocks(ctx) == used   # ... and the knob really switches it off
    finally:
        _lib.check(lib.crm_test_set_kinship_route(ctx, 1))
    assert np.array_equal(info["rho1"], info0["rho1"])
    scale = np.maximum(np.abs(st0["Q"]), np.trace(st0["F"], axis1=1, axis2=2))
    assert np.all(np.abs(st["Q"] - st0["Q"]) <= 1e-10 * scale)
    fs = np.abs(st0["F"]).max(axis=(1, 2), keepdims=True)
    assert np.all(np.abs(st["F"] - st0["F"]) <= 1e-10 * fs)
    assert np.all(np.abs(pv - pv0) <= 1e-6 * pv0 + 1e-13)
    sel = np.arange(0, variants, max(1, variants // 6))
    opv, oinfo = ocrm.OracleCellRegMap(c.y, c.E, W=c.W, Ls=ocrm.khatri_rao_halves(c.hK, c.E)).scan_interaction(G[:, sel])
    parity_bounds.assert_p_within(pv[sel], opv, parity_bounds.bounds(obj, panel, sel)[1])


def test_fits_without_a_kinship_term_need_no_rotated_test_direction(kernel_form):
    """A phenotype without a random effect ends its null fits at the upper clamp of delta: v0 = 2.2e-16 scale, K0 = v1 I up to
    1e-10 -- the rotated test direction A~ = Q0(rho*)'(g o E0) enters Q and F with weights below 1e-10 and is not formed
    (scan.hip: no_kinship_term; the Gram kernels read rows of zeros).  Against the form that forms it for every test: the
    same rho*, Q and F to 1e-9, p to 1e-7; the oracle's p-values; and, scanned together with a phenotype that has a random
    effect, bit for bit what each phenotype gets on its own."""
    import cellregmap_amd as crm
    from cellregmap_amd import _engine, _lib
    from cellregmap_amd.synth import make_cohort
    from oracle import crm as ocrm

    c = make_cohort(8, 150, 12, 96, seed=77)
    rng = np.random.default_rng(5)
    G = c.G + 0.05 * rng.normal(size=c.G.shape)
    y_flat = c.y[rng.permutation(c.y.size)]              # the kinship structure is gone: delta -> 1
    Ls = crm.get_L_values(c.hK, c.E)
    plain = crm.CellRegMap(c.y, c.E, W=c.W, Ls=Ls)
    flat = crm.CellRegMap(y_flat, c.E, W=c.W, Ls=Ls, background=plain._bg)
    lib, ctx = _lib.load(), _engine._context(0)
    panel = crm.GenotypePanel(G, groups=None)
    before = lib.crm_test_tests_without_pair(ctx)
    pv, info, st = flat.scan_interaction(panel, return_stats=True)
    skipped = lib.crm_test_tests_without_pair(ctx) - before
    assert skipped >= 10, skipped                         # a good part of its fits end at the clamp ...
    at_clamp = (info["e2"] + info["g2"]) <= 1e-13 * info["eps2"]
    assert at_clamp.sum() >= skipped
    kernel_form("pairs_without_kinship_term", 0)
    mark = lib.crm_test_tests_without_pair(ctx)
    pv0, info0, st0 = flat.scan_interaction(panel, return_stats=True)
    assert lib.crm_test_tests_without_pair(ctx) == mark   # ... and the knob forms every A~
    kernel_form("pairs_without_kinship_term", 0, reset=True)
    assert np.array_equal(info["rho1"], info0["rho1"])
    scale = np.maximum(np.abs(st0["Q"]), np.trace(st0["F"], axis1=1, axis2=2))
    assert np.all(np.abs(st["Q"] - st0["Q"]) <= 1e-9 * scale)
    fs = np.abs(st0["F"]).max(axis=(1, 2), keepdims=True)
    assert np.all(np.abs(st["F"] - st0["F"]) <= 1e-9 * fs)
    assert np.all(np.abs(st["lambda"] - st0["lambda"]) <= 1e-9 * np.abs(st0["lambda"]).max(axis=1, keepdims=True))
    assert np.all(np.abs(pv - pv0) <= 1e-7 * pv0 + 1e-13)
    sel = np.arange(0, G.shape[1], 12)
    opv, _ = ocrm.OracleCellRegMap(y_flat, c.E, W=c.W, Ls=ocrm.khatri_rao_halves(c.hK, c.E)).scan_interaction(G[:, sel])
    parity_bounds.assert_p_within(pv[sel], opv, parity_bounds.bounds(flat, panel, sel)[1])
    # both phenotypes in one pass
    both, _ = crm.scan_interaction_many([plain, flat], panel)
    pv_plain, _ = plain.scan_interaction(panel)
    assert np.array_equal(both[0], pv_plain) and np.array_equal(both[1], pv)


@pytest.mark.parametrize("r_mod", ["short", "ragged", "long"])
def test_four_null_fits_per_wavefront_give_the_bits_of_one(r_mod, kernel_form):
    """The LDS-shared null-fit kernel runs four fits per wavefront, one per row of sixteen lanes (nullfit.hip: G = 4): every
    lane stands for four lanes of the one-fit form -- the same spectrum entries in the same order into four accumulators,
    the same pairings of the butterfly -- so delta, lml and the scale must be those of the one-fit form ("nullfit_one_per_wave")
    to the last bit, whatever the length of the spectrum modulo the 256 entries of a trip: ~60 (shorter than a wavefront),
    ~1 170 (a ragged last trip), ~2 064.  1 100 variants: the queue's last ticket holds fewer than four."""
    import cellregmap_amd as crm
    from cellregmap_amd.synth import make_cohort

    donors, cells, k0 = {"short": (4, 300, 12), "ragged": (61, 30, 19), "long": (129, 17, 16)}[r_mod]
    c = make_cohort(donors, cells, k0, 1100, seed=91)
    rng = np.random.default_rng(4)
    G = c.G + 0.05 * rng.normal(size=c.G.shape)
    obj = crm.CellRegMap(c.y, c.E, W=c.W, Ls=crm.get_L_values(c.hK, c.E))
    lo, hi = {"short": (40, 63), "ragged": (1100, 1178), "long": (2049, 2080)}[r_mod]
    assert lo <= max(obj._bg.rank(i) for i in range(11)) <= hi
    panel = crm.GenotypePanel(G, groups=None)
    pv, info, st = obj.scan_interaction(panel, return_stats=True)
    xi = obj.scan_interaction_info(panel)[1]
    kernel_form("nullfit_one_per_wave", 1)
    pv1, info1, st1 = obj.scan_interaction(panel, return_stats=True)
    xi1 = obj.scan_interaction_info(panel)[1]
    kernel_form("nullfit_one_per_wave", 0, reset=True)
    for k in ("delta", "lml", "scale", "Q"):
        assert np.array_equal(st[k], st1[k]), k
    assert np.array_equal(pv, pv1) and np.array_equal(info["rho1"], info1["rho1"])
    # ... and the tracked kernels of the two forms leave the same trace
    for k in ("bound_Q", "bound_p", "model_flags"):
        assert np.array_equal(xi[k], xi1[k]), k


def test_resumable_scan_on_the_device(tmp_path):
    """``scan_interaction_resumable`` with the device as the per-chunk scan: chunks that are multiples of the scan's block give
    the one-panel scan bit for bit, and a second call on the finished checkpoint touches no GPU work at all."""
    from cellregmap_amd import CellRegMap, _engine, _lib, scan_interaction_resumable
    from cellregmap_amd.synth import make_cohort

    c = make_cohort(6, 20, 3, 600, seed=8)
    crm = CellRegMap(c.y, c.E, W=c.W, hK=c.hK)
    lib, ctx = _lib.load(), _engine._context(0)
    _lib.check(lib.crm_set_block_variants(ctx, 128))
    try:
        pv, info = crm.scan_interaction(c.G, groups=None)
        path = str(tmp_path / "scan.npz")
        rpv, rinfo = scan_interaction_resumable(crm, c.G, path, chunk=256, scan=lambda G, a, b: crm.scan_interaction(G, a, b, groups=None))
    finally:
        _lib.check(lib.crm_set_block_variants(ctx, 0))
    assert np.array_equal(rpv, pv) and all(np.array_equal(rinfo[k], info[k]) for k in info)

    def no_scan(*args):
        raise AssertionError("a finished checkpoint must not scan")

    again, _ = scan_interaction_resumable(crm, c.G, path, chunk=256, scan=no_scan)
    assert np.array_equal(again, pv)
