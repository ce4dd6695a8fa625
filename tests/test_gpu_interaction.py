"""End-to-end parity of the HIP interaction scan against the CPU oracle (same seeded inputs).

Tolerances (BASELINE.json north_star): score statistic Q rtol 1e-6, p-values rtol 1e-5.  The
p-value check carries an absolute floor of 1e-13 because Davies' result is 1 - (0.5 - sum): for
p < 1e-8 the last digits are summation-order noise in both implementations.

Two procedures are checked, each on both sides (oracle/lmm.py ``fit(polish=...)``,
``crm_set_null_fit_polish``):
  * default -- the reference's procedure verbatim (Brent on logit(delta), rtol = atol = 1e-6).
    Both sides take the same Brent path; where a comparison inside Brent is decided by the last
    bits of the likelihood the paths may part and the results then differ by up to the
    optimiser's own tolerance, which the north-star tolerances cover;
  * polish -- Brent followed by secant steps on the analytic derivative: both sides must then
    agree to ~1e-8 whatever the summation order."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

pytestmark = pytest.mark.gpu

P_RTOL, P_ATOL = 1e-5, 1e-13
Q_RTOL = 1e-6


def _assert_fit_info_matches(info, oinfo, pv, opv):
    """rho* and the variance components against the oracle.  Where the background explains nothing
    (v0 -> 0, delta -> 1) the covariance is v1 I whatever rho is: every grid point has the same likelihood
    and the reference's strict '>' (cellregmap/_cellregmap.py:354) is decided by rounding noise -- there
    rho* may differ, the p-value must not."""
    assert np.all(np.abs(pv - opv) <= P_RTOL * opv + P_ATOL), np.c_[pv, opv]
    total = oinfo["e2"] + oinfo["g2"] + oinfo["eps2"]
    flat = (oinfo["e2"] + oinfo["g2"] <= 1e-6 * total) & (info["e2"] + info["g2"] <= 1e-6 * total)
    same = info["rho1"] == oinfo["rho1"]
    assert np.all(same | flat), np.c_[info["rho1"], oinfo["rho1"], oinfo["e2"] + oinfo["g2"], total]
    assert_allclose(info["eps2"], oinfo["eps2"], rtol=1e-5)
    for k in ("e2", "g2"):  # (a component at the boundary only has absolute accuracy)
        assert np.all(~same | (np.abs(info[k] - oinfo[k]) <= 1e-5 * oinfo[k] + 1e-6 * total))


def _cohort(donors, cells, k, p, seed):
    from cellregmap_amd.synth import make_cohort

    return make_cohort(donors, cells, k, p, seed=seed)


def _compare(pv, info, stats, opv, oinfo, ostats, tight=False):
    """tight: both sides ran the polished procedure -> 1e-8 class agreement; otherwise the
    north-star tolerances."""
    assert_allclose(info["rho1"], oinfo["rho1"], rtol=0, atol=1e-12)
    assert_allclose(stats["delta"], ostats["delta"], rtol=1e-8 if tight else 5e-6)
    assert_allclose(stats["lml"], ostats["lml"], rtol=1e-11)
    for key in ("e2", "g2", "eps2"):
        assert_allclose(info[key], oinfo[key], rtol=1e-8 if tight else 1e-5, atol=1e-12)
    assert_allclose(stats["Q"], ostats["Q"], rtol=1e-8 if tight else Q_RTOL)
    rt = 1e-7 if tight else P_RTOL
    assert np.all(np.abs(pv - opv) <= rt * opv + P_ATOL), np.c_[pv, opv]


@pytest.mark.parametrize("genotypes", ["dense", "donor-collapsed"])
@pytest.mark.parametrize("mode", ["A", "B", "C", "C-eigh"])
def test_interaction_matches_oracle(mode, genotypes):
    from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib, get_L_values
    from oracle.crm import OracleCellRegMap, khatri_rao_halves

    if mode == "C-eigh":      # cols = k + k*donors >= n -> the reference's eigh branch
        c = _cohort(12, 10, 10, 24, seed=3)
    elif mode == "C":         # cols = 4 + 4*6 = 28 < n = 240 -> thin branch
        c = _cohort(6, 40, 4, 24, seed=4)
    else:
        c = _cohort(10, 20, 5, 24, seed=5)
    kw, okw = {}, {}
    if mode == "B":
        kw["hK"] = okw["hK"] = c.hK
    elif mode.startswith("C"):
        kw["Ls"] = get_L_values(c.hK, c.E)
        okw["Ls"] = khatri_rao_halves(c.hK, c.E)
    crm = CellRegMap(c.y, c.E, W=c.W, **kw)
    panel = GenotypePanel(c.G, groups=None if genotypes == "dense" else "auto")
    assert (panel.n_groups is None) == (genotypes == "dense")
    pv, info, stats = crm.scan_interaction(panel, return_stats=True)
    ocrm = OracleCellRegMap(c.y, c.E, W=c.W, **okw)
    opv, oinfo, ostats = ocrm.scan_interaction(c.G, return_stats=True)
    _compare(pv, info, stats, opv, oinfo, ostats)
    F = np.stack(ostats["F"])
    scale = np.abs(F).max(axis=(1, 2), keepdims=True)
    assert np.all(np.abs(stats["F"] - F) <= 1e-6 * scale)
    # polished procedure on both sides: the same mathematics to ~1e-8
    lib = _lib.load()
    _lib.check(lib.crm_set_null_fit_polish(_engine._context(0), 1))
    try:
        pv, info, stats = crm.scan_interaction(panel, return_stats=True)
    finally:
        _lib.check(lib.crm_set_null_fit_polish(_engine._context(0), 0))
    pcrm = OracleCellRegMap(c.y, c.E, W=c.W, polish=True, **okw)
    ppv, pinfo, pstats = pcrm.scan_interaction(c.G, return_stats=True)
    _compare(pv, info, stats, ppv, pinfo, pstats, tight=True)
    F = np.stack(pstats["F"])
    assert np.all(np.abs(stats["F"] - F) <= 1e-8 * scale)


def test_run_interaction_config1_subset():
    """BASELINE config 1 (500 cells, 10 contexts; first 48 of the 200 variants), through the
    functional wrapper, mode C with cols = 510 >= n."""
    from cellregmap_amd import run_interaction
    from cellregmap_amd.synth import make_config
    from oracle import crm as ocrm

    c = make_config("cfg1", n_variants=48)
    pv, info = run_interaction(c.y, c.E, c.G, W=c.W, hK=c.hK)
    opv, oinfo = ocrm.run_interaction(c.y, c.E, c.G, W=c.W, hK=c.hK)
    assert_allclose(info["rho1"], oinfo["rho1"], atol=1e-12)
    assert np.all(np.abs(pv - opv) <= P_RTOL * opv + P_ATOL), np.c_[pv, opv]
    # the planted GxC variants (10, 11) must be the most significant ones
    assert set(np.argsort(pv)[:2]) == {10, 11}


def test_permutation_hooks_and_extra_covariates():
    from cellregmap_amd import CellRegMap
    from oracle.crm import OracleCellRegMap

    c = _cohort(8, 25, 4, 16, seed=9)
    rng = np.random.default_rng(1)
    W = np.concatenate([c.W, rng.normal(size=(c.y.size, 2))], axis=1)
    idx_E = rng.permutation(c.y.size)
    idx_G = rng.permutation(c.y.size)
    crm = CellRegMap(c.y, c.E, W=W, hK=c.hK)
    ocrm = OracleCellRegMap(c.y, c.E, W=W, hK=c.hK)
    for kw in ({"idx_E": idx_E}, {"idx_G": idx_G}, {"idx_E": idx_E, "idx_G": idx_G}):
        pv, info, stats = crm.scan_interaction(c.G, return_stats=True, **kw)
        opv, oinfo, ostats = ocrm.scan_interaction(c.G, return_stats=True, **kw)
        _compare(pv, info, stats, opv, oinfo, ostats)


def test_collapsed_path_equals_dense_path():
    """Donor-constant genotypes: the collapsed path is an exact rearrangement of the dense one
    (also with the context permutation hook); the genotype permutation hook and general G run
    dense."""
    from cellregmap_amd import CellRegMap, GenotypePanel, detect_groups
    from cellregmap_amd.synth import make_cohort

    c = make_cohort(9, 30, 6, 40, seed=17)
    rng = np.random.default_rng(3)
    W = np.concatenate([c.W, rng.normal(size=(c.y.size, 1))], axis=1)
    crm = CellRegMap(c.y, c.E, W=W, hK=c.hK)
    dense = GenotypePanel(c.G, groups=None)
    auto = GenotypePanel(c.G)
    donors = GenotypePanel.from_donors(c.G[::30], c.donor_of_cell)
    assert auto.n_groups == 9 and donors.n_groups == 9
    idx_E = rng.permutation(c.y.size)
    idx_G = rng.permutation(c.y.size)
    from cellregmap_amd import _engine, _lib

    lib = _lib.load()
    for polish in (0, 1):
        # polish = 1 pins the null-fit optimum, so the two summation orders must agree to ~1e-9;
        # with the reference's Brent(1e-6) procedure they agree within its own tolerance
        _lib.check(lib.crm_set_null_fit_polish(_engine._context(0), polish))
        try:
            for kw in ({}, {"idx_E": idx_E}, {"idx_G": idx_G}):
                ref = crm.scan_interaction(dense, return_stats=True, **kw)
                for panel in (auto, donors):
                    got = crm.scan_interaction(panel, return_stats=True, **kw)
                    assert np.array_equal(got[1]["rho1"], ref[1]["rho1"])
                    # (no polish: two summation orders under Brent(1e-6) may part by its tolerance)
                    assert_allclose(got[2]["Q"], ref[2]["Q"], rtol=1e-8 if polish else 5e-6)
                    assert_allclose(got[2]["delta"], ref[2]["delta"], rtol=1e-8 if polish else 5e-6)
                    rt = 1e-7 if polish else P_RTOL
                    assert np.all(np.abs(got[0] - ref[0]) <= rt * ref[0] + P_ATOL)
        finally:
            _lib.check(lib.crm_set_null_fit_polish(_engine._context(0), 0))
    # general (not donor-constant) genotypes are never collapsed
    Gr = rng.normal(size=c.G.shape)
    assert detect_groups(Gr) is None
    pv_general, _ = crm.scan_interaction(Gr)
    assert np.all(np.isfinite(pv_general))


def test_blocks_and_ragged_tail():
    """Block boundaries must not matter: 300 variants in blocks of 128 == one block."""
    import ctypes

    from cellregmap_amd import CellRegMap, _engine, _lib

    c = _cohort(10, 12, 3, 300, seed=11)
    crm = CellRegMap(c.y, c.E, W=c.W, hK=c.hK)
    lib = _lib.load()
    pv1, info1 = crm.scan_interaction(c.G)
    _lib.check(lib.crm_set_block_variants(_engine._context(0), 128))
    try:
        pv2, info2 = crm.scan_interaction(c.G)
    finally:
        _lib.check(lib.crm_set_block_variants(_engine._context(0), 0))
    assert np.array_equal(pv1, pv2)
    assert np.array_equal(info1["rho1"], info2["rho1"])


def test_errors_are_loud():
    from cellregmap_amd import CellRegMap

    c = _cohort(5, 8, 3, 4, seed=2)
    with pytest.raises(AssertionError):
        CellRegMap(c.y, c.E[:-1], W=c.W)
    y = c.y.copy()
    y[3] = np.nan
    with pytest.raises(ValueError):
        CellRegMap(y, c.E, W=c.W).scan_interaction(c.G)
    with pytest.raises(ValueError):
        CellRegMap(c.y, c.E, W=c.W).scan_interaction(c.G[:-1])


@pytest.mark.parametrize("genotypes", ["dense", "donor-collapsed"])
def test_many_phenotypes_in_one_pass_equal_separate_scans(genotypes):
    """BASELINE config 4 in miniature: several genes x one panel.  The shared pass must return, per
    gene, exactly what the single-gene scan returns (same kernels on the same operands)."""
    from cellregmap_amd import CellRegMap, GenotypePanel, get_L_values, run_interaction_many, scan_interaction_many
    from cellregmap_amd.synth import make_cohort

    c = make_cohort(8, 25, 4, 70, seed=41)
    rng = np.random.default_rng(7)
    Y = np.stack([c.y, c.y[rng.permutation(c.y.size)], rng.normal(size=c.y.size), c.y + rng.normal(size=c.y.size)], axis=1)
    Ls = get_L_values(c.hK, c.E)
    first = CellRegMap(Y[:, 0], c.E, W=c.W, Ls=Ls)
    crms = [first] + [CellRegMap(Y[:, i], c.E, W=c.W, Ls=Ls, background=first._bg) for i in range(1, 4)]
    panel = GenotypePanel(c.G, groups=None if genotypes == "dense" else "auto")
    for kw in ({}, {"idx_E": rng.permutation(c.y.size)}, {"idx_G": rng.permutation(c.y.size)}):
        pv, info = scan_interaction_many(crms, panel, **kw)
        assert pv.shape == (4, 70)
        assert len(set(np.unique(info["rho1"]))) > 1  # the genes do not all agree on rho*
        for i, crm in enumerate(crms):
            spv, sinfo = crm.scan_interaction(panel, **kw)
            assert np.array_equal(pv[i], spv)
            for k in sinfo:
                assert np.array_equal(info[k][i], sinfo[k])
    pv2, _ = run_interaction_many(Y, c.E, c.G, W=c.W, hK=c.hK)
    assert_allclose(pv2, scan_interaction_many(crms, GenotypePanel(c.G))[0], rtol=1e-12)


def test_phenotypes_bound_in_a_batch_are_those_bound_one_by_one():
    """crm_gene_create_batch / crm_gene_create_like (the covariates and contexts of the first phenotype copied on the
    device, the rotations of a batch as one product) against crm_gene_create per phenotype: the same bits, with several
    covariate columns and correlated ones (orthogonalised inside the library on the first bind only)."""
    from cellregmap_amd import CellRegMap, GenotypePanel, get_L_values, scan_interaction_many
    from cellregmap_amd.synth import make_cohort

    c = make_cohort(8, 25, 4, 40, seed=43)
    rng = np.random.default_rng(9)
    n = c.y.size
    W = np.concatenate([c.W, rng.normal(size=(n, 2)) + 0.3], axis=1)
    Y = np.stack([c.y, c.y[rng.permutation(n)], rng.normal(size=n), c.y + rng.normal(size=n), c.y ** 2], axis=1)
    Ls = get_L_values(c.hK, c.E)
    first = CellRegMap(Y[:, 0], c.E, W=W, Ls=Ls)
    batch = [first] + [CellRegMap(Y[:, i], c.E, W=W, Ls=Ls, background=first._bg) for i in range(1, 5)]
    alone = [CellRegMap(Y[:, i], c.E, W=W, Ls=Ls, background=first._bg) for i in range(5)]
    panel = GenotypePanel(c.G, groups=None)
    pv, info = scan_interaction_many(batch, panel)                  # binds 1 .. 4 in one batch
    assert all(b._gene is not None for b in batch)
    one = CellRegMap(Y[:, 3], c.E, W=W, Ls=Ls, background=first._bg)
    one._bind_gene(like=first)                                      # ... and one through crm_gene_create_like
    for i, a in enumerate(alone):
        spv, sinfo = a.scan_interaction(panel)                      # crm_gene_create
        assert np.array_equal(pv[i], spv)
        for k in sinfo:
            assert np.array_equal(info[k][i], sinfo[k])
    assert np.array_equal(one.scan_interaction(panel)[0], pv[3])
    with pytest.raises(ValueError, match="non-finite"):
        bad = CellRegMap(np.where(np.arange(n) == 3, np.nan, c.y), c.E, W=W, Ls=Ls, background=first._bg)
        scan_interaction_many([first, bad], panel)


@pytest.mark.parametrize("hook", ["none", "E", "G"])
def test_mode_b_on_the_kinship_structure_route(hook):
    """Mode B (hS = [sqrt(rho) E1, sqrt(1 - rho) hK]) with a donor-expanded hK is the same structure with a single column
    of ones in the place of us: the folded form then takes the us rows as per-donor sums of the Khatri-Rao rows themselves
    -- a plain batched product G_d'(us o E0)_d -- and the E1 rows through the pair features.  Against the direct route
    (optimum pinned) and the oracle, dense full-rank donor-level factor, ragged donors, both hooks, several phenotypes."""
    from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib, scan_interaction_many
    from oracle.crm import OracleCellRegMap

    rng = np.random.default_rng(91)
    donors, k0, p = 11, 6, 45
    donor = np.repeat(np.arange(donors), rng.integers(9, 50, size=donors))
    n = donor.size
    hK = rng.normal(size=(donors, donors))[donor]
    E = rng.normal(size=(n, k0))
    W = np.concatenate([np.ones((n, 1)), rng.normal(size=(n, 1))], axis=1)
    G = rng.normal(size=(n, p))
    y = 0.5 * G[:, 3] * E[:, 0] + E @ rng.normal(size=k0) * 0.3 + hK @ rng.normal(size=donors) * 0.2 + rng.normal(size=n)
    idx = rng.permutation(n)
    hooks = {} if hook == "none" else ({"idx_E": idx} if hook == "E" else {"idx_G": idx})
    crm = CellRegMap(y, E, W=W, hK=hK)
    lib, ctx = _lib.load(), _engine._context(0)
    assert lib.crm_background_kinship_groups(crm._bg.handle) == donors and lib.crm_background_kinship_folded(crm._bg.handle) > 0
    panel = GenotypePanel(G, groups=None)
    try:
        _lib.check(lib.crm_set_null_fit_polish(ctx, 1))
        _lib.check(lib.crm_test_set_kinship_route(ctx, 2))
        try:
            ppv, pinfo, pst = crm.scan_interaction(panel, return_stats=True, progress=False, **hooks)
            _lib.check(lib.crm_test_set_kinship_route(ctx, 0))
            pv0, info0, st0 = crm.scan_interaction(panel, return_stats=True, progress=False, **hooks)
        finally:
            _lib.check(lib.crm_test_set_kinship_route(ctx, 2))
            _lib.check(lib.crm_set_null_fit_polish(ctx, 0))
        assert np.array_equal(pinfo["rho1"], info0["rho1"])
        scale = np.maximum(np.abs(st0["Q"]), np.trace(st0["F"], axis1=1, axis2=2))
        assert np.all(np.abs(pst["Q"] - st0["Q"]) <= 1e-9 * scale)
        assert np.all(np.abs(pst["F"] - st0["F"]) <= 1e-9 * np.abs(st0["F"]).max(axis=(1, 2), keepdims=True))
        pv, info, st = crm.scan_interaction(panel, return_stats=True, progress=False, **hooks)
        opv, oinfo, ost = OracleCellRegMap(y, E, W=W, hK=hK).scan_interaction(G, return_stats=True, **hooks)
        _compare(pv, info, st, opv, oinfo, ost)
        ys = [y, y[rng.permutation(n)], rng.normal(size=n)]
        crms = [crm] + [CellRegMap(v, E, W=W, hK=hK, background=crm._bg) for v in ys[1:]]
        mpv, minfo = scan_interaction_many(crms, panel, **hooks)
        for i, obj in enumerate(crms):
            spv, sinfo = obj.scan_interaction(panel, progress=False, **hooks)
            assert np.array_equal(minfo["rho1"][i], sinfo["rho1"])
            assert np.all(np.abs(mpv[i] - spv) <= 1e-7 * spv + 1e-13)
    finally:
        _lib.check(lib.crm_test_set_kinship_route(ctx, 1))
        _lib.check(lib.crm_set_null_fit_polish(ctx, 0))


@pytest.mark.parametrize("mode", ["B", "C"])
def test_folded_route_with_an_E1_of_its_own(mode, kernel_form):
    """E1 given and different from E (other columns, another count): the folded form's E1 rows then come from the general
    pair features E1_a o E0_i (k1 k0 columns), not from the symmetric ones the scan shares with E0'diag(g^2)E0 when E1 is E
    itself.  Against the oracle and against the direct route."""
    from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib, get_L_values
    from oracle.crm import OracleCellRegMap, khatri_rao_halves

    kernel_form("kin_fold", 2)      # (mode C folds by itself from 32 columns of us on)
    rng = np.random.default_rng(17 if mode == "B" else 18)
    donors, k0, k1, p = 9, 5, 3, 40
    donor = np.repeat(np.arange(donors), rng.integers(12, 40, size=donors))
    n = donor.size
    hK = rng.normal(size=(donors, donors))[donor]
    E, E1 = rng.normal(size=(n, k0)), rng.normal(size=(n, k1))
    W = np.ones((n, 1))
    G = rng.normal(size=(n, p))
    y = 0.5 * G[:, 3] * E[:, 0] + E1 @ rng.normal(size=k1) * 0.3 + hK @ rng.normal(size=donors) * 0.2 + rng.normal(size=n)
    kw, okw = (dict(hK=hK),) * 2 if mode == "B" else (dict(Ls=get_L_values(hK, E)), dict(Ls=khatri_rao_halves(hK, E)))
    crm = CellRegMap(y, E, W=W, E1=E1, **kw)
    lib, ctx = _lib.load(), _engine._context(0)
    assert lib.crm_background_kinship_groups(crm._bg.handle) == donors and lib.crm_background_kinship_folded(crm._bg.handle) > 0
    panel = GenotypePanel(G, groups=None)
    try:
        _lib.check(lib.crm_test_set_kinship_route(ctx, 2))
        pv, info, st = crm.scan_interaction(panel, return_stats=True, progress=False)
        _lib.check(lib.crm_test_set_kinship_route(ctx, 0))
        pv0, info0, st0 = crm.scan_interaction(panel, return_stats=True, progress=False)
    finally:
        _lib.check(lib.crm_test_set_kinship_route(ctx, 1))
    assert np.array_equal(info["rho1"], info0["rho1"])
    assert_allclose(st["lml"], st0["lml"], rtol=1e-12)
    # (the two routes are two faithful runs of the search: the statistic's tolerance, or the variant's own bound)
    import parity_bounds

    parity_bounds.assert_Q_within(st["Q"], st0["Q"], parity_bounds.bounds(crm, panel)[0])
    opv, oinfo, ost = OracleCellRegMap(y, E, W=W, E1=E1, **okw).scan_interaction(G, return_stats=True)
    _compare(pv, info, st, opv, oinfo, ost)


@pytest.mark.parametrize("route", [2, 0, "folded"])
def test_pair_stage_over_sub_ranges_of_a_block(route, monkeypatch, kernel_form):
    """Several phenotypes can ask for more (variant, rho*) pairs than the pair-ordered buffers hold (min(11, genes) per
    variant in the worst case): the block keeps its size for the stages before -- the per-phenotype null fits above all --
    and the pair stage runs over sub-ranges of its variants (scan.hip: pair_cap).  With the buffers cut down to the
    smallest size the library accepts (CRM_PAIR_BUFFER_GB), 700 variants x 5 phenotypes take several sub-ranges and must
    give exactly what the unrestricted pass gives, on the kinship-structure route and on the direct one, with and without
    the permutation hooks."""
    from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib, get_L_values, scan_interaction_many
    from cellregmap_amd.synth import make_cohort

    folded = route == "folded"
    if folded:     # the kinship-structure route with the donor-level factor folded into the mixing matrices (forced: the
        route = 2  # library folds by itself from 32 columns of us on); another seed keeps this background out of the cache
        kernel_form("kin_fold", 2)
    c = make_cohort(8, 30, 4, 700, seed=44 if folded else 43)
    rng = np.random.default_rng(9)
    n = c.y.size
    ys = [c.y, c.y[rng.permutation(n)], rng.normal(size=n), c.y + rng.normal(size=n), c.y[::-1].copy()]
    Ls = get_L_values(c.hK, c.E)
    first = CellRegMap(ys[0], c.E, W=c.W, Ls=Ls)
    assert (_lib.load().crm_background_kinship_folded(first._bg.handle) > 0) == folded
    crms = [first] + [CellRegMap(y, c.E, W=c.W, Ls=Ls, background=first._bg) for y in ys[1:]]
    G = c.G + 0.05 * rng.normal(size=c.G.shape)          # general genotypes
    panel = GenotypePanel(G, groups=None)
    lib, ctx = _lib.load(), _engine._context(0)
    _lib.check(lib.crm_test_set_kinship_route(ctx, route))
    try:
        for kw in ({}, {"idx_G": rng.permutation(n)}):
            monkeypatch.delenv("CRM_PAIR_BUFFER_GB", raising=False)
            pv, info = scan_interaction_many(crms, panel, **kw)
            assert np.mean([len(set(info["rho1"][:, j])) for j in range(700)]) > 1.5     # several rho* per variant
            monkeypatch.setenv("CRM_PAIR_BUFFER_GB", "1e-9")
            pv2, info2 = scan_interaction_many(crms, panel, **kw)
            assert np.array_equal(pv, pv2)
            for k in info:
                assert np.array_equal(info[k], info2[k])
    finally:
        _lib.check(lib.crm_test_set_kinship_route(ctx, 1))


def test_shared_h_route_of_the_multi_gene_scan():
    """With Q0(rho) = H Mix(rho) the n-length Khatri-Rao contraction can be done once per variant
    against H and finished per (variant, rho*) pair with Mix(rho*): same null fits (bit-identical
    info), score statistics equal to rounding."""
    from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib, get_L_values, scan_interaction_many
    from cellregmap_amd.synth import make_cohort

    c = make_cohort(8, 25, 4, 70, seed=41)
    rng = np.random.default_rng(7)
    Y = np.stack([c.y, c.y[rng.permutation(c.y.size)], rng.normal(size=c.y.size), c.y + rng.normal(size=c.y.size)], axis=1)
    Ls = get_L_values(c.hK, c.E)
    first = CellRegMap(Y[:, 0], c.E, W=c.W, Ls=Ls)      # thin branch (4 + 4*8 columns < 200 cells): keeps H
    crms = [first] + [CellRegMap(Y[:, i], c.E, W=c.W, Ls=Ls, background=first._bg) for i in range(1, 4)]
    panel = GenotypePanel(c.G, groups=None)
    lib = _lib.load()
    out = {}
    # (the routes under test are the two that contract over all cells; the donor-structured one is switched off here and
    # has its own test, test_kinship_structure_route_equals_the_direct_route)
    _lib.check(lib.crm_test_set_kinship_route(_engine._context(0), 0))
    try:
        for mode in (0, 1):
            _lib.check(lib.crm_test_set_shared_h(_engine._context(0), mode))
            for name, kw in (("plain", {}), ("idx_E", {"idx_E": rng.permutation(c.y.size)}),
                             ("idx_G", {"idx_G": np.random.default_rng(3).permutation(c.y.size)})):
                if name == "idx_E":
                    kw = {"idx_E": np.random.default_rng(5).permutation(c.y.size)}
                out[mode, name] = scan_interaction_many(crms, panel, **kw)
    finally:
        _lib.check(lib.crm_test_set_shared_h(_engine._context(0), -1))
        _lib.check(lib.crm_test_set_kinship_route(_engine._context(0), 1))
    for name in ("plain", "idx_E", "idx_G"):
        (pv0, info0), (pv1, info1) = out[0, name], out[1, name]
        for k in info0:
            assert np.array_equal(info0[k], info1[k])
        assert not np.array_equal(pv0, pv1)  # the other route really ran
        assert_allclose(pv1, pv0, rtol=1e-9)


def test_rotation_through_the_mixing_matrices_equals_direct_rotation():
    """T(rho) = Mix(rho)'(H'G) (default for device-built, well-conditioned backgrounds) against the
    direct G'Q0(rho): same results to ~1e-9 with the null-fit polish on."""
    from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib, get_L_values

    c = _cohort(8, 40, 5, 48, seed=51)
    crm = CellRegMap(c.y, c.E, W=c.W, Ls=get_L_values(c.hK, c.E))   # thin branch: 5 + 5*8 = 45 < 320
    panel = GenotypePanel(c.G, groups=None)
    lib = _lib.load()
    ctxh = _engine._context(0)
    _lib.check(lib.crm_set_null_fit_polish(ctxh, 1))
    try:
        fast = crm.scan_interaction(panel, return_stats=True)
        _lib.check(lib.crm_set_fast_rotation(ctxh, 0))
        direct = crm.scan_interaction(panel, return_stats=True)
    finally:
        _lib.check(lib.crm_set_fast_rotation(ctxh, 1))
        _lib.check(lib.crm_set_null_fit_polish(ctxh, 0))
    assert np.array_equal(fast[1]["rho1"], direct[1]["rho1"])
    assert_allclose(fast[2]["lml"], direct[2]["lml"], rtol=1e-12)
    assert_allclose(fast[2]["Q"], direct[2]["Q"], rtol=1e-8)
    assert np.all(np.abs(fast[0] - direct[0]) <= 1e-7 * direct[0] + P_ATOL)


def test_scan_on_lapack_decompositions_including_null_columns():
    """Background handed over as LAPACK's own economic decompositions (crm_background_create_qs): the
    reference's thin SVD keeps the null-space columns of the rank-deficient mode C half matrix
    (S0 ~ 1e-30) and arbitrary orthonormal vectors for them; they must be inert in the device path."""
    from cellregmap_amd import CellRegMap, GenotypePanel, get_L_values
    from cellregmap_amd._engine import background_from_qs
    from oracle.crm import OracleCellRegMap, khatri_rao_halves

    c = _cohort(6, 40, 4, 24, seed=4)
    ocrm = OracleCellRegMap(c.y, c.E, W=c.W, Ls=khatri_rao_halves(c.hK, c.E))
    S0 = ocrm._qs[0.5][1]
    assert S0.min() < 1e-20 * S0.max()          # rank deficient by construction (hard parts)
    bg = background_from_qs([ocrm._qs[r] for r in ocrm._rho], ocrm._rho)
    crm = CellRegMap(c.y, c.E, W=c.W, Ls=get_L_values(c.hK, c.E), background=bg)
    pv, info, st = crm.scan_interaction(GenotypePanel(c.G, groups=None), return_stats=True)
    opv, oinfo, ost = ocrm.scan_interaction(c.G, return_stats=True)
    _compare(pv, info, st, opv, oinfo, ost)
    # and the device's own decomposition (null columns dropped) gives the same answers
    pv2, info2 = CellRegMap(c.y, c.E, W=c.W, Ls=get_L_values(c.hK, c.E)).scan_interaction(GenotypePanel(c.G, groups=None))
    assert np.array_equal(info2["rho1"], info["rho1"])
    assert np.all(np.abs(pv2 - pv) <= P_RTOL * pv + P_ATOL)


def test_donor_tables_are_not_reused_across_panels():
    """One long-lived CellRegMap against a sequence of short-lived panels with different donor
    groupings (allocator addresses get recycled; the per-donor tables must follow the panel)."""
    import gc

    from cellregmap_amd import CellRegMap, GenotypePanel
    from cellregmap_amd.synth import make_cohort

    c = make_cohort(8, 20, 3, 32, seed=61)
    crm = CellRegMap(c.y, c.E, W=c.W, hK=c.hK)
    rng = np.random.default_rng(0)
    for trial in range(4):
        perm = rng.permutation(c.y.size)          # same genotypes, cells shuffled -> other grouping
        G = c.G[perm] if trial % 2 else c.G
        panel = GenotypePanel(G)
        assert panel.n_groups == 8
        pv, _ = crm.scan_interaction(panel)
        del panel
        gc.collect()
        ref, _ = CellRegMap(c.y, c.E, W=c.W, hK=c.hK).scan_interaction(GenotypePanel(G, groups=None))
        assert np.all(np.abs(pv - ref) <= P_RTOL * ref + P_ATOL)


def test_shared_donor_tables_follow_contexts_and_donor_structure():
    """The phenotype-free donor tables live in the background, keyed by the contents of E0 and of the
    donor index: further genes and fresh panels with the same donors reuse them, other contexts or
    another grouping must not."""
    from cellregmap_amd import CellRegMap, GenotypePanel
    from cellregmap_amd.synth import make_cohort

    c = make_cohort(8, 20, 3, 48, seed=71)
    rng = np.random.default_rng(5)
    first = CellRegMap(c.y, c.E, W=c.W, hK=c.hK)
    bg = first._bg

    def dense(y, E, G):
        return CellRegMap(y, E, W=c.W, E1=c.E, hK=c.hK, background=bg).scan_interaction(GenotypePanel(G, groups=None))[0]

    def close(a, b):
        return np.all(np.abs(a - b) <= P_RTOL * b + P_ATOL)

    # gene 1 builds the tables; gene 2 (other phenotype, other SNPs, same donors) reuses them
    G1, G2 = c.G[:, :24], c.G[:, 24:]
    assert close(first.scan_interaction(GenotypePanel(G1))[0], dense(c.y, c.E, G1))
    y2 = rng.permutation(c.y)
    second = CellRegMap(y2, c.E, W=c.W, hK=c.hK, background=bg)
    assert close(second.scan_interaction(GenotypePanel(G2))[0], dense(y2, c.E, G2))
    # other contexts E0 on the same background (E1 stays): new tables
    E_alt = c.E[rng.permutation(c.y.size)]
    third = CellRegMap(c.y, E_alt, W=c.W, E1=c.E, hK=c.hK, background=bg)
    assert close(third.scan_interaction(GenotypePanel(G1))[0], dense(c.y, E_alt, G1))
    # another donor structure (cells shuffled) with the first contexts again
    perm = rng.permutation(c.y.size)
    assert close(first.scan_interaction(GenotypePanel(G2[perm]))[0], dense(c.y, c.E, G2[perm]))
    # and back: a third distinct (E0, grouping) pair evicts the least recently used entry
    assert close(second.scan_interaction(GenotypePanel(G1))[0], dense(y2, c.E, G1))
    assert close(third.scan_interaction(GenotypePanel(G2))[0], dense(c.y, E_alt, G2))


@pytest.mark.parametrize("genotypes", ["dense", "donor-collapsed"])
def test_many_phenotypes_against_the_oracle(genotypes):
    """``scan_interaction_many`` / ``run_interaction_many`` against the ORACLE gene by gene (the per-gene
    calls of cellregmap/_cellregmap.py:547-587), not against the device's own single-gene scan."""
    from cellregmap_amd import CellRegMap, GenotypePanel, get_L_values, run_interaction_many, scan_interaction_many
    from cellregmap_amd.synth import make_cohort
    from oracle import crm as ocrm

    c = make_cohort(8, 25, 4, 40, seed=43)
    n = c.y.size
    rng = np.random.default_rng(11)
    Y = np.stack([c.y, c.y[rng.permutation(n)], rng.normal(size=n), c.y + rng.normal(size=n), 3.0 - 2.0 * c.y], axis=1)
    W = np.concatenate([c.W, rng.normal(size=(n, 1))], axis=1)
    G = c.G if genotypes == "donor-collapsed" else c.G + 0.05 * rng.normal(size=c.G.shape)  # general genotypes
    Ls = get_L_values(c.hK, c.E)
    first = CellRegMap(Y[:, 0], c.E, W=W, Ls=Ls)
    crms = [first] + [CellRegMap(Y[:, i], c.E, W=W, Ls=Ls, background=first._bg) for i in range(1, Y.shape[1])]
    panel = GenotypePanel(G)
    assert (panel.n_groups is not None) == (genotypes == "donor-collapsed")
    oLs = ocrm.khatri_rao_halves(c.hK, c.E)
    for hooks in ({}, {"idx_E": rng.permutation(n)}, {"idx_G": rng.permutation(n)}):
        pv, info = scan_interaction_many(crms, panel, **hooks)
        for i in range(Y.shape[1]):
            opv, oinfo = ocrm.OracleCellRegMap(Y[:, i], c.E, W=W, Ls=oLs).scan_interaction(G, **hooks)
            _assert_fit_info_matches({k: v[i] for k, v in info.items()}, oinfo, pv[i], opv)
    # the functional wrapper: one run_interaction per column of Y
    pv2, info2 = run_interaction_many(Y, c.E, G, W=W, hK=c.hK)
    for i in range(Y.shape[1]):
        opv, oinfo = ocrm.run_interaction(Y[:, i], c.E, G, W=W, hK=c.hK)
        _assert_fit_info_matches({k: v[i] for k, v in info2.items()}, oinfo, pv2[i], opv)


@pytest.mark.parametrize("genotypes", ["dense", "donor-collapsed"])
def test_cis_windows_of_many_phenotypes_against_the_oracle(genotypes):
    """An eQTL-shaped run: every phenotype is tested against its own cis window of one resident panel
    (``scan_interaction_many(Y, G, cis_index)``); each result must be what the reference's
    per-gene call on ``G[:, window]`` gives (cellregmap/_cellregmap.py:547-587) -- checked against the oracle."""
    from cellregmap_amd import CellRegMap, GenotypePanel, get_L_values, run_interaction_many, scan_interaction_many
    from cellregmap_amd.synth import make_cohort
    from oracle import crm as ocrm

    c = make_cohort(8, 25, 4, 60, seed=47)
    n, p = c.G.shape
    rng = np.random.default_rng(5)
    Y = np.stack([c.y, c.y[rng.permutation(n)], rng.normal(size=n), c.y + rng.normal(size=n), 1.0 - c.y,
                  rng.normal(size=n)], axis=1)
    G = c.G if genotypes == "donor-collapsed" else c.G + 0.05 * rng.normal(size=c.G.shape)
    mask = np.zeros(p, bool)
    mask[[0, 1, 2, 30, 31, 59]] = True
    cis = [(0, 25), slice(10, 45), mask, np.array([50, 12, 12, -1, 3]), np.array([], dtype=int), (40, 60)]
    Ls = get_L_values(c.hK, c.E)
    first = CellRegMap(Y[:, 0], c.E, W=c.W, Ls=Ls)
    crms = [first] + [CellRegMap(Y[:, i], c.E, W=c.W, Ls=Ls, background=first._bg) for i in range(1, Y.shape[1])]
    panel = GenotypePanel(G)
    oLs = ocrm.khatri_rao_halves(c.hK, c.E)
    idx_G = rng.permutation(n)
    for hooks in ({}, {"idx_G": idx_G}):
        pv, info = scan_interaction_many(crms, panel, cis_index=cis, **hooks)
        assert len(pv) == len(cis) and pv[4].size == 0
        for i, sel in enumerate(cis):
            cols = np.arange(p)[sel] if isinstance(sel, slice) else (np.arange(*sel) if isinstance(sel, tuple) else sel)
            Gi = G[:, cols]
            assert pv[i].shape == (Gi.shape[1],)
            if Gi.shape[1] == 0:
                continue
            opv, oinfo = ocrm.OracleCellRegMap(Y[:, i], c.E, W=c.W, Ls=oLs).scan_interaction(Gi, **hooks)
            _assert_fit_info_matches({k: v[i] for k, v in info.items()}, oinfo, pv[i], opv)
            # and what the same object gives when it scans its window alone (other launch shapes: same tolerance)
            dpv, _ = scan_interaction_many([crms[i]], panel, cis_index=[sel], **hooks)
            assert_allclose(dpv[0], pv[i], rtol=P_RTOL, atol=P_ATOL)
    pv2, _ = run_interaction_many(Y, c.E, G, W=c.W, hK=c.hK, cis_index=cis)
    pv_plain, _ = scan_interaction_many(crms, panel, cis_index=cis)
    for i in range(len(cis)):
        assert np.array_equal(pv2[i], pv_plain[i])


def test_many_phenotypes_must_share_covariates_and_contexts_by_content():
    """Same shapes are not enough: the shared pass takes g'W and the context features from the first
    object (advisor finding, round 1)."""
    import ctypes

    from cellregmap_amd import CellRegMap, GenotypePanel, _lib, scan_interaction_many
    from cellregmap_amd.synth import make_cohort

    c = make_cohort(6, 20, 3, 12, seed=47)
    rng = np.random.default_rng(2)
    W2 = np.concatenate([c.W, rng.normal(size=(c.y.size, 1))], axis=1)
    W3 = np.concatenate([c.W, rng.normal(size=(c.y.size, 1))], axis=1)
    a = CellRegMap(c.y, c.E, W=W2, hK=c.hK)
    b = CellRegMap(c.y[::-1].copy(), c.E, W=W3, hK=c.hK, background=a._bg)
    with pytest.raises(ValueError, match="same covariates"):
        scan_interaction_many([a, b], c.G)
    e = CellRegMap(c.y, c.E[::-1].copy(), W=W2, E1=c.E, hK=c.hK, background=a._bg)
    with pytest.raises(ValueError, match="same contexts"):
        scan_interaction_many([a, e], c.G)
    # and the C-ABI refuses it too
    lib = _lib.load()
    panel = GenotypePanel(c.G)
    handles = (ctypes.c_void_p * 2)(a._bind_gene().value, b._bind_gene().value)
    out = np.empty((2, 12))
    rc = lib.crm_scan_interaction_multi(handles, 2, panel.handle, 0, 12, None, None, _lib.ptr(out), None, None, None,
                                        None, None)
    assert rc == -2 and b"contents" in lib.crm_last_error()


def test_boolean_permutation_index_selects_rows_like_numpy():
    """A boolean idx_E / idx_G is a mask (numpy's ``E0[idx, :]``): with all entries True it is the identity,
    in the single-gene and in the multi-gene entry point alike."""
    from cellregmap_amd import CellRegMap, scan_interaction_many
    from cellregmap_amd.synth import make_cohort

    c = make_cohort(6, 20, 3, 10, seed=48)
    crm = CellRegMap(c.y, c.E, W=c.W, hK=c.hK)
    ref, _ = crm.scan_interaction(c.G)
    mask = np.ones(c.y.size, bool)
    assert np.array_equal(crm.scan_interaction(c.G, idx_E=mask)[0], ref)
    # (the genotype hook runs the dense / mixed-table path instead of the collapsed one: equal to rounding)
    assert_allclose(scan_interaction_many([crm], c.G, idx_G=mask)[0][0], ref, rtol=1e-6)
    with pytest.raises(ValueError):
        crm.scan_interaction(c.G, idx_E=mask[:-1])


def _decaying_cohort(n, k, m, p, seed, decades=6.0):
    """Ill-conditioned background: kinship factor hK = U diag(sigma) with sigma over `decades` decades (so the
    spectrum S0 of K spans twice that), contexts strongly correlated with each other (condition number
    ~1e4) -- the shape real kinship factors and principal-component contexts have, unlike the N(0,1)
    contexts and donor indicators of the other cohorts."""
    rng = np.random.default_rng(seed)
    U, _ = np.linalg.qr(rng.normal(size=(n, m)))
    hK = U * np.logspace(0.0, -decades, m)
    Z = rng.normal(size=(n, k))
    mix = np.linalg.qr(rng.normal(size=(k, k)))[0] * np.logspace(0.0, -2.0, k)  # column scales 1 .. 1e-2
    E = Z @ mix @ rng.normal(size=(k, k))
    E = (E - E.mean(0)) / E.std(0)
    G = rng.normal(size=(n, p))
    W = np.ones((n, 1))
    y = 0.5 * G[:, 0] * E[:, 0] + E @ rng.normal(size=k) * 0.3 + hK @ rng.normal(size=m) * 3.0 + rng.normal(size=n)
    return y, E, W, G, hK


@pytest.mark.parametrize("mode", ["B-thin", "C-thin", "C-eigh"])
def test_decaying_spectrum_background(mode):
    """Parity on an ill-conditioned background (the Gram route of the thin branch squares the condition
    number; the kept spectrum is far beyond S_max <= 1e6 S_min, so the scan must take the direct
    rotations G'Q0(rho) instead of the mixing matrices)."""
    from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib, get_L_values
    from oracle.crm import OracleCellRegMap, khatri_rao_halves

    if mode == "B-thin":
        y, E, W, G, hK = _decaying_cohort(400, 5, 24, 20, seed=1)
        kw, okw = {"hK": hK}, {"hK": hK}
    elif mode == "C-thin":      # cols = 3 + 3 * 12 = 39 < n
        y, E, W, G, hK = _decaying_cohort(300, 3, 12, 20, seed=2)
        kw, okw = {"Ls": get_L_values(hK, E)}, {"Ls": khatri_rao_halves(hK, E)}
    else:                       # cols = 6 + 6 * 40 = 246 >= n = 200: the reference's eigh branch
        y, E, W, G, hK = _decaying_cohort(200, 6, 40, 20, seed=3, decades=4.0)
        kw, okw = {"Ls": get_L_values(hK, E)}, {"Ls": khatri_rao_halves(hK, E)}
    crm = CellRegMap(y, E, W=W, **kw)
    o = OracleCellRegMap(y, E, W=W, **okw)
    S0 = o._qs[0.5][1]
    kept = S0[S0 > 1e-12 * S0.max()]
    assert kept.max() > 1e7 * kept.min()          # really ill-conditioned
    pv, info, st = crm.scan_interaction(GenotypePanel(G, groups=None), return_stats=True)
    opv, oinfo, ost = o.scan_interaction(G, return_stats=True)
    _compare(pv, info, st, opv, oinfo, ost)
    # and with the polished null fit on both sides: the algebra itself, to 1e-8
    lib = _lib.load()
    _lib.check(lib.crm_set_null_fit_polish(_engine._context(0), 1))
    try:
        pv, info, st = crm.scan_interaction(GenotypePanel(G, groups=None), return_stats=True)
    finally:
        _lib.check(lib.crm_set_null_fit_polish(_engine._context(0), 0))
    ppv, pinfo, pst = OracleCellRegMap(y, E, W=W, polish=True, **okw).scan_interaction(G, return_stats=True)
    _compare(pv, info, st, ppv, pinfo, pst, tight=True)


def _loaded_hip_runtime():
    """The HIP runtime libcrm_hip.so itself is linked against, opened by the path it is mapped from: a bare
    dlopen("libamdhip64.so") can resolve to a second copy of the runtime (torch bundles one), and device pointers of one
    runtime mean nothing to the other."""
    import ctypes

    from cellregmap_amd import _lib

    _lib.load()
    with open("/proc/self/maps") as fh:
        paths = {ln.split()[-1] for ln in fh if "libamdhip64.so" in ln}
    assert len(paths) == 1, paths          # exactly one HIP runtime in this process
    return ctypes.CDLL(paths.pop())


@pytest.mark.parametrize("mode", ["C-thin", "C-eigh", "B"])
def test_constructor_in_phases_with_exchanged_grid_points(mode):
    """The multi-GPU constructor on one GPU: two builders own the even / odd grid points, exchange their
    slots through device tensors (what RCCL broadcasts between ranks) and must both end up with the
    background a single constructor builds (to rounding: the batch size picks the contraction kernel's tile
    width) and with the same scan results."""
    import ctypes

    from cellregmap_amd import CellRegMap, GenotypePanel, get_L_values
    from cellregmap_amd._engine import _RHO_GRID, BackgroundBuilder

    # device buffers from the HIP runtime the library itself runs on (a torch tensor's data_ptr() is the same
    # kind of pointer; torch is kept out of this process because it has to initialise the GPU BEFORE the
    # library does, and other tests of the session have used the library already)
    hip = _loaded_hip_runtime()

    class DeviceBuffer:
        def __init__(self, doubles):
            self.ptr = ctypes.c_void_p()
            assert hip.hipMalloc(ctypes.byref(self.ptr), ctypes.c_size_t(8 * doubles)) == 0

        def data_ptr(self):
            return self.ptr.value

        def free(self):
            hip.hipFree(self.ptr)

    if mode == "C-eigh":
        c = _cohort(12, 10, 10, 12, seed=3)
    elif mode == "C-thin":
        c = _cohort(6, 40, 4, 12, seed=4)
    else:
        c = _cohort(10, 20, 5, 12, seed=5)
    B = get_L_values(c.hK, c.E) if mode.startswith("C") else c.hK
    kw = {"Ls": B} if mode.startswith("C") else {"hK": c.hK}
    even = np.arange(11) % 2 == 0
    a = BackgroundBuilder(c.E, B, _RHO_GRID, mine=even)
    b = BackgroundBuilder(c.E, B, _RHO_GRID, mine=~even)
    assert [a.rank(i) >= 0 for i in range(11)] == list(even)
    ranks = [a.rank(i) if even[i] else b.rank(i) for i in range(11)]
    a.complete(ranks)
    b.complete(ranks)
    layout = a.layout()
    assert layout == b.layout() and ("Mix" in layout) == (mode != "C-eigh")
    for what, size in layout.items():
        buf = DeviceBuffer(size)
        for i in range(11):
            src, dst = (a, b) if even[i] else (b, a)
            src.export_slot(i, what, buf)
            dst.import_slot(i, what, buf)
        buf.free()
    bga, bgb = a.seal(), b.seal()
    ref = CellRegMap(c.y, c.E, W=c.W, **kw)
    n = c.y.size
    probe = np.random.default_rng(0).normal(size=(n, 3))
    for i in range(11):
        Q0, S0 = ref._bg.read(i, n)
        want = Q0 @ (S0[:, None] * (Q0.T @ probe))
        for bg in (bga, bgb):
            q, s = bg.read(i, n)
            assert_allclose(s, S0, rtol=1e-10, atol=1e-12 * S0.max())
            assert np.abs(q.T @ q - np.eye(q.shape[1])).max() < 1e-12
            assert np.abs(q @ (s[:, None] * (q.T @ probe)) - want).max() <= 1e-11 * np.abs(want).max()
            # both builders hold the SAME bytes for every grid point: what one computed, the other imported
        qa, sa = bga.read(i, n)
        qb, sb = bgb.read(i, n)
        assert np.array_equal(qa, qb) and np.array_equal(sa, sb)
    panel = GenotypePanel(c.G, groups=None)
    pv, info = ref.scan_interaction(panel)
    for bg in (bga, bgb):
        pv2, info2 = CellRegMap(c.y, c.E, W=c.W, background=bg, **kw).scan_interaction(panel)
        assert np.array_equal(info2["rho1"], info["rho1"])
        assert np.all(np.abs(pv2 - pv) <= P_RTOL * pv + P_ATOL)
    # an unsealed background is refused
    half = BackgroundBuilder(c.E, B, _RHO_GRID, mine=even)
    with pytest.raises(Exception, match="under construction"):
        CellRegMap(c.y, c.E, W=c.W, background=half._bg, **kw).scan_interaction(panel)


@pytest.mark.parametrize("hook,shape", [("none", "small"), ("E", "small"), ("G", "small"), ("none", "wide"),
                                        ("none", "full"), ("G", "full")])
def test_kinship_structure_route_equals_the_direct_route(hook, shape, monkeypatch, kernel_form):
    """Mode C through get_L_values with an "expanded" kinship factor (rows of a donor-level factor repeated for the cells
    of each donor, ragged donors, a DENSE donor-level factor): the dense scan of general genotypes forms
    Q0(rho*)'(g o E0) = Mix(rho*)' [H'(g o E0)] with H'(g o E0) built donor by donor (crm_background_set_kinship_groups)
    instead of contracting against Q0(rho*) over all cells.  Same statistics as the direct route (test hook) to rounding,
    and the oracle's; one phenotype and several."""
    from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib, get_L_values, scan_interaction_many
    from oracle.crm import OracleCellRegMap, khatri_rao_halves

    rng = np.random.default_rng(77)
    # "wide": 70 contexts, so that [us | E1] takes 140 columns (two column tiles of the per-donor launch)
    # "full": a dense donor-level factor of full rank (as many columns as donors) -- the library then folds it into the
    # mixing matrices (crm_background::kin_fold) and the per-donor sums are the operand of the Mix product as they stand;
    # "small" / "wide" keep the contraction over the donors per block (rank-deficient factor: the folded operand would be longer)
    donors, k0, p, mcols, lo, hi = {"small": (9, 5, 40, 6, 7, 40), "wide": (12, 70, 8, 3, 40, 60), "full": (9, 5, 40, 9, 7, 40)}[shape]
    sizes = rng.integers(lo, hi, size=donors)
    donor = np.repeat(np.arange(donors), sizes)
    n = donor.size
    hKd = rng.normal(size=(donors, mcols))                 # donor-level factor of rank < donors, dense
    hK = hKd[donor]
    E = rng.normal(size=(n, k0))
    W = np.concatenate([np.ones((n, 1)), rng.normal(size=(n, 1))], axis=1)
    G = rng.normal(size=(n, p))                            # general genotypes: the dense path
    y = 0.5 * G[:, 3] * E[:, 0] + E @ rng.normal(size=k0) * 0.3 + hK @ rng.normal(size=mcols) * 0.2 + rng.normal(size=n)
    idx = rng.permutation(n)
    hooks = {} if hook == "none" else ({"idx_E": idx} if hook == "E" else {"idx_G": idx})
    if shape == "full":   # (the library folds by itself from 32 columns of us on: few columns do better unfolded)
        kernel_form("kin_fold", 2)
    crm = CellRegMap(y, E, W=W, Ls=get_L_values(hK, E))
    lib, ctx = _lib.load(), _engine._context(0)
    panel = GenotypePanel(G, groups=None)
    assert lib.crm_background_kinship_groups(crm._bg.handle) == donors      # the structure was found and is in use
    assert (lib.crm_background_kinship_folded(crm._bg.handle) > 0) == (shape == "full")
    # the two routes with the null-fit optimum pinned (polish), so that the comparison is not about where Brent stops
    try:
        _lib.check(lib.crm_set_null_fit_polish(ctx, 1))
        _lib.check(lib.crm_test_set_kinship_route(ctx, 2))      # (always: at these sizes the route's flop count does not pay)
        try:
            ppv, pinfo, pst = crm.scan_interaction(panel, return_stats=True, progress=False, **hooks)
            _lib.check(lib.crm_test_set_kinship_route(ctx, 0))
            pv0, info0, st0 = crm.scan_interaction(panel, return_stats=True, progress=False, **hooks)
        finally:
            _lib.check(lib.crm_test_set_kinship_route(ctx, 2))
            _lib.check(lib.crm_set_null_fit_polish(ctx, 0))
        assert np.array_equal(pinfo["rho1"], info0["rho1"])
        assert_allclose(pst["delta"], st0["delta"], rtol=1e-8)
        scale = np.maximum(np.abs(st0["Q"]), np.trace(st0["F"], axis1=1, axis2=2))
        assert np.all(np.abs(pst["Q"] - st0["Q"]) <= 1e-9 * scale)
        assert np.all(np.abs(pst["F"] - st0["F"]) <= 1e-9 * np.abs(st0["F"]).max(axis=(1, 2), keepdims=True))
        assert np.all(np.abs(ppv - pv0) <= 2e-6 * pv0 + 1e-13)
        # ... and the reference's procedure verbatim against the oracle
        pv, info, st = crm.scan_interaction(panel, return_stats=True, progress=False, **hooks)
        opv, oinfo, ost = OracleCellRegMap(y, E, W=W, Ls=khatri_rao_halves(hK, E)).scan_interaction(G, return_stats=True, **hooks)
        _compare(pv, info, st, opv, oinfo, ost)
        # several phenotypes in one pass take the same route for H'(g o E0)
        ys = [y, y[rng.permutation(n)], rng.normal(size=n)]
        crms = [crm] + [CellRegMap(v, E, W=W, Ls=get_L_values(hK, E), background=crm._bg) for v in ys[1:]]
        mpv, minfo = scan_interaction_many(crms, panel, **hooks)
        for i, obj in enumerate(crms):
            spv, sinfo = obj.scan_interaction(panel, progress=False, **hooks)
            assert np.array_equal(minfo["rho1"][i], sinfo["rho1"])
            assert np.all(np.abs(mpv[i] - spv) <= 1e-7 * spv + 1e-13)
    finally:
        _lib.check(lib.crm_test_set_kinship_route(ctx, 1))
        _lib.check(lib.crm_set_null_fit_polish(ctx, 0))


@pytest.mark.parametrize("genotypes", ["general", "donor-constant"])
def test_streamed_scan_of_a_host_matrix_equals_the_one_panel_scan(genotypes, monkeypatch):
    """A host matrix of many variants is uploaded in column chunks from a second thread while the chunks that have arrived
    are scanned (``CellRegMap._scan_streamed``).  With the chunks whole blocks of the scan the results are those of the
    one-panel scan bit for bit -- p-values, info, statistics, with a permutation hook, for general and for donor-constant
    genotypes (every chunk finds the donor structure by itself) --, one progress report runs over all chunks, a
    non-finite entry in a late chunk raises the reference's ValueError, and a block of columns of a row-major matrix
    makes the same panel as its contiguous copy."""
    from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib
    from cellregmap_amd.synth import make_cohort

    c = make_cohort(9, 30, 4, 700, seed=52)
    rng = np.random.default_rng(2)
    G = c.G if genotypes == "donor-constant" else c.G + 0.05 * rng.normal(size=c.G.shape)
    crm = CellRegMap(c.y, c.E, W=c.W, hK=c.hK)
    lib, ctx = _lib.load(), _engine._context(0)
    _lib.check(lib.crm_set_block_variants(ctx, 128))
    try:
        for kw in ({}, {"idx_G": rng.permutation(c.y.size)}):
            monkeypatch.setenv("CELLREGMAP_AMD_STREAM_CHUNK", "0")
            pv, info, st = crm.scan_interaction(G, return_stats=True, progress=False, **kw)
            monkeypatch.setenv("CELLREGMAP_AMD_STREAM_CHUNK", "256")       # 700 variants: chunks of 128, 256, 256, 60
            seen = []
            spv, sinfo, sst = crm.scan_interaction(G, return_stats=True, progress=lambda done, total: seen.append((done, total)), **kw)
            assert np.array_equal(pv, spv)
            for key in info:
                assert np.array_equal(info[key], sinfo[key]), key
            for key in st:
                assert np.array_equal(st[key], sst[key]), key
            assert seen[-1] == (700, 700) and all(t == 700 for _, t in seen) and [d for d, _ in seen] == sorted(d for d, _ in seen)
        bad = G.copy()
        bad[17, 650] = np.nan
        with pytest.raises(ValueError, match="non-finite"):
            crm.scan_interaction(bad, progress=False)
        assert np.array_equal(crm.scan_interaction(G, progress=False, **kw)[0], spv)   # the context is sound afterwards
        dense = crm.scan_interaction(G, progress=False, groups=None)[0]                # every chunk kept dense
        assert np.array_equal(dense, crm.scan_interaction(GenotypePanel(G, groups=None), progress=False)[0])
    finally:
        _lib.check(lib.crm_set_block_variants(ctx, 0))
    view = G[:, 100:356]                                                 # rows 700 doubles apart: no host copy is made
    assert not view.flags.c_contiguous
    a = crm.scan_interaction(GenotypePanel(view, groups=None), progress=False)[0]
    b = crm.scan_interaction(GenotypePanel(np.ascontiguousarray(view), groups=None), progress=False)[0]
    assert np.array_equal(a, b)
