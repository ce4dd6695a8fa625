"""Parity at BASELINE sizes: size-independent properties plus oracle checks on seeded random samples of the
variants -- config 2 against the oracle's OWN LAPACK decompositions (so that one BASELINE size does not share the
device's (Q0, S0)), configs 3 / 4-shape / 5 against the oracle bound to the decompositions the device built (the
oracle's SVDs alone would take ~14 min at config 3)."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

pytestmark = pytest.mark.gpu

P_RTOL, P_ATOL = 1e-5, 1e-13
# delta itself is only defined to the search's tolerance: tol = 1e-6 (|x| + 1) on x = logit(delta), d(delta) / delta = (1 - delta) dx;
# two tolerances at |x| <= 9
DELTA_RTOL = 2 * 1e-6 * (9 + 1)


def _oracle_on_device_decomposition(crm, y, E, W, Ls, only=None):
    """The CPU oracle bound to the decompositions the DEVICE built (the oracle's own LAPACK SVDs at these
    sizes would dominate the run: ~14 min at config 3).  ``only``: grid indices to read back (the others
    stay absent: use ``null_fit`` on a restricted grid then)."""
    from oracle.crm import OracleCellRegMap

    n = y.size
    qs = {}
    for i, rho in enumerate(crm._rho1):
        if only is None or i in only:
            Q0, S0 = crm._bg.read(i, n)
            qs[rho] = ((Q0,), S0)
    o = OracleCellRegMap.__new__(OracleCellRegMap)
    o._polish = False
    o._y, o._E0, o._W, o._E1 = y, E, W, E
    o._Ls, o._half, o._qs = Ls, {}, qs
    o._rho = [r for r in crm._rho1 if r in qs]
    return o


def _oracle_scan_allowing_ties(o, G, rho_device):
    """``o.scan_interaction(G)``; where the oracle's rho* differs from the device's, the two grid points must tie in the
    oracle's own likelihood (phenotypes without a random effect are flat over the grid: the first strictly larger value
    is then decided by the last bits of two different roundings -- tests/test_gpu_fuzz.py applies the same rule), and the
    variant is scanned again with the grid restricted to the device's choice so that everything else is still compared."""
    from oracle.lmm import LMM

    opv, oinfo = o.scan_interaction(G)
    for i in np.flatnonzero(np.abs(oinfo["rho1"] - rho_device) > 1e-12):
        X = np.concatenate((o._W, G[:, [i]]), axis=1)
        lml = {}
        for rho in (float(oinfo["rho1"][i]), min(o._rho, key=lambda r: abs(r - rho_device[i]))):
            lmm = LMM(o._y, X, o._qs[rho], restricted=True)
            lmm.fit(verbose=False, polish=o._polish)
            lml[rho] = lmm.lml()
        a, b = lml.values()
        assert abs(a - b) <= 1e-11 * abs(a), ("rho* differs without a tie", i, lml)
        grid, o._rho = o._rho, [r for r in lml if abs(r - rho_device[i]) < 1e-12]
        try:
            pv1, info1 = o.scan_interaction(G[:, [i]])
        finally:
            o._rho = grid
        opv[i] = pv1[0]
        for k in oinfo:
            oinfo[k][i] = info1[k][0]
    return opv, oinfo


def _compare_with_oracle(pv, info, st, pick, opv, oinfo, ost, k0, q_rtol=1e-6, delta_rtol=DELTA_RTOL, p_rtol=P_RTOL, bounds=None):
    """rho*, delta, lml, Q, F, the eigenvalues of F and p on the picked variants (north-star tolerances: statistics
    1e-6, p-values 1e-5).  ``bounds = (bound_Q, bound_p)`` of the device's ``scan_interaction_info`` at the picked variants:
    a variant whose own bound is wider than a tolerance is held to the bound (tests/parity_bounds.py)."""
    n_pick = len(pick)
    bq = np.zeros(n_pick) if bounds is None else np.asarray(bounds[0])
    bp = np.zeros(n_pick) if bounds is None else np.asarray(bounds[1])
    q_allow = np.maximum(q_rtol, 1.001 * bq)
    p_allow = np.maximum(p_rtol, 1.001 * bp + (2e-6 if bounds is not None else 0.0))
    assert_allclose(info["rho1"][pick], oinfo["rho1"], atol=1e-12)
    assert_allclose(st["lml"][pick], ost["lml"], rtol=1e-10)
    assert_allclose(st["delta"][pick], ost["delta"], rtol=delta_rtol)
    trF = np.array([np.trace(F) for F in ost["F"]])
    assert np.all(np.abs(st["Q"][pick] - ost["Q"]) <= q_allow * np.maximum(np.abs(ost["Q"]), trF)), np.c_[st["Q"][pick], ost["Q"], q_allow]
    total = oinfo["e2"] + oinfo["g2"] + oinfo["eps2"]
    for k in ("e2", "g2", "eps2"):
        assert np.all(np.abs(info[k][pick] - oinfo[k]) <= np.maximum(1e-5, q_allow) * oinfo[k] + 1e-6 * total), k
    for row, j in enumerate(pick):
        F = ost["F"][row]
        assert np.abs(st["F"][j] - F).max() <= q_allow[row] * np.abs(F).max(), j
        lam = np.linalg.eigvalsh(F)
        assert np.abs(st["lambda"][j] - lam).max() <= q_allow[row] * np.abs(lam).max(), j
    assert np.all(np.abs(pv[pick] - opv) <= p_allow * opv + P_ATOL), np.c_[pv[pick], opv, p_allow]


ROUTES = ["kinship", "direct"]


class _route:
    """The two product routes of the dense scan: the kinship-structure route (default whenever the kinship factor is
    donor-expanded: H'(g o E0) donor by donor, then Mix(rho*)') and the direct Khatri-Rao contraction against Q0(rho*) --
    what a cell-level kinship factor gets.  ``with _route("direct")`` switches the first one off on the test context."""

    def __init__(self, name):
        self.on = {"kinship": 1, "direct": 0}[name]

    def __enter__(self):
        from cellregmap_amd import _engine, _lib

        self.lib, self.ctx = _lib.load(), _engine._context(0)
        _lib.check(self.lib.crm_test_set_kinship_route(self.ctx, self.on))

    def __exit__(self, *exc):
        from cellregmap_amd import _lib

        _lib.check(self.lib.crm_test_set_kinship_route(self.ctx, 1))


@pytest.fixture(scope="module", params=ROUTES)
def cfg2(request):
    """BASELINE config 2, every test below once per route (the hook stays set while the parameter's tests run)."""
    from cellregmap_amd import CellRegMap, GenotypePanel, get_L_values
    from cellregmap_amd.synth import make_config

    c = make_config("cfg2", n_variants=384)
    Ls = get_L_values(c.hK, c.E)
    crm = CellRegMap(c.y, c.E, W=c.W, Ls=Ls)
    dense = GenotypePanel(c.G, groups=None)
    with _route(request.param):
        pv, info, st = crm.scan_interaction(dense, return_stats=True)
        yield c, Ls, crm, dense, pv, info, st


def test_oracle_spot_check_at_config2(cfg2):
    """24 seeded random variants (+ the planted ones) of the config-2 block against the oracle on the device's
    decompositions: rho*, delta, lml, Q, F, eigenvalues, p."""
    c, Ls, crm, dense, pv, info, st = cfg2
    o = _oracle_on_device_decomposition(crm, c.y, c.E, c.W, Ls)
    pick = sorted(set(np.random.default_rng(22).choice(384, size=24, replace=False).tolist()) | {10, 11})
    opv, oinfo, ost = o.scan_interaction(c.G[:, pick], return_stats=True)
    _compare_with_oracle(pv, info, st, pick, opv, oinfo, ost, c.E.shape[1])


def test_config2_against_the_oracles_own_decompositions(cfg2):
    """BASELINE config 2 with NOTHING shared: the oracle decomposes the eleven 5 000 x 1 020 half covariances itself
    (numpy / LAPACK: economic_qs_linear, _math.py:238-256) and scans 16 seeded random variants (+ the planted ones);
    the device's constructor (Gram -> tridiagonalisation -> divide & conquer -> mixing matrices) and scan must land on
    the same rho*, lml, Q, F spectrum and p.  Two decompositions of the same Sigma(rho) differ by a rotation inside
    eigenspaces, so F's entries and eigenvalues are compared, not Q0."""
    from cellregmap_amd import _engine, _lib
    from oracle.crm import OracleCellRegMap, khatri_rao_halves

    c, Ls, crm, dense, pv, info, st = cfg2
    o = OracleCellRegMap(c.y, c.E, W=c.W, Ls=khatri_rao_halves(c.hK, c.E))
    pick = sorted(set(np.random.default_rng(2).choice(384, size=16, replace=False).tolist()) | {10, 11})
    # the reference's procedure verbatim on both sides.  The two sides' likelihoods differ by the rounding of two
    # different eigenbases (~1e-13 relative), and Brent's 1e-6 search on logit(delta) can stop up to one tolerance apart
    # on them: every variant at the north-star tolerances or, where the device says so itself, at its own bounds
    import parity_bounds

    opv, oinfo, ost = o.scan_interaction(c.G[:, pick], return_stats=True)
    bq, bp, _ = parity_bounds.bounds(crm, dense, pick)
    parity_bounds.assert_bounds_are_informative(bq, bp, pv[pick])
    _compare_with_oracle(pv, info, st, pick, opv, oinfo, ost, c.E.shape[1], bounds=(bq, bp))
    # and with the optimum pinned on both sides (polish): the algebra itself, 1e-8 (p: Davies integrates to 1e-6)
    lib, ctx = _lib.load(), _engine._context(0)
    _lib.check(lib.crm_set_null_fit_polish(ctx, 1))
    try:
        ppv, pinfo, pst = crm.scan_interaction(dense, return_stats=True)
    finally:
        _lib.check(lib.crm_set_null_fit_polish(ctx, 0))
    o._polish = True
    opv, oinfo, ost = o.scan_interaction(c.G[:, pick], return_stats=True)
    _compare_with_oracle(ppv, pinfo, pst, pick, opv, oinfo, ost, c.E.shape[1], q_rtol=1e-8, delta_rtol=1e-7, p_rtol=2e-6)


@pytest.mark.parametrize("fast", [False, True])
def test_association_at_config2_against_the_oracle(cfg2, fast):
    """Row a13 at a BASELINE size: the persistent-effect LRT (_cellregmap.py:246-314, 443-469) on the config-2 cohort, full
    ML refit per SNP and FastScanner, 12 seeded random SNPs (+ the planted persistent ones) against the oracle on the
    device's decompositions: the null model's rho / variance components and every p-value."""
    c, Ls, crm, dense, *_ = cfg2
    o = _oracle_on_device_decomposition(crm, c.y, c.E, c.W, Ls)
    pick = sorted(set(np.random.default_rng(13).choice(384, size=12, replace=False).tolist()) | {5, 6})
    G = np.ascontiguousarray(c.G[:, pick])
    if fast:
        pv, info = crm.scan_association_fast(G)
        opv, oinfo = o.scan_association_fast(G)
    else:
        pv, info = crm.scan_association(G, progress=False)
        opv, oinfo = o.scan_association(G)
    for k in ("rho1", "e2", "g2", "eps2"):
        assert info[k].shape == (1,)
        assert_allclose(info[k], oinfo[k], rtol=1e-5, atol=1e-10)
    assert np.all(np.abs(pv - opv) <= P_RTOL * opv + 1e-300), np.c_[pv, opv]
    assert pv[[pick.index(5), pick.index(6)]].max() < 1e-6          # the planted persistent effects are found


def test_decomposition_is_an_orthonormal_factorisation(cfg2):
    c, Ls, crm, dense, *_ = cfg2
    i = 4
    rho = crm._rho1[i]
    Q0, S0 = crm._bg.read(i, c.y.size)
    assert np.abs(Q0.T @ Q0 - np.eye(Q0.shape[1])).max() < 1e-12
    hS = np.concatenate([np.sqrt(rho) * c.E] + [np.sqrt(1 - rho) * L for L in Ls], axis=1)
    v = np.random.default_rng(0).normal(size=(c.y.size, 3))
    lhs = hS @ (hS.T @ v)
    rhs = Q0 @ (S0[:, None] * (Q0.T @ v))
    assert np.abs(lhs - rhs).max() <= 1e-9 * np.abs(lhs).max()


def test_affine_invariance_of_the_phenotype(cfg2):
    """p-values of the score test do not change under y -> a*y + b (W holds the intercept)."""
    from cellregmap_amd import CellRegMap

    c, Ls, crm, dense, pv, info, st = cfg2
    crm2 = CellRegMap(2.5 * c.y + 3.0, c.E, W=c.W, Ls=Ls)
    pv2, info2 = crm2.scan_interaction(dense)
    assert np.array_equal(info2["rho1"], info["rho1"])
    assert np.all(np.abs(pv2 - pv) <= P_RTOL * pv + P_ATOL)
    assert_allclose(info2["e2"], 2.5 ** 2 * info["e2"], rtol=1e-5, atol=1e-12)


def test_cell_order_invariance(cfg2):
    """Permuting the cells consistently in every input leaves all outputs unchanged."""
    from cellregmap_amd import CellRegMap, GenotypePanel

    c, Ls, crm, dense, pv, info, st = cfg2
    perm = np.random.default_rng(5).permutation(c.y.size)
    crm2 = CellRegMap(c.y[perm], c.E[perm], W=c.W[perm], Ls=[L[perm] for L in Ls])
    pv2, info2, st2 = crm2.scan_interaction(GenotypePanel(c.G[perm], groups=None), return_stats=True)
    assert np.array_equal(info2["rho1"], info["rho1"])
    # two summation orders under the reference's Brent(1e-6) search: agreement within its tolerance
    assert_allclose(st2["Q"], st["Q"], rtol=5e-6)
    assert np.all(np.abs(pv2 - pv) <= P_RTOL * pv + P_ATOL)


def test_collapsed_equals_dense_at_config2(cfg2):
    from cellregmap_amd import GenotypePanel

    c, Ls, crm, dense, pv, info, st = cfg2
    panel = GenotypePanel(c.G)
    assert panel.n_groups == 50
    pv2, info2, st2 = crm.scan_interaction(panel, return_stats=True)
    assert np.array_equal(info2["rho1"], info["rho1"])
    assert_allclose(st2["Q"], st["Q"], rtol=5e-6)
    assert np.all(np.abs(pv2 - pv) <= P_RTOL * pv + P_ATOL)


def test_planted_effects_are_found(cfg2):
    """Statistical acceptance in the spirit of cellregmap/test/test_struct_lmm2.py:118-119."""
    c, Ls, crm, dense, pv, info, st = cfg2
    assert set(np.argsort(pv)[:2]) == {10, 11}
    assert np.all(pv[[10, 11]] < 1e-7)
    others = np.delete(pv, [10, 11])
    assert np.median(others) > 0.1


@pytest.mark.parametrize("route", ROUTES)
def test_config3_block_headline_size(route):
    """BASELINE config 3 (the bench workload: 20 000 cells x 50 contexts, mode C, r ~ 5 000) on one
    block of variants, once per product route: the oracle on 32 seeded random variants (device's decompositions),
    dense path == donor-collapsed path, affine invariance of the phenotype, and the factorisation behind it."""
    with _route(route):
        _config3_block(route)


def _config3_block(route):
    from cellregmap_amd import CellRegMap, GenotypePanel, get_L_values
    from cellregmap_amd.synth import make_config

    c = make_config("cfg3", n_variants=256)
    n = c.y.size
    Ls = get_L_values(c.hK, c.E)
    crm = CellRegMap(c.y, c.E, W=c.W, Ls=Ls)
    dense = GenotypePanel(c.G, groups=None)
    from cellregmap_amd import _engine, _lib
    fallbacks = _lib.load().crm_test_sync_fallbacks(_engine._context(0))
    pv, info, st = crm.scan_interaction(dense, return_stats=True)
    assert np.all(np.isfinite(pv)) and np.all((pv > 0) & (pv <= 1))
    # the direct route's contraction of this block runs in its persistent per-XCD form (3 900 tiles); on a GPU this process
    # has to itself none of its bounded waits runs out (a context that saw them time out would leave that form:
    # crm_test_sync_fallbacks)
    assert _lib.load().crm_test_sync_fallbacks(_engine._context(0)) == fallbacks

    if route == "kinship":   # (the background does not depend on the scan's route: once is enough)
        # factorisation: Q0 S0 Q0' v == hS hS' v for one interior grid point, Q0 orthonormal
        i = 6
        rho = crm._rho1[i]
        Q0, S0 = crm._bg.read(i, n)
        v = np.random.default_rng(0).normal(size=(n, 2))
        KE = c.E @ (c.E.T @ v)
        u = Ls.us
        lhs = rho * KE + (1 - rho) * sum(u[:, [j]] * (c.hK @ (c.hK.T @ (u[:, [j]] * v))) for j in range(u.shape[1]))
        rhs = Q0 @ (S0[:, None] * (Q0.T @ v))
        assert np.abs(lhs - rhs).max() <= 1e-9 * np.abs(lhs).max()
        G = Q0.T @ Q0
        assert np.abs(G - np.eye(G.shape[0])).max() < 1e-11

    # oracle on 32 seeded random variants (+ the planted ones), sharing the device's decompositions:
    # rho*, delta, lml, Q, F, the eigenvalues of F and p
    o = _oracle_on_device_decomposition(crm, c.y, c.E, c.W, Ls)
    pick = sorted(set(np.random.default_rng(2024).choice(256, size=32, replace=False).tolist()) | {10, 11})
    opv, oinfo, ost = o.scan_interaction(c.G[:, pick], return_stats=True)
    _compare_with_oracle(pv, info, st, pick, opv, oinfo, ost, c.E.shape[1])
    del o

    # donor-collapsed path
    pv_c, info_c, st_c = crm.scan_interaction(GenotypePanel(c.G), return_stats=True)
    assert_allclose(info_c["rho1"], info["rho1"], atol=1e-12)
    assert_allclose(st_c["Q"], st["Q"], rtol=5e-6)
    assert np.all(np.abs(pv_c - pv) <= P_RTOL * pv + P_ATOL)

    # y -> a y + b on the same background
    crm2 = CellRegMap(2.5 * c.y + 3.0, c.E, W=c.W, Ls=Ls, background=crm._bg)
    pv2, info2, st2 = crm2.scan_interaction(dense, return_stats=True)
    assert_allclose(info2["rho1"], info["rho1"], atol=1e-12)
    assert np.all(np.abs(pv2 - pv) <= P_RTOL * pv + P_ATOL)


@pytest.mark.parametrize("route", ROUTES)
def test_config4_per_gpu_shape_64_genes_against_one_panel(route):
    """BASELINE config 4's per-GPU shape: 64 phenotypes x one block of the config-3 panel in one pass
    (``scan_interaction_many``; over 8 GPUs each rank runs exactly this on its shard of the variants), once per product
    route (direct: the cost model picks between one contraction per (variant, rho*) pair and the shared-H form).
    A few (gene, variant) pairs against the oracle on the device's decompositions; the pass against the
    single-gene scan for one gene; the genes must not all agree on rho* (else the pair logic is idle)."""
    with _route(route):
        _config4_shape()


def _config4_shape():
    from cellregmap_amd import CellRegMap, GenotypePanel, get_L_values, scan_interaction_many
    from cellregmap_amd.synth import make_config

    c = make_config("cfg3", n_variants=256)
    n = c.y.size
    rng = np.random.default_rng(64)
    Ls = get_L_values(c.hK, c.E)
    first = CellRegMap(c.y, c.E, W=c.W, Ls=Ls)
    ys = [c.y]
    for g in range(1, 64):  # other genes: shuffled / noisier / pure-noise phenotypes on the same cohort
        kind = g % 3
        ys.append(c.y[rng.permutation(n)] if kind == 0 else (c.y + (0.5 + g / 16) * rng.normal(size=n) if kind == 1
                                                              else rng.normal(size=n)))
    crms = [first] + [CellRegMap(y, c.E, W=c.W, Ls=Ls, background=first._bg) for y in ys[1:]]
    panel = GenotypePanel(c.G, groups=None)
    pv, info = scan_interaction_many(crms, panel)
    assert pv.shape == (64, 256) and np.all(np.isfinite(pv)) and np.all((pv > 0) & (pv <= 1))
    assert np.mean([len(set(info["rho1"][:, j])) for j in range(256)]) > 2.0
    # one gene through the single-gene scan
    spv, sinfo = crms[17].scan_interaction(panel)
    assert np.array_equal(info["rho1"][17], sinfo["rho1"])
    assert np.all(np.abs(pv[17] - spv) <= 1e-7 * spv + P_ATOL)
    # (gene, variant) pairs against the oracle
    prng = np.random.default_rng(416)   # 8 genes x 2 variants = 16 (gene, variant) pairs, seeded
    pairs = [(int(g), sorted(prng.choice(256, size=2, replace=False).tolist()))
             for g in sorted(prng.choice(64, size=8, replace=False).tolist())]
    for g, pick in pairs:
        o = _oracle_on_device_decomposition(first, ys[g], c.E, c.W, Ls)
        opv, oinfo = _oracle_scan_allowing_ties(o, c.G[:, pick], info["rho1"][g, pick])
        assert_allclose(info["rho1"][g, pick], oinfo["rho1"], atol=1e-12)
        assert np.all(np.abs(pv[g, pick] - opv) <= P_RTOL * opv + P_ATOL), (g, np.c_[pv[g, pick], opv])
        total = oinfo["e2"] + oinfo["g2"] + oinfo["eps2"]
        for k in ("e2", "g2", "eps2"):  # (a component at the boundary, v0 -> 0, only has absolute accuracy)
            assert np.all(np.abs(info[k][g, pick] - oinfo[k]) <= 1e-5 * oinfo[k] + 1e-6 * total)
        del o


@pytest.mark.parametrize("route", ROUTES)
def test_config5_hundred_thousand_cells(route):
    with _route(route):
        _config5(route)


def _config5(route):
    """BASELINE config 5 (100 000 cells x 50 contexts, mode C: 10 050 columns, Q0 set ~ 89 GB in HBM) on one
    block of variants: the factorisation behind the background (random probes: the r x r Gram is 2e13 flop
    on the host), dense path == donor-collapsed path, affine invariance of the phenotype, and the oracle on
    eight variants over two distinct rho* (two phenotypes on one background; per variant the null fit at rho* and
    at a neighbouring grid point, the score statistic, F, its spectrum and the p-value; 8 GB per grid point cross
    PCIe, so four grid points are read back, not all eleven)."""
    from cellregmap_amd import CellRegMap, GenotypePanel, get_L_values
    from cellregmap_amd.synth import make_config
    from oracle.davies import davies_pvalue
    from oracle.lmm import LMM
    from oracle.scoretest import LowRankCov, Projection, score_F, score_Q

    c = make_config("cfg5", n_variants=192)
    n = c.y.size
    assert n == 100_000
    Ls = get_L_values(c.hK, c.E)
    crm = CellRegMap(c.y, c.E, W=c.W, Ls=Ls)
    assert max(crm._bg.rank(i) for i in range(11)) >= 10_000
    dense = GenotypePanel(c.G, groups=None)
    pv, info, st = crm.scan_interaction(dense, return_stats=True)
    # (at 100 000 cells the planted effects are beyond the range of a double: Q / lambda ~ 2 500, p = 0.0 from
    # Davies and from the Liu fall-back alike)
    assert np.all(np.isfinite(pv)) and np.all((pv >= 0) & (pv <= 1))
    # planted GxC variants (10, 11) come out on top
    assert set(np.argsort(pv)[:2]) == {10, 11}

    if route == "kinship":   # (neither depends on the dense scan's route: once is enough)
        # donor-collapsed path
        pv_c, info_c, st_c = crm.scan_interaction(GenotypePanel(c.G), return_stats=True)
        assert_allclose(info_c["rho1"], info["rho1"], atol=1e-12)
        assert_allclose(st_c["Q"], st["Q"], rtol=5e-6)
        assert np.all(np.abs(pv_c - pv) <= P_RTOL * pv + P_ATOL)

        # y -> a y + b on the same background
        crm2 = CellRegMap(2.5 * c.y + 3.0, c.E, W=c.W, Ls=Ls, background=crm._bg)
        pv2, info2 = crm2.scan_interaction(dense)
        assert_allclose(info2["rho1"], info["rho1"], atol=1e-12)
        assert np.all(np.abs(pv2 - pv) <= P_RTOL * pv + P_ATOL)
        del crm2

    # the grid point most variants selected, and a neighbour
    idx = np.rint(info["rho1"] * 10).astype(int)
    i_star = int(np.bincount(idx, minlength=11).argmax())
    rho = crm._rho1[i_star]
    Q0, S0 = crm._bg.read(i_star, n)
    # factorisation Q0 S0 Q0' v == Sigma(rho) v and orthonormality along random probes
    rng = np.random.default_rng(0)
    v = rng.normal(size=(n, 2))
    u = Ls.us
    KE = c.E @ (c.E.T @ v)
    lhs = rho * KE + (1 - rho) * sum(u[:, [j]] * (c.hK @ (c.hK.T @ (u[:, [j]] * v))) for j in range(u.shape[1]))
    rhs = Q0 @ (S0[:, None] * (Q0.T @ v))
    assert np.abs(lhs - rhs).max() <= 1e-9 * np.abs(lhs).max()
    x = rng.normal(size=(Q0.shape[1], 3))
    assert np.abs(Q0.T @ (Q0 @ x) - x).max() <= 1e-10 * np.abs(x).max()
    del Q0, S0

    # The oracle on eight variants over two distinct rho*: four of this phenotype at its modal grid point and four of
    # a second phenotype on the same background whose modal grid point is another one (a shuffled / noisier / pure-noise
    # outcome).  Per variant: the null fit at rho* (lml, delta) and at a neighbouring grid point (which must lose),
    # the score statistic, F, its spectrum and the p-value.
    def check(y, pv, info, st, i_star, count):
        i_nb = i_star + 1 if i_star < 10 else i_star - 1
        Q0, S0 = crm._bg.read(i_star, n)
        Q0n, S0n = crm._bg.read(i_nb, n)
        sel = np.flatnonzero(np.rint(info["rho1"] * 10).astype(int) == i_star)
        pick = [int(j) for j in np.random.default_rng(5 + i_star).choice(sel, size=min(count, sel.size), replace=False)]
        for j in pick:
            g = c.G[:, [j]]
            X = np.concatenate((c.W, g), axis=1)
            lmm = LMM(y, X, ((Q0,), S0), restricted=True)
            lmm.fit(verbose=False)
            assert_allclose(st["lml"][j], lmm.lml(), rtol=1e-11)
            assert_allclose(st["delta"][j], lmm.delta, rtol=5e-6)
            other = LMM(y, X, ((Q0n,), S0n), restricted=True)
            other.fit(verbose=False)
            assert other.lml() < lmm.lml()          # the device's argmax beats the neighbouring grid point
            P = Projection(LowRankCov(Q0, S0, lmm.v0, lmm.v1), X)
            half_dK = g * c.E
            Q = score_Q(P, half_dK, y)
            F = score_F(P, half_dK)
            assert_allclose(st["Q"][j], Q, rtol=1e-6)
            assert np.abs(st["F"][j] - F).max() <= 1e-6 * np.abs(F).max()
            lam = np.linalg.eigvalsh(F)
            assert np.abs(st["lambda"][j] - lam).max() <= 1e-6 * np.abs(lam).max()
            opv = davies_pvalue(Q, F, True)[0]
            assert abs(pv[j] - opv) <= P_RTOL * opv + P_ATOL, (pv[j], opv)
        return len(pick)

    checked = check(c.y, pv, info, st, i_star, 4)
    second = None
    for y2 in (rng.normal(size=n), c.y[rng.permutation(n)], c.y + 4.0 * rng.normal(size=n)):
        crm_b = CellRegMap(y2, c.E, W=c.W, Ls=Ls, background=crm._bg)
        pv_b, info_b, st_b = crm_b.scan_interaction(dense, return_stats=True)
        i_b = int(np.bincount(np.rint(info_b["rho1"] * 10).astype(int), minlength=11).argmax())
        if i_b != i_star:
            second = (y2, pv_b, info_b, st_b, i_b)
            break
    assert second is not None, "no second phenotype with another modal rho* found"
    checked += check(*second, 4)
    assert checked >= 8


def test_fold_of_a_dense_ragged_donor_level_factor_at_config2_size(kernel_form):
    """The kinship factor as the reference's simulator forms it (_simulate.py:83-102, 477-479: hK = U sqrt(S) of the
    donor-block K) is DENSE at donor level -- m x m, every donor against every component -- and real cohorts are ragged.
    At BASELINE config 2's size (5 000 cells, 20 contexts; 48 donors of 40 .. 160 cells) the library must find the donor
    structure, fold the dense donor-level block into its mixing matrices (MixK; with 20 contexts the fold is not the default
    -- one launch over [us | E1] per donor measured faster at config 2 -- so the form "kin_fold" asks for it) and scan
    general genotypes through the folded route to the oracle's numbers -- and to the direct route's (the contraction
    against Q0(rho*) over all cells); the unfolded kinship-structure route, config 2's default, must agree as well."""
    import parity_bounds
    from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib, get_L_values
    from cellregmap_amd.synth import column_normalize
    from oracle.crm import khatri_rao_halves

    rng = np.random.default_rng(77)
    donors, n, k0 = 48, 5000, 20
    counts = rng.integers(40, 161, size=donors)
    counts = np.maximum(20, (counts * (n / counts.sum())).astype(int))
    counts[-1] += n - counts.sum()
    donor = np.repeat(np.arange(donors), counts)
    assert donor.size == n and counts.min() >= 20 and counts.max() > 2 * counts.min()
    # U sqrt(S) of the donor-block K = Z Z' / mean diag + 1e-8 I restricted to its m leading directions: Z D R with R an
    # orthogonal m x m matrix (cellregmap_amd/synth.py: kinship_factor "rotated") -- dense at donor level, ragged donors
    R, _ = np.linalg.qr(rng.normal(size=(donors, donors)))
    hKd = np.sqrt((counts + 1e-8) / counts)[:, None] * R
    hK = hKd[donor]
    E = column_normalize(rng.normal(size=(n, k0)))
    W = np.ones((n, 1))
    G = column_normalize(rng.normal(size=(n, 96)))                      # general (cell-level) genotypes
    yk = sum(E[:, i] * (hK @ rng.normal(size=donors)) for i in range(k0))
    y = 0.3 + 0.4 * G[:, 3] * (E @ rng.normal(size=k0)) + yk / yk.std() + E @ rng.normal(size=k0) * 0.3 + rng.normal(size=n)
    Ls = get_L_values(hK, E)
    lib, ctx = _lib.load(), _engine._context(0)
    plain = CellRegMap(y, E, W=W, Ls=Ls)                               # the default at this size: structure used, not folded
    assert lib.crm_background_kinship_groups(plain._bg.handle) == donors
    assert lib.crm_background_kinship_folded(plain._bg.handle) == 0
    panel = GenotypePanel(G, groups=None)
    pv_plain, info_plain = plain.scan_interaction(panel)
    del plain
    _engine._bg_cache.clear()
    kernel_form("kin_fold", 2)                                         # (read when the structure is announced)
    crm = CellRegMap(y, E, W=W, Ls=Ls)
    assert lib.crm_background_kinship_groups(crm._bg.handle) == donors
    assert lib.crm_background_kinship_folded(crm._bg.handle) > 0
    pv, info, st = crm.scan_interaction(panel, return_stats=True)
    assert np.array_equal(info["rho1"], info_plain["rho1"])
    with _route("direct"):
        pv0, info0, st0 = crm.scan_interaction(panel, return_stats=True)
    assert np.array_equal(info["rho1"], info0["rho1"])
    bq, bp, _ = parity_bounds.bounds(crm, panel)
    parity_bounds.assert_bounds_are_informative(bq, bp, pv)
    assert_allclose(st["lml"], st0["lml"], rtol=1e-11)
    parity_bounds.assert_Q_within(st["Q"], st0["Q"], bq, np.maximum(np.abs(st0["Q"]), np.trace(st0["F"], axis1=1, axis2=2)), "routes")
    parity_bounds.assert_p_within(pv, pv0, bp, "routes")
    parity_bounds.assert_p_within(pv_plain, pv0, bp, "unfolded")
    o = _oracle_on_device_decomposition(crm, y, E, W, khatri_rao_halves(hK, E))
    pick = sorted(set(np.random.default_rng(5).choice(96, size=10, replace=False).tolist()) | {3})
    opv, oinfo, ost = o.scan_interaction(G[:, pick], return_stats=True)
    _compare_with_oracle(pv, info, st, pick, opv, oinfo, ost, k0, bounds=(bq[pick], bp[pick]))
    assert pv[3] < 1e-6                                                  # the planted GxE variant is found
