"""Parity at BASELINE sizes through size-independent properties (config 2: 5 000 cells x 20
contexts, mode C, r ~ 1 000) plus an oracle spot check on a handful of variants that shares the
device's decomposition (the oracle's own LAPACK SVD of 5 000 x 1 020 x 11 would dominate the
run time)."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

pytestmark = pytest.mark.gpu

P_RTOL, P_ATOL = 1e-5, 1e-13


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


@pytest.fixture(scope="module")
def cfg2():
    from cellregmap_amd import CellRegMap, GenotypePanel, get_L_values
    from cellregmap_amd.synth import make_config

    c = make_config("cfg2", n_variants=384)
    Ls = get_L_values(c.hK, c.E)
    crm = CellRegMap(c.y, c.E, W=c.W, Ls=Ls)
    dense = GenotypePanel(c.G, groups=None)
    pv, info, st = crm.scan_interaction(dense, return_stats=True)
    return c, Ls, crm, dense, pv, info, st


def test_oracle_spot_check_at_config2(cfg2):
    from oracle.crm import OracleCellRegMap

    c, Ls, crm, dense, pv, info, st = cfg2
    qs = {}
    for i, rho in enumerate(crm._rho1):
        Q0, S0 = crm._bg.read(i, c.y.size)
        qs[rho] = ((Q0,), S0)
    o = OracleCellRegMap.__new__(OracleCellRegMap)
    o._polish = False
    o._y, o._E0, o._W, o._E1 = c.y, c.E, c.W, c.E
    o._Ls, o._rho, o._half, o._qs = Ls, list(crm._rho1), {}, qs
    pick = [0, 5, 10, 11, 200, 383]
    opv, oinfo, ost = o.scan_interaction(c.G[:, pick], return_stats=True)
    assert_allclose(info["rho1"][pick], oinfo["rho1"], atol=1e-12)
    assert_allclose(st["Q"][pick], ost["Q"], rtol=1e-6)
    assert np.all(np.abs(pv[pick] - opv) <= P_RTOL * opv + P_ATOL), np.c_[pv[pick], opv]


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


def test_config3_block_headline_size():
    """BASELINE config 3 (the bench workload: 20 000 cells x 50 contexts, mode C, r ~ 5 000) on one
    block of variants: oracle spot check on the device's decompositions, dense path == donor-collapsed
    path, affine invariance of the phenotype, and the factorisation behind it."""
    from cellregmap_amd import CellRegMap, GenotypePanel, get_L_values
    from cellregmap_amd.synth import make_config
    from oracle.crm import OracleCellRegMap

    c = make_config("cfg3", n_variants=256)
    n = c.y.size
    Ls = get_L_values(c.hK, c.E)
    crm = CellRegMap(c.y, c.E, W=c.W, Ls=Ls)
    dense = GenotypePanel(c.G, groups=None)
    pv, info, st = crm.scan_interaction(dense, return_stats=True)
    assert np.all(np.isfinite(pv)) and np.all((pv > 0) & (pv <= 1))

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

    # oracle on three variants, sharing the device's (Q0, S0)
    qs = {crm._rho1[i]: ((Q0,), S0)}
    for j, r in enumerate(crm._rho1):
        if j != i:
            q, s = crm._bg.read(j, n)
            qs[r] = ((q,), s)
    o = OracleCellRegMap.__new__(OracleCellRegMap)
    o._polish = False
    o._y, o._E0, o._W, o._E1 = c.y, c.E, c.W, c.E
    o._Ls, o._rho, o._half, o._qs = Ls, list(crm._rho1), {}, qs
    pick = [0, 101, 255]
    opv, oinfo, ost = o.scan_interaction(c.G[:, pick], return_stats=True)
    assert_allclose(info["rho1"][pick], oinfo["rho1"], atol=1e-12)
    assert_allclose(st["Q"][pick], ost["Q"], rtol=1e-6)
    assert np.all(np.abs(pv[pick] - opv) <= P_RTOL * opv + P_ATOL), np.c_[pv[pick], opv]
    del qs, o

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


def test_config4_per_gpu_shape_64_genes_against_one_panel():
    """BASELINE config 4's per-GPU shape: 64 phenotypes x one block of the config-3 panel in one pass
    (``scan_interaction_many``; over 8 GPUs each rank runs exactly this on its shard of the variants).
    A few (gene, variant) pairs against the oracle on the device's decompositions; the pass against the
    single-gene scan for one gene; the genes must not all agree on rho* (else the pair logic is idle)."""
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
    for g, pick in ((0, [10, 200]), (5, [3]), (40, [255])):
        o = _oracle_on_device_decomposition(first, ys[g], c.E, c.W, Ls)
        opv, oinfo = o.scan_interaction(c.G[:, pick])
        assert_allclose(info["rho1"][g, pick], oinfo["rho1"], atol=1e-12)
        assert np.all(np.abs(pv[g, pick] - opv) <= P_RTOL * opv + P_ATOL), (g, np.c_[pv[g, pick], opv])
        total = oinfo["e2"] + oinfo["g2"] + oinfo["eps2"]
        for k in ("e2", "g2", "eps2"):  # (a component at the boundary, v0 -> 0, only has absolute accuracy)
            assert np.all(np.abs(info[k][g, pick] - oinfo[k]) <= 1e-5 * oinfo[k] + 1e-6 * total)
        del o


def test_config5_hundred_thousand_cells():
    """BASELINE config 5 (100 000 cells x 50 contexts, mode C: 10 050 columns, Q0 set ~ 89 GB in HBM) on one
    block of variants: the factorisation behind the background (random probes: the r x r Gram is 2e13 flop
    on the host), dense path == donor-collapsed path, affine invariance of the phenotype, and the oracle on
    variants whose rho* is the grid point read back (the null fit at rho* and at a neighbouring grid
    point, the score statistic, F and the p-value; 8 GB per grid point cross PCIe, so not all eleven)."""
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
    i_nb = i_star + 1 if i_star < 10 else i_star - 1
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

    pick = [int(j) for j in np.flatnonzero(idx == i_star)[:2]]
    Q0n, S0n = crm._bg.read(i_nb, n)
    for j in pick:
        g = c.G[:, [j]]
        X = np.concatenate((c.W, g), axis=1)
        lmm = LMM(c.y, X, ((Q0,), S0), restricted=True)
        lmm.fit(verbose=False)
        assert_allclose(st["lml"][j], lmm.lml(), rtol=1e-11)
        assert_allclose(st["delta"][j], lmm.delta, rtol=5e-6)
        other = LMM(c.y, X, ((Q0n,), S0n), restricted=True)
        other.fit(verbose=False)
        assert other.lml() < lmm.lml()          # the device's argmax beats the neighbouring grid point
        P = Projection(LowRankCov(Q0, S0, lmm.v0, lmm.v1), X)
        half_dK = g * c.E
        Q = score_Q(P, half_dK, c.y)
        F = score_F(P, half_dK)
        assert_allclose(st["Q"][j], Q, rtol=1e-6)
        assert np.abs(st["F"][j] - F).max() <= 1e-6 * np.abs(F).max()
        opv = davies_pvalue(Q, F, True)[0]
        assert abs(pv[j] - opv) <= P_RTOL * opv + P_ATOL, (pv[j], opv)
