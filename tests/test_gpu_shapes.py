"""Odd shapes through the whole scan vs the oracle: cell counts that are not multiples of any
tile size, many covariates, many contexts, ragged donors, sub-ranges of a panel."""
import ctypes

import numpy as np
import pytest
from numpy.testing import assert_allclose

pytestmark = pytest.mark.gpu

P_RTOL, P_ATOL = 1e-5, 1e-13


from fuzz_cases import random_problem as _random_problem  # noqa: E402


@pytest.mark.parametrize("n,k0,c,p,donors,mode", [
    (257, 7, 3, 5, 11, "B"),
    (131, 2, 8, 9, 6, "B"),
    (300, 33, 2, 6, 5, "A"),
    (199, 3, 1, 130, 9, "C"),
    (150, 70, 1, 3, 4, "A"),
    (260, 120, 2, 4, 5, "A"),
    (300, 128, 1, 3, 6, "B"),
])
def test_odd_shapes_match_oracle(n, k0, c, p, donors, mode):
    from cellregmap_amd import CellRegMap, GenotypePanel
    from oracle.crm import OracleCellRegMap

    y, E, W, G, kw = _random_problem(n, k0, c, p, donors, seed=n + k0, mode=mode)
    crm = CellRegMap(y, E, W=W, **kw)
    opv, oinfo, ost = OracleCellRegMap(y, E, W=W, **kw).scan_interaction(G, return_stats=True)
    for groups in (None, "auto"):
        pv, info, st = crm.scan_interaction(GenotypePanel(G, groups=groups), return_stats=True)
        assert_allclose(info["rho1"], oinfo["rho1"], atol=1e-12)
        assert_allclose(st["Q"], ost["Q"], rtol=1e-6)
        assert np.all(np.abs(pv - opv) <= P_RTOL * opv + P_ATOL), np.c_[pv, opv]


def _sweep_cases():
    rng = np.random.default_rng(2026)
    cases = []
    for i in range(20):
        mode = "ABC"[i % 3]
        n = int(rng.integers(40, 420))
        k0 = int(rng.choice([1, 2, 3, 5, 9, 13, 17, 31, 50, 63, 64, 65])) if mode != "C" else int(rng.integers(1, 7))
        c = int(rng.choice([1, 1, 2, 4, 8, 9, 12]))
        p = int(rng.integers(1, 40))
        donors = int(rng.integers(3, 12))
        perm = ["none", "E", "G"][int(rng.integers(0, 3))]
        cases.append((i, n, k0, c, p, donors, mode, perm))
    return cases


@pytest.mark.parametrize("i,n,k0,c,p,donors,mode,perm", _sweep_cases())
def test_randomised_sweep_against_the_oracle(i, n, k0, c, p, donors, mode, perm):
    """Seeded sweep over cell counts, context counts (tile-edge cases of the Khatri-Rao operand), covariate
    counts (both null-fit kernels), panel widths, background modes and the permutation hooks; dense and
    donor-collapsed paths."""
    from cellregmap_amd import CellRegMap, GenotypePanel
    from oracle.crm import OracleCellRegMap

    from cellregmap_amd import _engine, _lib

    y, E, W, G, kw = _random_problem(n, k0, c, p, donors, seed=1000 + i, mode=mode)
    idx = np.random.default_rng(i).permutation(n)
    hooks = {} if perm == "none" else ({"idx_E": idx} if perm == "E" else {"idx_G": idx})
    crm = CellRegMap(y, E, W=W, **kw)
    opv, oinfo, ost = OracleCellRegMap(y, E, W=W, **kw).scan_interaction(G, return_stats=True, **hooks)
    for groups in (None, "auto"):
        pv, info, st = crm.scan_interaction(GenotypePanel(G, groups=groups), return_stats=True, **hooks)
        assert_allclose(info["rho1"], oinfo["rho1"], atol=1e-12)
        # verbatim procedure: two roundings of the objective may end Brent's search 1e-6 apart (tests/test_oracle_spread.py
        # measures that on the oracle alone): 5e-6 here, the sharp comparison is the polished one below
        assert_allclose(st["Q"], ost["Q"], rtol=5e-6)
        assert np.all(np.abs(pv - opv) <= P_RTOL * opv + P_ATOL), np.c_[pv, opv]
    if c <= 8:   # (the polish is built for the register null-fit kernel)
        lib = _lib.load()
        _lib.check(lib.crm_set_null_fit_polish(_engine._context(0), 1))
        try:
            ppv, pinfo, pst = OracleCellRegMap(y, E, W=W, polish=True, **kw).scan_interaction(G, return_stats=True, **hooks)
            for groups in (None, "auto"):
                pv, info, st = crm.scan_interaction(GenotypePanel(G, groups=groups), return_stats=True, **hooks)
                assert_allclose(info["rho1"], pinfo["rho1"], atol=1e-12)
                assert_allclose(st["Q"], pst["Q"], rtol=1e-8)
                assert np.all(np.abs(pv - ppv) <= 2e-6 * ppv + P_ATOL), np.c_[pv, ppv]
        finally:
            _lib.check(lib.crm_set_null_fit_polish(_engine._context(0), 0))


def test_sub_range_of_a_panel_through_the_c_abi():
    """crm_scan_interaction(first, count) on an odd, unaligned sub-range equals the full scan."""
    from cellregmap_amd import CellRegMap, GenotypePanel, _lib

    y, E, W, G, kw = _random_problem(180, 4, 2, 77, 8, seed=5, mode="B")
    crm = CellRegMap(y, E, W=W, **kw)
    lib = _lib.load()
    for groups in (None, "auto"):
        panel = GenotypePanel(G, groups=groups)
        full, _ = crm.scan_interaction(panel)
        gene = crm._bind_gene()
        first, count = 13, 51
        pv = np.empty(count)
        rho = np.empty(count)
        _lib.check(lib.crm_scan_interaction(gene, panel.handle, first, count, None, None, _lib.ptr(pv), _lib.ptr(rho),
                                            None, None, None, None, None, None, None, None, None))
        assert np.array_equal(pv, full[first:first + count])
        rc = lib.crm_scan_interaction(gene, panel.handle, 70, 20, None, None, _lib.ptr(pv), None, None, None, None,
                                      None, None, None, None, None, None)
        assert rc != 0 and b"outside the panel" in lib.crm_last_error()
