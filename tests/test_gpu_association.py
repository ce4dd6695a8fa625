"""Association LRT paths on the GPU vs the CPU oracle (cellregmap/_cellregmap.py:246-314, 443-531).

Tolerance: p-values rtol 1e-5 (north star); the LRT statistic is a difference of two
log-likelihoods of size ~n, so p carries |d lml| ~ 1e-11 * n of rounding on top."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

pytestmark = pytest.mark.gpu


def _cohort(donors, cells, k, p, seed):
    from cellregmap_amd.synth import make_cohort

    return make_cohort(donors, cells, k, p, seed=seed)


@pytest.mark.parametrize("fast", [False, True])
@pytest.mark.parametrize("mode", ["A", "B"])
def test_scan_association_matches_oracle(fast, mode):
    from cellregmap_amd import CellRegMap
    from oracle.crm import OracleCellRegMap

    c = _cohort(10, 20, 3, 24, seed=21)
    rng = np.random.default_rng(0)
    W = np.concatenate([c.W, rng.normal(size=(c.y.size, 2))], axis=1)
    kw = {"hK": c.hK} if mode == "B" else {}
    crm = CellRegMap(c.y, c.E, W=W, **kw)
    ocrm = OracleCellRegMap(c.y, c.E, W=W, **kw)
    if fast:
        pv, info, st = crm.scan_association_fast(c.G, return_stats=True)
        opv, oinfo = ocrm.scan_association_fast(c.G)
    else:
        pv, info, st = crm.scan_association(c.G, return_stats=True)
        opv, oinfo = ocrm.scan_association(c.G)
    for k in ("rho1", "e2", "g2", "eps2"):
        assert info[k].shape == (1,)
        assert_allclose(info[k], oinfo[k], rtol=1e-5, atol=1e-12)
    assert np.all(np.abs(pv - opv) <= 1e-5 * opv + 1e-300), np.c_[pv, opv]
    assert_allclose(st["alt_lml"], st["alt_lml"])  # finite
    assert np.all(np.isfinite(st["alt_lml"])) and np.isfinite(st["null_lml"])


@pytest.mark.parametrize("fast", [False, True])
def test_run_association_wrapper_keeps_the_positional_swap(fast):
    """run_association(y, W, E, G, hK) binds W to the contexts slot and E to the fixed effects
    (_cellregmap.py:498, :529): the 6 contexts become covariates -> the wide null-fit kernel."""
    from cellregmap_amd import run_association, run_association_fast
    from oracle import crm as ocrm

    c = _cohort(12, 15, 6, 16, seed=22)
    f, of = (run_association_fast, ocrm.run_association_fast) if fast else (run_association, ocrm.run_association)
    pv, info = f(c.y, c.W, c.E, c.G, hK=c.hK)
    opv, oinfo = of(c.y, c.W, c.E, c.G, hK=c.hK)
    assert_allclose(info["rho1"], oinfo["rho1"], atol=1e-12)
    assert np.all(np.abs(pv - opv) <= 1e-5 * opv + 1e-300), np.c_[pv, opv]


def test_wide_and_register_null_fit_agree():
    """c = 9 > CRM_MAX_COV forces the LDS kernel; compare with the oracle directly."""
    from cellregmap_amd import CellRegMap
    from oracle.crm import OracleCellRegMap

    c = _cohort(10, 20, 3, 8, seed=23)
    rng = np.random.default_rng(1)
    W = np.concatenate([c.W, rng.normal(size=(c.y.size, 8))], axis=1)  # 9 columns
    pv, info, st = CellRegMap(c.y, c.E, W=W, hK=c.hK).scan_association(c.G, return_stats=True)
    ocrm = OracleCellRegMap(c.y, c.E, W=W, hK=c.hK)
    opv, oinfo = ocrm.scan_association(c.G)
    assert_allclose(info["rho1"], oinfo["rho1"], atol=1e-12)
    assert np.all(np.abs(pv - opv) <= 1e-5 * opv + 1e-300), np.c_[pv, opv]


def _assoc_sweep():
    rng = np.random.default_rng(77)
    out = []
    for i in range(10):
        out.append((i, int(rng.integers(5, 14)), int(rng.integers(8, 30)), int(rng.integers(1, 9)), int(rng.integers(1, 40)),
                    int(rng.choice([1, 2, 5, 9, 20])), "ABC"[i % 3], bool(i % 2)))
    return out


@pytest.mark.parametrize("i,donors,cells,k,p,c,mode,fast", _assoc_sweep())
def test_association_sweep_against_the_oracle(i, donors, cells, k, p, c, mode, fast):
    """Seeded sweep: cohort shapes, covariate counts on both sides of the register / LDS null-fit split,
    the three background modes, full and fast scans."""
    from cellregmap_amd import CellRegMap, get_L_values
    from oracle import crm as ocrm

    co = _cohort(donors, cells, k, p, seed=300 + i)
    rng = np.random.default_rng(i)
    W = np.concatenate([co.W, rng.normal(size=(co.y.size, c - 1))], axis=1) if c > 1 else co.W
    if mode == "A":
        kw, okw = {}, {}
    elif mode == "B":
        kw, okw = {"hK": co.hK}, {"hK": co.hK}
    else:
        kw, okw = {"Ls": get_L_values(co.hK, co.E)}, {"Ls": ocrm.khatri_rao_halves(co.hK, co.E)}
    dev = CellRegMap(co.y, co.E, W=W, **kw)
    ora = ocrm.OracleCellRegMap(co.y, co.E, W=W, **okw)
    pv, info = (dev.scan_association_fast if fast else dev.scan_association)(co.G)
    opv, oinfo = (ora.scan_association_fast if fast else ora.scan_association)(co.G)
    assert_allclose(info["rho1"], oinfo["rho1"], atol=1e-12)
    for key in ("e2", "g2", "eps2"):
        assert_allclose(info[key], oinfo[key], rtol=1e-5, atol=1e-12)
    assert np.all(np.abs(pv - opv) <= 1e-5 * opv + 1e-300), np.c_[pv, opv]


@pytest.mark.parametrize("fast", [False, True])
@pytest.mark.parametrize("contexts", [70, 100, 126])
def test_run_association_with_many_contexts(contexts, fast):
    """run_association(y, W, E, G, hK) binds the contexts to the fixed-effect slot (_cellregmap.py:498, :529), so a cohort
    with 100 contexts fits LMMs with 100 covariate columns; round 3 refused more than 61.  Now 63 .. 128 columns go through
    the slower null-fit kernel whose scratch lives in global memory (nullfit_xwide.hip): null model and every p-value
    against the oracle."""
    from cellregmap_amd import run_association, run_association_fast
    from oracle import crm as ocrm

    c = _cohort(10, 40, contexts, 12, seed=24)                # 400 cells
    f, of = (run_association_fast, ocrm.run_association_fast) if fast else (run_association, ocrm.run_association)
    pv, info = f(c.y, c.W, c.E, c.G, hK=c.hK)
    opv, oinfo = of(c.y, c.W, c.E, c.G, hK=c.hK)
    assert_allclose(info["rho1"], oinfo["rho1"], atol=1e-12)
    for key in ("e2", "g2", "eps2"):
        assert_allclose(info[key], oinfo[key], rtol=1e-5, atol=1e-10)
    assert np.all(np.abs(pv - opv) <= 1e-5 * opv + 1e-300), np.c_[pv, opv]


@pytest.mark.parametrize("fast", [False, True])
def test_streamed_association_scan_equals_the_one_panel_scan(fast, monkeypatch):
    """A host matrix of many SNPs goes to the device in column chunks beside the scan of the chunks that have arrived
    (``CellRegMap._streamed_panels``), for the association scans as for the interaction scan: same p-values, alternative
    likelihoods and null model as the one-panel scan, one progress report over all chunks."""
    from cellregmap_amd import CellRegMap
    from cellregmap_amd.synth import make_cohort

    c = make_cohort(9, 30, 4, 700, seed=61)
    G = c.G + 0.05 * np.random.default_rng(3).normal(size=c.G.shape)
    crm = CellRegMap(c.y, c.E, W=c.W, hK=c.hK)
    scan = crm.scan_association_fast if fast else crm.scan_association
    monkeypatch.setenv("CELLREGMAP_AMD_STREAM_CHUNK", "0")
    pv, info, st = scan(G, return_stats=True, progress=False)
    monkeypatch.setenv("CELLREGMAP_AMD_STREAM_CHUNK", "256")
    seen = []
    spv, sinfo, sst = scan(G, return_stats=True, progress=lambda done, total: seen.append((done, total)))
    assert_allclose(spv, pv, rtol=1e-12, atol=0)
    assert_allclose(sst["alt_lml"], st["alt_lml"], rtol=1e-13)
    assert sst["null_lml"] == st["null_lml"] and sst["null_delta"] == st["null_delta"]
    for key in info:
        assert np.array_equal(info[key], sinfo[key])
    assert seen and seen[-1] == (700, 700) and all(t == 700 for _, t in seen)
