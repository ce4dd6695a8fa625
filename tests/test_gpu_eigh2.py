"""The constructor's two-stage eigen-solver (eigh2_band.hip, eigh2_chase.hip) on the device, phase by phase against
its numpy statement (tools/eigh2_prototype.py: same Householder conventions, so the band and the tridiagonals agree
entry by entry) and end to end against LAPACK."""
import ctypes
import os
import sys

import numpy as np
import pytest

from cellregmap_amd import _engine, _lib

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import eigh2_prototype as proto  # noqa: E402

pytestmark = pytest.mark.gpu


def _family(dim, k1, seed, deficient=0):
    """A Gram matrix whose first 64 coordinates are the leading block: 64 - k1 zero rows, then k1 'contexts'."""
    rng = np.random.default_rng(seed)
    H = rng.normal(size=(dim + 40, dim))
    H[:, : 64 - k1] = 0.0
    if deficient:
        H[:, -deficient:] = H[:, 64:64 + deficient]
    return H.T @ H


def _call(C, wa, wb, stage):
    lib = _lib.load()
    ctx = _engine._context(0)
    nq, dim = len(wa), C.shape[0]
    wa, wb = np.asarray(wa, float), np.asarray(wb, float)
    lam = np.empty((nq, dim))
    Z = np.empty((nq, dim, dim)) if stage == 0 else None
    d, e, band = np.empty((nq, dim)), np.empty((nq, dim)), np.empty((dim, dim))
    _lib.check(lib.crm_test_eigh2(ctx, nq, dim, _lib.ptr(np.ascontiguousarray(C)), _lib.ptr(wa), _lib.ptr(wb), _lib.ptr(lam),
                                  _lib.ptr(Z), stage, _lib.ptr(d), _lib.ptr(e), _lib.ptr(band)))
    return lam, Z, d, e, band


@pytest.mark.parametrize("dim,k1", [(200, 10), (333, 64), (1100, 50)])
def test_dense_to_band_matches_the_prototype(dim, k1):
    C = _family(dim, k1, seed=dim)
    _, _, _, _, band = _call(C, [1.0], [1.0], 1)
    ref, _ = proto.stage1(C, 64, 64)
    low = np.tril(band)
    assert np.abs(np.tril(low, -65)).max() == 0.0
    assert np.abs(low - np.tril(ref)).max() <= 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize("dim,k1", [(200, 10), (333, 64), (700, 50)])
def test_chase_matches_the_prototype(dim, k1):
    C = _family(dim, k1, seed=dim + 1)
    rho = [0.0, 0.4, 0.9]
    wa, wb = np.sqrt(rho), np.sqrt(1 - np.asarray(rho))
    _, _, d, e, _ = _call(C, wa, wb, 2)
    band, _ = proto.stage1(C, 64, 64)
    for q in range(len(rho)):
        dsc = np.r_[np.full(64, wa[q]), np.full(dim - 64, wb[q])]
        A = band * np.outer(dsc, dsc)
        dr, er, _ = proto.chase(A, 64)
        scale = np.abs(dr).max()
        assert np.abs(d[q] - dr).max() <= 1e-10 * scale
        assert np.abs(e[q, : dim - 1] - er).max() <= 1e-10 * scale
        # ... and whatever the conventions: the tridiagonal has the spectrum of the scaled matrix
        from scipy.linalg import eigvalsh_tridiagonal

        ref = np.linalg.eigvalsh(C * np.outer(dsc, dsc))
        assert np.abs(eigvalsh_tridiagonal(d[q], e[q, : dim - 1]) - ref).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("dim,k1,deficient", [(200, 10, 0), (333, 64, 5), (1100, 50, 20), (1600, 20, 0), (2400, 50, 7)])
def test_family_solver_against_lapack(dim, k1, deficient):
    """(2 400: the back-transformation through stage 1 in blocks of 256 reflectors, with a trailing block of 128.)"""
    C = _family(dim, k1, seed=dim + 2, deficient=deficient)
    rho = [0.0, 0.1, 0.5, 0.9]
    wa, wb = np.sqrt(rho), np.sqrt(1 - np.asarray(rho))
    lam, Z, _, _, _ = _call(C, wa, wb, 0)
    for q in range(len(rho)):
        dsc = np.r_[np.full(64, wa[q]), np.full(dim - 64, wb[q])]
        A = C * np.outer(dsc, dsc)
        ref = np.linalg.eigvalsh(A)
        scale = np.abs(ref).max()
        assert np.abs(lam[q] - ref).max() <= 2e-13 * scale
        assert np.abs(Z[q].T @ Z[q] - np.eye(dim)).max() <= 5e-12
        assert np.abs(A @ Z[q] - Z[q] * lam[q]).max() <= 1e-12 * scale


def test_constructor_forms_agree_at_config2(kernel_form):
    """BASELINE config 2's background through the two-stage family solver (the default from order 1024 on: 1 020 columns
    + 44 of padding) and with every grid point tridiagonalised on its own (the form "eigh_one_stage"): the same ranks and
    spectra, and scans that agree to the polished null fit's accuracy.  Also flips "nullfit_exact" on the way."""
    from cellregmap_amd import CellRegMap, get_L_values
    from cellregmap_amd.synth import make_config

    c = make_config("cfg2", n_variants=96)
    Ls = get_L_values(c.hK, c.E)
    lib = _lib.load()
    ctx = _engine._context(0)
    _lib.check(lib.crm_set_null_fit_polish(ctx, 1))
    try:
        res = {}
        for form in ("two-stage", "one-stage"):
            _engine._bg_cache.clear()
            kernel_form("eigh_one_stage", 1 if form == "one-stage" else 0)
            crm = CellRegMap(c.y, c.E, W=c.W, Ls=Ls)
            ranks = [crm._bg.rank(i) for i in range(len(crm._rho1))]
            S0 = crm._bg.read(4, c.y.size)[1]
            pv, info = crm.scan_interaction(c.G)
            res[form] = (ranks, S0, pv, info["rho1"])
            del crm
        kernel_form("eigh_one_stage", 0, reset=True)
        a, b = res["two-stage"], res["one-stage"]
        assert a[0] == b[0]
        assert np.abs(a[1] - b[1]).max() <= 1e-12 * a[1].max()
        assert np.array_equal(a[3], b[3])
        assert np.all(np.abs(a[2] - b[2]) <= 1e-8 * b[2])
        # the likelihood in the reference's own operations (IEEE division, one log per entry): same answers
        _engine._bg_cache.clear()
        kernel_form("nullfit_exact", 1)
        pv_exact, _ = CellRegMap(c.y, c.E, W=c.W, Ls=Ls).scan_interaction(c.G)
        assert np.all(np.abs(pv_exact - a[2]) <= 1e-8 * a[2])
    finally:
        _lib.check(lib.crm_set_null_fit_polish(ctx, 0))
        _engine._bg_cache.clear()


def test_a_chase_that_gives_up_falls_back_to_the_one_stage_solver(kernel_form):
    """The two-stage solver's bulge chase runs its sweeps as a pipeline of co-resident workgroups; on a device shared with
    another process a workgroup's predecessor may never be scheduled, and the wait -- bounded by wall clock (2 s,
    eigh.h: E2_WAIT_TICKS) -- raises the abort flag.  The form "chase_abort" leaves the chase no patience at all: the
    constructor must notice, fall back to the one-stage solver and deliver the spectra the one-stage form delivers."""
    from cellregmap_amd import CellRegMap, get_L_values
    from cellregmap_amd.synth import make_config

    c = make_config("cfg2", n_variants=8)
    Ls = get_L_values(c.hK, c.E)
    got = {}
    for form in ("eigh_one_stage", "chase_abort"):
        _engine._bg_cache.clear()
        kernel_form(form, 1)
        crm = CellRegMap(c.y, c.E, W=c.W, Ls=Ls)
        got[form] = ([crm._bg.rank(i) for i in range(11)], crm._bg.read(4, c.y.size)[1], crm._bg.read(9, c.y.size)[1])
        kernel_form(form, 0, reset=True)
        del crm
    _engine._bg_cache.clear()
    a, b = got["eigh_one_stage"], got["chase_abort"]
    assert a[0] == b[0]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])    # (the same solver ran: the same bits)
