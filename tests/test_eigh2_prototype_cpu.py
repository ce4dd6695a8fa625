"""tools/eigh2_prototype.py (the numpy statement of the constructor's two-stage eigen-solver, which the HIP kernels
of eigh2_band.hip / eigh2_chase.hip follow) against LAPACK: the family argument (one band for every grid point), the
chase, the regrouped back-transformation."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import eigh2_prototype as proto  # noqa: E402


@pytest.mark.parametrize("n,k1,w,g", [(61, 5, 8, 8), (97, 7, 8, 5), (130, 16, 16, 16), (50, 3, 4, 4), (40, 0, 8, 8)])
def test_family_solver_against_lapack(n, k1, w, g):
    rng = np.random.default_rng(n)
    H = rng.normal(size=(n + 30, n))
    H[:, -3:] = H[:, :3]                      # a rank-deficient Gram matrix, like a background's
    C = H.T @ H
    rhos = [0.0, 0.3, 0.9]
    for rho, (lam, Z) in zip(rhos, proto.eigh_family(C, k1, rhos, w=w, g=g)):
        dscale = np.r_[np.full(k1, np.sqrt(rho)), np.full(n - k1, np.sqrt(1.0 - rho))]
        A = C * np.outer(dscale, dscale)
        ref = np.linalg.eigvalsh(A)
        scale = np.abs(ref).max()
        assert np.abs(lam - ref).max() <= 1e-13 * scale
        assert np.abs(Z.T @ Z - np.eye(n)).max() <= 1e-11
        assert np.abs(A @ Z - Z * lam).max() <= 1e-13 * scale * n


def test_one_band_serves_every_grid_point():
    """Q1' (D C D) Q1 = D (Q1' C Q1) D when the first panel is the leading block: the band of the scaled matrix is the
    scaled band."""
    rng = np.random.default_rng(5)
    n, k1, w = 70, 6, 8
    H = rng.normal(size=(n + 10, n))
    C = H.T @ H
    band, _ = proto.stage1(C, k1, w)
    assert np.abs(np.tril(band, -w - 1)).max() == 0.0
    for rho in (0.2, 0.7):
        d = np.r_[np.full(k1, np.sqrt(rho)), np.full(n - k1, np.sqrt(1 - rho))]
        scaled_then_reduced, _ = proto.stage1(C * np.outer(d, d), k1, w)
        assert np.abs(scaled_then_reduced - band * np.outer(d, d)).max() <= 1e-12 * np.abs(band).max()


def test_back_transformation_tasks_cover_every_tile_once():
    """eigh2_back.hip: back_tasks -- whole rounds of eight-tile workgroups, the rest as four-tile workgroups when one round of
    those takes it.  Whatever the sizes: every tile of sixteen eigenvectors of every matrix exactly once, at most eight per
    workgroup (host-only hook, no GPU)."""
    import ctypes

    from cellregmap_amd import _lib

    lib = _lib.load()
    for batch, dim, cus in [(10, 5064, 256), (10, 10064, 256), (10, 1064, 256), (1, 1100, 256), (3, 2400, 256), (11, 1024, 256),
                            (10, 16448, 256), (2, 3, 256), (1, 16, 4), (7, 1999, 64), (10, 5064, 304), (1, 17, 1)]:
        cap = 3 * (batch * ((dim + 15) // 16) + 8)
        tasks = np.zeros(cap, np.int32)
        count = ctypes.c_int()
        _lib.check(lib.crm_test_back_tasks(batch, dim, cus, _lib.ptr(tasks), cap, ctypes.byref(count)))
        t = tasks[: 3 * count.value].reshape(-1, 3)
        nt = (dim + 15) // 16
        seen = np.zeros((batch, nt), int)
        for b, t0, n in t:
            assert 1 <= n <= 8 and 0 <= b < batch and t0 + n <= nt
            seen[b, t0:t0 + n] += 1
        assert (seen == 1).all(), (batch, dim, cus)
        if (batch, dim, cus) == (10, 10064, 256):     # three whole rounds of eight tiles, then 44 short workgroups
            assert len(t) == 812 and (t[:768, 2] == 8).all() and (t[768:, 2] <= 4).all()
        if (batch, dim, cus) == (10, 5064, 256):      # 290 short workgroups would not fit one round: eight tiles throughout
            assert len(t) == 400 and (t[:, 2] == 8).sum() == 390
    # too small a buffer is refused with the count reported
    count = ctypes.c_int()
    small = np.zeros(3, np.int32)
    assert lib.crm_test_back_tasks(10, 5064, 256, _lib.ptr(small), 3, ctypes.byref(count)) != 0 and count.value == 400
