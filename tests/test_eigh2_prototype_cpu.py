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
