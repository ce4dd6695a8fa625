"""oracle.crm effect-size functions (cellregmap/_cellregmap.py:137-244, :589-682) against dense
algebra (parity unpinned by the reference: glimix-core is not importable here)."""
import numpy as np
from numpy.testing import assert_allclose

from cellregmap_amd.synth import make_cohort
from oracle import crm as ocrm
from oracle.lmm import LMM
from oracle.sugar import economic_qs_linear


def test_compute_maf_doctest_vector():
    # the docstring example of the reference (_cellregmap.py:600-609)
    X = np.random.RandomState(0).randint(0, 3, size=(100, 10))
    assert_allclose(ocrm.compute_maf(X), [0.49, 0.49, 0.445, 0.495, 0.5, 0.45, 0.48, 0.48, 0.47, 0.435])


def test_predict_interaction_is_the_dense_blup():
    c = make_cohort(6, 12, 3, 3, seed=4)
    maf = np.array([0.2, 0.3, 0.4])
    Ls = ocrm.khatri_rao_halves(c.hK, c.E)
    o = ocrm.OracleCellRegMap(c.y, c.E, W=c.W, Ls=Ls)
    bg, bgxe = o.predict_interaction(c.G, maf)
    assert bg.shape == (3,) and bgxe.shape == (1, c.y.size, 3)
    n = c.y.size
    for i in range(3):
        g = c.G[:, [i]]
        gE = g * c.E
        M = np.concatenate((c.W, g, c.E), axis=1)
        # replay the grid search, then the dense formulas at the kept fit
        best = None
        for rho in o._rho:
            hS = np.concatenate([np.sqrt(rho) * gE] + [np.sqrt(1 - rho) * L for L in Ls], axis=1)
            lmm = LMM(o._y, M, economic_qs_linear(hS, return_q1=False), restricted=True)
            lmm.fit(verbose=False)
            if best is None or lmm.lml() > best[0]:
                best = (lmm.lml(), rho, lmm.v0, lmm.v1, hS)
        _, rho, v0, v1, hS = best
        K = v0 * hS @ hS.T + v1 * np.eye(n)
        Ki = np.linalg.inv(K)
        beta = np.linalg.solve(M.T @ Ki @ M, M.T @ Ki @ c.y)
        assert_allclose(bg[i], beta[c.W.shape[1]], rtol=1e-7, atol=1e-10)
        # BLUP of the GxC effects: cov(beta_gxe, y) K^-1 (y - M beta), cov = v0 rho E0 (g o E0)'
        blup = v0 * rho * c.E @ (gE.T @ (Ki @ (c.y - M @ beta))) / np.sqrt(2 * maf[i] * (1 - maf[i]))
        assert_allclose(bgxe[0, :, i], blup, rtol=1e-6, atol=1e-9)


def test_compute_maf_keeps_the_container_like_the_reference():
    """_cellregmap.py:620-638: a DataFrame gives a Series named "maf"; missing calls (NaN) leave the denominator."""
    import pandas as pd

    from cellregmap_amd import compute_maf

    X = np.random.RandomState(0).randint(0, 3, size=(100, 10)).astype(float)
    X[3, 2] = np.nan
    X[7, 2] = np.nan
    df = pd.DataFrame(X, columns=[f"snp{i}" for i in range(10)])
    maf = compute_maf(df)
    assert isinstance(maf, pd.Series) and maf.name == "maf" and list(maf.index) == list(df.columns)
    want = np.nansum(X, axis=0) / (2 * (~np.isnan(X)).sum(axis=0))
    want = np.minimum(want, 1 - want)
    np.testing.assert_allclose(maf.values, want, rtol=0, atol=1e-15)
    np.testing.assert_allclose(compute_maf(X), want, rtol=0, atol=1e-15)
    assert isinstance(compute_maf(X), np.ndarray)
