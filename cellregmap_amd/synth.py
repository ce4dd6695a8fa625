"""Synthetic single-cell cohort generator for tests and benchmarks.

Own counterpart of the reference's simulator (cellregmap/_simulate.py:315-397,
not exported by the reference package): donors -> cells expansion, HWE
genotypes, column-normalised contexts, donor-block kinship factor, phenotype as
a sum of moment-normalised variance components.  Sizes follow
``BASELINE.json.configs``; SURVEY.md section 8(d) fixes the recipe.
"""
from collections import namedtuple

import numpy as np

Cohort = namedtuple("Cohort", "y W E G hK donor_of_cell mafs variances")

CONFIGS = {
    # name: (donors, cells_per_donor, contexts, variants)
    "cfg1": (50, 10, 10, 200),
    "cfg2": (50, 100, 20, 10_000),
    "cfg3": (100, 200, 50, 50_000),
    "cfg5": (200, 500, 50, 10_000),
}


def column_normalize(X):
    X = np.asarray(X, float)
    return (X - X.mean(0)) / X.std(0)


def _moments(v, variance):
    v = v - v.mean()
    return v / v.std() * np.sqrt(variance)


def variances(r0=0.5, v0=0.5):
    """create_variances(r0, v0) of the reference (_simulate.py:104-158)."""
    g, gxe = v0 * (1 - r0), v0 * r0
    rest = (1 - g - gxe) / 3
    return {"g": g, "gxe": gxe, "e": rest, "k": rest, "n": rest}


def donor_genotypes(n_donors, n_variants, rng, maf_min=0.05, maf_max=0.45):
    mafs = rng.random(n_variants) * (maf_max - maf_min) + maf_min
    u = rng.random((n_donors, n_variants))
    p0 = (1 - mafs) ** 2
    p1 = 1 - mafs ** 2
    G = (u > p0).astype(np.int8) + (u > p1).astype(np.int8)
    # resample monomorphic columns (their normalisation would divide by zero)
    bad = np.flatnonzero(G.std(0) == 0)
    while bad.size:
        u = rng.random((n_donors, bad.size))
        G[:, bad] = (u > p0[bad]).astype(np.int8) + (u > p1[bad]).astype(np.int8)
        bad = bad[G[:, bad].std(0) == 0]
    return G, mafs


def kinship_factor(donor_of_cell, n_donors, kind="indicator", seed=20):
    """The kinship factor hK with hK hK' = K, K the donor-block relatedness matrix Z Z' / mean diag + 1e-8 I of the
    reference's simulator (``sample_covariance_matrix`` + ``jitter``, _simulate.py:83-102) cut at sqrt(eps) like its
    ``_symmetric_decomp`` (:477-479: ``economic_svd`` drops the jitter's n - m directions).

    ``indicator``: hK = Z (donor indicators; Z Z' has unit diagonal) -- the sparse factor of that K.
    ``rotated``: hK = U sqrt(S) as the reference forms it.  The m kept eigenvalues of K are the donors' cell counts
    (+ 1e-8); for equal counts the eigenspace is degenerate and LAPACK returns some rotation of Z / sqrt(n_d) inside
    it, so U sqrt(S) = Z D R with D = diag(sqrt(n_d + 1e-8) / sqrt(n_d)) and R orthogonal: a DENSE n x m factor whose
    rows are constant within a donor.  R is drawn here from a seeded stream (the n x n decomposition itself is out of
    reach at 20 000 cells; tests/test_synth_cpu.py checks this form against ``economic_svd`` of K at small n)."""
    n = donor_of_cell.shape[0]
    Z = np.zeros((n, n_donors))
    Z[np.arange(n), donor_of_cell] = 1.0
    if kind == "indicator":
        return Z
    if kind != "rotated":
        raise ValueError(f"kinship factor {kind!r}: 'indicator' or 'rotated'")
    counts = np.bincount(donor_of_cell, minlength=n_donors).astype(float)
    R, _ = np.linalg.qr(np.random.default_rng(seed + 977).normal(size=(n_donors, n_donors)))
    return Z @ (np.sqrt((counts + 1e-8) / counts)[:, None] * R)


def make_cohort(n_donors, cells_per_donor, n_contexts, n_variants, seed=20,
                g_causals=(5, 6), gxe_causals=(10, 11), with_phenotype=True, dtype=np.float64, kinship="indicator",
                components=None, columns=None):
    """``kinship``: see ``kinship_factor``.  ``components``: a dict that receives the phenotype's moment-normalised
    parts (offset, y_g, y_gxe, y_k, y_e, y_n), as the reference's ``Simulation`` tuple exposes them.
    ``columns = (first, count)`` (only with ``with_phenotype=False``): ``G`` holds just those columns of the panel --
    the donor-level draw (donors x variants, small) is made whole so that every column is what the full panel would
    hold, only the expansion to cells is restricted (a rank of a multi-GPU job never forms the n x p matrix)."""
    rng = np.random.default_rng(seed)
    n = n_donors * cells_per_donor
    Gd, mafs = donor_genotypes(n_donors, n_variants, rng)
    donor_of_cell = np.repeat(np.arange(n_donors), cells_per_donor)
    # normalise at donor level == normalising the expanded matrix (equal group sizes)
    Gd = column_normalize(Gd)
    if columns is not None:
        if with_phenotype:
            raise ValueError("columns=...: only for panels without a phenotype (the causal variants are columns of G)")
        first, count = columns
        G = np.ascontiguousarray(Gd[:, first:first + count][donor_of_cell, :], dtype=dtype)
    else:
        G = np.ascontiguousarray(Gd[donor_of_cell, :], dtype=dtype)
    E = column_normalize(rng.normal(size=(n, n_contexts)))
    W = np.ones((n, 1))
    # donor-block kinship: K = Z Z' / mean diag + 1e-8 I ;  hK = U sqrt(S) (rank n_donors)
    hK = kinship_factor(donor_of_cell, n_donors, kinship, seed)
    var = variances()
    if not with_phenotype:
        return Cohort(None, W, E, G, hK, donor_of_cell, mafs, var)
    offset = 0.3
    parts = {"y_g": np.zeros(n), "y_gxe": np.zeros(n)}
    g_causals = [c for c in g_causals if c < n_variants]
    gxe_causals = [c for c in gxe_causals if c < n_variants]
    if g_causals:
        beta = rng.choice([1.0, -1.0], size=len(g_causals)) * np.sqrt(var["g"] / len(g_causals))
        parts["y_g"] = _moments(G[:, g_causals] @ beta, var["g"])
    if gxe_causals:
        ygxe = np.zeros(n)
        for c in gxe_causals:
            alpha = rng.normal(size=n_contexts) * np.sqrt(var["gxe"] / len(gxe_causals))
            ygxe += G[:, c] * (E @ alpha)
        parts["y_gxe"] = _moments(ygxe, var["gxe"])
    # population-structure x context term: sum_i diag(E[:, i]) hK u_i
    yk = np.zeros(n)
    for i in range(n_contexts):
        yk += E[:, i] * (hK @ rng.normal(size=n_donors))
    parts["y_k"] = _moments(yk, var["k"])
    parts["y_e"] = _moments(E @ rng.normal(size=n_contexts), var["e"])
    parts["y_n"] = _moments(rng.normal(size=n), var["n"])
    # (the same order of additions as before the parts were kept: goldens built on y stay bit-identical)
    y = np.full(n, offset)
    for key in ("y_g", "y_gxe", "y_k", "y_e", "y_n"):
        if key in ("y_g", "y_gxe") and not (g_causals if key == "y_g" else gxe_causals):
            continue
        y += parts[key]
    if components is not None:
        components.update(parts, offset=offset)
    return Cohort(y, W, E, G, hK, donor_of_cell, mafs, var)


def make_config(name, seed=20, n_variants=None, **kw):
    d, c, k, p = CONFIGS[name]
    return make_cohort(d, c, k, p if n_variants is None else n_variants, seed=seed, **kw)
