"""Seeded random problems shared by the CPU spread test, the GPU fuzz test and tools/fuzz_scan.py
(plain helper module, not a test)."""
import numpy as np


def random_problem(n, k0, c, p, donors, seed, mode):
    """Cohort with ragged donors, normal contexts / covariates, donor-constant genotypes and a
    phenotype with a persistent and a GxC effect; ``mode``: background A / B / C
    (cellregmap/_cellregmap.py:101-131)."""
    rng = np.random.default_rng(seed)
    donor = np.sort(rng.integers(0, donors, size=n))
    donor[:donors] = np.arange(donors)
    donor = np.sort(donor)
    Gd = rng.normal(size=(donors, p))
    G = Gd[donor]
    E = rng.normal(size=(n, k0))
    W = np.concatenate([np.ones((n, 1)), rng.normal(size=(n, c - 1))], axis=1) if c > 1 else np.ones((n, 1))
    hK = np.zeros((n, donors))
    hK[np.arange(n), donor] = 1.0
    y = 0.4 * G[:, 0] + 0.5 * (G[:, 1 % p] * E[:, 0]) + E @ rng.normal(size=k0) * 0.3 + rng.normal(size=n)
    kw = {}
    if mode == "B":
        kw["hK"] = hK
    elif mode == "C":
        kw["Ls"] = [E[:, [i]] * hK for i in range(k0)]
    return y, E, W, G, kw


def fuzz_cases(count, seed=7, max_cells=600, max_contexts=128, max_variants=70, wide_covariates=True, max_rows=144,
               extra_covariates=()):
    """``count`` problem descriptions (i, n, k0, c, p, donors, mode, perm) drawn from one seeded stream:
    30..max_cells cells, 1..max_contexts contexts (1..8 in mode C), 1..14 covariate columns (1..8
    without ``wide_covariates``: the null-fit polish is built for the register kernel only), the three
    background modes, no / context / genotype permutation hook.  ``max_rows`` (contexts + covariates + 2) and
    ``extra_covariates`` (more choices for c) open the stream to the sizes of the slower kernel forms
    (tools/fuzz_scan.py; the defaults keep the streams of the committed records)."""
    rng = np.random.default_rng(seed)
    covs = ([1, 1, 1, 2, 3, 5, 8, 9, 14] if wide_covariates else [1, 1, 1, 2, 3, 5, 8]) + list(extra_covariates)
    out = []
    i = 0
    while len(out) < count:
        mode = "ABC"[int(rng.integers(0, 3))]
        n = int(rng.integers(30, max_cells))
        k0 = int(rng.integers(1, max_contexts + 1)) if mode != "C" else int(rng.integers(1, 9))
        c = int(rng.choice(covs))
        p = int(rng.integers(1, max_variants))
        donors = int(rng.integers(2, 14))
        perm = ["none", "E", "G"][int(rng.integers(0, 3))]
        i += 1
        if k0 + c + 2 > max_rows or n <= c + 2 or donors > n:
            continue
        # Mode A with at least as many contexts as cells: Sigma = E1 E1' has full rank n, Q0 is square and the
        # complement terms (u'v - (Q0'u)'(Q0'v)) / delta of the reference's likelihood are rounding noise
        # divided by delta -- the optimum glimix-core reports there is decided by that noise (the device and
        # the oracle can land in different basins); not a parity case.
        if mode == "A" and k0 >= n:
            continue
        # The same in modes B and C: once the background's columns and the fixed effects [W, g] together span
        # all n cells (cols + c + 1 >= n) the model is saturated -- the residual outside span(Q0) has no degrees
        # of freedom left, the profiled likelihood grows without bound as delta -> 0 and where a finite-precision
        # search stops is decided by rounding (seed 2026 of tools/fuzz_scan.py: n = 43 with 42-rank Q0 and 9
        # fixed effects -- device delta 4e-14 / lml 237, oracle delta 0.92 / lml -60).  Not a parity case either.
        cols = k0 if mode == "A" else (k0 + donors if mode == "B" else k0 + k0 * donors)
        if cols + c + 1 >= n:
            continue
        # Mode B with two donors: the genotype is constant within a donor, so span(1, g) IS the span of the two donor
        # indicators -- at rho = 0 the whole random effect hK hK' is absorbed by the fixed effects and the restricted
        # likelihood does not depend on delta at all (flat to 1e-12 over delta in [1e-9, 0.99]).  Brent then walks
        # to delta ~ 1e-13, where the complement terms' rounding noise divided by delta is O(1) in the likelihood
        # and decides which "optimum" is reported (seed 2026: device -61.2, oracle -66.4 at the same delta).
        if mode == "B" and donors <= 2:
            continue
        out.append((i - 1, n, k0, c, p, donors, mode, perm))
    return out


def build_case(case):
    i, n, k0, c, p, donors, mode, perm = case
    y, E, W, G, kw = random_problem(n, k0, c, p, donors, seed=5000 + i, mode=mode)
    idx = np.random.default_rng(i).permutation(n)
    hooks = {} if perm == "none" else ({"idx_E": idx} if perm == "E" else {"idx_G": idx})
    return y, E, W, G, kw, hooks
