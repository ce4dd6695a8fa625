"""The build's own generator (cellregmap_amd/synth.py) against the invariants the reference asserts on
its simulator (cellregmap/test/test_simulation.py:197-214: the components' variances sum to one, pairwise correlations
stay small, the phenotype's mean is the offset) and against the reference's construction of the kinship factor
(``_simulate.py:83-102, 477-479``: hK = U sqrt(S) of the jittered donor-block K cut at sqrt(eps))."""
import itertools

import numpy as np
import pytest
from numpy.testing import assert_allclose


@pytest.mark.parametrize("kinship", ["indicator", "rotated"])
def test_moment_invariants_of_the_phenotype(kinship):
    from cellregmap_amd.synth import make_cohort

    parts = {}
    c = make_cohort(50, 40, 6, 40, seed=20, kinship=kinship, components=parts)      # 2 000 cells
    names = ("y_g", "y_gxe", "y_n", "y_e", "y_k")
    assert_allclose(sum(parts[k].var() for k in names), 1.0)
    assert_allclose([parts[k].var() for k in names], [c.variances[k[2:]] for k in names])
    for a, b in itertools.combinations(names, 2):
        assert abs(np.corrcoef(parts[a], parts[b])[0, 1]) < 0.1, (a, b)
    assert_allclose(c.y, parts["offset"] + sum(parts[k] for k in names))
    assert_allclose(c.y.mean(), parts["offset"])
    # contexts and genotypes are column-normalised (_simulate.py:50-54); W is the intercept
    assert_allclose(c.E.mean(0), 0.0, atol=1e-12)
    assert_allclose(c.E.std(0), 1.0)
    assert_allclose(c.G.mean(0), 0.0, atol=1e-12)
    assert_allclose(c.G.std(0), 1.0)
    assert np.all(c.W == 1.0) and c.W.shape == (2000, 1)
    assert np.all((c.mafs >= 0.05) & (c.mafs <= 0.45))


def test_both_kinship_factors_are_factors_of_the_references_K():
    """hK hK' equals the reference's K = Z Z' / mean diag + 1e-8 I on span(Z) (the jitter's n - m directions are what
    ``economic_svd`` cuts off at sqrt(eps)); the rotated factor is dense, its rows constant within a donor, and it spans
    what ``U sqrt(S)`` of the literal decomposition spans, with the same Gram matrix."""
    from cellregmap_amd.synth import kinship_factor
    from oracle.sugar import economic_svd

    donors, cells = 7, 6
    donor_of_cell = np.repeat(np.arange(donors), cells)
    n = donors * cells
    Z = kinship_factor(donor_of_cell, donors, "indicator")
    K = Z @ Z.T
    K /= K.diagonal().mean()
    K += 1e-8 * np.eye(n)
    U, S, _ = economic_svd(K)
    ref = U * np.sqrt(S)                                   # _symmetric_decomp (_simulate.py:477-479)
    assert ref.shape == (n, donors)                        # the jitter's directions (1e-8 < sqrt(eps)) are gone
    rot = kinship_factor(donor_of_cell, donors, "rotated", seed=3)
    assert_allclose(rot @ rot.T, ref @ ref.T, atol=1e-12)  # the same covariance
    assert_allclose(Z @ Z.T, ref @ ref.T, atol=2e-8)       # (the indicator factor leaves the 1e-8 out)
    assert np.count_nonzero(np.abs(rot) > 1e-12) > 0.9 * rot.size      # dense ...
    for d in range(donors):                                             # ... and donor-expanded
        block = rot[donor_of_cell == d]
        assert np.array_equal(block, np.repeat(block[:1], cells, axis=0))
    # same column space as the literal factor
    assert np.linalg.matrix_rank(np.c_[rot, ref], tol=1e-9) == donors


def test_default_cohort_is_unchanged_by_the_new_options():
    """The committed goldens (tests/golden/*.npz) were generated with the indicator factor: same stream, same numbers."""
    import os

    from cellregmap_amd.synth import make_cohort
    from conftest import GOLDEN

    gold = np.load(os.path.join(GOLDEN, "e2e_golden.npz"))
    c = make_cohort(10, 20, 5, 24, seed=5)
    assert np.array_equal(c.y, gold["A/y"]) and np.array_equal(c.G, gold["A/G"]) and np.array_equal(c.hK, gold["A/hK"])
