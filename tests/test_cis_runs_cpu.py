"""Host logic of the cis-window pass of scan_interaction_many: the variant axis is cut into maximal runs with a
constant set of active phenotypes (no GPU; the scans themselves are in test_gpu_interaction.py)."""
import numpy as np
import pytest

from cellregmap_amd._engine import _cis_runs


def _brute(columns, ngenes, p):
    active = np.zeros((ngenes, p), bool)
    for i, c in enumerate(columns):
        active[i, c] = True
    return active


def _check(cis, ngenes, p, **kw):
    columns, runs = _cis_runs(cis, ngenes, p, **kw)
    active = _brute(columns, ngenes, p)
    covered = np.zeros(p, bool)
    prev_end, prev_set = -1, None
    for first, count, genes in runs:
        assert count > 0 and first >= max(prev_end, 0)
        assert not covered[first:first + count].any()
        covered[first:first + count] = True
        for v in range(first, first + count):
            assert np.array_equal(np.flatnonzero(active[:, v]), genes)
        if first == prev_end:                        # maximal: adjacent runs differ in their phenotypes
            assert not np.array_equal(prev_set, genes)
        prev_end, prev_set = first + count, genes
    assert np.array_equal(covered, active.any(axis=0))
    return columns, runs


@pytest.mark.parametrize("dense_limit", [1 << 26, 0])
def test_ranges_slices_masks_and_index_arrays(dense_limit):
    p = 40
    mask = np.zeros(p, bool)
    mask[[3, 4, 5, 20, 39]] = True
    cis = [(0, 10), slice(5, 25), mask, np.array([7, 7, 2, -1, 30]), np.array([], dtype=int), (12, 12)]
    columns, runs = _check(cis, len(cis), p, dense_limit=dense_limit)
    assert np.array_equal(columns[3], [7, 7, 2, 39, 30])      # order and repeats of the caller are kept
    assert columns[4].size == 0 and columns[5].size == 0
    assert runs[0][0] == 0 and np.array_equal(runs[0][2], [0])


@pytest.mark.parametrize("seed", range(5))
def test_random_windows_both_routes_agree(seed):
    rng = np.random.default_rng(seed)
    p, ng = 300, 12
    cis = []
    for i in range(ng):
        a = int(rng.integers(0, p))
        cis.append((a, min(p, a + int(rng.integers(0, 120)))) if i % 3 else rng.integers(0, p, size=int(rng.integers(0, 40))))
    _, dense = _check(cis, ng, p)
    _, sweep = _check(cis, ng, p, dense_limit=0)
    assert len(dense) == len(sweep)
    for (a, c, g), (a2, c2, g2) in zip(dense, sweep):
        assert (a, c) == (a2, c2) and np.array_equal(g, g2)


def test_bad_windows_are_rejected():
    with pytest.raises(ValueError):
        _cis_runs([(0, 5)], 2, 10)
    with pytest.raises(ValueError):
        _cis_runs([(3, 11)], 1, 10)
    with pytest.raises(ValueError):
        _cis_runs([np.array([10])], 1, 10)
    with pytest.raises(ValueError):
        _cis_runs([np.ones(9, bool)], 1, 10)
