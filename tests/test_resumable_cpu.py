"""``scan_interaction_resumable``: the chunked scan with a checkpoint file (optional hook), its
bookkeeping on the CPU with the oracle as the per-chunk scan: a run that is killed after some chunks and started again scans
only what is missing and returns what an uninterrupted run returns; a checkpoint of another problem is refused."""
import types

import numpy as np
import pytest

from cellregmap_amd import scan_interaction_resumable
from cellregmap_amd.synth import make_cohort


def _problem():
    from oracle.crm import OracleCellRegMap

    c = make_cohort(6, 12, 3, 23, seed=14)
    o = OracleCellRegMap(c.y, c.E, W=c.W, hK=c.hK)
    crm = types.SimpleNamespace(_y=c.y, _E0=c.E, _W=c.W)     # (what the fingerprint reads of a CellRegMap)
    return c, o, crm


def test_a_killed_run_resumes_where_it_stopped(tmp_path):
    c, o, crm = _problem()
    path = str(tmp_path / "scan.npz")
    calls = []

    class Killed(Exception):
        pass

    def scan(G, idx_E, idx_G, die_after=None):
        calls.append(G.shape[1])
        if die_after is not None and len(calls) > die_after:
            raise Killed()
        return o.scan_interaction(G, idx_E, idx_G)

    want_pv, want_info = o.scan_interaction(c.G)
    with pytest.raises(Killed):
        scan_interaction_resumable(crm, c.G, path, chunk=5, scan=lambda G, a, b: scan(G, a, b, die_after=2))
    assert calls == [5, 5, 5]                                 # two chunks done, the third died
    with np.load(path) as f:
        assert f["done"].tolist() == [True, True, False, False, False]
    calls.clear()
    pv, info = scan_interaction_resumable(crm, c.G, path, chunk=5, scan=scan)
    assert calls == [5, 5, 3]                                 # only the missing chunks (23 = 4 x 5 + 3)
    assert np.array_equal(pv, want_pv)
    for k in want_info:
        assert np.array_equal(info[k], want_info[k])
    calls.clear()
    pv2, _ = scan_interaction_resumable(crm, c.G, path, chunk=5, scan=scan)
    assert calls == [] and np.array_equal(pv2, want_pv)       # a finished checkpoint answers without scanning


def test_a_checkpoint_of_another_problem_is_refused(tmp_path):
    c, o, crm = _problem()
    path = str(tmp_path / "scan.npz")
    scan_interaction_resumable(crm, c.G[:, :6], path, chunk=4, scan=o.scan_interaction)
    with pytest.raises(ValueError, match="another problem"):
        scan_interaction_resumable(crm, c.G[:, :7], path, chunk=4, scan=o.scan_interaction)          # other panel
    with pytest.raises(ValueError, match="another problem"):
        scan_interaction_resumable(crm, c.G[:, :6], path, chunk=3, scan=o.scan_interaction)          # other chunking
    perm = np.random.default_rng(0).permutation(c.y.size)
    with pytest.raises(ValueError, match="another problem"):
        scan_interaction_resumable(crm, c.G[:, :6], path, chunk=4, idx_E=perm, scan=o.scan_interaction)
    other = types.SimpleNamespace(_y=c.y + 1.0, _E0=c.E, _W=c.W)
    with pytest.raises(ValueError, match="another problem"):
        scan_interaction_resumable(other, c.G[:, :6], path, chunk=4, scan=o.scan_interaction)


def test_no_variants(tmp_path):
    c, o, crm = _problem()
    pv, info = scan_interaction_resumable(crm, c.G[:, :0], str(tmp_path / "empty.npz"), scan=o.scan_interaction)
    assert pv.shape == (0,) and info["rho1"].shape == (0,)
