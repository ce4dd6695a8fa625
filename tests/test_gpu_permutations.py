"""The batched permutation driver (second half): ``scan_interaction_permutations`` /
``crm_scan_interaction_permuted`` against B separate ``scan_interaction(G, idx_E=perm)`` calls -- the loop the reference's
calibration test runs (cellregmap/test/test_struct_lmm2.py:208-211) around the hooks at cellregmap/_cellregmap.py:398-413 --
bit for bit, against the oracle on a sample, and the reference test's calibration thresholds on its output."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from cellregmap_amd.synth import make_cohort  # noqa: E402


def _objects(mode, c, crm, ocrm):
    if mode == "A":
        return {}, {}
    if mode == "B":
        return dict(hK=c.hK), dict(hK=c.hK)
    return dict(Ls=crm.get_L_values(c.hK, c.E)), dict(Ls=ocrm.khatri_rao_halves(c.hK, c.E))


@pytest.mark.parametrize("mode", ["A", "B", "C"])
@pytest.mark.parametrize("hook", ["E", "G", "both"])
def test_permutations_in_one_call_equal_separate_calls_and_the_oracle(mode, hook):
    import cellregmap_amd as crm
    from oracle import crm as ocrm

    c = make_cohort(10, 18, 4, 29, seed=53)
    n = c.y.size
    kw, okw = _objects(mode, c, crm, ocrm)
    obj = crm.CellRegMap(c.y, c.E, W=c.W, **kw)
    rng = np.random.default_rng(8)
    B = 5
    perms = [rng.permutation(n) for _ in range(B)]
    perms2 = [rng.permutation(n) for _ in range(B)]
    perms[2] = None            # (an identity entry inside a list)
    lists = {"E": dict(idx_E_list=perms), "G": dict(idx_G_list=perms), "both": dict(idx_E_list=perms, idx_G_list=perms2)}[hook]
    for groups in (None, "auto"):       # dense and donor-collapsed passes
        panel = crm.GenotypePanel(c.G, groups=groups)
        pv, info, Q = obj.scan_interaction_permutations(panel, return_Q=True, **lists)
        assert pv.shape == (B, c.G.shape[1]) and Q.shape == pv.shape
        for b in range(B):
            # (a None inside a list stands for the identity: the call with the identity spelled out)
            pb = np.arange(n) if perms[b] is None else perms[b]
            one = {"E": dict(idx_E=pb), "G": dict(idx_G=pb), "both": dict(idx_E=pb, idx_G=perms2[b])}[hook]
            pv1, info1, st1 = obj.scan_interaction(panel, return_stats=True, **one)
            assert np.array_equal(pv[b], pv1), (groups, b, np.abs(pv[b] / pv1 - 1).max())
            assert np.array_equal(Q[b], st1["Q"])
            for k in info1:
                assert np.array_equal(info[k], info1[k]), k
    # the oracle on two of the permutations (north-star tolerances)
    o = ocrm.OracleCellRegMap(c.y, c.E, W=c.W, **okw)
    for b in (0, 2, B - 1):
        one = {"E": dict(idx_E=perms[b]), "G": dict(idx_G=perms[b]), "both": dict(idx_E=perms[b], idx_G=perms2[b])}[hook]
        opv, oinfo = o.scan_interaction(c.G, **one)
        assert np.array_equal(info["rho1"], oinfo["rho1"])
        assert np.all(np.abs(pv[b] - opv) <= 1e-5 * opv + 1e-13), (b, np.c_[pv[b], opv])


def test_permuted_contexts_are_calibrated_like_the_reference_test_asks():
    """cellregmap/test/test_struct_lmm2.py:190-211: a phenotype WITH GxE effects scanned against row-permuted contexts
    looks null -- median p > 0.3 and min p > 0.04 over its 20 variants (one seeded permutation there; sixteen here, in one
    call: both bounds in the shares a uniform sample gives -- the median of 20 uniform p-values is below 0.3 four times in a
    hundred, P(min of 20 > 0.04) = 0.44 -- and the pooled median near one half)."""
    import cellregmap_amd as crm

    c = make_cohort(50, 10, 3, 20, seed=2)      # 500 cells, 20 variants, causal GxE variants inside (synth.py)
    obj = crm.CellRegMap(c.y, c.E, W=c.W, hK=c.hK)
    rng = np.random.default_rng(1)
    perms = [rng.permutation(c.y.size) for _ in range(16)]
    pv, info = obj.scan_interaction_permutations(c.G, idx_E_list=perms)
    plain, _ = obj.scan_interaction(c.G)
    assert plain.min() < 1e-3                    # (the unpermuted scan does see the GxE variants)
    med = np.median(pv, axis=1)
    assert np.mean(med > 0.3) >= 0.8, med
    assert 0.35 < np.median(pv) < 0.65, np.median(pv)
    assert np.mean(pv.min(axis=1) > 0.04) >= 0.2, pv.min(axis=1)
    assert pv.min() > 1e-4


def test_permutation_lists_are_checked():
    import cellregmap_amd as crm
    from cellregmap_amd._lib import CrmError

    c = make_cohort(6, 10, 2, 5, seed=1)
    obj = crm.CellRegMap(c.y, c.E, W=c.W)
    with pytest.raises(ValueError):
        obj.scan_interaction_permutations(c.G)
    with pytest.raises(ValueError):
        obj.scan_interaction_permutations(c.G, idx_E_list=[np.arange(60)], idx_G_list=[np.arange(60), np.arange(60)])
    with pytest.raises(ValueError):
        obj.scan_interaction_permutations(c.G, idx_E_list=[np.arange(7)])
    with pytest.raises(CrmError):
        obj.scan_interaction_permutations(c.G, idx_E_list=[np.full(60, 99)])
    # a scan after a refused one finds the context in order
    pv, _ = obj.scan_interaction_permutations(c.G, idx_E_list=[np.arange(60)[::-1].copy()])
    pv1, _ = obj.scan_interaction(c.G, idx_E=np.arange(60)[::-1].copy())
    assert np.array_equal(pv[0], pv1)
