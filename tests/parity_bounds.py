"""How device results are held against the oracle's under the reference's verbatim procedure (plain helper module, not a
test).  The reference stops its null fit at a tolerance of 1e-6 on logit(delta); where exactly a search stops within that
tolerance is decided by rounding noise, and the library reports per variant how far two faithful runs may differ
(``scan_interaction_info``: ``bound_Q`` / ``bound_p``; include/crm_hip.h: crm_scan_interaction_bounds).  A variant is held to
the north-star tolerance -- statistics 1e-6, p-values 1e-5 -- wherever its bound is within it (no flag), and to its own bound
where it is not."""
import numpy as np

Q_TOL, P_TOL, P_ATOL = 1e-6, 1e-5, 1e-13
DAVIES = 2e-6     # two roundings of one (Q, lambda) through Davies' integration to acc = 1e-6 (tests/test_gpu_fuzz.py)


def bounds(crm, panel, sel=None, **hooks):
    """(bound_Q, bound_p, info) of ``crm.scan_interaction_info(panel)``, optionally at the variants ``sel``."""
    xi = crm.scan_interaction_info(panel, **hooks)[1]
    take = slice(None) if sel is None else np.asarray(sel)
    return xi["bound_Q"][take], xi["bound_p"][take], xi


def assert_p_within(pv, opv, bound_p, what=""):
    pv, opv = np.asarray(pv), np.asarray(opv)
    allowed = np.maximum(P_TOL, 1.001 * np.asarray(bound_p) + DAVIES)
    assert np.all(np.abs(pv - opv) <= allowed * np.abs(opv) + P_ATOL), (what, np.c_[pv, opv, np.abs(pv / opv - 1), allowed])


def assert_Q_within(Q, oQ, bound_Q, scale=None, what=""):
    """``scale``: max(|Q|, tr F) per variant (the bounds are relative to it); default |oQ|."""
    Q, oQ = np.asarray(Q), np.asarray(oQ)
    scale = np.abs(oQ) if scale is None else np.asarray(scale)
    allowed = np.maximum(Q_TOL, 1.001 * np.asarray(bound_Q))
    assert np.all(np.abs(Q - oQ) <= allowed * scale), (what, np.c_[Q, oQ, np.abs(Q - oQ) / scale, allowed])


def assert_bounds_are_informative(bound_Q, bound_p, pv=None):
    """The bounds are not a blanket excuse: finite, and of the order of the search's tolerance at most -- the statistic moves
    by a few 1e-6 of its value per tolerance (measured 99th percentile 4.7e-6), a p-value by that times its own steepness
    d ln p / d ln Q, which grows like |ln p| for strongly associated variants."""
    bq, bp = np.asarray(bound_Q), np.asarray(bound_p)
    assert np.all(np.isfinite(bq)) and np.all(np.isfinite(bp))
    assert np.all(bq <= 20 * Q_TOL), bq.max()
    steep = 1.0 if pv is None else np.maximum(1.0, np.abs(np.log(np.maximum(np.asarray(pv), 1e-300))))
    assert np.all(bp <= 20 * P_TOL * steep), (bp / steep).max()
