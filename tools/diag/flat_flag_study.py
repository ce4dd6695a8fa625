"""What the reproducibility bounds of ``scan_interaction_info`` (include/crm_hip.h: crm_scan_interaction_bounds) measure
against what actually happens between device and oracle (verbatim Brent on both sides), per variant of a fuzz stream: the
movement of Q and p over one stopping tolerance, the relative gain of the objective over one tolerance at the stopping
point, the decision distance of the search (smallest margin / noise bound), the distance of rho* from the runner-up grid
point -- and the actual differences of Q, p, lml and of the stopping point itself against the oracle.  Prints the shares
flagged, the scans beyond the tolerances that carry no flag, the scans beyond their own bounds, the distribution of
(distance of the stopping points in tolerances) x (relative gain) -- the quantity whose observed maximum the rule's constant
STOP_SHIFT_C covers -- and the trade-off for other constants; the rows go to gpurun_out/flat_flag_study_<seed>.npy.
    [CRM_FUZZ_MANY_CONTEXTS=1] python tools/diag/flat_flag_study.py [count 150] [seed 7] [max_variants] [max_cells]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_cases import build_case, fuzz_cases  # noqa: E402

from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib  # noqa: E402
from oracle.crm import OracleCellRegMap  # noqa: E402

REC = 10   # scan.hip: FLAT_REC
COLUMNS = ("decision", "Q_move_one_tol", "p_move_one_tol", "margin", "noise_bound_roundings", "rho_decision", "rho_gap", "lml",
           "curvature", "delta",
           "actual_rel_dQ", "actual_rel_dp", "covariates", "actual_rel_dlml", "same_rho", "flag", "rho_tie_flag", "mode",
           "cells", "problem", "oracle_delta", "bound_Q", "bound_p", "statistic_flag")


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    limits = {}
    if len(sys.argv) > 3:
        limits["max_variants"] = int(sys.argv[3])
    if len(sys.argv) > 4:
        limits["max_cells"] = int(sys.argv[4])
    if os.environ.get("CRM_FUZZ_MANY_CONTEXTS"):
        limits.update(max_contexts=256, max_rows=288, extra_covariates=(30, 70))
    lib = _lib.load()
    ctx = _engine._context(0)
    rows = []
    for idx, case in enumerate(fuzz_cases(count, seed=seed, wide_covariates=True, **limits)):
        y, E, W, G, kw, hooks = build_case(case)
        crm = CellRegMap(y, E, W=W, **kw)
        try:
            opv, oinfo, ost = OracleCellRegMap(y, E, W=W, **kw).scan_interaction(G, return_stats=True, **hooks)
        except ValueError:
            continue
        panel = GenotypePanel(G, groups=None)
        pv, info, st = crm.scan_interaction(panel, return_stats=True, **hooks)
        _, xi = crm.scan_interaction_info(panel, **hooks)
        rec = np.empty(REC * G.shape[1])
        got = lib.crm_test_null_fit_probe_read(ctx, _lib.ptr(rec), rec.size)
        if got != rec.size:
            continue
        rec = rec.reshape(-1, REC)
        for j in range(G.shape[1]):
            qscale = max(abs(ost["Q"][j]), float(np.trace(ost["F"][j])))
            rows.append((*rec[j], abs(st["Q"][j] - ost["Q"][j]) / qscale, abs(pv[j] - opv[j]) / opv[j], W.shape[1],
                         abs(st["lml"][j] - ost["lml"][j]) / abs(ost["lml"][j]), float(info["rho1"][j] == oinfo["rho1"][j]),
                         float(xi["flat_optimum"][j]), float(xi["rho_tie"][j]), "ABC".index(case[6]), y.size, idx, ost["delta"][j],
                         xi["bound_Q"][j], xi["bound_p"][j], float(xi["statistic_at_tolerance"][j])))
    a = np.array(rows)
    col = {k: i for i, k in enumerate(COLUMNS)}
    same = a[:, col["same_rho"]] > 0
    dec = a[:, col["decision"]]
    dQ, dp = a[:, col["actual_rel_dQ"]], a[:, col["actual_rel_dp"]]
    bq, bp = a[:, col["bound_Q"]], a[:, col["bound_p"]]
    fp, fq = a[:, col["flag"]] != 0, a[:, col["statistic_flag"]] != 0
    d = np.clip(a[:, col["delta"]], 1e-300, 1 - 1e-16)
    od = np.clip(a[:, col["oracle_delta"]], 1e-300, 1 - 1e-16)
    x, ox = np.log(d) - np.log1p(-d), np.log(od) - np.log1p(-od)
    shift = np.abs(x - ox) / (1e-6 * np.abs(x) + 1e-6)          # distance of the two stopping points in tolerances
    gain = a[:, col["curvature"]] / np.abs(a[:, col["lml"]])     # relative gain of the objective over one tolerance
    ok = same & np.isfinite(gain) & (gain > 0)
    out = {"scans": int(a.shape[0]), "seed": seed, "problems": count, "same_rho": int(same.sum()),
           "beyond_1e-6_on_Q": int((same & (dQ > 1e-6)).sum()), "beyond_1e-5_on_p": int((same & (dp > 1e-5)).sum()),
           "share_flat_optimum_p_flag": float(fp[same].mean()), "share_statistic_at_tolerance_flag": float(fq[same].mean()),
           "unflagged_beyond_1e-5_on_p": int((same & ~fp & (dp > 1e-5)).sum()),
           "unflagged_beyond_1e-6_on_Q": int((same & ~fq & (dQ > 1e-6)).sum()),
           "worst_unflagged_rel_dp": float(dp[same & ~fp].max()), "worst_unflagged_rel_dQ": float(dQ[same & ~fq].max()),
           "beyond_own_bound_on_Q": int((same & (dQ > np.maximum(1e-6, 1.001 * bq))).sum()),
           "beyond_own_bound_on_p": int((same & (dp > np.maximum(1e-5, 1.001 * bp + 2e-6))).sum()),
           "shift_in_tolerances_percentiles_50_90_99_99.9_100": [float(v) for v in np.percentile(shift[same], [50, 90, 99, 99.9, 100])],
           "shift_times_relative_gain_percentiles_50_90_99_99.9_100": [float(v) for v in np.percentile((shift * gain)[ok], [50, 90, 99, 99.9, 100])],
           "relative_gain_percentiles_1_5_50_95": [float(v) for v in np.percentile(gain[ok], [1, 5, 50, 95])],
           "Q_move_one_tolerance_percentiles_50_90_99": [float(v) for v in np.nanpercentile(a[same, col["Q_move_one_tol"]], [50, 90, 99])],
           "p_move_one_tolerance_percentiles_50_90_99": [float(v) for v in np.nanpercentile(a[same, col["p_move_one_tol"]], [50, 90, 99])],
           "decision_within_noise_bound_share": float((~(dec > 1.0))[same].mean()),
           "rho_differs": int((~same).sum()), "rho_differs_without_tie_flag": int((~same & (a[:, col["rho_tie_flag"]] == 0)).sum()),
           "rho_tie_flag_share": float(a[:, col["rho_tie_flag"]].mean()),
           "actual_rel_dlml_percentiles_50_95_100": [float(x_) for x_ in np.percentile(a[same, col["actual_rel_dlml"]], [50, 95, 100])]}
    # the trade-off the constant of the rule sits on: share flagged / scans beyond left unflagged, per constant C
    Sq, Sp = a[:, col["Q_move_one_tol"]], a[:, col["p_move_one_tol"]]
    for C in (2e-14, 4e-14, 8e-14, 1.2e-13, 2.5e-13):
        sh = np.where(gain > 0, np.minimum(1.0, C / np.maximum(gain, 1e-300)), 1.0)
        sh = np.where(dec > 1.0, sh, 1.0)
        flag_p, flag_q = ~(Sp * sh <= 1e-5), ~(Sq * sh <= 1e-6)
        out["C_%g" % C] = {"p_flag_share": float(flag_p[same].mean()), "p_missed": int((same & ~flag_p & (dp > 1e-5)).sum()),
                           "Q_flag_share": float(flag_q[same].mean()), "Q_missed": int((same & ~flag_q & (dQ > 1e-6)).sum())}
    print(json.dumps(out, indent=1))
    dest = os.path.join(ROOT, "gpurun_out")
    os.makedirs(dest, exist_ok=True)
    tag = "%d%s" % (seed, "_many_contexts" if os.environ.get("CRM_FUZZ_MANY_CONTEXTS") else "")
    np.save(os.path.join(dest, "flat_flag_study_%s.npy" % tag), a)
    with open(os.path.join(dest, "flat_flag_study_%s.json" % tag), "w") as fh:
        json.dump({"columns": COLUMNS, **out}, fh, indent=1)


if __name__ == "__main__":
    main()
