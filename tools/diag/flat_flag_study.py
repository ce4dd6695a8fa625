"""What the CRM_MODEL_FLAT_OPTIMUM rule measures against what actually happens between device and oracle (verbatim Brent
on both sides), per variant of a fuzz stream: the decision distance of the fit at rho* (smallest margin of the search's
decisions / noise bound of the objective, include/crm_hip.h), the raw margin and bound, how far Q and p move one stopping
tolerance away, the distance of rho* from the runner-up grid point, and the actual relative differences of Q, p and lml
against the oracle.  Prints, for a ladder of factors kappa on the noise bound, the share of scans a rule `decision <= kappa
and sensitive` would flag and how many scans beyond the north-star tolerances it would leave unflagged; the rows go to
gpurun_out/flat_flag_study_<seed>.npy for calibration off the box.
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
           "xunc", "delta",
           "actual_rel_dQ", "actual_rel_dp", "covariates", "actual_rel_dlml", "same_rho", "flag", "rho_tie_flag", "mode",
           "cells", "problem", "oracle_delta")


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
                         float(xi["flat_optimum"][j]), float(xi["rho_tie"][j]), "ABC".index(case[6]), y.size, idx, ost["delta"][j]))
    a = np.array(rows)
    col = {k: i for i, k in enumerate(COLUMNS)}
    same = a[:, col["same_rho"]] > 0
    bad = same & ((a[:, col["actual_rel_dQ"]] > 1e-6) | (a[:, col["actual_rel_dp"]] > 1e-5))
    sens = (a[:, col["Q_move_one_tol"]] > 5e-7) | (a[:, col["p_move_one_tol"]] > 5e-6) | ~np.isfinite(a[:, col["Q_move_one_tol"]])
    dec = a[:, col["decision"]]
    out = {"scans": int(a.shape[0]), "seed": seed, "problems": count, "same_rho": int(same.sum()), "beyond_north_star": int(bad.sum()),
           "sensitive_to_one_tolerance": float(sens[same].mean()), "beyond_but_not_sensitive": int((bad & ~sens).sum()),
           "flag_as_shipped": {"share": float(a[same, col["flag"]].mean()), "missed": int((bad & (a[:, col["flag"]] == 0)).sum())},
           "rho_differs": int((~same).sum()), "rho_differs_without_tie_flag": int((~same & (a[:, col["rho_tie_flag"]] == 0)).sum()),
           "rho_tie_flag_share": float(a[:, col["rho_tie_flag"]].mean())}
    for kappa in (0.01, 0.02, 0.05, 0.1, 0.2, 0.5, 1.0, 2.0, 5.0):
        und = ~(dec > kappa)
        out["kappa_%g" % kappa] = {"flagged_share": float((sens & und)[same].mean()), "missed": int((bad & ~(sens & und)).sum()),
                                   "undecided_share": float(und[same].mean())}
    rel = a[:, col["margin"]] / np.abs(a[:, col["lml"]])
    for theta in (1e-15, 2e-15, 4e-15, 8e-15, 1.6e-14, 3e-14):
        und = ~(rel > theta)
        out["relative_margin_%g" % theta] = {"flagged_share": float((sens & und)[same].mean()), "missed": int((bad & ~(sens & und)).sum())}
    out["decision_of_the_scans_beyond"] = sorted(float(x) for x in dec[bad])[-12:] if bad.any() else []
    out["decision_percentiles_1_5_25_50_75"] = [float(x) for x in np.nanpercentile(dec[same], [1, 5, 25, 50, 75])]
    out["noise_bound_over_abs_lml_percentiles_5_50_95_100"] = [
        float(x) for x in np.nanpercentile(a[same, col["noise_bound_roundings"]] / np.abs(a[same, col["lml"]]), [5, 50, 95, 100])]
    out["actual_rel_dlml_percentiles_50_95_100"] = [float(x) for x in np.percentile(a[same, col["actual_rel_dlml"]], [50, 95, 100])]
    print(json.dumps(out, indent=1))
    dest = os.path.join(ROOT, "gpurun_out")
    os.makedirs(dest, exist_ok=True)
    tag = "%d%s" % (seed, "_many_contexts" if os.environ.get("CRM_FUZZ_MANY_CONTEXTS") else "")
    np.save(os.path.join(dest, "flat_flag_study_%s.npy" % tag), a)
    with open(os.path.join(dest, "flat_flag_study_%s.json" % tag), "w") as fh:
        json.dump({"columns": COLUMNS, **out}, fh, indent=1)


if __name__ == "__main__":
    main()
