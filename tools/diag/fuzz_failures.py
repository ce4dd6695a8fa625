"""Which problems of a fuzz stream disagree with the oracle, and how (verbatim procedure):
    [CRM_FUZZ_MANY_CONTEXTS=1] python tools/diag/fuzz_failures.py count seed [max_variants] [max_cells]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_cases import build_case, fuzz_cases  # noqa: E402

from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib  # noqa: E402
from oracle.crm import OracleCellRegMap  # noqa: E402

count, seed = int(sys.argv[1]), int(sys.argv[2])
limits = {}
if len(sys.argv) > 3:
    limits["max_variants"] = int(sys.argv[3])
if len(sys.argv) > 4:
    limits["max_cells"] = int(sys.argv[4])
if os.environ.get("CRM_FUZZ_MANY_CONTEXTS"):
    limits.update(max_contexts=256, max_rows=288, extra_covariates=(30, 70))
for idx, case in enumerate(fuzz_cases(count, seed=seed, wide_covariates=True, **limits)):
    y, E, W, G, kw, hooks = build_case(case)
    crm = CellRegMap(y, E, W=W, **kw)
    try:
        opv, oinfo, ost = OracleCellRegMap(y, E, W=W, **kw).scan_interaction(G, return_stats=True, **hooks)
    except ValueError:
        continue
    xi = crm.scan_interaction_info(GenotypePanel(G, groups=None), **hooks)[1]
    flat, loose = xi["flat_optimum"], xi["statistic_at_tolerance"]      # (the p-value's flag, the statistic's)
    rec = np.zeros(10 * G.shape[1])
    got = _lib.load().crm_test_null_fit_probe_read(_engine._context(0), _lib.ptr(rec), rec.size)
    rec = rec.reshape(-1, 10) if got == rec.size else np.full((G.shape[1], 10), np.nan)
    for groups in (None, "auto"):
        pv, info, st = crm.scan_interaction(GenotypePanel(G, groups=groups), return_stats=True, **hooks)
        same = info["rho1"] == oinfo["rho1"]
        qs = np.array([max(abs(ost["Q"][j]), float(np.trace(ost["F"][j]))) for j in range(G.shape[1])])
        dq = np.abs(st["Q"] - ost["Q"]) / qs
        dp = np.abs(pv - opv) / opv
        dl = np.abs(st["lml"] - ost["lml"]) / np.abs(ost["lml"])
        bad = (~same & (dl > 1e-11)) | (same & ((~loose & (dq > 1e-6)) | (~flat & (dp > 1e-5))))
        if bad.any():
            j = int(np.argmax(np.where(bad, np.maximum(dq, dp), 0)))
            print("problem %d %s: cells %d contexts %d covariates %d variants %d mode %s hooks %s path %s: %d bad; worst variant %d: "
                  "rho %g vs %g, dlml %.2e, dQ %.2e, dp %.2e, flat %d, delta %.3e vs %.3e; probes: decision %.2e Q move %.2e p move %.2e" %
                  (idx, case[:8] if isinstance(case, tuple) else "", y.size, E.shape[1], W.shape[1], G.shape[1], case[6], sorted(hooks),
                   "dense" if groups is None else "auto", int(bad.sum()), j, info["rho1"][j], oinfo["rho1"][j], dl[j], dq[j], dp[j],
                   int(flat[j]), st["delta"][j], ost["delta"][j], rec[j, 0], rec[j, 1], rec[j, 2]), flush=True)
print("done")
