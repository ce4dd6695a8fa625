"""What the flat-optimum probes measure against what actually happens between device and oracle (verbatim Brent on both
sides): per variant of the fuzz stream (drop of the likelihood one tolerance away / its value, relative move of Q and of p
one tolerance away, actual relative differences of Q and p against the oracle).  python tools/diag/flat_flag_study.py [count] [seed]"""
import ctypes
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

count = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
lib = _lib.load()
ctx = _engine._context(0)
rows = []
for case in fuzz_cases(count, seed=seed, wide_covariates=True):
    y, E, W, G, kw, hooks = build_case(case)
    crm = CellRegMap(y, E, W=W, **kw)
    try:
        opv, oinfo, ost = OracleCellRegMap(y, E, W=W, **kw).scan_interaction(G, return_stats=True, **hooks)
    except ValueError:
        continue
    panel = GenotypePanel(G, groups=None)
    pv, info, st = crm.scan_interaction(panel, return_stats=True, **hooks)
    crm.scan_interaction_info(panel, **hooks)
    rec = np.empty(3 * G.shape[1])
    got = lib.crm_test_null_fit_probe_read(ctx, _lib.ptr(rec), rec.size)
    if got != rec.size:
        continue
    rec = rec.reshape(-1, 3)
    for j in range(G.shape[1]):
        if info["rho1"][j] != oinfo["rho1"][j]:
            continue
        qscale = max(abs(ost["Q"][j]), float(np.trace(ost["F"][j])))
        rows.append((rec[j, 0], rec[j, 1], rec[j, 2], abs(st["Q"][j] - ost["Q"][j]) / qscale, abs(pv[j] - opv[j]) / opv[j], W.shape[1],
                     abs(st["lml"][j] - ost["lml"][j]) / abs(ost["lml"][j])))
a = np.array(rows)
out = {"scans": int(a.shape[0])}
bad = (a[:, 3] > 1e-6) | (a[:, 4] > 1e-5)
out["beyond_north_star"] = int(bad.sum())
sens = (a[:, 1] > 5e-7) | (a[:, 2] > 5e-6)
out["sensitive"] = int(sens.sum())
out["bad_not_sensitive"] = int((bad & ~sens).sum())
for thr in (1e-16, 3e-16, 1e-15, 3e-15, 1e-14, 3e-14, 1e-13, 2e-13):
    und = ~(a[:, 0] > thr)
    out["thr_%g" % thr] = {"flagged": int((sens & und).sum()), "bad_unflagged": int((bad & ~(sens & und)).sum())}
out["actual_lml_difference_percentiles"] = [float(x) for x in np.percentile(a[:, 6], [0, 25, 50, 75, 95, 100])]
out["drop_of_bad_percentiles"] = [float(x) for x in np.percentile(a[bad, 0], [0, 25, 50, 75, 90, 100])] if bad.any() else None
out["drop_of_all_percentiles"] = [float(x) for x in np.percentile(a[:, 0], [0, 5, 25, 50, 75, 95, 100])]
out["ratio_actual_dQ_over_probe_move_of_bad"] = [float(x) for x in np.percentile(a[bad, 3] / np.maximum(a[bad, 1], 1e-300), [0, 50, 100])] if bad.any() else None
print(json.dumps(out, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "flat_flag_study.npy"), a)
