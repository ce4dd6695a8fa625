"""Do the variants whose score statistic differs from the oracle's by more than 1e-6 (verbatim procedure) coincide with
null fits whose objective is FLAT over the reference's stopping tolerance?  Per variant of the fuzz stream: the relative
difference of Q, and  c = (1/2 f''(x*) tol^2) / (eps |f(x*)|)  with tol = 1e-6 |x*| + 1e-6 (what Brent's last comparisons
see of the objective, in units of its rounding error), f'' from the oracle's analytic gradient.  GPU only (the device
results); prints the table of c-quantiles for the variants beyond / within 1e-6.
    python tools/flatness_report.py [count 400] [seed 2026]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_cases import build_case, fuzz_cases  # noqa: E402
from test_gpu_fuzz import _oracle_on_device_decomposition  # noqa: E402

from cellregmap_amd import CellRegMap, GenotypePanel  # noqa: E402
from oracle.lmm import LMM  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
EPS = np.finfo(float).eps
rows = []
for case in fuzz_cases(count, seed=seed, wide_covariates=True):
    y, E, W, G, kw, hooks = build_case(case)
    crm = CellRegMap(y, E, W=W, **kw)
    try:
        o = _oracle_on_device_decomposition(crm, y, E, W, False)
        opv, oinfo, ost = o.scan_interaction(G, return_stats=True, **hooks)
    except ValueError:
        continue
    pv, info, st = crm.scan_interaction(GenotypePanel(G, groups=None), return_stats=True, progress=False, **hooks)
    same = info["rho1"] == oinfo["rho1"]
    trF = np.array([np.trace(F) for F in ost["F"]])
    dq = np.abs(st["Q"] - ost["Q"]) / np.maximum(np.abs(ost["Q"]), trF)
    for j in range(G.shape[1]):
        if not same[j]:
            continue
        X = np.concatenate((W, G[:, [j]]), axis=1)
        lm = LMM(y, X, o._qs[oinfo["rho1"][j]], restricted=True)
        lm.fit()
        x = lm._x
        f0 = abs(lm._neg_lml_at(x))
        h = 1e-3
        curv = (lm._neg_lml_grad_at(x + h) - lm._neg_lml_grad_at(x - h)) / (2 * h)
        tol = 1e-6 * abs(x) + 1e-6
        c = 0.5 * abs(curv) * tol * tol / (EPS * f0)
        rows.append((float(dq[j]), float(c), float(x), "ABC".index(case[6]), case[5]))
a = np.array(rows)
beyond = a[:, 0] > 1e-6
q = [0.0, 0.01, 0.05, 0.25, 0.5, 0.75, 0.95, 0.99, 1.0]
out = {"variants": int(a.shape[0]), "beyond_1e-6": int(beyond.sum()),
       "c_quantiles": {"levels": q, "beyond": [float(v) for v in np.quantile(a[beyond, 1], q)] if beyond.any() else None,
                       "within": [float(v) for v in np.quantile(a[~beyond, 1], q)]},
       "share_flagged_at_threshold": {str(t): {"of_beyond": float((a[beyond, 1] < t).mean()) if beyond.any() else None,
                                               "of_all": float((a[:, 1] < t).mean())} for t in (1, 2, 4, 8, 16, 32, 64, 128, 256)},
       "beyond_rows": [[float(v) for v in r] for r in a[beyond][np.argsort(-a[beyond, 0])][:60]]}
print(json.dumps(out, indent=0))
