"""Oracle against itself on the fuzz stream of tools/fuzz_scan.py (CPU only), the objective of the null fits multiplied by
(1 + eps * N(0, 1)) at every evaluation: how often does rounding noise of a given relative size in the likelihood move
the score statistic by more than 1e-6 under the reference's Brent(1e-6) search?  The device's likelihood agrees with the
oracle's to 1e-15 typical, 4e-15 worst, AT FIXED POINTS (tools/probe_objective.py), i.e. this experiment at eps = 1e-15 ..
4e-15 is the oracle-only counterpart of the device-vs-oracle comparison.
    python tools/oracle_noise_spread.py [count 400] [seed 2026] [eps ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_cases import build_case, fuzz_cases  # noqa: E402
from oracle import lmm as olmm  # noqa: E402
from oracle.crm import OracleCellRegMap  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
levels = [float(v) for v in sys.argv[3:]] or [1e-15, 4e-15]
exact = olmm.LMM._neg_lml_at
rows = {eps: [] for eps in levels}
raised = 0
for case in fuzz_cases(count, seed=seed, wide_covariates=True):
    y, E, W, G, kw, hooks = build_case(case)
    try:
        base = OracleCellRegMap(y, E, W=W, **kw)
        pa, ia, sa = base.scan_interaction(G, return_stats=True, **hooks)
        trF = np.array([np.trace(F) for F in sa["F"]])
        for eps in levels:
            rng = np.random.default_rng(case[0])
            olmm.LMM._neg_lml_at = lambda self, x, _e=eps, _r=rng: exact(self, x) * (1.0 + _e * _r.normal())
            try:
                pb, ib, sb = base.scan_interaction(G, return_stats=True, **hooks)
            finally:
                olmm.LMM._neg_lml_at = exact
            same = ia["rho1"] == ib["rho1"]
            dq = np.abs(sa["Q"] - sb["Q"]) / np.maximum(np.abs(sa["Q"]), trF)
            rows[eps].extend((float(dq[j]), bool(same[j]), "ABC".index(case[6])) for j in range(G.shape[1]))
    except ValueError:
        raised += 1
out = {"what": "oracle vs oracle with relative noise eps on every evaluation of the null-fit objective", "problems": count - raised,
       "seed": seed, "levels": {}}
for eps, r in rows.items():
    a = np.array(r, float)
    same = a[:, 1] > 0
    out["levels"]["%g" % eps] = {"variant_scans": int(a.shape[0]), "rho_star_differs": int((~same).sum()),
                                 "worst_rel_Q": float(a[same, 0].max()), "median_rel_Q": float(np.median(a[same, 0])),
                                 "share_Q_beyond_1e-6": float((a[same, 0] > 1e-6).mean()),
                                 "share_Q_beyond_1e-6_by_mode": {m: float((a[same & (a[:, 2] == k), 0] > 1e-6).mean())
                                                                 for k, m in enumerate("ABC")}}
print(json.dumps(out, indent=1))
