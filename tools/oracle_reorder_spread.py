"""Oracle against itself on the fuzz stream of tools/fuzz_scan.py (CPU only): the same problem with the cells listed in
another order (all inputs and the rows of Q0 permuted consistently -- identical mathematics, every n-length inner product
summed in another order).  Per background mode: the share of variants whose score statistic moves by more than 1e-6 and
the worst relative difference of the null model's lml.   python tools/oracle_reorder_spread.py [count 400] [seed 2026]"""
import copy
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_cases import build_case, fuzz_cases  # noqa: E402
from oracle.crm import OracleCellRegMap  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
rows = []
raised = 0
for case in fuzz_cases(count, seed=seed, wide_covariates=True):
    y, E, W, G, kw, hooks = build_case(case)
    try:
        base = OracleCellRegMap(y, E, W=W, **kw)
        pa, ia, sa = base.scan_interaction(G, return_stats=True, **hooks)
        n = y.shape[0]
        order = np.random.default_rng(case[0]).permutation(n)
        inv = np.argsort(order)
        cp = copy.copy(base)
        cp._y, cp._W, cp._E0, cp._E1 = base._y[order], base._W[order], base._E0[order], base._E1[order]
        cp._qs = {rho: ((np.ascontiguousarray(q[0][order]),), sv) for rho, (q, sv) in base._qs.items()}
        chooks = {k: inv[np.asarray(v)[order]] for k, v in hooks.items()}
        pb, ib, sb = cp.scan_interaction(G[order], return_stats=True, **chooks)
    except ValueError:
        raised += 1
        continue
    same = ia["rho1"] == ib["rho1"]
    for j in range(G.shape[1]):
        qscale = max(abs(sa["Q"][j]), float(np.trace(sa["F"][j])))
        rows.append((abs(sa["Q"][j] - sb["Q"][j]) / qscale, abs(pa[j] - pb[j]) / pa[j],
                     abs(sa["lml"][j] - sb["lml"][j]) / abs(sa["lml"][j]), bool(same[j]), "ABC".index(case[6])))
a = np.array(rows, float)
same = a[:, 3] > 0
out = {"what": "oracle vs oracle, cells in another order (identical mathematics)", "problems": count - raised, "seed": seed,
       "variant_scans": int(a.shape[0]), "rho_star_differs": int((~same).sum()),
       "worst_rel_Q": float(a[same, 0].max()), "worst_rel_p": float(a[same, 1].max()),
       "share_Q_beyond_1e-6": float((a[same, 0] > 1e-6).mean()),
       "share_Q_beyond_1e-6_by_mode": {m: float((a[same & (a[:, 4] == k), 0] > 1e-6).mean()) for k, m in enumerate("ABC")},
       "worst_rel_lml_by_mode": {m: float(a[same & (a[:, 4] == k), 2].max()) for k, m in enumerate("ABC")},
       "median_rel_lml": float(np.median(a[same, 2]))}
print(json.dumps(out, indent=1))
