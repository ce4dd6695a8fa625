"""Which problems of the fuzz stream carry the device-vs-oracle differences?  Per problem (dense path, verbatim procedure,
oracle on the device's decomposition): the worst relative differences of lml and Q with the problem's parameters, sorted.
GPU only.   python tools/fuzz_case_report.py [count 400] [seed 2026]"""
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

count = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
out = []
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
    dl = np.abs(st["lml"] - ost["lml"]) / np.abs(ost["lml"])
    dd = np.abs(st["delta"] - ost["delta"]) / ost["delta"]
    j = int(np.argmax(dl))
    ranks = [crm._bg.rank(i) for i in range(len(crm._rho1))]
    S0 = crm._bg.read(int(round(oinfo["rho1"][j] * 10)) if len(ranks) > 1 else 0, y.size)[1]
    out.append({"case": [int(v) if not isinstance(v, str) else v for v in case], "rmax": max(ranks), "worst_rel_lml": float(dl.max()),
                "worst_rel_Q": float(dq[same].max()) if same.any() else None, "beyond_1e-6": int((dq[same] > 1e-6).sum()),
                "variants": int(G.shape[1]), "delta_at_worst_lml": float(ost["delta"][j]), "rel_ddelta_at_worst": float(dd[j]),
                "lml_at_worst": float(ost["lml"][j]), "rho_at_worst": float(oinfo["rho1"][j]),
                "cond_S0": float(S0.max() / S0.min()) if S0.size else None})
print(json.dumps({"columns": "case = (index, n, k0, c, p, donors, mode, permutation hook)",
                  "by_lml": sorted(out, key=lambda r: -r["worst_rel_lml"])[:40],
                  "by_Q": sorted(out, key=lambda r: -(r["worst_rel_Q"] or 0))[:40]}, indent=0))
