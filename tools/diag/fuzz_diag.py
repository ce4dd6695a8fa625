import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from fuzz_cases import build_case, fuzz_cases
from cellregmap_amd import CellRegMap, GenotypePanel
from oracle.crm import OracleCellRegMap
for case in fuzz_cases(150, seed=7):
    y, E, W, G, kw, hooks = build_case(case)
    try:
        opv, oinfo, ost = OracleCellRegMap(y, E, W=W, **kw).scan_interaction(G, return_stats=True, **hooks)
    except ValueError as e:
        continue
    crm = CellRegMap(y, E, W=W, **kw)
    for groups in (None, "auto"):
        pv, info, st = crm.scan_interaction(GenotypePanel(G, groups=groups), return_stats=True, **hooks)
        rl = np.abs(st["lml"] - ost["lml"]) / np.abs(ost["lml"])
        rq = np.abs(st["Q"] - ost["Q"]) / np.abs(ost["Q"])
        bad = ~np.isfinite(rq) | (rl > 1e-10) | ~np.isfinite(pv) | ~np.isfinite(opv) | ((rq > 5e-6) & (info['rho1'] == oinfo['rho1']))
        if bad.any():
            j = np.flatnonzero(bad)
            print("CASE", case, "groups", groups, "variants", j[:5], "dev lml", st["lml"][j[:3]], "or lml", ost["lml"][j[:3]],
                  "dev Q", st["Q"][j[:3]], "or Q", ost["Q"][j[:3]], "dev p", pv[j[:3]], "or p", opv[j[:3]],
                  "rho", info["rho1"][j[:3]], oinfo["rho1"][j[:3]], "delta", st["delta"][j[:3]], ost["delta"][j[:3]], "relQ", rq[j[:3]], "ranks", [crm._bg.rank(i) for i in range(len(crm._rho1))], flush=True)
