"""Problems of a fuzz stream where the device and the oracle pick different rho* with different likelihoods
(python tools/diag/fuzz_outliers.py [count 1000] [seed 2026] [polish 1])."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from fuzz_cases import build_case, fuzz_cases
from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib
from oracle.crm import OracleCellRegMap
count = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
polish = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
lib = _lib.load()
_lib.check(lib.crm_set_null_fit_polish(_engine._context(0), 1 if polish else 0))
for case in fuzz_cases(count, seed=seed, wide_covariates=not polish):
    y, E, W, G, kw, hooks = build_case(case)
    try:
        opv, oinfo, ost = OracleCellRegMap(y, E, W=W, polish=polish, **kw).scan_interaction(G, return_stats=True, **hooks)
    except ValueError:
        continue
    crm = CellRegMap(y, E, W=W, **kw)
    pv, info, st = crm.scan_interaction(GenotypePanel(G, groups=None), return_stats=True, **hooks)
    differs = info["rho1"] != oinfo["rho1"]
    rl = np.abs(st["lml"] - ost["lml"]) / np.abs(ost["lml"])
    rp = np.abs(pv - opv) / opv
    bad = (differs & (rl > 1e-10)) | (~differs & (rp > 5e-5))
    if bad.any():
        j = np.flatnonzero(bad)[:3]
        print("CASE", case, "n_bad", int(bad.sum()), "of", G.shape[1], "ranks", [crm._bg.rank(i) for i in range(11)],
              "n", y.size, "\n   dev rho", info["rho1"][j], "or rho", oinfo["rho1"][j], "dev lml", st["lml"][j], "or lml", ost["lml"][j],
              "dev delta", st["delta"][j], "or delta", ost["delta"][j], "dev p", pv[j], "or p", opv[j], "dev Q", st["Q"][j], "or Q", ost["Q"][j], flush=True)
