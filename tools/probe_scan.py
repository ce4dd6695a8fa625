"""GPU probe: end-to-end scan vs oracle with verbose diffs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellregmap_amd import CellRegMap, get_L_values
from cellregmap_amd.synth import make_cohort
from oracle.crm import OracleCellRegMap, khatri_rao_halves

np.set_printoptions(linewidth=200, precision=6)
for mode, args in [("B", (10, 20, 5, 24, 5)), ("A", (10, 20, 5, 24, 5)), ("C", (6, 40, 4, 24, 4)), ("C-eigh", (12, 10, 10, 24, 3))]:
    c = make_cohort(*args[:4], seed=args[4])
    kw, okw = {}, {}
    if mode == "B":
        kw["hK"] = okw["hK"] = c.hK
    elif mode.startswith("C"):
        kw["Ls"] = get_L_values(c.hK, c.E); okw["Ls"] = khatri_rao_halves(c.hK, c.E)
    t = time.time()
    crm = CellRegMap(c.y, c.E, W=c.W, **kw)
    print(mode, "ctor", time.time() - t, "ranks", [crm._bg.rank(i) for i in range(len(crm._rho1))])
    t = time.time()
    pv, info, st = crm.scan_interaction(c.G, return_stats=True)
    print(mode, "scan", time.time() - t)
    o = OracleCellRegMap(c.y, c.E, W=c.W, **okw)
    print("oracle ranks", [o._qs[r][1].shape[0] for r in o._rho])
    opv, oinfo, ost = o.scan_interaction(c.G, return_stats=True)
    print(" rho  gpu", info["rho1"][:12]); print(" rho  ora", oinfo["rho1"][:12])
    print(" max rel delta", np.max(np.abs(st["delta"] / ost["delta"] - 1)), "lml", np.max(np.abs(st["lml"] / ost["lml"] - 1)))
    print(" max rel Q", np.max(np.abs(st["Q"] / ost["Q"] - 1)))
    F = np.stack(ost["F"]); print(" max F err", np.max(np.abs(st["F"] - F)) / np.abs(F).max())
    print(" max rel p", np.max(np.abs(pv / opv - 1)), "max abs p", np.max(np.abs(pv - opv)))
    print(" pv gpu", pv[:8]); print(" pv ora", opv[:8])
