"""More than 128 contexts at a size where the work buffers are gigabytes (20 000 cells, 50 donors, 160 contexts, mode C:
8 160 columns of the half factor, 600 dense general variants): the folded kinship-structure route, the unfolded one and the
direct contraction against H must agree with each other (the small shapes are checked against the oracle in
tests/test_gpu_edges.py::test_interaction_scan_with_many_contexts).  GPU only.

    python tools/diag/many_contexts_at_size.py [contexts 160] [variants 600]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

k0 = int(sys.argv[1]) if len(sys.argv) > 1 else 160
p = int(sys.argv[2]) if len(sys.argv) > 2 else 600
from cellregmap_amd import CellRegMap, GenotypePanel, get_L_values  # noqa: E402
from cellregmap_amd.synth import make_cohort  # noqa: E402

co = make_cohort(50, 400, k0, p, seed=3)   # 20 000 cells: the 8 160 columns stay below the cell count (thin branch, factored H)
rng = np.random.default_rng(0)
G = co.G + 0.05 * rng.normal(size=co.G.shape)          # general genotypes: the dense path
from cellregmap_amd import _engine, _lib  # noqa: E402

lib, ctx = _lib.load(), _engine._context(0)
res, secs = {}, {}
for name, route, fold in (("folded", 2, None), ("unfolded", 2, "0"), ("direct", 0, None)):
    os.environ.pop("CRM_KIN_FOLD", None)
    if fold is not None:
        os.environ["CRM_KIN_FOLD"] = fold       # read when the donor structure is announced
    _engine._bg_cache.clear()                   # ... so every form gets a background of its own
    t0 = time.time()
    crm = CellRegMap(co.y, co.E, W=co.W, Ls=get_L_values(co.hK, co.E))   # mode C: K o EE' through its factored halves
    t1 = time.time()
    assert lib.crm_background_kinship_groups(crm._bg.handle) == 50
    assert (lib.crm_background_kinship_folded(crm._bg.handle) > 0) == (fold is None)
    _lib.check(lib.crm_test_set_kinship_route(ctx, route))
    pv, info, st = crm.scan_interaction(GenotypePanel(G, groups=None), return_stats=True, progress=False)
    _lib.check(lib.crm_test_set_kinship_route(ctx, 1))
    secs[name] = {"constructor_s": round(t1 - t0, 2), "scan_s": round(time.time() - t1, 2)}
    res[name] = (pv, info["rho1"], st["Q"], st["lml"])
    del crm
out = {"cells": int(co.y.size), "contexts": k0, "variants": p, "columns_of_H": k0 + 50 * k0, "seconds": secs}
ref = res["direct"]
for name in ("folded", "unfolded"):
    pv, rho, Q, lml = res[name]
    same = rho == ref[1]
    out[name + "_vs_direct"] = {"rho_star_differs": int((~same).sum()),
                                "max_rel_lml": float(np.max(np.abs(lml - ref[3]) / np.abs(ref[3]))),
                                "max_rel_Q": float(np.max(np.abs(Q[same] - ref[2][same]) / np.abs(ref[2][same]))),
                                "max_rel_p": float(np.nanmax(np.abs(pv[same] - ref[0][same]) / np.maximum(ref[0][same], 1e-300)))}
out["p_values"] = {"nan": int(np.isnan(ref[0]).sum()), "zero": int((ref[0] == 0).sum()), "min_positive": float(ref[0][ref[0] > 0].min()),
                   "max": float(np.nanmax(ref[0]))}
print(json.dumps(out, indent=1))
