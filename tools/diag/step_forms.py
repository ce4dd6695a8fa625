"""One timed step (4096 general variants, dense path) of a BASELINE configuration under kernel forms of the library
(include/crm_hip_test.h: crm_test_set_form), to see what a form is worth before it becomes the default.
    python tools/diag/step_forms.py cfg2 [C|B] [steps 8]  ->  one JSON line per form set"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib, get_L_values  # noqa: E402
from cellregmap_amd.synth import CONFIGS, make_cohort  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
mode = sys.argv[2] if len(sys.argv) > 2 else "C"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
donors, cells, k0, _ = CONFIGS[cfg]
c = make_cohort(donors, cells, k0, 4096, seed=20)
G = c.G + 0.05 * np.random.default_rng(3).normal(size=c.G.shape)
lib, ctx = _lib.load(), _engine._context(0)
FORMS = [{}, {"kin_fold": 2}, {"kin_fold": 2, "donor_pairs": 2}, {"kin_fold": 0}, {"nullfit_one_per_wave": 1}]
base = None
for forms in FORMS:
    for k, v in forms.items():
        _lib.check(lib.crm_test_set_form(k.encode(), v, 0))
    _engine._bg_cache.clear()
    kw = {"Ls": get_L_values(c.hK, c.E)} if mode == "C" else {"hK": c.hK}
    crm = CellRegMap(c.y, c.E, W=c.W, **kw)
    panel = GenotypePanel(G, groups=None)
    pv, _ = crm.scan_interaction(panel, progress=False)
    _lib.check(lib.crm_ctx_synchronize(ctx))
    t0 = time.perf_counter()
    for _ in range(steps):
        pv, _ = crm.scan_interaction(panel, progress=False)
    dt = (time.perf_counter() - t0) / steps
    base = pv if base is None else base
    print(json.dumps({"config": cfg, "mode": mode, "forms": forms, "ms_per_step": round(dt * 1e3, 3), "rate": round(4096 / dt, 1),
                      "folded": int(lib.crm_background_kinship_folded(crm._bg.handle)),
                      "max_rel_dp_vs_default": float(np.max(np.abs(pv - base) / base))}), flush=True)
    for k in forms:
        lib.crm_test_set_form(k.encode(), 0, 1)
    del crm, panel
