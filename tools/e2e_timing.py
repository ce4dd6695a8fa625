"""GPU probe: wall-clock anatomy of run_interaction(y, E, G, W, hK) on raw numpy inputs, config 3."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cellregmap_amd as crm
from cellregmap_amd.synth import make_cohort

c = make_cohort(100, 200, 50, 16, seed=20)
s = make_cohort(100, 200, 50, 8192, seed=1000, with_phenotype=False)
crm.run_interaction(c.y, c.E, s.G[:, :128], W=c.W, hK=c.hK)   # warm-up: context, code objects
crm._engine._bg_cache.clear()
for label in ("cold background", "cached background"):
    t0 = time.time(); Ls = crm.get_L_values(c.hK, c.E); t1 = time.time()
    obj = crm.CellRegMap(c.y, c.E, W=c.W, Ls=Ls); t2 = time.time()
    panel = crm.GenotypePanel(s.G); t3 = time.time()
    pv, info = obj.scan_interaction(panel); t4 = time.time()
    print(f"{label}: get_L_values {t1-t0:.3f}  CellRegMap {t2-t1:.3f}  GenotypePanel(8192 variants, auto) {t3-t2:.3f} "
          f"(groups {panel.n_groups})  scan {t4-t3:.3f}  total {t4-t0:.3f} s")
t = time.time(); pv2, _ = crm.run_interaction(c.y, c.E, s.G, W=c.W, hK=c.hK); print("run_interaction end to end", round(time.time()-t, 3), "s", np.array_equal(pv, pv2))
t = time.time(); p2 = crm.GenotypePanel(s.G, groups=None); print("dense panel", round(time.time()-t,3))
