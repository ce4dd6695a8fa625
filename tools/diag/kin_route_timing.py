import sys; sys.path.insert(0, "/root/repo")
import numpy as np, time
from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib, get_L_values
from cellregmap_amd.synth import make_cohort
lib = _lib.load()
for (d, c, k, p) in ((8, 30, 4, 64), (100, 200, 50, 4096)):
    co = make_cohort(d, c, k, p, seed=20)
    crm = CellRegMap(co.y, co.E, W=co.W, Ls=get_L_values(co.hK, co.E))
    print("groups in use:", lib.crm_background_kinship_groups(crm._bg.handle), "found:", _engine._kinship_groups(co.hK) is not None, flush=True)
    panel = GenotypePanel(co.G, groups=None)
    for on in (1, 0, 1):
        _lib.check(lib.crm_test_set_kinship_route(_engine._context(0), on))
        crm.scan_interaction(panel, progress=False)
        t = time.perf_counter(); pv, _ = crm.scan_interaction(panel, progress=False); dt = time.perf_counter() - t
        print(" route", on, "%.1f ms for %d variants = %.0f/s" % (dt * 1e3, p, p / dt), flush=True)
