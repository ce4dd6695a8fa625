"""rocprofv3 target: donor-collapsed scan at config 3 (20 000 cells, 50 contexts, mode C)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellregmap_amd import CellRegMap, GenotypePanel, get_L_values
from cellregmap_amd.synth import make_cohort

c = make_cohort(100, 200, 50, 16, seed=20)
s = make_cohort(100, 200, 50, 4096, seed=1000, with_phenotype=False)
crm = CellRegMap(c.y, c.E, W=c.W, Ls=get_L_values(c.hK, c.E))
panel = GenotypePanel.from_donors(s.G[::200], s.donor_of_cell)
crm.scan_interaction(panel)
t = time.time(); pv, _ = crm.scan_interaction(panel); print("collapsed scan 4096 variants:", time.time() - t, "s")
