import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from cellregmap_amd import CellRegMap, GenotypePanel, get_L_values, scan_interaction_many
from cellregmap_amd.synth import make_cohort
from oracle import crm as ocrm
c = make_cohort(8, 25, 4, 40, seed=43)
n = c.y.size
rng = np.random.default_rng(11)
Y = np.stack([c.y, c.y[rng.permutation(n)], rng.normal(size=n), c.y + rng.normal(size=n), 3.0 - 2.0 * c.y], axis=1)
W = np.concatenate([c.W, rng.normal(size=(n, 1))], axis=1)
G = c.G + 0.05 * rng.normal(size=c.G.shape)
Ls = get_L_values(c.hK, c.E)
oLs = ocrm.khatri_rao_halves(c.hK, c.E)
first = CellRegMap(Y[:, 0], c.E, W=W, Ls=Ls)
crms = [first] + [CellRegMap(Y[:, i], c.E, W=W, Ls=Ls, background=first._bg) for i in range(1, Y.shape[1])]
panel = GenotypePanel(G)
pv, info = scan_interaction_many(crms, panel)
for i in range(Y.shape[1]):
    spv, sinfo, sst = crms[i].scan_interaction(panel, return_stats=True)
    opv, oinfo, ost = ocrm.OracleCellRegMap(Y[:, i], c.E, W=W, Ls=oLs).scan_interaction(G, return_stats=True)
    print("gene", i, "multi==single rho", np.array_equal(info["rho1"][i], sinfo["rho1"]), "single==oracle rho", np.array_equal(sinfo["rho1"], oinfo["rho1"]),
          "max rel p multi/single", np.max(np.abs(pv[i]-spv)/spv), "single/oracle", np.max(np.abs(spv-opv)/opv))
    bad = np.flatnonzero(sinfo["rho1"] != oinfo["rho1"])
    if bad.size:
        print("   variants", bad, "dev rho", sinfo["rho1"][bad], "or rho", oinfo["rho1"][bad], "dev lml", sst["lml"][bad], "or lml", ost["lml"][bad])
