import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cellregmap_amd import CellRegMap, GenotypePanel, get_L_values
from cellregmap_amd.synth import make_config
t0=time.time()
c = make_config("cfg5", n_variants=192)
Ls = get_L_values(c.hK, c.E)
print("data", time.time()-t0); t0=time.time()
crm = CellRegMap(c.y, c.E, W=c.W, Ls=Ls)
print("ctor", time.time()-t0, [crm._bg.rank(i) for i in range(11)]); t0=time.time()
pv, info, st = crm.scan_interaction(GenotypePanel(c.G, groups=None), return_stats=True)
print("scan", time.time()-t0)
bad = np.flatnonzero(~((pv > 0) & (pv <= 1)))
print("bad", bad, pv[bad], st["Q"][bad], info["rho1"][bad], st["lambda"][bad][:, -5:] if bad.size else "")
print("pv sorted head", np.sort(pv)[:5], np.argsort(pv)[:5])
