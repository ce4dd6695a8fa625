"""Timing of estimate_betas (per SNP) at a BASELINE config; GPU only."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import cellregmap_amd as crm
from cellregmap_amd.synth import make_config

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
nsnp = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c = make_config(name, n_variants=nsnp, seed=0)
maf = np.clip(crm.compute_maf(c.G), 0.05, 0.5)
for rep in range(2):
    t = time.time()
    bg, bgxe = crm.estimate_betas(c.y, c.W, c.E, c.G, maf=maf, hK=c.hK)
    dt = time.time() - t
    print(f"{name}: n={c.y.size} k0={c.E.shape[1]} snps={nsnp}: {dt:.2f} s total, {dt / nsnp:.2f} s per SNP", flush=True)
print("beta_g", bg, "beta_gxe sd", bgxe.std(axis=1))
