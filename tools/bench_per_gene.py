"""eQTL-style loop: many genes, each against its own cis window (a fresh panel) on one cohort.
Per-gene wall time of run_interaction at a BASELINE config; GPU only."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import cellregmap_amd as crm
from cellregmap_amd.synth import make_config

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
genes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
window = int(sys.argv[3]) if len(sys.argv) > 3 else 256
c = make_config(name, n_variants=genes * window, seed=0)
rng = np.random.default_rng(1)
for g in range(genes):
    y = c.y if g == 0 else rng.permutation(c.y)
    G = np.ascontiguousarray(c.G[:, g * window:(g + 1) * window])
    t = time.time()
    pv, info = crm.run_interaction(y, c.E, G, W=c.W, hK=c.hK)
    print(f"{name} gene {g}: {window} SNPs in {time.time() - t:.3f} s  (min p {pv.min():.3g})", flush=True)
