"""Per-call anatomy of the cis-window pass: a few per-gene scans of 1024-variant windows of a resident panel (tables built
before), for a kernel trace.   rocprofv3 --kernel-trace ... -- python3 tools/diag/cis_call_trace.py [general 0|1]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import cellregmap_amd as crm
from cellregmap_amd.synth import make_config

general = bool(int(sys.argv[1])) if len(sys.argv) > 1 else False
c = make_config("cfg3", n_variants=8192, seed=0)
rng = np.random.default_rng(1)
G = c.G + (0.05 * rng.normal(size=c.G.shape) if general else 0.0)
Ls = crm.get_L_values(c.hK, c.E)
first = crm.CellRegMap(c.y, c.E, W=c.W, Ls=Ls)
genes = [first] + [crm.CellRegMap(rng.permutation(c.y), c.E, W=c.W, Ls=Ls, background=first._bg) for _ in range(7)]
panel = crm.GenotypePanel(G)
for g in genes:
    crm.scan_interaction_many([g], panel, cis_index=[(0, 256)])      # tables, work buffers
print("MARK warm", flush=True)
t = time.time()
for i, g in enumerate(genes):
    crm.scan_interaction_many([g], panel, cis_index=[(256 * i, 256 * i + 1024)])
dt = time.time() - t
print("8 per-gene scans of 1024 variants: %.2f ms each" % (dt / 8 * 1e3), flush=True)
t = time.time()
crm.scan_interaction_many(genes, panel, cis_index=[(256 * i, 256 * i + 1024) for i in range(8)])
print("the same through scan_interaction_many(cis_index): %.2f ms" % ((time.time() - t) * 1e3), flush=True)
