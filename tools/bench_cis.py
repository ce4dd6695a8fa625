"""eQTL-shaped pass: many phenotypes, each against its own (overlapping) cis window of ONE resident panel,
through scan_interaction_many(..., cis_index=...).  Compared with one scan_interaction per gene on the
same windows.  GPU only.   python tools/bench_cis.py [cfg3] [genes 64] [window 1024] [stride 256] [general 0|1]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import cellregmap_amd as crm
from cellregmap_amd.synth import make_config

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
genes = int(sys.argv[2]) if len(sys.argv) > 2 else 64
window = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
stride = int(sys.argv[4]) if len(sys.argv) > 4 else 256
general = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
p = stride * (genes - 1) + window
c = make_config(name, n_variants=p, seed=0)
rng = np.random.default_rng(1)
G = c.G + (0.05 * rng.normal(size=c.G.shape) if general else 0.0)
Y = np.stack([c.y if g == 0 else rng.permutation(c.y) for g in range(genes)], axis=1)
cis = [(g * stride, g * stride + window) for g in range(genes)]

t = time.time()
Ls = crm.get_L_values(c.hK, c.E)
first = crm.CellRegMap(Y[:, 0], c.E, W=c.W, Ls=Ls)
crms = [first] + [crm.CellRegMap(Y[:, g], c.E, W=c.W, Ls=Ls, background=first._bg) for g in range(1, genes)]
print(f"{name}: background + {genes} phenotypes bound in {time.time() - t:.2f} s", flush=True)
t = time.time()
panel = crm.GenotypePanel(G)
print(f"panel of {p} variants resident in {time.time() - t:.2f} s (donor-level: {panel.n_groups is not None})", flush=True)
crm.scan_interaction_many(crms[:2], panel, cis_index=[(0, 64), (32, 96)])          # warm-up
t = time.time()
from cellregmap_amd import _engine
_engine._bind_genes_like(crms[0], crms[2:])
print(f"{genes - 2} phenotypes bound in one batch from the first one's device copies in {time.time() - t:.3f} s", flush=True)

t = time.time()
pv, info = crm.scan_interaction_many(crms, panel, cis_index=cis)
dt = time.time() - t
t = time.time()
pv_again, _ = crm.scan_interaction_many(crms, panel, cis_index=cis)
dt_again = time.time() - t
print(f"the same pass again (per-phenotype donor tables built): {dt_again:.3f} s = {sum(v.size for v in pv) / dt_again:.0f} /s", flush=True)
tests = sum(v.size for v in pv)
print(f"one pass over the panel: {tests} variant-tests ({genes} genes x {window}) in {dt:.3f} s = {tests / dt:.0f} /s", flush=True)

t = time.time()
worst = 0.0
for g in range(genes):     # (each window from the host again: upload + structure detection + scan)
    q, _ = crms[g].scan_interaction(G[:, cis[g][0]:cis[g][1]], progress=False)
    worst = max(worst, float(np.max(np.abs(q - pv[g]) / np.maximum(q, 1e-300))))
dt2 = time.time() - t
print(f"gene by gene from host windows: {dt2:.3f} s = {tests / dt2:.0f} /s;  max rel dp between the two = {worst:.2e}", flush=True)
