"""Achieved rate of the MixK(rho*) products inside a multi-phenotype pass (config 4's shape), by the library's own HIP events
around them: is the multi-problem launch as efficient as the single-phenotype one?   python tools/diag/cfg4_mix_rate.py [genes 64] [variants 8192]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("CELLREGMAP_AMD_PROGRESS", "0")
from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib, get_L_values  # noqa: E402
from cellregmap_amd.synth import CONFIGS, make_cohort  # noqa: E402

genes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
lib = _lib.load()
donors, cells, k0, _ = CONFIGS["cfg3"]
n = donors * cells
c = make_cohort(donors, cells, k0, 16, seed=20)
G = make_cohort(donors, cells, k0, nv, seed=77, with_phenotype=False).G
Ls = get_L_values(c.hK, c.E)
crm = CellRegMap(c.y, c.E, W=c.W, Ls=Ls)
crm._bind_gene()
rng = np.random.default_rng(99)
crms = [crm]
for i in range(1, genes):
    yi = c.y[rng.permutation(n)] if i % 2 else c.y + rng.normal(size=n)
    ci = CellRegMap(yi, c.E, W=c.W, background=crm._bg, Ls=Ls)
    ci._bind_gene()
    crms.append(ci)
handles = (ctypes.c_void_p * len(crms))(*[x._gene.value for x in crms])
panel = GenotypePanel(G, groups=None)
ctx = _engine._context(0)
pv = np.empty((genes, nv)); rho = np.empty((genes, nv))


def run():
    _lib.check(lib.crm_scan_interaction_multi(handles, genes, panel.handle, 0, nv, None, None, _lib.ptr(pv), _lib.ptr(rho), None, None, None, None))
    _lib.check(lib.crm_ctx_synchronize(ctx))


run()
_lib.check(lib.crm_kernel_timer_reset(ctx))
t0 = time.perf_counter()
run()
el = time.perf_counter() - t0
ms, cnt, fl, tot = ctypes.c_double(), ctypes.c_long(), ctypes.c_double(), ctypes.c_double()
_lib.check(lib.crm_kernel_timer_read(ctx, ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl), ctypes.byref(tot)))
_lib.check(lib.crm_kernel_timer_stop(ctx))
pairs = float(np.mean([len(set(rho[:, j])) for j in range(nv)]))
print("genes %d variants %d: %.3f s = %.0f variant-tests/s; %.2f distinct rho* per variant; timed launches %d, %.1f ms each, "
      "%.1f TFLOP/s on their own flops (%.3f of 78.6); share of the pass %.3f" % (
          genes, nv, el, genes * nv / el, pairs, cnt.value, ms.value / max(cnt.value, 1), fl.value / (ms.value * 1e-3) * 1e-12,
          fl.value / (ms.value * 1e-3) * 1e-12 / 78.6, ms.value * 1e-3 / el))
