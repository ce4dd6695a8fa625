"""Fixed cost of one scan call at BASELINE config 3: the same 16 384 resident variants scanned as 4 calls of one block, 2 calls
of two blocks and 1 call of four (sub-ranges of one panel through the C-ABI).  GPU only.
    python tools/diag/per_call_overhead.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib, get_L_values  # noqa: E402
from cellregmap_amd.synth import make_config  # noqa: E402

c = make_config("cfg3", n_variants=16384)
G = c.G + 0.05 * np.random.default_rng(0).normal(size=c.G.shape)
crm = CellRegMap(c.y, c.E, W=c.W, Ls=get_L_values(c.hK, c.E))
panel = GenotypePanel(G, groups=None)
lib, ctx = _lib.load(), _engine._context(0)
gene = crm._bind_gene()
pv, rho = np.empty(16384), np.empty(16384)


def scan(first, count):
    _lib.check(lib.crm_scan_interaction(gene, panel.handle, first, count, None, None, _lib.ptr(pv[first:]), _lib.ptr(rho[first:]),
                                        None, None, None, None, None, None, None, None, None))


scan(0, 16384)
for calls in (1, 2, 4, 1, 2, 4):
    per = 16384 // calls
    _lib.check(lib.crm_ctx_synchronize(ctx))
    t0 = time.perf_counter()
    for k in range(calls):
        scan(k * per, per)
    _lib.check(lib.crm_ctx_synchronize(ctx))
    print(f"{calls} call(s) of {per} variants: {time.perf_counter() - t0:.4f} s", flush=True)
os.environ["CRM_TRACE_SETUP"] = "1"
