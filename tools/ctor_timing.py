"""Background constructor (11 decompositions) at a BASELINE config, phases printed by the library
(CRM_TRACE_SETUP=1).   python tools/ctor_timing.py cfg3|cfg5|cfg2 [B]"""
import os
import sys
import time

os.environ["CRM_TRACE_SETUP"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from cellregmap_amd import CellRegMap, _engine, get_L_values  # noqa: E402
from cellregmap_amd.synth import make_config  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
mode = sys.argv[2] if len(sys.argv) > 2 else "C"
c = make_config(cfg, n_variants=16)
kw = {"Ls": get_L_values(c.hK, c.E)} if mode == "C" else {"hK": c.hK}
for rep in range(2):
    _engine._bg_cache.clear()
    t0 = time.time()
    crm = CellRegMap(c.y, c.E, W=c.W, **kw)
    print(f"[{cfg} mode {mode}] constructor run {rep}: {time.time() - t0:.3f} s, ranks",
          [crm._bg.rank(i) for i in range(len(crm._rho1))], flush=True)
    n = c.y.size
    if rep == 1 and n <= 20000:
        Q0, S0 = crm._bg.read(5, n)
        print("   orthonormality defect at rho[5]:", np.abs(Q0.T @ Q0 - np.eye(Q0.shape[1])).max())
    del crm
