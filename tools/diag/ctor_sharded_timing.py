"""What one rank of an N-GPU job pays in the constructor: decompositions of 1, 2 and 11 owned grid points at a
BASELINE config (phases printed by the library).  python tools/diag/ctor_sharded_timing.py [cfg3]"""
import os, sys, time
os.environ["CRM_TRACE_SETUP"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cellregmap_amd import _engine, get_L_values
from cellregmap_amd._engine import BackgroundBuilder
from cellregmap_amd.synth import make_config
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
c = make_config(cfg, n_variants=16)
Ls = get_L_values(c.hK, c.E)
rho = _engine._RHO_GRID
for owned in (11, 2, 1, 2, 1):
    mine = np.zeros(11, np.int32); mine[:owned] = 1
    t0 = time.time()
    b = BackgroundBuilder(c.E, Ls, rho, device=0, mine=mine)
    print(f"[{cfg}] begin with {owned} owned grid points: {time.time() - t0:.3f} s", flush=True)
    del b
