"""B permutations of one scan: ``scan_interaction_permutations`` (one call: null fits, rho*, rotations once per block) against
B separate ``scan_interaction(G, idx_E=perm)`` calls, at a BASELINE configuration.  GPU only.
    python tools/bench_permutations.py [cfg2|cfg3] [C|B] [B 16] [variants 4096]"""
import json
import sys
import time
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib, get_L_values  # noqa: E402
from cellregmap_amd.synth import CONFIGS, make_cohort  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
mode = sys.argv[2] if len(sys.argv) > 2 else "C"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
p = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
donors, cells, k0, _ = CONFIGS[cfg]
c = make_cohort(donors, cells, k0, p, seed=20)
rng = np.random.default_rng(3)
G = c.G + 0.05 * rng.normal(size=c.G.shape)            # general genotypes: the dense path
kw = {"Ls": get_L_values(c.hK, c.E)} if mode == "C" else {"hK": c.hK}
crm = CellRegMap(c.y, c.E, W=c.W, **kw)
panel = GenotypePanel(G, groups=None)
lib, ctx = _lib.load(), _engine._context(0)
perms = [rng.permutation(c.y.size) for _ in range(B)]
crm.scan_interaction(panel, idx_E=perms[0])            # warm-up (workspaces, Q0 of the selected grid points)
crm.scan_interaction_permutations(panel, idx_E_list=perms[:2])
_lib.check(lib.crm_ctx_synchronize(ctx))
t0 = time.perf_counter()
one = [crm.scan_interaction(panel, idx_E=q)[0] for q in perms]
t_sep = time.perf_counter() - t0
t0 = time.perf_counter()
pv, info = crm.scan_interaction_permutations(panel, idx_E_list=perms)
t_one = time.perf_counter() - t0
same = all(np.array_equal(pv[b], one[b]) for b in range(B))
out = {"config": cfg, "mode": mode, "permutations": B, "variants": p,
       "separate_calls_s": round(t_sep, 4), "one_call_s": round(t_one, 4), "speedup": round(t_sep / t_one, 3),
       "separate_rate": round(B * p / t_sep, 1), "one_call_rate": round(B * p / t_one, 1), "unit": "variant-tests/s (permuted scans)",
       "bit_identical": bool(same), "median_p_over_permutations": float(np.median(pv)),
       "note": "idx_E permutations of the contexts (the reference's calibration loop, cellregmap/test/test_struct_lmm2.py:208-209); "
               "general genotypes, dense path; the one-call form computes null fits, rho* and the rotations of a block once"}
print(json.dumps(out))
