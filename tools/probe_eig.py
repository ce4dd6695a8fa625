"""GPU probe: where does eig_davies_kernel spend its time? (eigenvalues vs Davies), realistic F."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellregmap_amd import CellRegMap, _engine, _lib
from cellregmap_amd.synth import make_cohort

c = make_cohort(20, 100, 50, 64, seed=5)
crm = CellRegMap(c.y, c.E, W=c.W, hK=c.hK)
pv, info, st = crm.scan_interaction(c.G, return_stats=True)
F = np.ascontiguousarray(np.tile(st["F"], (16, 1, 1)))      # 1024 matrices
Q = np.ascontiguousarray(np.tile(st["Q"], 16))
lam = np.ascontiguousarray(np.tile(st["lambda"], (16, 1)))
lib = _lib.load(); h = _engine._context(0)
count, k = F.shape[0], F.shape[1]
out = np.empty((count, k)); p = np.empty(count); ifl = np.empty(count, np.int32); liu = np.empty(count)
for name, fn in (("eig+davies(Q=0)", lambda: lib.crm_test_eigvalsh(h, count, k, _lib.ptr(F), _lib.ptr(out))),
                 ("davies only", lambda: lib.crm_test_davies(h, count, k, _lib.ptr(Q), _lib.ptr(lam), _lib.ptr(p), _lib.ptr(ifl), _lib.ptr(liu)))):
    fn(); t = time.time(); fn(); dt = time.time() - t
    print(f"{name}: {dt*1e3:.2f} ms for {count} matrices k={k}")
print("eig max err", np.abs(out[:64] - st["lambda"]).max(), "p match", np.abs(p[:64] / pv - 1).max(), "ifault", np.bincount(ifl[:64] + 2))
print("p-values", np.sort(pv)[:5], "Q/sum(lam)", (st["Q"] / st["lambda"].sum(1))[:5])
