"""Phase timings of the constructor's two-stage eigen-solver on a random family of the size of a BASELINE background
(CRM_TRACE_SETUP=1 prints the phases on stderr), through the test hook:  python tools/diag/eigh2_timing.py [dim 5064] [nq 10]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["CRM_TRACE_SETUP"] = "1"
from cellregmap_amd import _engine, _lib  # noqa: E402

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 5064
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(1)
H = rng.normal(size=(dim + 500, dim))
H[:, :14] = 0.0
C = H.T @ H
rho = np.linspace(0.0, 0.9, nq)
wa, wb = np.sqrt(rho), np.sqrt(1 - rho)
lib = _lib.load()
ctx = _engine._context(0)
lam = np.empty((nq, dim))
for rep in range(2):
    t0 = time.perf_counter()
    _lib.check(lib.crm_test_eigh2(ctx, nq, dim, _lib.ptr(C), _lib.ptr(wa), _lib.ptr(wb), _lib.ptr(lam), None, 0, None, None, None))
    print("two-stage, %d x %d: %.3f s" % (nq, dim, time.perf_counter() - t0), flush=True)
ref = np.linalg.eigvalsh(C * np.outer(np.r_[np.full(64, wa[3]), np.full(dim - 64, wb[3])], np.r_[np.full(64, wa[3]), np.full(dim - 64, wb[3])]))
print("max relative eigenvalue difference vs LAPACK (grid point 3): %.2e" % (np.abs(lam[3] - ref).max() / np.abs(ref).max()))
