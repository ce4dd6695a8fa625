"""Random sizes through the constructor's two-stage family solver (test hook), against LAPACK: eigenvalues, orthonormality,
residual.  Sizes are drawn so that every remainder class of the back-transformation's tiles (16), windows (64) and task
rounds shows up:   python tools/diag/eigh2_soak.py [cases 24] [seed 5] [max order 3600]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cellregmap_amd import _engine, _lib  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
top = int(sys.argv[3]) if len(sys.argv) > 3 else 3600
rng = np.random.default_rng(seed)
lib = _lib.load()
ctx = _engine._context(0)
worst = {"eigenvalues": 0.0, "orthonormality": 0.0, "residual": 0.0}
rows = []
for case in range(cases):
    dim = int(rng.integers(130, top)) if case % 4 else int(rng.integers(130, 400))
    k1 = int(rng.integers(1, 65))
    nq = int(rng.integers(1, 12))
    deficient = int(rng.integers(0, 30)) if case % 3 == 0 else 0
    H = rng.normal(size=(dim + 40, dim))
    H[:, : 64 - k1] = 0.0
    if deficient:
        H[:, -deficient:] = H[:, 64:64 + deficient]
    C = np.ascontiguousarray(H.T @ H)
    rho = np.sort(rng.uniform(0.0, 1.0, nq))
    rho[0] = 0.0 if case % 2 else rho[0]
    wa, wb = np.sqrt(rho), np.sqrt(1 - rho)
    lam = np.empty((nq, dim))
    Z = np.empty((nq, dim, dim))
    t0 = time.perf_counter()
    _lib.check(lib.crm_test_eigh2(ctx, nq, dim, _lib.ptr(C), _lib.ptr(wa), _lib.ptr(wb), _lib.ptr(lam), _lib.ptr(Z), 0, None, None, None))
    dt = time.perf_counter() - t0
    rec = {"order": dim, "contexts": k1, "grid_points": nq, "deficient": deficient, "seconds": round(dt, 3)}
    for q in range(nq):
        dsc = np.r_[np.full(64, wa[q]), np.full(dim - 64, wb[q])]
        A = C * np.outer(dsc, dsc)
        ref = np.linalg.eigvalsh(A)
        scale = np.abs(ref).max()
        ev = float(np.abs(lam[q] - ref).max() / scale)
        orth = float(np.abs(Z[q].T @ Z[q] - np.eye(dim)).max())
        res = float(np.abs(A @ Z[q] - Z[q] * lam[q]).max() / scale)
        worst["eigenvalues"] = max(worst["eigenvalues"], ev)
        worst["orthonormality"] = max(worst["orthonormality"], orth)
        worst["residual"] = max(worst["residual"], res)
        rec.update(eigenvalues=max(rec.get("eigenvalues", 0.0), ev), orthonormality=max(rec.get("orthonormality", 0.0), orth),
                   residual=max(rec.get("residual", 0.0), res))
    rows.append(rec)
    print(json.dumps(rec), file=sys.stderr, flush=True)
ok = worst["eigenvalues"] <= 2e-13 and worst["orthonormality"] <= 5e-12 and worst["residual"] <= 1e-12
print(json.dumps({"cases": cases, "seed": seed, "worst": worst, "within_the_suite_tolerances": ok, "rows": rows}, indent=1))
sys.exit(0 if ok else 1)
