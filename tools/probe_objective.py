"""The null-fit objective itself, device vs oracle, at the SAME points x = logit(delta) (test hook
crm_test_null_fit_probe), on the problems of the fuzz stream whose score statistics differ most under the verbatim
procedure: is it the likelihood's value that differs (and by how much), or only where the two Brent searches stop?
Oracle bound to the device's decomposition.  GPU only.   python tools/probe_objective.py [count 400] [seed 2026] [top 12]"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_cases import build_case, fuzz_cases  # noqa: E402
from test_gpu_fuzz import _oracle_on_device_decomposition  # noqa: E402

from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib  # noqa: E402
from oracle.lmm import LMM  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lib, ctx = _lib.load(), _engine._context(0)

found = []
for case in fuzz_cases(count, seed=seed, wide_covariates=True):
    if case[3] > 8:
        continue                      # the probe is built into the register kernels (c <= 8)
    y, E, W, G, kw, hooks = build_case(case)
    crm = CellRegMap(y, E, W=W, **kw)
    try:
        o = _oracle_on_device_decomposition(crm, y, E, W, False)
        opv, oinfo, ost = o.scan_interaction(G, return_stats=True, **hooks)
    except ValueError:
        continue
    pv, info, st = crm.scan_interaction(GenotypePanel(G, groups=None), return_stats=True, progress=False, **hooks)
    same = info["rho1"] == oinfo["rho1"]
    trF = np.array([np.trace(F) for F in ost["F"]])
    dq = np.where(same, np.abs(st["Q"] - ost["Q"]) / np.maximum(np.abs(ost["Q"]), trF), 0.0)
    found.append((float(dq.max()), int(np.argmax(dq)), case))
found.sort(key=lambda t: -t[0])
picked = found[:top] + found[len(found) // 2: len(found) // 2 + 3]      # the worst ones and three typical ones
report = []
xs = [-6.0, -4.0, -3.0, -2.0, -1.0, -0.5, 0.0, 0.5, 1.0, 2.0, 3.0, 4.0, 6.0]
for dqmax, j, case in picked:
    y, E, W, G, kw, hooks = build_case(case)
    crm = CellRegMap(y, E, W=W, **kw)
    o = _oracle_on_device_decomposition(crm, y, E, W, False)
    panel = GenotypePanel(G[:, [j]], groups=None)
    X = np.concatenate((W, G[:, [j]]), axis=1)
    nrho = len(o._rho)
    worst = {"lml": 0.0, "scale": 0.0}
    per_x = []
    for x in xs:
        _lib.check(lib.crm_test_null_fit_probe(ctx, 1, x))
        try:
            crm.scan_interaction(panel, progress=False)
            buf = np.empty(2 * nrho)
            got = lib.crm_test_null_fit_probe_read(ctx, _lib.ptr(buf), buf.size)
            assert got == 2 * nrho, got
        finally:
            _lib.check(lib.crm_test_null_fit_probe(ctx, 0, 0.0))
        dev = buf.reshape(nrho, 2)
        rel = []
        for i, rho in enumerate(o._rho):
            lm = LMM(y, X, o._qs[rho], restricted=True)
            ref = -lm._neg_lml_at(x)
            rel.append((abs(dev[i, 0] - ref) / abs(ref), abs(dev[i, 1] - lm.scale) / lm.scale))
        rel = np.array(rel)
        per_x.append({"x": x, "worst_rel_lml_over_rho": float(rel[:, 0].max()), "worst_rel_scale_over_rho": float(rel[:, 1].max())})
        worst["lml"] = max(worst["lml"], float(rel[:, 0].max()))
        worst["scale"] = max(worst["scale"], float(rel[:, 1].max()))
    report.append({"case": [v if isinstance(v, str) else int(v) for v in case], "variant": j, "rel_dQ_verbatim": dqmax,
                   "objective_worst_rel_lml": worst["lml"], "objective_worst_rel_scale": worst["scale"], "per_x": per_x})
    print(case, "dQ %.2e   objective: lml %.2e scale %.2e" % (dqmax, worst["lml"], worst["scale"]), file=sys.stderr, flush=True)
print(json.dumps({"what": "null-fit objective at fixed x, device vs oracle (oracle on the device's Q0, S0); case = (index, n, k0, c, p, "
                          "donors, mode, hook)", "cases": report}, indent=0))
