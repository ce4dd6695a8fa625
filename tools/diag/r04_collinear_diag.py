"""Round 4 diagnostics: the nearly collinear two-donor problems (fuzz stream 2026, problems 238 / 344) and the variants
built at the reference's rank rule -- per-variant differences to the oracle, the share of each variant outside span(W),
and the null-fit objective at fixed points for the worst ones.  GPU only."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_cases import build_case, fuzz_cases, random_problem  # noqa: E402

from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib  # noqa: E402
from oracle.crm import OracleCellRegMap  # noqa: E402
from oracle.lmm import LMM  # noqa: E402
from oracle.sugar import economic_svd, epsilon  # noqa: E402

lib, ctx = _lib.load(), _engine._context(0)
out = {}
for problem in (238, 344):
    case = [c for c in fuzz_cases(400, seed=2026) if c[0] == problem][0]
    y, E, W, G, kw, hooks = build_case(case)
    crm = CellRegMap(y, E, W=W, **kw)
    o = OracleCellRegMap(y, E, W=W, **kw)
    opv, oinfo, ost = o.scan_interaction(G, return_stats=True, **hooks)
    Qw, _ = np.linalg.qr(W)
    share = np.array([np.sum((g - Qw @ (Qw.T @ g)) ** 2) / np.sum(g * g) for g in G.T])
    rec = {"case": [v if isinstance(v, str) else int(v) for v in case], "share_outside_span_W": share.tolist()}
    for groups in (None, "auto"):
        pv, info, st = crm.scan_interaction(GenotypePanel(G, groups=groups), return_stats=True, **hooks)
        qscale = np.maximum(np.abs(ost["Q"]), [np.trace(F) for F in ost["F"]])
        dq = np.abs(st["Q"] - ost["Q"]) / qscale
        rec["dense" if groups is None else "collapsed"] = {
            "rel_dQ": dq.tolist(), "rel_dlml": (np.abs(st["lml"] - ost["lml"]) / np.abs(ost["lml"])).tolist(),
            "rel_ddelta": (np.abs(st["delta"] - ost["delta"]) / ost["delta"]).tolist(), "delta": st["delta"].tolist()}
        print(problem, groups, "worst dQ %.2e at %d (share %.2e); beyond 1e-6: %s" % (
            dq.max(), dq.argmax(), share[dq.argmax()], np.flatnonzero(dq > 1e-6).tolist()), flush=True)
    # objective at fixed points for the three worst dense variants
    dq = np.array(rec["dense"]["rel_dQ"])
    probes = []
    for j in np.argsort(-dq)[:3]:
        panel = GenotypePanel(G[:, [j]], groups=None)
        X = np.concatenate((W, G[:, [j]]), axis=1)
        nrho = len(o._rho)
        worst = 0.0
        for x in (-6.0, -3.0, -1.0, 0.0, 1.0, 3.0, 6.0, float(np.log(ost["delta"][j] / (1 - ost["delta"][j])))):
            _lib.check(lib.crm_test_null_fit_probe(ctx, 1, x))
            try:
                crm.scan_interaction(panel, progress=False, **hooks)
                buf = np.empty(2 * nrho)
                assert lib.crm_test_null_fit_probe_read(ctx, _lib.ptr(buf), buf.size) == 2 * nrho
            finally:
                _lib.check(lib.crm_test_null_fit_probe(ctx, 0, 0.0))
            dev = buf.reshape(nrho, 2)
            for i, rho in enumerate(o._rho):
                lm = LMM(y, X, o._qs[rho], restricted=True)
                ref = -lm._neg_lml_at(x)
                worst = max(worst, abs(dev[i, 0] - ref) / abs(ref))
        probes.append({"variant": int(j), "rel_dQ": float(dq[j]), "share": float(share[j]), "objective_worst_rel_lml": worst})
        print("  variant %d dQ %.2e share %.2e objective %.2e" % (j, dq[j], share[j], worst), flush=True)
    rec["objective_probes"] = probes
    out[str(problem)] = rec

# the rank-rule variants
y, E, W, G, kw = random_problem(120, 3, 2, 4, 6, seed=11, mode="B")
rng = np.random.default_rng(3)
Qw, _ = np.linalg.qr(W)
u = rng.normal(size=y.size)
u -= Qw @ (Qw.T @ u)
u /= np.linalg.norm(u)
base = W @ np.array([0.7, -0.4])
G = G.copy()
for col, target in ((1, 0.5), (3, 2.0)):
    lo, hi = 1e-12, 1e-4
    for _ in range(200):
        mid = np.sqrt(lo * hi)
        sm = np.linalg.svd(np.c_[W, base + mid * u], compute_uv=False)[-1]
        lo, hi = (mid, hi) if sm < target * epsilon.small else (lo, mid)
    G[:, col] = base + hi * u
crm = CellRegMap(y, E, W=W, **kw)
opv, oinfo, ost = OracleCellRegMap(y, E, W=W, **kw).scan_interaction(G, return_stats=True)
pv, info = crm.scan_interaction_info(GenotypePanel(G, groups=None))
pv2, info2, st = crm.scan_interaction(GenotypePanel(G, groups=None), return_stats=True)
print("flags", info["model_flags"], "\npv dev", pv2, "\npv ora", opv, "\nQ dev", st["Q"], "\nQ ora", ost["Q"], "\nlml dev", st["lml"],
      "\nlml ora", ost["lml"], "\ndelta", st["delta"], ost["delta"])
out["rank_rule"] = {"flags": info["model_flags"].tolist(), "pv": pv2.tolist(), "opv": opv.tolist(), "Q": st["Q"].tolist(),
                    "oQ": ost["Q"].tolist(), "lml": st["lml"].tolist(), "olml": ost["lml"].tolist()}
dest = os.path.join(ROOT, "gpurun_out", "r04_collinear_diag.json")
with open(dest, "w") as fh:
    json.dump(out, fh, indent=1)
