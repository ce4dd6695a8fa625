"""CPU-only study behind the round-6 CRM_MODEL_FLAT_OPTIMUM rule: the oracle's Brent search instrumented so that every
decision that depends on objective VALUES leaves its margin behind (comparisons of the bracketing phase and of localmin,
the sign of a sub-tolerance parabolic step), run on the fuzz stream next to a second run whose objective carries relative
noise eps.  Prints, per threshold theta on the smallest margin of the (variant, rho*) fit, the share of scans the rule
would flag and how many scans whose Q moved by more than 1e-6 between the two runs it would miss.
    python tools/diag/brent_margins_cpu.py [count 150] [seed 2026] [eps 2e-15]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_cases import build_case, fuzz_cases  # noqa: E402
from oracle import brent  # noqa: E402
from oracle import lmm as olmm  # noqa: E402
from oracle.crm import OracleCellRegMap  # noqa: E402

GOLDEN = brent.GOLDEN
REC = {"cmp": np.inf, "sign": np.inf, "evals": 0}


def note(kind, margin):
    if margin < REC[kind]:
        REC[kind] = margin


def minimize(f, a=-np.inf, b=np.inf, rtol=1e-6, atol=1e-6):
    """oracle/brent.py: minimize, statement for statement, plus the margins."""
    REC["cmp"], REC["sign"], REC["evals"] = np.inf, np.inf, 0
    count = [0]

    def g(x):
        count[0] += 1
        return f(x)

    x0 = min(max(brent.START, a), b)
    x1 = min(max(x0 + brent.FIRST_STEP, a), b)
    f0, f1 = g(x0), g(x1)
    note("cmp", abs(f1 - f0))
    if f1 > f0:
        x0, x1, f0, f1 = x1, x0, f1, f0
    lo = hi = None
    for _ in range(brent.MAXITER):
        x2 = min(max(x1 + brent.GROWTH * (x1 - x0), a), b)
        if x2 == x1:
            break
        f2 = g(x2)
        note("cmp", abs(f2 - f1))
        if f2 > f1:
            lo, hi = (x0, x2) if x0 < x2 else (x2, x0)
            break
        x0, f0, x1, f1 = x1, f1, x2, f2
    if lo is None:
        lo, hi = (x0, x1) if x0 < x1 else (x1, x0)
    xm, fm = x1, f1
    # localmin
    a_, b_ = lo, hi
    x0, f0 = xm, fm
    x1 = x2 = x0
    f1 = f2 = f0
    d = e = 0.0
    for _ in range(brent.MAXITER):
        m = 0.5 * (a_ + b_)
        tol = rtol * abs(x0) + atol
        tol2 = 2.0 * tol
        if abs(x0 - m) <= tol2 - 0.5 * (b_ - a_):
            break
        p = q = r = 0.0
        if tol < abs(e):
            r = (x0 - x1) * (f0 - f2)
            q = (x0 - x2) * (f0 - f1)
            p = (x0 - x2) * q - (x0 - x1) * r
            q = 2.0 * (q - r)
            if 0.0 < q:
                p = -p
            q = abs(q)
            r = e
            e = d
        if abs(p) < abs(0.5 * q * r) and q * (a_ - x0) < p and p < q * (b_ - x0):
            d = p / q
            u = x0 + d
            if (u - a_) < tol2 or (b_ - u) < tol2:
                d = tol if x0 < m else -tol
            elif abs(d) < tol:
                # the step is replaced by +-tol: only its SIGN survives; its size in f-units: how far the values that made
                # it would have to move to turn it round (p = 0): |p| / (|x0-x2|^2 + |x0-x1|^2) bounds it from below
                den = abs(x0 - x2) * abs(x0 - x2) + abs(x0 - x1) * abs(x0 - x1) + abs(x0 - x2) * abs(x0 - x1) * 2
                note("sign", abs(p) / den if den > 0 else 0.0)
        else:
            e = (b_ - x0) if x0 < m else (a_ - x0)
            d = GOLDEN * e
        if tol <= abs(d):
            u = x0 + d
        elif 0.0 < d:
            u = x0 + tol
        else:
            u = x0 - tol
        fu = g(u)
        note("cmp", abs(fu - f0))
        if fu <= f0:
            if u < x0:
                b_ = x0
            else:
                a_ = x0
            x2, f2, x1, f1, x0, f0 = x1, f1, x0, f0, u, fu
        else:
            if u < x0:
                a_ = u
            else:
                b_ = u
            note("cmp", abs(fu - f1))
            if fu <= f1 or x1 == x0:
                x2, f2, x1, f1 = x1, f1, u, fu
            else:
                note("cmp", abs(fu - f2))
                if fu <= f2 or x2 == x0 or x2 == x1:
                    x2, f2 = u, fu
    REC["evals"] = count[0]
    return x0, f0, count[0]


class MarginOracle(OracleCellRegMap):
    """null_fit that keeps the margins of the winning grid point's search and the lml gap to the runner-up."""

    def null_fit(self, X, restricted=True):
        best_lml, best_rho, best, second = -np.inf, 0, None, -np.inf
        marg = None
        for rho in self._rho:
            lmm = olmm.LMM(self._y, X, self._qs[rho], restricted=restricted)
            lmm.fit(verbose=False, polish=False)
            val = lmm.lml()
            rec = (REC["cmp"], REC["sign"])
            if val > best_lml:
                second = best_lml
                best_lml, best_rho, best, marg = val, rho, lmm, rec
            elif val > second:
                second = val
        self.margins.append((marg[0] / abs(best_lml), marg[1] / abs(best_lml), (best_lml - second) / abs(best_lml)))
        return best_rho, best, best_lml


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
    eps = float(sys.argv[3]) if len(sys.argv) > 3 else 2e-15
    exact = olmm.LMM._neg_lml_at
    plain_minimize = brent.minimize
    rows = []
    for case in fuzz_cases(count, seed=seed, wide_covariates=True, max_variants=24):
        y, E, W, G, kw, hooks = build_case(case)
        try:
            brent.minimize = minimize
            a = MarginOracle(y, E, W=W, **kw)
            a.margins = []
            pa, ia, sa = a.scan_interaction(G, return_stats=True, **hooks)
            brent.minimize = plain_minimize
            rng = np.random.default_rng(case[0])
            olmm.LMM._neg_lml_at = lambda self, x, _r=rng: exact(self, x) * (1.0 + eps * _r.uniform(-1.0, 1.0))
            try:
                pb, ib, sb = a.scan_interaction(G, return_stats=True, **hooks)
            finally:
                olmm.LMM._neg_lml_at = exact
        except ValueError:
            continue
        finally:
            brent.minimize = plain_minimize
        trF = np.array([np.trace(F) for F in sa["F"]])
        dq = np.abs(sa["Q"] - sb["Q"]) / np.maximum(np.abs(sa["Q"]), trF)
        dp = np.abs(pa - pb) / pa
        dx = np.abs(sa["delta"] - sb["delta"]) / sa["delta"]
        for j in range(G.shape[1]):
            rows.append((*a.margins[j], dq[j], dp[j], dx[j], float(ia["rho1"][j] == ib["rho1"][j])))
    r = np.array(rows)
    out = {"scans": int(r.shape[0]), "eps": eps, "moved_beyond_1e-6_on_Q": int((r[:, 3] > 1e-6).sum()),
           "moved_beyond_1e-5_on_p": int((r[:, 4] > 1e-5).sum()), "rho_flips": int((r[:, 6] == 0).sum()),
           "cmp_margin_percentiles": [float(v) for v in np.percentile(r[:, 0], [1, 5, 25, 50, 75, 95])]}
    bad = (r[:, 3] > 1e-6) | (r[:, 4] > 1e-5) | (r[:, 6] == 0)
    for theta in (2e-15, 4e-15, 8e-15, 1.6e-14, 3e-14, 1e-13, 3e-13):
        flag = (r[:, 0] < theta) | (r[:, 1] < theta) | (r[:, 2] < theta)
        flag_cmp = r[:, 0] < theta
        out["theta_%g" % theta] = {"flagged_share": float(flag.mean()), "missed": int((bad & ~flag).sum()),
                                   "flagged_share_cmp_only": float(flag_cmp.mean()), "missed_cmp_only": int((bad & ~flag_cmp).sum())}
    print(json.dumps(out, indent=1))
    np.save("/tmp/brent_margins_cpu.npy", r)


if __name__ == "__main__":
    main()
