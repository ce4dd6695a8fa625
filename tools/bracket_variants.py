"""The one guessed piece of the oracle, turned into a number.

brent-search (the scalar minimiser behind glimix-core's ``LMM.fit``; reference call site cellregmap/_cellregmap.py:352)
is absent from this image and the first step of its downhill bracketing phase is not recoverable here
(oracle/brent.py).  Any bracketing of the same basin hands Brent's ``localmin`` another triple to start from; the
search then stops somewhere else inside its own tolerance (rtol = atol = 1e-6 on logit delta), and Q and the p-value move
with it.  This script reruns the oracle's interaction scan with other plausible bracketing phases -- first step
0.5 / 2 / golden ratio / tolerance-sized, another start, another growth factor -- on the end-to-end goldens and on
problems of the fuzz stream, and records, per variant scan, the envelope of Q and p over the variants relative to the
restatement's choice (start 0, step 1, growth 2).

    python tools/bracket_variants.py [problems 200] [seed 2026] [out.json]      (CPU only)

tests/test_oracle_brackets.py runs a smaller sample of the same code in the CPU suite.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN_RATIO = 1.618033988749895
# name: (start, first step, growth)
VARIANTS = {
    "step 0.5": (0.0, 0.5, 2.0),
    "step 2": (0.0, 2.0, 2.0),
    "step golden ratio": (0.0, GOLDEN_RATIO, 2.0),
    "step -1 (first probe on the other side)": (0.0, -1.0, 2.0),
    "tolerance-sized first step (2e-6), growth 2": (0.0, 2e-6, 2.0),
    "start -0.5": (-0.5, 1.0, 2.0),
    "start +0.5": (0.5, 1.0, 2.0),
    "growth golden ratio": (0.0, 1.0, GOLDEN_RATIO),
    "growth 3": (0.0, 1.0, 3.0),
}


def _scan(make, G, hooks):
    return make().scan_interaction(G, return_stats=True, **hooks)


def problems(count, seed, **limits):
    """(label, builder of the oracle object, G, hooks): the e2e goldens first, then `count` fuzz problems."""
    from fuzz_cases import build_case, fuzz_cases
    from oracle.crm import OracleCellRegMap, khatri_rao_halves

    gold = np.load(os.path.join(ROOT, "tests", "golden", "e2e_golden.npz"))
    for name in sorted({k.split("/")[0] for k in gold.files}):
        g = {k.split("/", 1)[1]: gold[k] for k in gold.files if k.startswith(name + "/")}
        mode = str(g["mode"])
        kw = {"hK": g["hK"]} if mode == "B" else ({"Ls": khatri_rao_halves(g["hK"], g["E"])} if mode == "C" else {})
        yield "golden " + name, mode, (lambda g=g, kw=kw: OracleCellRegMap(g["y"], g["E"], W=g["W"], **kw)), g["G"], {}
    for case in fuzz_cases(count, seed=seed, wide_covariates=False, **limits):
        y, E, W, G, kw, hooks = build_case(case)
        yield "fuzz %d" % case[0], case[6], (lambda y=y, E=E, W=W, kw=kw: OracleCellRegMap(y, E, W=W, **kw)), G, hooks


def run(count=200, seed=2026, **limits):
    from oracle import brent

    saved = (brent.START, brent.FIRST_STEP, brent.GROWTH)
    rows = {name: [] for name in VARIANTS}   # per variant scan: rel dQ, rel dp, same rho*, mode, nfev ratio
    raised = 0
    nprob = 0
    try:
        for label, mode, make, G, hooks in problems(count, seed, **limits):
            try:
                base = _scan(make, G, hooks)
            except ValueError:   # the reference's LMM raises on degenerate variants
                raised += 1
                continue
            nprob += 1
            pb, ib, sb = base
            scale = np.maximum(np.abs(sb["Q"]), [np.trace(F) for F in sb["F"]])
            for name, (start, step, growth) in VARIANTS.items():
                brent.START, brent.FIRST_STEP, brent.GROWTH = start, step, growth
                try:
                    pv, info, st = _scan(make, G, hooks)
                except ValueError:
                    continue
                finally:
                    brent.START, brent.FIRST_STEP, brent.GROWTH = saved
                same = info["rho1"] == ib["rho1"]
                for j in range(G.shape[1]):
                    rows[name].append((abs(st["Q"][j] - sb["Q"][j]) / scale[j], abs(pv[j] - pb[j]) / pb[j], bool(same[j]),
                                       "ABC".index(mode), abs(st["lml"][j] - sb["lml"][j]) / abs(sb["lml"][j])))
    finally:
        brent.START, brent.FIRST_STEP, brent.GROWTH = saved
    out = {"what": "oracle vs oracle: the interaction scan with other bracketing phases before Brent's localmin, relative to "
                   "start 0 / first step 1 / growth 2 (oracle/brent.py)",
           "problems": nprob, "seed": seed, "oracle_raised": raised, "limits": limits, "variants": {}}
    env_q, env_p = None, None
    for name, r in rows.items():
        a = np.array(r, float)
        same = a[:, 2] > 0
        q, p = a[same, 0], a[same, 1]
        out["variants"][name] = {
            "start_step_growth": VARIANTS[name], "variant_scans": int(a.shape[0]), "rho_star_differs": int((~same).sum()),
            "worst_rel_lml_where_rho_differs": float(a[~same, 4].max()) if (~same).any() else 0.0,
            "worst_rel_Q": float(q.max()), "median_rel_Q": float(np.median(q)),
            "worst_rel_p": float(p.max()), "share_Q_beyond_1e-6": float((q > 1e-6).mean()),
            "share_p_beyond_1e-5": float((p > 1e-5).mean()), "count_p_beyond_1e-5": int((p > 1e-5).sum()),
            "share_Q_beyond_1e-6_by_mode": {m: float((a[same & (a[:, 3] == k), 0] > 1e-6).mean()) if (same & (a[:, 3] == k)).any() else 0.0
                                            for k, m in enumerate("ABC")}}
        qq = np.where(same, a[:, 0], 0.0)
        pp = np.where(same, a[:, 1], 0.0)
        env_q = qq if env_q is None else np.maximum(env_q, qq)
        env_p = pp if env_p is None else np.maximum(env_p, pp)
    out["envelope_over_variants"] = {
        "variant_scans": int(env_q.size), "worst_rel_Q": float(env_q.max()), "worst_rel_p": float(env_p.max()),
        "share_Q_beyond_1e-6": float((env_q > 1e-6).mean()), "share_p_beyond_1e-5": float((env_p > 1e-5).mean()),
        "count_p_beyond_1e-5": int((env_p > 1e-5).sum()),
        "quantiles_rel_Q": {str(q): float(np.quantile(env_q, q)) for q in (0.5, 0.9, 0.99, 0.999)},
        "quantiles_rel_p": {str(q): float(np.quantile(env_p, q)) for q in (0.5, 0.9, 0.99, 0.999)}}
    return out


if __name__ == "__main__":
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
    res = run(count, seed)
    text = json.dumps(res, indent=1)
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as fh:
            fh.write(text + "\n")
    print(text)
