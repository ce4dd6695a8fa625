"""Reference-side pins for rho*, v0, v1, Q-independent outputs and the Davies p-value -- to be run on any machine
where ``pip install cellregmap==0.0.3`` (with glimix-core >= 3.1.12, numpy-sugar, chiscore) works; this image has
no network and cannot, so the goldens under tests/golden/ come from this repo's restatement.

    python tests/golden/regen_with_reference.py [--write]

Loads the inputs of tests/golden/e2e_golden.npz, runs them through the REAL package (``CellRegMap(...)
.scan_interaction(G)`` -- the same constructor arguments the oracle got) and prints, per case, the largest
differences against the stored oracle outputs.  With ``--write`` the reference's outputs go to
tests/golden/e2e_reference.npz (commit that file: tests/test_oracle_golden.py and tests/test_gpu_golden.py
pick it up and then compare against the reference itself).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    try:
        from cellregmap import CellRegMap
        from numpy_sugar import ddot
        from numpy_sugar.linalg import economic_svd
    except ImportError as e:  # pragma: no cover
        raise SystemExit(f"the reference package is not importable here ({e}); run this where cellregmap==0.0.3 is installed")
    gold = np.load(os.path.join(HERE, "e2e_golden.npz"))
    names = sorted({k.split("/")[0] for k in gold.files})
    out = {}
    for name in names:
        g = {k.split("/", 1)[1]: gold[k] for k in gold.files if k.startswith(name + "/")}
        mode = str(g["mode"])
        kw = {}
        if mode == "B":
            kw["hK"] = g["hK"]
        elif mode == "C":   # get_L_values(hK, E), cellregmap/_cellregmap.py:533-545
            U, S, _ = economic_svd(g["E"])
            us = U * S
            kw["Ls"] = [ddot(us[:, i], g["hK"]) for i in range(us.shape[1])]
        crm = CellRegMap(g["y"], g["E"], W=g["W"], **kw)
        pv, info = crm.scan_interaction(g["G"])
        pv = np.asarray(pv, float)
        same = np.asarray(info["rho1"]) == g["rho1"]
        print(f"{name}: rho* equal on {same.sum()} / {same.size} variants; max rel dp {np.max(np.abs(pv - g['pv']) / g['pv']):.3g}; "
              f"max rel d(eps2) {np.max(np.abs(info['eps2'] - g['eps2']) / g['eps2']):.3g}")
        out.update({f"{name}/pv": pv, **{f"{name}/{k}": np.asarray(v, float) for k, v in info.items()}})
    if "--write" in sys.argv:
        np.savez_compressed(os.path.join(HERE, "e2e_reference.npz"), **out)
        print("wrote", os.path.join(HERE, "e2e_reference.npz"))


if __name__ == "__main__":
    main()
