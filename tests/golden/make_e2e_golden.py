"""End-to-end golden vectors of the interaction scan, produced by THIS repo's CPU oracle (oracle/crm.py):
inputs and (rho*, delta, v0 rho*, v0 (1 - rho*), v1, lml, Q, eigenvalues of F, p) for the three background
modes and the eigh branch, n <= 500.  They pin the oracle against drift (tests/test_oracle_golden.py) and the
device against a frozen target (tests/test_gpu_golden.py).

They are NOT outputs of the reference package: cellregmap cannot be imported in this image (glimix-core,
numpy-sugar, chiscore missing).  tests/golden/regen_with_reference.py reruns the same inputs
through the real package wherever ``pip install cellregmap==0.0.3`` works and reports the differences.

    python tests/golden/make_e2e_golden.py        # rewrites tests/golden/e2e_golden.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from cellregmap_amd.synth import make_cohort  # noqa: E402
from oracle.crm import OracleCellRegMap, khatri_rao_halves  # noqa: E402

CASES = {
    # name: (donors, cells per donor, contexts, variants, seed, mode)
    "A": (10, 20, 5, 24, 5, "A"),
    "B": (10, 20, 5, 24, 5, "B"),
    "C_thin": (6, 40, 4, 24, 4, "C"),       # 4 + 4 * 6 = 28 columns < 240 cells: thin SVD branch
    "C_eigh": (12, 10, 10, 24, 3, "C"),     # 10 + 10 * 12 = 130 columns >= 120 cells: eigh branch
    "cfg1": (50, 10, 10, 32, 20, "C"),      # BASELINE config 1's cohort (500 cells), first 32 variants
}


def run(name):
    donors, cells, k, p, seed, mode = CASES[name]
    c = make_cohort(donors, cells, k, p, seed=seed)
    rng = np.random.default_rng(seed)
    W = np.concatenate([c.W, rng.normal(size=(c.y.size, 1))], axis=1) if name in ("B", "C_thin") else c.W
    kw = {}
    if mode == "B":
        kw["hK"] = c.hK
    elif mode == "C":
        kw["Ls"] = khatri_rao_halves(c.hK, c.E)
    pv, info, st = OracleCellRegMap(c.y, c.E, W=W, **kw).scan_interaction(c.G, return_stats=True)
    lam = np.stack([np.linalg.eigvalsh(F) for F in st["F"]])
    out = {"y": c.y, "E": c.E, "W": W, "hK": c.hK, "G": c.G, "mode": np.array(mode),
           "pv": pv, "rho1": info["rho1"], "e2": info["e2"], "g2": info["g2"], "eps2": info["eps2"],
           "Q": st["Q"], "delta": st["delta"], "lml": st["lml"], "scale": st["scale"], "lambda": lam}
    return {f"{name}/{k}": v for k, v in out.items()}


if __name__ == "__main__":
    blob = {}
    for name in CASES:
        blob.update(run(name))
        print(name, "p range", blob[f"{name}/pv"].min(), blob[f"{name}/pv"].max(), "rho*", np.unique(blob[f"{name}/rho1"]))
    np.savez_compressed(os.path.join(HERE, "e2e_golden.npz"), **blob)
    print("wrote", os.path.join(HERE, "e2e_golden.npz"), os.path.getsize(os.path.join(HERE, "e2e_golden.npz")), "bytes")
