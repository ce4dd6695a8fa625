"""Are the null fits of this build the ones of another build of the library, bit for bit?  Runs the verbatim scans of a fuzz
stream under the library given by CRM_OTHER_LIB (default tools/_r05/libcrm_hip_r05.so: round 5's HEAD, built by
`git worktree add /tmp/r05 d67abc4 && python -m cellregmap_amd.build` there) and under the current one, each in its own
process, and compares delta, lml, scale, rho*, Q and the p-values entry by entry.
    python tools/diag/compare_builds.py [count 150] [seed 2026]         (parent)
    python tools/diag/compare_builds.py --child <lib or ''> <count> <seed> <out.npz>"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(lib_path, count, seed, out):
    from cellregmap_amd import _lib

    if lib_path:
        import ctypes

        _lib.LIB_PATH = lib_path
        probe = ctypes.CDLL(lib_path)     # (an older build lacks the entry points added since: bind what it has)
        for name in [k for k in _lib.SIGNATURES if not hasattr(probe, k)]:
            del _lib.SIGNATURES[name]
    from fuzz_cases import build_case, fuzz_cases

    from cellregmap_amd import CellRegMap, GenotypePanel

    keep = {k: [] for k in ("pv", "rho1", "Q", "lml", "delta", "scale")}
    for case in fuzz_cases(count, seed=seed, wide_covariates=True):
        y, E, W, G, kw, hooks = build_case(case)
        crm = CellRegMap(y, E, W=W, **kw)
        for groups in (None, "auto"):
            pv, info, st = crm.scan_interaction(GenotypePanel(G, groups=groups), return_stats=True, **hooks)
            keep["pv"].append(pv)
            keep["rho1"].append(info["rho1"])
            for k in ("Q", "lml", "delta", "scale"):
                keep[k].append(st[k])
    np.savez(out, **{k: np.concatenate(v) for k, v in keep.items()})


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
    other = os.environ.get("CRM_OTHER_LIB", os.path.join(ROOT, "tools", "_r05", "libcrm_hip_r05.so"))
    outs = []
    this = os.environ.get("CRM_THIS_LIB", "")     # (a diagnostic build instead of the package's own library)
    for tag, lib in (("other", other), ("this", this)):
        out = os.path.join("/tmp", "compare_builds_%s_%d.npz" % (tag, os.getpid()))
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", lib, str(count), str(seed), out])
        outs.append(np.load(out))
    a, b = outs
    rep = {"other_library": other, "this_library": this or "cellregmap_amd/libcrm_hip.so", "problems": count, "seed": seed, "variant_scans": int(a["pv"].size)}
    for k in a.files:
        same = (a[k] == b[k]) | (np.isnan(a[k]) & np.isnan(b[k]))
        rep[k] = {"identical": int(same.sum()), "different": int((~same).sum()),
                  "worst_rel_difference": float(np.nanmax(np.abs(a[k] - b[k]) / np.maximum(np.abs(a[k]), 1e-300))) if (~same).any() else 0.0}
    print(json.dumps(rep, indent=1))
    dest = os.path.join(ROOT, "gpurun_out")
    os.makedirs(dest, exist_ok=True)
    tag = os.path.basename(this).replace("libcrm_hip_", "").replace(".so", "") if this else "package"
    with open(os.path.join(dest, "compare_builds_seed%d_%s.json" % (seed, tag)), "w") as fh:
        json.dump(rep, fh, indent=1)
    return 0 if all(rep[k]["different"] == 0 for k in ("delta", "lml", "scale", "rho1")) else 1


if __name__ == "__main__":
    sys.exit(main())
