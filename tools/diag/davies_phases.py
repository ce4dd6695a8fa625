"""Where eig_davies_kernel (csrc/davies.hip) spends its time at k0 contexts: the same 4096 matrices through
crm_test_eigvalsh (reduction + bisection + Davies at Q = 0) and their eigenvalues through crm_test_davies (Davies alone, at
null-like Q); run under  rocprofv3 --kernel-trace  and read the per-dispatch durations with tools/diag/davies_phases.sh.
    python tools/diag/davies_phases.py [k0 50] [count 4096]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cellregmap_amd import _engine, _lib  # noqa: E402

args = [a for a in sys.argv[1:] if a != "--stamps"]
k = int(args[0]) if len(args) > 0 else 50
count = int(args[1]) if len(args) > 1 else 4096
if "--stamps" in sys.argv:     # the build of tools/diag/build_davies_stamps.sh: phase durations instead of eigenvalues
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "_r05", "libcrm_hip_davies_stamps.so")
rng = np.random.default_rng(3)
A = rng.standard_normal((count, k, 4 * k)) * 10 ** rng.uniform(-1, 1, (count, 1, 4 * k))
F = A @ A.transpose(0, 2, 1) / (4 * k)
lib = _lib.load()
ctx = _engine._context(0)
lam = np.empty((count, k))
for _ in range(3):
    _lib.check(lib.crm_test_eigvalsh(ctx, count, k, _lib.ptr(F), _lib.ptr(lam)))
if "--stamps" in sys.argv:
    t = lam[:, :3] * 0.01      # us
    start = (lam[:, 3] - lam[:, 3].min()) * 0.01
    if k >= 5:
        print("  of the reduction, loading the matrix: %.0f / %.0f us" % (np.median(lam[:, 4]) * 0.01, lam[:, 4].max() * 0.01))
    print("per wavefront, us (median / max): reduction %.0f / %.0f, bisection %.0f / %.0f, Davies %.0f / %.0f; start of the "
          "last wavefront %.0f us after the first" % (np.median(t[:, 0]), t[:, 0].max(), np.median(t[:, 1]), t[:, 1].max(),
                                                   np.median(t[:, 2]), t[:, 2].max(), start.max()))
    sys.exit(0)
ref = np.linalg.eigvalsh(F[:64])
print("eigenvalues vs numpy, max |d| / |T|:", np.abs(lam[:64] - ref).max() / np.abs(ref).max())
Q = (lam * rng.chisquare(1, lam.shape)).sum(1)
pv = np.empty(count)
ifault = np.empty(count, dtype=np.int32)
liu = np.empty(count)
for _ in range(3):
    _lib.check(lib.crm_test_davies(ctx, count, k, _lib.ptr(Q), _lib.ptr(lam), _lib.ptr(pv), _lib.ptr(ifault), _lib.ptr(liu)))
print("p-values: median %.3f, faults %d" % (np.median(pv), int((ifault != 0).sum())))
