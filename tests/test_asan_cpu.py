"""Host-side logic of libcrm_hip.so without a GPU: the divide-and-conquer deflation plan (C++, eigh_dc.hip) against
the numpy prototype of the same algorithm (tools/eigh_prototype.py), and the same calls plus the no-GPU error paths
under AddressSanitizer (host-only build, ``python -m cellregmap_amd.build --asan``; GPU ASan is not available on
this pool)."""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _plan(lib, lam, z, n1, beta):
    n = lam.size
    k, nrot, rho = ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
    rows = np.zeros(n, np.int32)
    dl, w, rots = np.zeros(n), np.zeros(n), np.zeros(4 * n)
    vp = ctypes.c_void_p
    lib.crm_test_dc_plan.restype = ctypes.c_int
    lib.crm_test_dc_plan.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_int),
                                     ctypes.POINTER(ctypes.c_double), vp, vp, vp, ctypes.POINTER(ctypes.c_int), vp]
    rc = lib.crm_test_dc_plan(lam.ctypes.data, z.ctypes.data, n1, n, float(beta), ctypes.byref(k), ctypes.byref(rho),
                              rows.ctypes.data, dl.ctypes.data, w.ctypes.data, ctypes.byref(nrot), rots.ctypes.data)
    assert rc == 0
    return k.value, rho.value, rows, dl, w, rots[: 4 * nrot.value].reshape(-1, 4)


def _cases():
    rng = np.random.default_rng(0)
    out = []
    for trial in range(40):
        n1, n2 = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        lam = np.concatenate([np.sort(rng.normal(size=n1)), np.sort(rng.normal(size=n2))])
        z = rng.normal(size=n1 + n2)
        kind = trial % 4
        if kind == 1:      # clusters: repeated eigenvalues across and inside the halves
            lam = np.round(lam * 2) / 2
        elif kind == 2:    # tiny components of the update vector
            z[rng.random(z.size) < 0.4] *= 1e-18
        elif kind == 3:    # nearly equal poles
            lam[n1:] = lam[:n1][rng.integers(0, n1, size=n2)] + 1e-17 * rng.normal(size=n2)
        out.append((lam, z, n1, float(rng.normal()) if trial % 7 else 0.0))
    return out


def test_deflation_plan_matches_the_prototype():
    from cellregmap_amd import _lib
    from eigh_prototype import merge

    lib = _lib.load()
    for lam, z, n1, beta in _cases():
        k, rho, rows, dl, w, rots = _plan(lib, lam, z, n1, beta)
        ref = merge(lam[:n1], lam[n1:], z[:n1], z[n1:], beta) if beta != 0.0 else None
        if ref is None:
            assert k == 0          # rho = 0: everything deflates
            continue
        assert k == ref["k"]
        assert list(rows[:k]) == ref["nondefl"] and sorted(rows[k:]) == sorted(ref["defl"])
        assert len(rots) == len(ref["rotations"])
        for got, want in zip(rots, ref["rotations"]):
            assert (int(got[0]), int(got[1])) == (want[0], want[1])
            np.testing.assert_allclose(got[2:], want[2:], rtol=1e-14, atol=1e-300)
        if k:
            np.testing.assert_allclose(dl[:k], ref["dl"], rtol=1e-15, atol=1e-300)
            np.testing.assert_allclose(w[:k], ref["w"], rtol=1e-13)
            assert np.all(np.diff(dl[:k]) > 0)
            nrm2 = float(np.sum(ref["w"] ** 2))
            assert abs(nrm2 - 1.0) < 1e-13 and rho > 0.0


def test_host_side_under_address_sanitizer(tmp_path):
    from cellregmap_amd import build

    lib = build.build_asan(verbose=False)
    rt = subprocess.run([build.HIPCC, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(rt):
        import pytest

        pytest.skip("AddressSanitizer runtime of the ROCm clang not found")
    script = tmp_path / "drive.py"
    script.write_text(f"""
import ctypes, sys
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
import numpy as np
from test_asan_cpu import _plan, _cases
lib = ctypes.CDLL({lib!r})
for lam, z, n1, beta in _cases():
    _plan(lib, lam, z, n1, beta)
# no-GPU error paths: loud, with a message, nothing dereferenced
h = ctypes.c_void_p()
lib.crm_last_error.restype = ctypes.c_char_p
assert lib.crm_ctx_create(0, ctypes.byref(h)) != 0 and lib.crm_last_error()
assert lib.crm_ctx_create(0, None) != 0
assert lib.crm_scan_interaction(None, None, 0, 0, None, None, *([None] * 11)) != 0
assert lib.crm_background_complete(None, None) != 0 and lib.crm_background_seal(None) != 0
print("asan drive ok")
""")
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "asan drive ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
