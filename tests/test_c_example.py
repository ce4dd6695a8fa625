"""The C-ABI driven from plain C (examples/scan_from_c.c): it must compile against include/crm_hip.h
and link against libcrm_hip.so everywhere; on a GPU box its output must equal the Python host's."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "scan_from_c")
    lib_dir = os.path.join(ROOT, "cellregmap_amd")
    if not os.path.exists(os.path.join(lib_dir, "libcrm_hip.so")):
        from cellregmap_amd import build

        build.build()
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "scan_from_c.c"), "-L" + lib_dir, "-lcrm_hip",
                    "-Wl,-rpath," + lib_dir, "-o", exe], check=True)
    return exe


def test_c_example_compiles_and_links(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 2 and "usage" in out.stderr


@pytest.mark.gpu
def test_c_example_matches_the_python_host(tmp_path):
    from cellregmap_amd import CellRegMap, GenotypePanel
    from cellregmap_amd.synth import make_cohort

    c = make_cohort(10, 20, 4, 24, seed=77)
    n, k0 = c.E.shape
    blob = np.concatenate([np.array([n, k0, c.hK.shape[1], c.W.shape[1], c.G.shape[1]], float), c.y.ravel(),
                           c.E.ravel(), c.hK.ravel(), c.W.ravel(), c.G.ravel()])
    path = tmp_path / "cohort.bin"
    blob.astype(np.float64).tofile(path)
    out = subprocess.run([_build(tmp_path), str(path)], capture_output=True, text=True, check=True)
    got = np.array([[float(x) for x in line.split()] for line in out.stdout.strip().splitlines()])
    # The C driver makes the minimal sequence of calls (context, background, gene, panel, scan) and does not announce the
    # donor structure of the kinship factor, which the Python host finds by itself (crm_background_set_kinship_groups: the
    # same statistics through another order of summation): the same route on both sides for the bit-for-bit comparison,
    # the host's default route at the north-star tolerance
    from cellregmap_amd import _engine, _lib

    lib, ctx = _lib.load(), _engine._context(0)
    crm = CellRegMap(c.y, c.E, W=c.W, hK=c.hK)
    panel = GenotypePanel(c.G, groups=None)
    _lib.check(lib.crm_test_set_kinship_route(ctx, 0))
    try:
        pv, info = crm.scan_interaction(panel, progress=False)
    finally:
        _lib.check(lib.crm_test_set_kinship_route(ctx, 1))
    assert np.array_equal(got[:, 0], pv)
    assert np.array_equal(got[:, 1], info["rho1"])
    pv_host, info_host = crm.scan_interaction(panel, progress=False)
    assert np.array_equal(info_host["rho1"], info["rho1"])
    assert np.all(np.abs(pv_host - pv) <= 1e-5 * pv + 1e-13)
