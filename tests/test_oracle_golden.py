"""Drift detection for the CPU oracle: the committed end-to-end goldens (tests/golden/e2e_golden.npz, written by
tests/golden/make_e2e_golden.py from the oracle itself) must be reproduced; if tests/golden/e2e_reference.npz
exists (outputs of the real cellregmap package on the same inputs, tests/golden/regen_with_reference.py), the
oracle is compared with the reference itself at the north-star tolerances."""
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose

from conftest import GOLDEN


def _cases():
    gold = np.load(os.path.join(GOLDEN, "e2e_golden.npz"))
    names = sorted({k.split("/")[0] for k in gold.files})
    return gold, names


def _inputs(gold, name):
    return {k.split("/", 1)[1]: gold[k] for k in gold.files if k.startswith(name + "/")}


def oracle_kwargs(g):
    from oracle.crm import khatri_rao_halves

    mode = str(g["mode"])
    if mode == "B":
        return {"hK": g["hK"]}
    if mode == "C":
        return {"Ls": khatri_rao_halves(g["hK"], g["E"])}
    return {}


@pytest.mark.parametrize("name", _cases()[1])
def test_oracle_reproduces_its_goldens(name):
    from oracle.crm import OracleCellRegMap

    gold, _ = _cases()
    g = _inputs(gold, name)
    pv, info, st = OracleCellRegMap(g["y"], g["E"], W=g["W"], **oracle_kwargs(g)).scan_interaction(g["G"], return_stats=True)
    assert np.array_equal(info["rho1"], g["rho1"])
    # (another BLAS build / thread count may round the likelihood differently: Brent-path level agreement)
    assert_allclose(st["lml"], g["lml"], rtol=1e-11)
    assert_allclose(st["delta"], g["delta"], rtol=5e-6)
    assert_allclose(st["Q"], g["Q"], rtol=1e-6)
    assert np.all(np.abs(pv - g["pv"]) <= 1e-5 * g["pv"] + 1e-13)
    for k in ("e2", "g2", "eps2"):
        assert_allclose(info[k], g[k], rtol=1e-5, atol=1e-9)


def test_oracle_against_the_reference_package_when_its_outputs_are_committed():
    path = os.path.join(GOLDEN, "e2e_reference.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/e2e_reference.npz not present: cellregmap==0.0.3 cannot be installed in this image "
                    "(run tests/golden/regen_with_reference.py where it can)")
    ref = np.load(path)
    gold, names = _cases()
    for name in names:
        assert np.array_equal(ref[f"{name}/rho1"], gold[f"{name}/rho1"]), name
        assert np.all(np.abs(ref[f"{name}/pv"] - gold[f"{name}/pv"]) <= 1e-5 * ref[f"{name}/pv"] + 1e-13), name
