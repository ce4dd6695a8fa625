"""The HIP path against the committed end-to-end goldens (tests/golden/e2e_golden.npz; and against the reference
package's own outputs when tests/golden/e2e_reference.npz has been committed)."""
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _load():
    gold = np.load(os.path.join(GOLDEN, "e2e_golden.npz"))
    return gold, sorted({k.split("/")[0] for k in gold.files})


@pytest.mark.parametrize("name", _load()[1])
def test_device_matches_the_goldens(name):
    from cellregmap_amd import CellRegMap, GenotypePanel, get_L_values

    gold, _ = _load()
    g = {k.split("/", 1)[1]: gold[k] for k in gold.files if k.startswith(name + "/")}
    mode = str(g["mode"])
    kw = {"hK": g["hK"]} if mode == "B" else ({"Ls": get_L_values(g["hK"], g["E"])} if mode == "C" else {})
    crm = CellRegMap(g["y"], g["E"], W=g["W"], **kw)
    for groups in (None, "auto"):
        pv, info, st = crm.scan_interaction(GenotypePanel(g["G"], groups=groups), return_stats=True)
        assert np.array_equal(info["rho1"], g["rho1"])
        assert_allclose(st["lml"], g["lml"], rtol=1e-11)
        assert_allclose(st["Q"], g["Q"], rtol=1e-6)
        assert np.all(np.abs(pv - g["pv"]) <= 1e-5 * g["pv"] + 1e-13)
        for k in ("e2", "g2", "eps2"):
            assert_allclose(info[k], g[k], rtol=1e-5, atol=1e-9)
        lam = st["lambda"]
        assert np.abs(lam - g["lambda"]).max() <= 1e-6 * np.abs(g["lambda"]).max()
    ref_path = os.path.join(GOLDEN, "e2e_reference.npz")
    if os.path.exists(ref_path):
        ref = np.load(ref_path)
        assert np.array_equal(info["rho1"], ref[f"{name}/rho1"])
        assert np.all(np.abs(pv - ref[f"{name}/pv"]) <= 1e-5 * ref[f"{name}/pv"] + 1e-13)
