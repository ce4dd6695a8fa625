"""The ``nccl`` (= RCCL) process group carrying this code's collectives on hardware: world size 1 is what a one-GPU box
can offer, and it is the same code path the driver's 8-GPU run takes -- exchange buffers as CUDA tensors handed to the
library by ``data_ptr``, ``all_reduce`` / ``all_gather`` on torch's NCCL stream, the bench's canary, the final gathers.
A fresh child process per case (never a re-exec of the pytest process, which has initialised the GPU).
Results against the oracle and against a plain one-process build (cellregmap/_cellregmap.py:340)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["C-thin", "C-eigh", "B"])
def test_nccl_group_of_one_carries_exchange_and_gathers(mode, tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "nccl_world_one.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "workers", "nccl_world_one.py"), mode, out]
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    got = np.load(out)

    import cellregmap_amd as crm
    from cellregmap_amd.synth import make_cohort
    from oracle import crm as ocrm

    donors, cells, k, p = (12, 20, 4, 37) if mode != "C-eigh" else (12, 10, 10, 21)
    c = make_cohort(donors, cells, k, p, seed=31)
    n = c.y.size
    if mode == "B":
        kw, okw = dict(hK=c.hK), dict(hK=c.hK)
    else:
        kw, okw = dict(Ls=crm.get_L_values(c.hK, c.E)), dict(Ls=ocrm.khatri_rao_halves(c.hK, c.E))
    one = crm.CellRegMap(c.y, c.E, W=c.W, **kw)
    # every slot went out into a CUDA tensor, through RCCL's all_gather and back in: the spectra are the solver's
    spectra = [one._bg.read(i, n)[1] for i in range(11)]
    assert int(got["exchanged_bytes"]) > 0
    assert np.array_equal(got["ranks"], [s.size for s in spectra])
    np.testing.assert_allclose(got["spectra"], np.concatenate(spectra), rtol=1e-9, atol=1e-12)
    # the scans: equal to the single-process scan (same device, same launches), and within the north star of the oracle
    pv1, info1 = one.scan_interaction(c.G)
    assert np.array_equal(got["rho1"], info1["rho1"])
    assert np.all(np.abs(got["pv"] - pv1) <= 1e-7 * pv1 + 1e-15), np.c_[got["pv"], pv1]
    opv, oinfo = ocrm.OracleCellRegMap(c.y, c.E, W=c.W, **okw).scan_interaction(c.G)
    assert np.array_equal(got["rho1"], oinfo["rho1"])
    assert np.all(np.abs(got["pv"] - opv) <= 1e-5 * opv + 1e-13), np.c_[got["pv"], opv]
    rng = np.random.default_rng(3)
    Y = np.stack([c.y, rng.permutation(c.y), rng.normal(size=n)], axis=1)
    assert got["pvm"].shape == (3, p)
    for i in range(3):
        opv, _ = ocrm.OracleCellRegMap(Y[:, i], c.E, W=c.W, **okw).scan_interaction(c.G)
        assert np.all(np.abs(got["pvm"][i] - opv) <= 1e-5 * opv + 1e-13), (i, np.c_[got["pvm"][i], opv])
