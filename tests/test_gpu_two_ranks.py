"""Two real processes on the one GPU of the test box: the rho-sharded constructor exchanging device buffers, shard-only
panels and the final gather -- the N > 1 path of ``bench.py --gpus N`` with gloo in place of RCCL (RCCL does not
take two ranks on one device; the 8-GPU run is the driver's).  Results against the oracle and a one-process run."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["C-thin", "C-eigh", "B", "C-thin-host"])
def test_two_ranks_share_one_gpu(mode, tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "two_ranks.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "workers", "two_ranks_one_gpu.py"), mode, out]
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    got = np.load(out)
    mode = mode.replace("-host", "")      # (exchange through CPU tensors: same results expected)

    import cellregmap_amd as crm
    from cellregmap_amd.synth import make_cohort
    from oracle import crm as ocrm

    donors, cells, k, p = (12, 20, 4, 37) if mode != "C-eigh" else (12, 10, 10, 21)   # eigh: k + k*donors >= n
    c = make_cohort(donors, cells, k, p, seed=31)
    n = c.y.size
    if mode == "B":
        kw, okw = dict(hK=c.hK), dict(hK=c.hK)
    else:
        kw, okw = dict(Ls=crm.get_L_values(c.hK, c.E)), dict(Ls=ocrm.khatri_rao_halves(c.hK, c.E))
    one = crm.CellRegMap(c.y, c.E, W=c.W, **kw)
    # the constructor: same ranks, same spectra as the one-process build (each grid point comes from ONE rank's solver)
    spectra = [one._bg.read(i, n)[1] for i in range(11)]
    assert np.array_equal(got["ranks"], [s.size for s in spectra])
    np.testing.assert_allclose(got["spectra"], np.concatenate(spectra), rtol=1e-9, atol=1e-12)
    # the scan: against the oracle (north-star tolerances) and against one process (other launch shapes: same tolerance)
    opv, oinfo = ocrm.OracleCellRegMap(c.y, c.E, W=c.W, **okw).scan_interaction(c.G)
    assert np.all(np.abs(got["pv"] - opv) <= 1e-5 * opv + 1e-13), np.c_[got["pv"], opv]
    pv1, info1 = one.scan_interaction(c.G)
    assert np.all(np.abs(got["pv"] - pv1) <= 1e-5 * pv1 + 1e-13)
    rng = np.random.default_rng(3)
    Y = np.stack([c.y, rng.permutation(c.y), rng.normal(size=n)], axis=1)
    assert got["pvm"].shape == (3, p)
    for i in range(3):
        opv, _ = ocrm.OracleCellRegMap(Y[:, i], c.E, W=c.W, **okw).scan_interaction(c.G)
        assert np.all(np.abs(got["pvm"][i] - opv) <= 1e-5 * opv + 1e-13), (i, np.c_[got["pvm"][i], opv])
