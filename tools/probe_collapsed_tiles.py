"""Donor-collapsed scan at config 3 (4096 variants per step) under the contraction kernel variants the test hook can
force: does the donor-table contraction (112 donors to contract over, 64 000 tiles of 128 x 128 with a 131 KB store
each) prefer narrower tiles / the register-staged kernel?   python tools/probe_collapsed_tiles.py"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib, get_L_values  # noqa: E402
from cellregmap_amd.synth import make_cohort  # noqa: E402

c = make_cohort(100, 200, 50, 16, seed=20)
s = make_cohort(100, 200, 50, 4096, seed=1000, with_phenotype=False)
crm = CellRegMap(c.y, c.E, W=c.W, Ls=get_L_values(c.hK, c.E))
panel = GenotypePanel.from_donors(s.G[::200], s.donor_of_cell)
lib, ctx = _lib.load(), _engine._context(0)
pv0, _ = crm.scan_interaction(panel, progress=False)
for tile, dma in ((0, 1), (128, 1), (128, 0), (64, 0), (64, 1)):
    _lib.check(lib.crm_test_set_contraction(ctx, tile, dma))
    crm.scan_interaction(panel, progress=False)
    t = time.perf_counter()
    for _ in range(5):
        pv, _ = crm.scan_interaction(panel, progress=False)
    dt = (time.perf_counter() - t) / 5
    print(f"tile {tile:3d} dma {dma}: {dt * 1e3:7.2f} ms per 4096 variants = {4096 / dt:9.0f} variant-tests/s, "
          f"max rel dp vs default {np.max(np.abs(pv - pv0) / pv0):.2e}", flush=True)
_lib.check(lib.crm_test_set_contraction(ctx, 0, 1))
