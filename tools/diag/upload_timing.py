"""Host -> device time of a panel chunk: a block of columns of a row-major matrix as it lies (rows p doubles apart) against
its contiguous copy, for chunk widths of the streamed scan.  GPU only.   python tools/diag/upload_timing.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cellregmap_amd import GenotypePanel  # noqa: E402

n, p = 20000, 50000
G = np.random.default_rng(0).normal(size=(n, p))
GenotypePanel(G[:, :256], groups=None)          # start-up
for width in (4096, 8192, 16384):
    for rep in range(2):
        view = G[:, 4096:4096 + width]
        t0 = time.perf_counter(); P = GenotypePanel(view, groups=None); t1 = time.perf_counter(); del P
        t2 = time.perf_counter()
        cont = np.ascontiguousarray(view)
        t3 = time.perf_counter(); P = GenotypePanel(cont, groups=None); t4 = time.perf_counter(); del P
        gb = n * width * 8 / 1e9
        print(f"{width:6d} columns ({gb:.2f} GB): as it lies {t1 - t0:.3f} s = {gb / (t1 - t0):5.1f} GB/s | host copy {t3 - t2:.3f} s, "
              f"contiguous upload {t4 - t3:.3f} s = {gb / (t4 - t3):5.1f} GB/s", flush=True)

# what releasing a chunk's panel costs the thread that does it
for width in (8192, 8192, 4096):
    P = GenotypePanel(G[:, :width], groups=None)
    t0 = time.perf_counter()
    del P
    print(f"release of a {width}-column panel: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
