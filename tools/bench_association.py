"""GPU probe: association LRT throughput at config 3 (run_association binds the 50 contexts to the
fixed effects and W to the background, mode B with hK)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellregmap_amd import CellRegMap, GenotypePanel
from cellregmap_amd.synth import make_cohort

c = make_cohort(100, 200, 50, 16, seed=20)
s = make_cohort(100, 200, 50, 4096, seed=1000, with_phenotype=False)
t = time.time()
crm = CellRegMap(c.y, c.W, c.E, hK=c.hK)       # the wrapper's positional binding: E <- W, W <- E
crm._bind_gene()
print("ctor", round(time.time() - t, 2), "s; ranks", [crm._bg.rank(i) for i in range(11)])
panel = GenotypePanel(s.G, groups=None)
for name, fn, nv in (("fast", crm.scan_association_fast, 4096), ("full", crm.scan_association, 1024)):
    fn(GenotypePanel(s.G[:, :256], groups=None))
    t = time.time(); pv, info = fn(GenotypePanel(s.G[:, :nv], groups=None)) if nv < 4096 else fn(panel); dt = time.time() - t
    print(f"scan_association_{name}: {nv} SNPs in {dt:.3f} s -> {nv/dt:.0f} SNPs/s; rho {info['rho1']}, min p {pv.min():.3g}")

# end to end from a host matrix of 32 768 SNPs (5.2 GB): uploaded whole, then scanned -- against streamed in column chunks
# beside the scan (the host's default, CELLREGMAP_AMD_STREAM_CHUNK)
big = make_cohort(100, 200, 50, 32768, seed=1001, with_phenotype=False).G
big = big + 0.05 * np.random.default_rng(0).normal(size=big.shape)      # general genotypes: every chunk stays dense
for name, fn in (("fast", crm.scan_association_fast), ("full", crm.scan_association)):
    for chunk in ("0", "8192"):
        os.environ["CELLREGMAP_AMD_STREAM_CHUNK"] = chunk
        t = time.time(); pv, _ = fn(big, progress=False); dt = time.time() - t
        print(f"scan_association_{name} from the host matrix, {'streamed' if chunk != '0' else 'one panel'}: "
              f"{big.shape[1]} SNPs in {dt:.3f} s -> {big.shape[1] / dt:.0f} SNPs/s end to end")
