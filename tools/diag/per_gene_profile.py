import sys, time, cProfile, pstats
sys.path.insert(0, ".")
import numpy as np
import cellregmap_amd as crm
from cellregmap_amd.synth import make_config
genes, window = 24, 256
c = make_config("cfg3", n_variants=genes * window, seed=0)
rng = np.random.default_rng(1)
ys = [c.y if g == 0 else rng.permutation(c.y) for g in range(genes)]
Gs = [np.ascontiguousarray(c.G[:, g * window:(g + 1) * window]) for g in range(genes)]
crm.run_interaction(ys[0], c.E, Gs[0], W=c.W, hK=c.hK)
crm.run_interaction(ys[1], c.E, Gs[1], W=c.W, hK=c.hK)
times = []
pr = cProfile.Profile()
pr.enable()
for g in range(2, genes):
    t = time.time()
    crm.run_interaction(ys[g], c.E, Gs[g], W=c.W, hK=c.hK)
    times.append(time.time() - t)
pr.disable()
print("per gene ms:", " ".join("%.0f" % (1e3 * t) for t in times))
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
