"""Worker of tests/test_gpu_two_ranks.py: one of TWO processes that share the single GPU of the test box.

The process group is gloo (RCCL refuses two ranks on one device) but everything else is the real N > 1 path of
bench.py --gpus N: rank r decomposes the grid points i % 2 == r with the HIP eigen-solver, spectra / mixing
matrices (or Q0 on the eigh branch) are packed per owner and exchanged in one all_gather (exported to / imported from
CUDA tensors; gloo carries host copies of them), each rank uploads only its
own shard of the panel, scans it, and the per-variant results are all-gathered.  Rank 0 writes the gathered result
and the constructor's spectra to ``out`` (npz); the test compares them with the oracle and a one-process run."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    mode, out = sys.argv[1], sys.argv[2]
    # exchange buffers: CUDA tensors (what RCCL broadcasts), or -- mode "...-host" -- the CPU tensors that
    # sharded_background allocates by default under gloo: crm_background_export / _import then copy to and from HOST memory
    host = mode.endswith("-host")
    mode = mode[:-5] if host else mode
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)                    # torch first, then the library (one HIP runtime for both)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 2
    torch.zeros(1, device="cuda").add_(1.0)
    import cellregmap_amd as crm
    from cellregmap_amd.distributed import (scan_interaction_distributed, scan_interaction_many_distributed,
                                            sharded_background, variant_shard)
    from cellregmap_amd.synth import make_cohort

    donors, cells, k, p = (12, 20, 4, 37) if mode != "C-eigh" else (12, 10, 10, 21)   # eigh: k + k*donors >= n
    c = make_cohort(donors, cells, k, p, seed=31)
    n = c.y.size
    rho = np.linspace(0.0, 1.0, 11)
    if mode == "B":
        B, kw = c.hK, dict(hK=c.hK)
    else:
        B = crm.get_L_values(c.hK, c.E)
        kw = dict(Ls=B)
    info = {}
    bg = sharded_background(c.E, B, rho, device=0, tensor_device=None if host else torch.device("cuda", 0), info=info,
                            overlap=lambda: "ran beside the collective")
    # the packed exchange itself must have carried the grid points (a silent fall-back to a local build would pass the
    # comparisons below just as well)
    assert info["exchange"] == "ok" and info["collectives"] == 3 and info["exchanged_bytes"] > 0, info
    assert info["overlap_result"] == "ran beside the collective"
    obj = crm.CellRegMap(c.y, c.E, W=c.W, background=bg, **kw)
    first, count = variant_shard(p, rank, world)
    shard = np.ascontiguousarray(c.G[:, first:first + count])
    pv, info = scan_interaction_distributed(obj, shard, p_total=p)
    rng = np.random.default_rng(3)
    Y = np.stack([c.y, rng.permutation(c.y), rng.normal(size=n)], axis=1)
    objs = [obj] + [crm.CellRegMap(Y[:, i], c.E, W=c.W, background=bg, **kw) for i in (1, 2)]
    pvm, infom = scan_interaction_many_distributed(objs, shard, p_total=p)
    spectra = [bg.read(i, n)[1] for i in range(len(rho))]
    if rank == 0:
        np.savez(out, pv=pv, pvm=pvm, rho1=info["rho1"], rho1m=infom["rho1"],
                 ranks=np.array([s.size for s in spectra]), spectra=np.concatenate(spectra))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
