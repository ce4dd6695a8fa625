"""Worker of tests/test_gpu_nccl_world1.py: ONE process, process group ``nccl`` (= RCCL) at world size 1, started as a
fresh child (the pytest process has touched the GPU; this one initialises torch first, then the library).

Everything the real N > 1 run of ``bench.py --gpus N`` does on RCCL runs here on device tensors: the canary collectives,
``sharded_background(force_exchange=True)`` -- every slot exported into a CUDA tensor on the library's stream, one
``all_gather`` on torch's NCCL stream, every slot imported again --, ``gather_variant_results`` and
``gather_many_results``.  The exchange buffer is poisoned through torch's caching allocator first and a long fill is
queued on torch's current stream right before the call, so an export that is not ordered behind torch's stream
(the race the round-5 review found at ``distributed.py:183``) shows up as a wrong spectrum, not as luck.
Writes what the test compares to ``out`` (npz)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    mode, out = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0), world_size=1, rank=0)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1

    sys.path.insert(0, ROOT)
    import bench                                  # (Comm.canary is the bench's own first-collectives check)
    import cellregmap_amd as crm
    from cellregmap_amd.distributed import (gather_many_results, gather_variant_results, scan_interaction_distributed,
                                            scan_interaction_many_distributed, sharded_background)
    from cellregmap_amd.synth import make_cohort

    comm = bench.Comm.__new__(bench.Comm)
    comm.dist, comm.torch, comm.group, comm.backend, comm.note = dist, torch, None, "nccl", None
    comm.safe = dist.new_group(backend="gloo")
    comm.canary()
    assert comm.backend == "nccl" and comm.note is None, comm.note
    assert comm.max(3.5) == 3.5 and comm.table([1.0, 2.0], 0, 1) == [[1.0, 2.0]]

    donors, cells, k, p = (12, 20, 4, 37) if mode != "C-eigh" else (12, 10, 10, 21)   # eigh: k + k*donors >= n
    c = make_cohort(donors, cells, k, p, seed=31)
    n = c.y.size
    rho = np.linspace(0.0, 1.0, 11)
    if mode == "B":
        B, kw = c.hK, dict(hK=c.hK)
    else:
        B = crm.get_L_values(c.hK, c.E)
        kw = dict(Ls=B)
    # poison what the exchange buffer is likely to be carved from, then keep torch's stream busy while the call starts
    junk = torch.full((1 << 24,), float("nan"), dtype=torch.float64, device="cuda")
    del junk
    busy = torch.empty(1 << 27, dtype=torch.float64, device="cuda")
    for _ in range(8):
        busy.fill_(1.0)
    info = {}
    bg = sharded_background(c.E, B, rho, device=0, force_exchange=True, info=info, overlap=lambda: "beside the collective")
    assert info["exchange"] == "ok" and info["collectives"] == 3 and info["exchanged_bytes"] > 0, info
    assert info["overlap_result"] == "beside the collective"
    obj = crm.CellRegMap(c.y, c.E, W=c.W, background=bg, **kw)
    pv, sinfo = scan_interaction_distributed(obj, c.G)
    rng = np.random.default_rng(3)
    Y = np.stack([c.y, rng.permutation(c.y), rng.normal(size=n)], axis=1)
    objs = [obj] + [crm.CellRegMap(Y[:, i], c.E, W=c.W, background=bg, **kw) for i in (1, 2)]
    pvm, infom = scan_interaction_many_distributed(objs, c.G)
    # the two gathers on their own, on arrays whose every bit must come back
    probe = {"a": rng.normal(size=p), "b": rng.normal(size=p)}
    back = gather_variant_results(probe, p)
    assert all(np.array_equal(back[k_], probe[k_]) for k_ in probe)
    mpv, minfo = gather_many_results(pvm, infom, p)
    assert np.array_equal(mpv, pvm) and all(np.array_equal(minfo[k_], infom[k_]) for k_ in infom)
    spectra = [bg.read(i, n)[1] for i in range(len(rho))]
    np.savez(out, pv=pv, pvm=pvm, rho1=sinfo["rho1"], rho1m=infom["rho1"], ranks=np.array([s.size for s in spectra]),
             spectra=np.concatenate(spectra), exchanged_bytes=info["exchanged_bytes"])
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
