"""Multi-GPU layer of the scan: variants are independent (cellregmap/_cellregmap.py:340), so
they are sharded across ranks as contiguous blocks with no data-path collective; the one
exchange is the final gather of the per-variant results (RCCL all_gather over xGMI when the
process group is ``nccl``; ``gloo`` on CPU in the tests).  One process per GPU.
"""
import numpy as np


def variant_shard(p, rank, world):
    """Contiguous shard [first, first + count) of p variants for ``rank`` of ``world``."""
    base, extra = divmod(int(p), int(world))
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return first, count


def gather_variant_results(local, p, group=None):
    """All-gather per-variant float64 arrays.

    ``local`` maps names to 1-D arrays holding this rank's shard (``variant_shard`` order).
    Returns the same mapping with the full length-``p`` arrays, on every rank."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        return {k: np.asarray(v, float) for k, v in local.items()}
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    names = sorted(local)
    first, count = variant_shard(p, rank, world)
    width = variant_shard(p, 0, world)[1]  # rank 0 holds the largest shard
    pack = torch.zeros((len(names), width), dtype=torch.float64)
    for i, k in enumerate(names):
        v = np.asarray(local[k], float)
        if v.shape != (count,):
            raise ValueError(f"{k}: expected {count} entries for rank {rank}, got {v.shape}")
        pack[i, :count] = torch.from_numpy(v)
    pack = pack.to(device)
    parts = [torch.empty_like(pack) for _ in range(world)]
    dist.all_gather(parts, pack, group=group)
    out = {k: np.empty(p) for k in names}
    for r, part in enumerate(parts):
        f, c = variant_shard(p, r, world)
        part = part.cpu().numpy()
        for i, k in enumerate(names):
            out[k][f:f + c] = part[i, :c]
    return out


def grid_point_owner(i, world):
    """Rank that decomposes grid point i (round robin over the rho grid)."""
    return i % world


def owned_grid_points(nrho, rank, world):
    return [i for i in range(nrho) if grid_point_owner(i, world) == rank]


def sharded_background(E1, B, rho, device=0, group=None, builder=None, tensor_device=None, force_exchange=False,
                       overlap=None, info=None):
    """The background of ``CellRegMap(...)`` (cellregmap/_cellregmap.py:101-131: one economic
    eigendecomposition per grid point of rho) built ONCE per job instead of once per rank: rank r decomposes
    the grid points i with i % world == r, the ranks are all-reduced (they fix the common leading
    dimension), and what the others need of every grid point -- spectrum and mixing matrix (0.2 GB per grid point
    at BASELINE config 3; Q0 itself only where there is no mixing matrix) -- is exchanged in ONE collective: every
    rank packs the slots of its grid points into one buffer and a single ``all_gather`` (RCCL over xGMI when the
    group is ``nccl``) hands every rank every buffer.  Returns the sealed background; pass it as
    ``CellRegMap(..., background=...)``.

    ``B``: what the constructor concatenates behind sqrt(rho) E1 -- ``None`` (mode A), hK (mode B), the
    ``HadamardHalves`` of ``get_L_values`` or their concatenation (mode C).
    ``builder``: factory ``mine -> object with rank / complete / layout / export_slot / import_slot / seal``
    (tests inject a numpy one); default: the HIP library's ``BackgroundBuilder``.
    ``overlap``: callable run while the collective is in flight (the caller's panel upload: PCIe beside xGMI);
    its return value lands in ``info["overlap_result"]``.
    ``info``: dict that receives the phase timings (``decompose_s``, ``exchange_s`` = pack + collectives + unpack,
    ``overlap_s``), the bytes this rank contributed, and ``exchange`` = "ok" / "not needed" / "failed: ...".

    Failure protocol.  Every rank issues the SAME sequence of collectives whatever happens to it locally, so a rank in
    trouble never leaves its peers waiting and never gets one collective out of step with them:
      1. local: decompose the owned grid points (may fail: out of memory, a solver error);
      2. ``all_reduce(MAX)`` of the ranks of the grid points with a failure flag in the last entry;
      3. local: common leading dimension, pack the owned slots (may fail);
      4. ``all_reduce(MIN)`` of an ok flag;
      5. one ``all_gather`` of the packed buffers -- only if every rank said ok in 2 and 4;
      6. local: import what the others sent (may fail).
    A flag raised in 2 or 4 is seen by everybody: all ranks then decompose every grid point themselves (the single-GPU
    constructor) and no further collective is issued.  A failure in 6 concerns this rank alone and comes after the
    last collective: it alone rebuilds.  A collective that itself raises (a communicator that cannot start, a
    timeout) also ends in the local build; the caller's next collective is then on its own (``bench.py`` keeps a
    gloo group behind RCCL for that).  ``overlap`` runs outside all of this: what it raises is the caller's error
    and propagates (after the collective in flight has been waited for).

    Order of initialisation in a process that uses torch on the GPU and this library: torch first
    (``torch.cuda.set_device`` / ``init_process_group(..., device_id=...)``), then the first call into the
    library -- both then share one HIP runtime and device pointers can be handed across."""
    import time
    import warnings

    import torch
    import torch.distributed as dist

    info = {} if info is None else info
    rho = np.asarray(rho, float)
    nrho = rho.shape[0]
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    mine = np.array([grid_point_owner(i, world) == rank for i in range(nrho)], np.int32)
    if builder is None:
        from ._engine import BackgroundBuilder

        def builder(flags):
            return BackgroundBuilder(E1, B, rho, device=device, mine=flags)

    def everything_here():
        full = builder(np.ones(nrho, np.int32))
        full.complete([full.rank(i) for i in range(nrho)])
        return full.seal()

    def run_overlap():
        if overlap is not None and "overlap_result" not in info:
            t2 = time.perf_counter()
            info["overlap_result"] = overlap()
            info["overlap_s"] = time.perf_counter() - t2

    t0 = time.perf_counter()
    exchanging = world > 1 or (force_exchange and dist.is_available() and dist.is_initialized())
    if not exchanging:
        bg = everything_here()
        info.update(decompose_s=time.perf_counter() - t0, exchange_s=0.0, exchange="not needed", exchanged_bytes=0)
        run_overlap()
        return bg
    # (force_exchange: a world of one still runs the collective calls and copies every slot out and back in --
    # the whole exchange path on a single GPU)

    def local_build(why):
        warnings.warn(f"rank {rank}: the exchange of the background failed ({why}); decomposing every grid "
                      "point on this rank instead", RuntimeWarning, stacklevel=3)
        t3 = time.perf_counter()
        bg_ = everything_here()
        info.update(exchange="failed: " + why[:240], fallback_s=time.perf_counter() - t3)
        return bg_

    def say(exc):
        return "%s: %s" % (type(exc).__name__, exc)

    nccl = dist.get_backend(group) == "nccl"
    dev = tensor_device if tensor_device is not None else (torch.device("cuda", device) if nccl else torch.device("cpu"))
    # 1. local: my grid points
    b, trouble = None, None
    try:
        b = builder(mine)
        mine_ranks = [b.rank(i) if mine[i] else -1 for i in range(nrho)]
    except Exception as exc:  # noqa: BLE001 -- reported to the peers through the flag below, then retried alone
        b, trouble, mine_ranks = None, say(exc), [-1] * nrho
    info["decompose_s"] = time.perf_counter() - t0
    t1 = time.perf_counter()
    # 2. ranks of all grid points + failure flag
    try:
        ranks = torch.tensor(mine_ranks + [1 if trouble else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(ranks, op=dist.ReduceOp.MAX, group=group)
        ranks = ranks.cpu().numpy()
    except Exception as exc:  # noqa: BLE001 -- the collective itself
        b = None
        bg = local_build(trouble or say(exc))
        run_overlap()
        return bg
    if ranks[-1] != 0:
        b = None
        bg = local_build(trouble or "another rank could not decompose its grid points")
        run_overlap()
        return bg
    # 3. local: common layout, pack
    pack = per_point = most = None
    try:
        b.complete(ranks[:nrho])
        layout = b.layout()                                   # slot name -> doubles, the same on every rank
        per_point = int(sum(layout.values()))
        most = (nrho + world - 1) // world                    # grid points of the busiest rank: every piece has that size
        own = owned_grid_points(nrho, rank, world)
        pack = torch.empty(most * per_point, dtype=torch.float64, device=dev)
        if len(own) < most:
            pack[len(own) * per_point:].zero_()               # only the padding behind this rank's last piece
        if pack.is_cuda:
            # the fill above runs on torch's current stream, the exports below on the library's own (non-blocking) stream:
            # nothing else orders the two, so torch's stream is drained before the first export touches the buffer
            torch.cuda.current_stream(pack.device).synchronize()
        for k, i in enumerate(own):
            off = k * per_point
            for what, size in layout.items():
                b.export_slot(i, what, pack[off:off + size])
                off += size
    except Exception as exc:  # noqa: BLE001
        trouble = say(exc)
    # 4. does everybody have its piece?
    try:
        ok = torch.tensor([0 if trouble else 1], dtype=torch.int64, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        ok = int(ok.cpu().item())
    except Exception as exc:  # noqa: BLE001
        ok, trouble = 0, trouble or say(exc)
    if not ok:
        b = pack = None
        bg = local_build(trouble or "another rank could not pack its grid points")
        run_overlap()
        return bg
    # 5. the exchange: one all_gather, the caller's overlap hook beside it
    work = None
    try:
        # (gloo has no all_gather of GPU tensors: device buffers -- the two-ranks-on-one-GPU test -- travel through host copies)
        staged = (not nccl) and pack.is_cuda
        wire = pack.cpu() if staged else pack
        everything = torch.empty(world * most * per_point, dtype=torch.float64, device=wire.device)
        work = dist.all_gather(list(everything.chunk(world)), wire, group=group, async_op=True)
    except Exception as exc:  # noqa: BLE001
        trouble = say(exc)
    try:
        run_overlap()
    finally:
        if work is not None:
            try:
                work.wait()
            except Exception as exc:  # noqa: BLE001
                trouble = trouble or say(exc)
    # 6. local: take what the others sent
    if trouble is None:
        try:
            if staged:
                everything = everything.to(dev)
            if everything.is_cuda:
                # the wait orders torch's current stream behind the collective; the library copies on its own stream
                torch.cuda.current_stream(everything.device).synchronize()
            for r in range(world):
                if r == rank and not force_exchange:
                    continue
                for k, i in enumerate(owned_grid_points(nrho, r, world)):
                    off = (r * most + k) * per_point
                    for what, size in layout.items():
                        b.import_slot(i, what, everything[off:off + size])
                        off += size
            bg = b.seal()
            info.update(exchange_s=time.perf_counter() - t1 - info.get("overlap_s", 0.0), exchange="ok",
                        exchanged_bytes=8 * int(mine.sum()) * per_point, collectives=3)
            return bg
        except Exception as exc:  # noqa: BLE001 -- after the last collective: this rank's business alone
            trouble = say(exc)
    b = pack = everything = None
    return local_build(trouble)


def _my_columns(G, p_total, rank, world):
    """This rank's shard of the panel: either cut from the full matrix (``p_total is None``), or ``G`` IS the
    shard already -- ``variant_shard(p_total, rank, world)`` columns, loaded by the rank itself so that no
    rank ever holds the whole n x p matrix in host memory (8 GB at BASELINE config 3)."""
    if p_total is None:
        G = np.asarray(G)
        p = G.shape[1]
        first, count = variant_shard(p, rank, world)
        return p, np.ascontiguousarray(G[:, first:first + count], dtype=float)
    first, count = variant_shard(p_total, rank, world)
    if hasattr(G, "shape") and G.shape[1] != count:
        raise ValueError(f"rank {rank} holds {G.shape[1]} columns, its shard of {p_total} variants has {count}")
    return int(p_total), G


def scan_interaction_distributed(crm, G, idx_E=None, idx_G=None, group=None, scan=None, p_total=None):
    """``crm.scan_interaction`` over this rank's shard of the columns of ``G`` followed by the
    gather; returns the reference's ``(pvalues, info)`` for all variants on every rank.

    ``G``: the full n x p matrix on every rank, or -- with ``p_total`` -- only this rank's shard
    (``variant_shard(p_total, rank, world)``; array or ``GenotypePanel``).
    ``scan`` overrides the per-shard call (tests inject the CPU oracle)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    p, mine = _my_columns(G, p_total, rank, world)
    first, count = variant_shard(p, rank, world)
    fn = scan if scan is not None else crm.scan_interaction
    if count > 0:
        pv, info = fn(mine, idx_E, idx_G)
    else:
        pv, info = np.empty(0), {k: np.empty(0) for k in ("rho1", "e2", "g2", "eps2")}
    full = gather_variant_results({"pv": pv, **info}, p, group)
    return full.pop("pv"), full


def scan_interaction_many_distributed(crms, G, idx_E=None, idx_G=None, group=None, scan_many=None, p_total=None):
    """BASELINE config 4's shape: several genes (``CellRegMap`` objects sharing one background) against
    one panel, the variants sharded over the ranks.  Every rank runs ``scan_interaction_many`` -- all
    genes, its shard of the columns of ``G`` -- so the phenotype-free work of a variant is done once,
    on one GPU; the gather returns ``(pvalues (genes x p), info of (genes x p) arrays)`` on every rank.

    ``G`` / ``p_total``: as in ``scan_interaction_distributed``.
    ``scan_many`` overrides the per-shard call (tests inject the CPU oracle)."""
    import torch.distributed as dist

    from ._engine import scan_interaction_many

    ng = len(crms)
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    p, mine = _my_columns(G, p_total, rank, world)
    first, count = variant_shard(p, rank, world)
    keys = ("rho1", "e2", "g2", "eps2")
    fn = scan_many if scan_many is not None else scan_interaction_many
    if count > 0:
        pv, info = fn(crms, mine, idx_E, idx_G)
    else:
        pv, info = np.empty((ng, 0)), {k: np.empty((ng, 0)) for k in keys}
    return gather_many_results(pv, info, p, group)


def gather_many_results(pv, info, p, group=None):
    """The gather of BASELINE config 4: this rank's (genes x shard) p-values and info arrays in, the (genes x p)
    arrays of all shards out, on every rank -- every (gene, array) row packed into ONE ``all_gather``
    (64 genes x 5 arrays x 6 250 variants x 8 bytes = 16 MB per rank at 8 GPUs)."""
    ng = pv.shape[0]
    keys = tuple(info)
    local = {}
    for gi in range(ng):
        local[f"pv:{gi:06d}"] = pv[gi]
        for k in keys:
            local[f"{k}:{gi:06d}"] = info[k][gi]
    full = gather_variant_results(local, p, group)
    out_pv = np.stack([full[f"pv:{gi:06d}"] for gi in range(ng)])
    out_info = {k: np.stack([full[f"{k}:{gi:06d}"] for gi in range(ng)]) for k in keys}
    return out_pv, out_info
