"""The N > 1 path on CPU: gloo processes (world sizes 2, 3 and 8 -- remainders in the variant shards, uneven
ownership of the eleven grid points: 2,2,2,1,1,1,1,1 at world 8) shard the variants, scan their shard (the
CPU oracle stands in for the GPU scan here) and all-gather; every rank must end up with exactly
the single-process result.  The sharded constructor's protocol, including what happens when a rank gets into
trouble at each of its stages, runs with a numpy stand-in for the device's builder."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_variant_shard_partitions_everything():
    from cellregmap_amd.distributed import variant_shard

    for p in (0, 1, 7, 64, 1001):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                f, c = variant_shard(p, r, world)
                cover.extend(range(f, f + c))
            assert cover == list(range(p))


WORLDS = [2, 3, 8]


def _spawn(target, world, *args, timeout=300):
    """``world`` gloo ranks as spawned processes; returns their queue items sorted by rank."""
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q, *args)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        results = sorted([q.get(timeout=timeout) for _ in procs], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    return results


def _init(rank, world, port, timeout_s=120):
    import datetime

    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
    return dist


P_SINGLE = 11   # 11 variants: 6 + 5 at world 2, 4 + 4 + 3 at world 3, 2,2,2,1,1,1,1,1 at world 8


def _worker(rank, world, port, q):
    dist = _init(rank, world, port)
    try:
        from cellregmap_amd.distributed import scan_interaction_distributed, variant_shard
        from cellregmap_amd.synth import make_cohort
        from oracle.crm import OracleCellRegMap

        c = make_cohort(6, 10, 3, P_SINGLE, seed=13)
        ocrm = OracleCellRegMap(c.y, c.E, W=c.W, hK=c.hK)
        pv, info = scan_interaction_distributed(None, c.G, scan=ocrm.scan_interaction)
        # the same with every rank holding only its own columns
        f, cnt = variant_shard(P_SINGLE, rank, world)
        pv2, info2 = scan_interaction_distributed(None, np.ascontiguousarray(c.G[:, f:f + cnt]), scan=ocrm.scan_interaction,
                                                  p_total=P_SINGLE)
        assert np.array_equal(pv, pv2) and all(np.array_equal(info[k], info2[k]) for k in info)
        q.put((rank, pv, info))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", WORLDS)
def test_gather_equals_single_process(world):
    from cellregmap_amd.synth import make_cohort
    from oracle.crm import OracleCellRegMap

    results = _spawn(_worker, world)
    c = make_cohort(6, 10, 3, P_SINGLE, seed=13)
    ref_pv, ref_info = OracleCellRegMap(c.y, c.E, W=c.W, hK=c.hK).scan_interaction(c.G)
    for rank, pv, info in results:
        assert np.array_equal(pv, ref_pv)
        for k in ref_info:
            assert np.array_equal(info[k], ref_info[k])


def _worker_fewer_variants_than_ranks(rank, world, port, q):
    """5 variants over 8 ranks: three ranks hold nothing and still take part in the gather."""
    dist = _init(rank, world, port)
    try:
        from cellregmap_amd.distributed import scan_interaction_distributed
        from cellregmap_amd.synth import make_cohort
        from oracle.crm import OracleCellRegMap

        c = make_cohort(6, 10, 3, 5, seed=13)
        ocrm = OracleCellRegMap(c.y, c.E, W=c.W, hK=c.hK)
        pv, info = scan_interaction_distributed(None, c.G, scan=ocrm.scan_interaction)
        q.put((rank, pv, info))
    finally:
        dist.destroy_process_group()


def test_gather_with_empty_shards():
    from cellregmap_amd.synth import make_cohort
    from oracle.crm import OracleCellRegMap

    results = _spawn(_worker_fewer_variants_than_ranks, 8)
    c = make_cohort(6, 10, 3, 5, seed=13)
    ref_pv, ref_info = OracleCellRegMap(c.y, c.E, W=c.W, hK=c.hK).scan_interaction(c.G)
    for rank, pv, info in results:
        assert np.array_equal(pv, ref_pv)
        for k in ref_info:
            assert np.array_equal(info[k], ref_info[k])


P_MANY = 10   # 10 variants, 2 genes: 5 + 5, 4 + 3 + 3, 2,2,1,1,1,1,1,1


def _worker_many(rank, world, port, q):
    dist = _init(rank, world, port)
    try:
        from cellregmap_amd.distributed import scan_interaction_many_distributed
        from cellregmap_amd.synth import make_cohort
        from oracle.crm import OracleCellRegMap

        c = make_cohort(6, 10, 3, P_MANY, seed=17)
        ys = [c.y, c.y[::-1].copy()]
        oracles = [OracleCellRegMap(y, c.E, W=c.W, hK=c.hK) for y in ys]

        def scan_many(crms, G, idx_E, idx_G):
            res = [o.scan_interaction(G, idx_E, idx_G) for o in crms]
            return np.stack([r[0] for r in res]), {k: np.stack([r[1][k] for r in res]) for k in res[0][1]}

        pv, info = scan_interaction_many_distributed(oracles, c.G, scan_many=scan_many)
        q.put((rank, pv, info))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", WORLDS)
def test_multi_gene_gather_equals_single_process(world):
    """Config 4's shape (several genes x one panel, variants sharded over the ranks)."""
    from cellregmap_amd.synth import make_cohort
    from oracle.crm import OracleCellRegMap

    results = _spawn(_worker_many, world)
    c = make_cohort(6, 10, 3, P_MANY, seed=17)
    for gi, y in enumerate((c.y, c.y[::-1].copy())):
        ref_pv, ref_info = OracleCellRegMap(y, c.E, W=c.W, hK=c.hK).scan_interaction(c.G)
        for _, pv, info in results:
            assert pv.shape == (2, P_MANY)
            assert np.array_equal(pv[gi], ref_pv)
            for k in ref_info:
                assert np.array_equal(info[k][gi], ref_info[k])


class _NumpyBuilder:
    """Stand-in for the HIP library's ``BackgroundBuilder`` in the protocol test: the same three phases
    (owned grid points decomposed by the oracle's economic_qs_linear, common leading dimension from the ranks
    of all, slots exported / imported as flat float64 tensors), no GPU."""

    def __init__(self, E1, B, rho, mine):
        from oracle.sugar import economic_qs_linear

        self.rho, self.mine, self.n = list(rho), list(mine), E1.shape[0]
        self.qs = {}
        for i, r in enumerate(self.rho):
            if self.mine[i]:
                (Q0,), S0 = economic_qs_linear(np.concatenate([np.sqrt(r) * E1, np.sqrt(1 - r) * B], axis=1), return_q1=False)
                keep = S0 > 1e-12 * S0.max()
                self.qs[i] = (Q0[:, keep], S0[keep])
        self.slots = None
        self.decomposed = sorted(self.qs)

    def rank(self, i):
        return self.qs[i][0].shape[1] if i in self.qs else -1

    def complete(self, ranks):
        self.ranks = [int(r) for r in ranks]
        assert all(r >= 0 for r in self.ranks)
        self.ldq = max(self.ranks) + 3      # some padding, like the device's round_up(rmax, 128)
        self.slots = {}
        for i in range(len(self.rho)):
            Q = np.zeros((self.n, self.ldq))
            S = np.zeros(self.ldq)
            if i in self.qs:
                Q[:, : self.ranks[i]], S[: self.ranks[i]] = self.qs[i]
            self.slots[i] = {"Q0": Q, "S0": S}

    def layout(self):
        return {"Q0": self.n * self.ldq, "S0": self.ldq}

    def export_slot(self, i, what, tensor):
        assert self.mine[i]
        tensor.numpy()[:] = self.slots[i][what].ravel()

    def import_slot(self, i, what, tensor):
        self.slots[i][what][...] = tensor.numpy().reshape(self.slots[i][what].shape)

    def seal(self):
        return self



def _problem():
    from cellregmap_amd.synth import make_cohort

    return make_cohort(6, 10, 3, 4, seed=13), np.linspace(0, 1, 11)


def _snapshot(b):
    return b.decomposed, b.ranks, {i: (s["Q0"].copy(), s["S0"].copy()) for i, s in b.slots.items()}


def _ctor_worker(rank, world, port, q):
    dist = _init(rank, world, port)
    try:
        from cellregmap_amd.distributed import sharded_background

        c, rho = _problem()
        info = {}
        b = sharded_background(c.E, c.hK, rho, builder=lambda mine: _NumpyBuilder(c.E, c.hK, rho, mine),
                               overlap=lambda: "uploaded while the collective ran", info=info)
        # one packed all_gather (+ the two small all_reduces: ranks, ok flag), the overlap hook ran, timings recorded
        assert info["exchange"] == "ok" and info["collectives"] == 3, info
        assert info["overlap_result"] == "uploaded while the collective ran"
        assert info["exchanged_bytes"] == 8 * len(b.decomposed) * (b.n * b.ldq + b.ldq)
        assert all(k in info for k in ("decompose_s", "exchange_s", "overlap_s"))
        q.put((rank, *_snapshot(b)))
    finally:
        dist.destroy_process_group()


def _reference_background():
    from cellregmap_amd.distributed import sharded_background

    c, rho = _problem()
    ref = sharded_background(c.E, c.hK, rho, builder=lambda mine: _NumpyBuilder(c.E, c.hK, rho, mine))  # world of one
    assert ref.decomposed == list(range(11))
    return ref


def _same_background(ref, ranks, slots):
    assert ranks == ref.ranks
    for i in range(11):
        assert np.array_equal(slots[i][0], ref.slots[i]["Q0"]) and np.array_equal(slots[i][1], ref.slots[i]["S0"])


@pytest.mark.parametrize("world", WORLDS)
def test_sharded_constructor(world):
    """Rank r decomposes the grid points i % world == r (uneven at world 3 and 8: the busiest rank's count sizes every
    piece of the packed exchange, the others pad); afterwards every rank holds all eleven decompositions, identical to
    a single process's."""
    results = _spawn(_ctor_worker, world)
    ref = _reference_background()
    for rank, decomposed, ranks, slots in results:
        assert decomposed == [i for i in range(11) if i % world == rank]
        _same_background(ref, ranks, slots)


def _ctor_worker_failing(rank, world, port, q, how):
    """Something goes wrong at one stage of the protocol (cellregmap_amd/distributed.py: sharded_background), on rank 1
    only unless stated:
      "decompose"   the owned grid points cannot be decomposed -- BEFORE the first collective;
      "pack"        the slots cannot be exported -- between the first and the second collective;
      "collective"  the all_gather itself raises, on every rank;
      "import"      what was received cannot be copied in -- after the last collective.
    Every rank must come back with the full background, without waiting for a collective timeout, and having issued the
    same number of collectives as its peers (the all_reduce after the call would otherwise pair with a stale one)."""
    import time
    import warnings

    dist = _init(rank, world, port, timeout_s=60)
    try:
        import torch

        from cellregmap_amd import distributed

        c, rho = _problem()
        calls = {"n": 0}

        class Builder(_NumpyBuilder):
            def __init__(self, E1, B, rho_, mine):
                calls["n"] += 1
                if how == "decompose" and rank == 1 and calls["n"] == 1:
                    raise MemoryError("simulated: out of device memory in the decomposition")
                super().__init__(E1, B, rho_, mine)

            def export_slot(self, i, what, tensor):
                if how == "pack" and rank == 1:
                    raise RuntimeError("simulated: the slot cannot be exported")
                super().export_slot(i, what, tensor)

            def import_slot(self, i, what, tensor):
                if how == "import" and rank == 1:
                    raise RuntimeError("simulated: the received slot cannot be copied in")
                super().import_slot(i, what, tensor)

        if how == "collective":
            def broken(*a, **k):
                raise RuntimeError("simulated: ncclCommInitRank failed")
            real_all_gather, dist.all_gather = dist.all_gather, broken
        info = {}
        t0 = time.perf_counter()
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            b = distributed.sharded_background(c.E, c.hK, rho, builder=lambda mine: Builder(c.E, c.hK, rho, mine), info=info,
                                               overlap=lambda: 7)
        took = time.perf_counter() - t0
        everybody = how in ("decompose", "pack", "collective")
        failed = everybody or rank == 1
        assert info["exchange"].startswith("failed: ") == failed, info
        if failed and (rank == 1 or how == "collective"):
            assert "simulated" in info["exchange"], info
        assert any("decomposing every grid point on this rank" in str(w.message) for w in caught) == failed
        assert info["overlap_result"] == 7
        assert took < 30.0, f"rank {rank} waited {took:.1f} s: a peer's trouble must not cost a collective timeout"
        if how == "collective":
            dist.all_gather = real_all_gather
        # the ranks are still in step: the next collective pairs up
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        assert int(t.item()) == world * (world + 1) // 2
        q.put((rank, *_snapshot(b)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("how", ["decompose", "pack", "collective", "import"])
def test_sharded_constructor_when_a_rank_gets_into_trouble(how, world):
    """Trouble on one rank before, between or after the collectives -- or in the collective itself -- ends with the
    same background on every rank: all of them rebuild alone when the trouble is announced through the protocol's flags
    (before the exchange), only the affected rank when it comes after the last collective."""
    results = _spawn(_ctor_worker_failing, world, how)
    ref = _reference_background()
    for rank, decomposed, ranks, slots in results:
        alone = how in ("decompose", "pack", "collective") or rank == 1
        assert decomposed == (list(range(11)) if alone else [i for i in range(11) if i % world == rank])
        _same_background(ref, ranks, slots)


def _ctor_worker_overlap_raises(rank, world, port, q):
    dist = _init(rank, world, port, timeout_s=60)
    try:
        import torch

        from cellregmap_amd import distributed

        c, rho = _problem()
        built = {"n": 0}

        def make(mine):
            built["n"] += 1
            return _NumpyBuilder(c.E, c.hK, rho, mine)

        def upload():
            raise ValueError("genotypes must be finite")     # the caller's own error (the reference's ValueError)

        try:
            distributed.sharded_background(c.E, c.hK, rho, builder=make, overlap=upload)
            outcome = "returned"
        except ValueError as exc:
            outcome = str(exc)
        t = torch.tensor([1], dtype=torch.int64)
        dist.all_reduce(t)        # still in step
        q.put((rank, outcome, built["n"], int(t.item())))
    finally:
        dist.destroy_process_group()


def test_an_error_of_the_overlap_hook_is_the_callers_and_is_not_retried():
    for rank, outcome, builds, total in _spawn(_ctor_worker_overlap_raises, 2):
        assert outcome == "genotypes must be finite"
        assert builds == 1          # no rebuild, the hook ran once
        assert total == 2
