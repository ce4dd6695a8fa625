"""The N > 1 path on CPU: world_size-2 gloo processes shard the variants, scan their shard (the
CPU oracle stands in for the GPU scan here) and all-gather; every rank must end up with exactly
the single-process result."""
import os
import socket
import sys

import numpy as np

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


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cellregmap_amd.distributed import scan_interaction_distributed
        from cellregmap_amd.synth import make_cohort
        from oracle.crm import OracleCellRegMap

        c = make_cohort(6, 10, 3, 7, seed=13)  # 7 variants over 2 ranks: ragged shards (4 + 3)
        ocrm = OracleCellRegMap(c.y, c.E, W=c.W, hK=c.hK)
        pv, info = scan_interaction_distributed(None, c.G, scan=ocrm.scan_interaction)
        # the same with every rank holding only its own columns
        from cellregmap_amd.distributed import variant_shard

        f, cnt = variant_shard(7, rank, world)
        pv2, info2 = scan_interaction_distributed(None, np.ascontiguousarray(c.G[:, f:f + cnt]), scan=ocrm.scan_interaction,
                                                  p_total=7)
        assert np.array_equal(pv, pv2) and all(np.array_equal(info[k], info2[k]) for k in info)
        q.put((rank, pv, info))
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_equals_single_process():
    import torch.multiprocessing as mp

    from cellregmap_amd.synth import make_cohort
    from oracle.crm import OracleCellRegMap

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    c = make_cohort(6, 10, 3, 7, seed=13)
    ref_pv, ref_info = OracleCellRegMap(c.y, c.E, W=c.W, hK=c.hK).scan_interaction(c.G)
    for rank, pv, info in results:
        assert np.array_equal(pv, ref_pv)
        for k in ref_info:
            assert np.array_equal(info[k], ref_info[k])


def _worker_many(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cellregmap_amd.distributed import scan_interaction_many_distributed
        from cellregmap_amd.synth import make_cohort
        from oracle.crm import OracleCellRegMap

        c = make_cohort(6, 10, 3, 5, seed=17)  # 5 variants over 2 ranks (3 + 2), 2 genes
        ys = [c.y, c.y[::-1].copy()]
        oracles = [OracleCellRegMap(y, c.E, W=c.W, hK=c.hK) for y in ys]

        def scan_many(crms, G, idx_E, idx_G):
            res = [o.scan_interaction(G, idx_E, idx_G) for o in crms]
            return np.stack([r[0] for r in res]), {k: np.stack([r[1][k] for r in res]) for k in res[0][1]}

        pv, info = scan_interaction_many_distributed(oracles, c.G, scan_many=scan_many)
        q.put((rank, pv, info))
    finally:
        dist.destroy_process_group()


def test_two_rank_multi_gene_gather_equals_single_process():
    """Config 4's shape (several genes x one panel, variants sharded over the ranks)."""
    import torch.multiprocessing as mp

    from cellregmap_amd.synth import make_cohort
    from oracle.crm import OracleCellRegMap

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_many, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    c = make_cohort(6, 10, 3, 5, seed=17)
    for gi, y in enumerate((c.y, c.y[::-1].copy())):
        ref_pv, ref_info = OracleCellRegMap(y, c.E, W=c.W, hK=c.hK).scan_interaction(c.G)
        for _, pv, info in results:
            assert pv.shape == (2, 5)
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


def _ctor_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cellregmap_amd.distributed import sharded_background
        from cellregmap_amd.synth import make_cohort

        c = make_cohort(6, 10, 3, 4, seed=13)
        rho = np.linspace(0, 1, 11)
        info = {}
        b = sharded_background(c.E, c.hK, rho, builder=lambda mine: _NumpyBuilder(c.E, c.hK, rho, mine),
                               overlap=lambda: "uploaded while the collective ran", info=info)
        # one packed all_gather (+ the all_reduce of the ranks), the overlap hook ran, timings recorded
        assert info["exchange"] == "ok" and info["collectives"] == 2, info
        assert info["overlap_result"] == "uploaded while the collective ran"
        assert info["exchanged_bytes"] == 8 * len(b.decomposed) * (b.n * b.ldq + b.ldq)
        assert all(k in info for k in ("decompose_s", "exchange_s", "overlap_s"))
        q.put((rank, b.decomposed, b.ranks, {i: (s["Q0"].copy(), s["S0"].copy()) for i, s in b.slots.items()}))
    finally:
        dist.destroy_process_group()


def _ctor_worker_failing(rank, world, port, q, how):
    """The exchange goes wrong -- the collective itself raises on every rank ("collective"), or one rank cannot take
    what it received ("import", rank 1 only): the affected ranks decompose every grid point themselves."""
    sys.path.insert(0, ROOT)
    import warnings

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cellregmap_amd import distributed
        from cellregmap_amd.synth import make_cohort

        c = make_cohort(6, 10, 3, 4, seed=13)
        rho = np.linspace(0, 1, 11)

        class Builder(_NumpyBuilder):
            def import_slot(self, i, what, tensor):
                if how == "import" and rank == 1:
                    raise RuntimeError("simulated: the received slot cannot be copied in")
                super().import_slot(i, what, tensor)

        if how == "collective":
            def broken(*a, **k):
                raise RuntimeError("simulated: ncclCommInitRank failed")
            dist.all_gather = broken
        info = {}
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            b = distributed.sharded_background(c.E, c.hK, rho, builder=lambda mine: Builder(c.E, c.hK, rho, mine), info=info,
                                               overlap=lambda: 7)
        failed = how == "collective" or rank == 1
        assert info["exchange"].startswith("failed: RuntimeError: simulated") == failed, info
        assert any("decomposing every grid point on this rank" in str(w.message) for w in caught) == failed
        assert info["overlap_result"] == 7
        q.put((rank, b.decomposed, b.ranks, {i: (s["Q0"].copy(), s["S0"].copy()) for i, s in b.slots.items()}))
    finally:
        dist.destroy_process_group()


def test_sharded_constructor_two_ranks():
    """Rank r decomposes the grid points i % 2 == r; after the exchange both ranks hold all eleven
    decompositions, identical to a single process's."""
    import torch.multiprocessing as mp

    from cellregmap_amd.distributed import sharded_background
    from cellregmap_amd.synth import make_cohort

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ctor_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    c = make_cohort(6, 10, 3, 4, seed=13)
    rho = np.linspace(0, 1, 11)
    ref = sharded_background(c.E, c.hK, rho, builder=lambda mine: _NumpyBuilder(c.E, c.hK, rho, mine))  # world of one
    assert ref.decomposed == list(range(11))
    assert results[0][1] == [0, 2, 4, 6, 8, 10] and results[1][1] == [1, 3, 5, 7, 9]
    for rank, _, ranks, slots in results:
        assert ranks == ref.ranks
        for i in range(11):
            assert np.array_equal(slots[i][0], ref.slots[i]["Q0"]) and np.array_equal(slots[i][1], ref.slots[i]["S0"])


import pytest  # noqa: E402


@pytest.mark.parametrize("how", ["collective", "import"])
def test_sharded_constructor_falls_back_to_a_local_build_when_the_exchange_fails(how):
    """A rank whose exchange raises -- the collective itself (RCCL that cannot start: every rank), or the import of what
    it received (one rank) -- decomposes all eleven grid points itself and ends up with the same background as the
    ranks that exchanged; nobody has to agree on which way was taken."""
    import torch.multiprocessing as mp

    from cellregmap_amd.distributed import sharded_background
    from cellregmap_amd.synth import make_cohort

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ctor_worker_failing, args=(r, 2, port, q, how)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    c = make_cohort(6, 10, 3, 4, seed=13)
    rho = np.linspace(0, 1, 11)
    ref = sharded_background(c.E, c.hK, rho, builder=lambda mine: _NumpyBuilder(c.E, c.hK, rho, mine))
    for rank, decomposed, ranks, slots in results:
        assert decomposed == (list(range(11)) if how == "collective" or rank == 1 else [0, 2, 4, 6, 8, 10])
        assert ranks == ref.ranks
        for i in range(11):
            assert np.array_equal(slots[i][0], ref.slots[i]["Q0"]) and np.array_equal(slots[i][1], ref.slots[i]["S0"])
