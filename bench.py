#!/usr/bin/env python3
"""Throughput of the interaction score test on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = the whole hot path (null fits over the rho grid, Khatri-Rao contraction, score statistic,
eigenvalues, Davies) over one batch of `--batch` synthetic variants of BASELINE config 3 (20 000 cells,
50 contexts, mode C background K o EE' + EE', r ~ 5 000), dense general-genotype path.  Inputs (background
decomposition, phenotype, genotype panel) are resident in HBM before the timed region.  N > 1: the script
starts one process per GPU itself (or runs under torch.distributed.run); variants are sharded across the
ranks with no data-path collective, the grid points of the background are decomposed by different ranks
and broadcast (RCCL), and the per-variant results are all-gathered at the end.

Legs of one run (one JSON line on rank 0):
  value / ms_per_step   weak scaling: every rank times K steps on its own shard (the driver's contract)
  full_panel            strong scaling on the FIXED 50 000-variant panel of the config, sharded over the ranks:
                        scan-only and end-to-end (constructor + upload + scan + gather) rates
  config4               64 phenotypes against one panel, variants sharded (BASELINE config 4's shape)
  donor_collapsed, cpu_baseline (N = 1)

CRM_BENCH_SHARE_GPU=1 python bench.py --gpus 2 ...: dry run of the N > 1 code path on a box with ONE GPU (both ranks
on device 0, gloo instead of RCCL; flagged in the line's "data" -- not a scaling measurement).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("CELLREGMAP_AMD_PROGRESS", "0")   # no tqdm bars around the benchmark's scans

PEAK_FP64_MFMA_TFLOPS = 78.6  # MI355X FP64 matrix peak (vendor sheet)


def algorithmic_flops(n, r_list, r_star, k0, c):
    """F_alg per variant-test (dense general G)."""
    R = float(sum(r_list))
    return 2.0 * n * R + 2.0 * n * r_star * k0 + n * k0 * (k0 + 1) + 2.0 * n * k0 * (c + 2) + r_star * k0 * (k0 + 1)


def executed_flops(n, cols, r_list, r_star, k0, c, fast_rotation, kin=None):
    """What the engine executes per variant-test.  fast_rotation: the rotations go through the mixing matrices,
    T(rho) = Mix(rho)'(H'g) -- one n-length product plus eleven cols-length ones.  kin = (k1, k2, donors_padded, m, folded, pairs): the
    background knows the donor structure of its kinship factor, so H'g and H'(g o E0) are formed donor by donor
    (2 n (k1 + k2) flops per product column plus the contraction over the donors with the donor-level factor) and
    Q0(rho*)'(g o E0) is taken as Mix(rho*)'[H'(g o E0)]: 2 cols r* k0 instead of 2 n r* k0."""
    R = float(sum(r_list))
    if kin is not None:
        k1, k2, dpad, m, folded, pairs = kin
        if folded:   # the contraction over the donors sits in the mixing matrices: their products are k1 + donors k2 long
            per_column = 2.0 * n * (k1 + k2)
            cols = folded
        else:
            per_column = 2.0 * n * (k1 + k2) + 2.0 * dpad * m * k2    # donor sums + contraction over the donors
        rot = per_column + 2.0 * cols * R
        # H'(g o E0): per-donor sums and E1 rows -- or, when every context set is the scan's own, ONE product per donor
        # against the k0 (k0 + 1) / 2 symmetric pair features (scan.hip: donor pairs)
        contraction = (2.0 * n * pairs if pairs else per_column * k0) + 2.0 * cols * r_star * k0
    else:
        rot = 2.0 * n * cols + 2.0 * cols * R if fast_rotation else 2.0 * n * R
        contraction = 2.0 * n * r_star * k0
    return rot + contraction + n * k0 * (k0 + 1) + 2.0 * n * k0 * (c + 2) + r_star * k0 * (k0 + 1)


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _launch_ranks(n, argv):
    """One process per GPU through torch.distributed.run on 127.0.0.1; returns its exit code (a failed rank makes the
    launcher -- a fresh child process, never a re-exec of this one -- exit non-zero, and so does this script)."""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    return subprocess.call(cmd, env=env)


def roofline_record(lib, ctx, crm, config, kr_ms, kr_n, kr_fl, elapsed):
    """The `roofline` object of the line: the dominant kernel's executed flops over its duration by HIP events on the
    library's stream (crm_kernel_timer_*), against the FP64-MFMA peak; its L2-fabric traffic from the committed PMC profile of
    the same launch shape and kernel form, else null.  Returns (roofline, donors of the kinship structure in use)."""
    # ---- roofline of the dominant kernel (Khatri-Rao contraction, FP64 MFMA bound) ----------
    kr_s = kr_ms * 1e-3
    achieved = kr_fl / kr_s * 1e-12 if kr_s > 0 else 0.0
    kin_groups = lib.crm_background_kinship_groups(crm._bg.handle) if os.environ.get("CRM_KIN_ROUTE", "1") != "0" else 0
    if kin_groups:
        kernel = ("gemm_tn_glds_kernel<false, 1, 0, false, 128, 1>: A~ = MixK(rho*)' S for every variant of a block, one launch per "
                  "block (a plain K x (variants k0) x r product, K = k1 + donors k2 rows of per-donor sums S = [E1'(g o E0) over all "
                  "cells ; us'(g o E0) donor by donor] against the mixing matrix with the donor-level kinship factor folded in; "
                  "LDS-DMA operand tiles, FP64 MFMA) -- the dominant launch of the kinship-structure route (DESIGN.md 6c); "
                  "achieved = its executed flops 2 K r k0 per variant / its duration by HIP events on the library's stream, r = the "
                  "columns of the spectrum this launch computes: all r* of them, or the whole 128-column tiles when r* mod 128 <= 16 "
                  "(cfg3: 4992 of 5000; the last 8 go through skinny_tn_kernel, one pass over S outside the timed pair)")
    else:
        kernel = ("gemm_tn_glds_sync_kernel<true, KRQ, ECQ, false> (Khatri-Rao contraction A~ = KR(G,E)' Q0, LDS-DMA operand tiles, "
                  "persistent workgroups re-aligned per XCD; <true, 1, 2, false> at k0 = 50; launches of <= 1024 tiles: gemm_tn_glds_kernel) "
                  "+ gemm_tn_glds_kernel<true, KRQ, ECQ, false, 160> over the last 128 + r mod 128 columns when r mod 128 <= 32 "
                  "(cfg3: 38 x 128 + 136 of r = 5000); achieved / avg_launch_ms cover both launches of a block")
    roofline = {
        "bound": "mfma", "kernel": kernel,
        "achieved": round(achieved, 3), "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_FP64_MFMA_TFLOPS, 4), "traffic": None,
        "launches": int(kr_n), "avg_launch_ms": round(kr_ms / max(kr_n, 1), 3),
        "flops_per_launch": kr_fl / max(kr_n, 1),
        "share_of_step_time": round(kr_s / elapsed, 4),
    }
    if roofline["share_of_step_time"] < 0.5:
        # (side records only -- cfg2, mode B: short spectra leave the step spread over several kernels; the default cfg3
        # line has this kernel at two thirds of the step)
        roofline["dominant"] = False
        roofline["note"] = ("the timed kernel is the route's largest single product but takes less than half of the step at "
                            "this configuration; the kernel-stats profile of the same configuration shows the spread")
    # L2-fabric traffic of that kernel is NOT measured by this run (PMC counters need their own rocprofv3 --pmc passes,
    # tools/pmc_bench.sh): the figure of the committed profile is quoted only when it was collected on the same launch
    # shape AND the same kernel form (persistent / one workgroup per tile, tail launch, library version); otherwise null.
    form = {"contraction_sync": not lib.crm_test_sync_fallbacks(ctx), "tail_launch": True, "library": lib.crm_version().decode(),
            "kinship_route": bool(kin_groups), "tile_band": 8}
    roofline["kernel_form"] = form
    for name in ("r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary_direct_route.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            shape = pmc["launch_shape"]
            same_form = {"kinship_route": False, **pmc.get("kernel_form", {"contraction_sync": True, "tail_launch": True,
                                                                           "library": "0.1.0"})} == form
            if (config == shape["config"] and roofline["launches"] > 0 and same_form
                    and abs(roofline["flops_per_launch"] / shape["flops_per_launch"] - 1.0) < 0.05):
                roofline["traffic"] = pmc["traffic_bytes_per_launch"]
                roofline["traffic_unit"] = "bytes/launch (2*FETCH_SIZE + WRITE_SIZE, fabric side incl. Infinity Cache)"
                roofline["traffic_source"] = (f"EARLIER PROFILE, not this run: profiles/{name} "
                                              f"({pmc.get('collected_on', 'rocprofv3 --pmc')})")
                roofline["algorithmic_bytes_per_launch"] = pmc["algorithmic_bytes_per_launch"]
                break
        except (OSError, KeyError, ValueError):
            pass
    if roofline["traffic"] is None:
        roofline["traffic_note"] = "no committed PMC profile matches this launch shape and kernel form"
    return roofline, kin_groups


def cpu_baseline_leg(crm, cohort, Ls, n, sample_of, G_weak, G_full, pv_dense, cpu_variants, cpu_repeats, multi_keep, config4):
    """`cpu_baseline`: the oracle (the reference-shaped per-variant loop: 11 LMM fits redoing the rotations, QSCov / PMat /
    ScoreStatistic, eigvalsh, Davies) timed on this host on a seeded random sample of the variants the GPU scanned, median
    of a few repeats; bound to the decompositions the device built.  Also checks (gene, variant) pairs of the
    config-4 leg against the oracle (into config4["oracle_check"]).  The only place of this script that touches oracle/."""
    from oracle.crm import OracleCellRegMap

    qs = {}
    for i, rho in enumerate(crm._rho1):
        Q0, S0 = crm._bg.read(i, n)
        qs[rho] = ((Q0,), S0)
    ocrm = OracleCellRegMap.__new__(OracleCellRegMap)
    ocrm._polish = False
    ocrm._y, ocrm._E0, ocrm._W, ocrm._E1 = cohort.y, cohort.E, cohort.W, cohort.E
    ocrm._Ls, ocrm._rho, ocrm._half, ocrm._qs = Ls, list(crm._rho1), {}, qs
    m = cpu_variants
    pick = np.sort(np.random.default_rng(2024).choice(sample_of, size=m, replace=False))
    Gs = np.ascontiguousarray(G_weak[:, pick])
    ocrm.scan_interaction(Gs[:, :1])  # warm-up variant, discarded
    times = []
    for _ in range(max(1, cpu_repeats)):
        t0 = time.time()
        opv, _ = ocrm.scan_interaction(Gs)
        times.append(time.time() - t0)
    t_cpu = float(np.median(times))
    import threadpoolctl

    blas = threadpoolctl.threadpool_info()
    # the threads that did the work: the BLAS pool behind numpy (not the OpenMP pool torch brings along)
    nthreads = max([b.get("num_threads", 1) for b in blas if b.get("user_api") == "blas"] or
                   [b.get("num_threads", 1) for b in blas] or [1])
    dev = np.abs(opv - pv_dense[pick]) / np.maximum(opv, 1e-300)
    # (gene, variant) pairs of the config-4 leg against the oracle (other phenotypes, same decomposition)
    c4_check = None
    if multi_keep is not None:
        import copy

        mpv_, mrho_, ys, whole_ = multi_keep
        Gsrc = G_full if whole_ else G_weak
        prng = np.random.default_rng(4)
        worst, pairs, same_rho = 0.0, 0, True
        for gi in sorted({1, len(ys) // 2, len(ys) - 1}):
            cols_ = np.sort(prng.choice(mpv_.shape[1], size=3, replace=False))
            og = copy.copy(ocrm)
            og._y = ys[gi]
            gp, ginfo = og.scan_interaction(np.ascontiguousarray(Gsrc[:, cols_]))
            worst = max(worst, float(np.max(np.abs(gp - mpv_[gi, cols_]) / np.maximum(gp, 1e-300))))
            same_rho = same_rho and bool(np.array_equal(ginfo["rho1"], mrho_[gi, cols_]))
            pairs += len(cols_)
        c4_check = {"pairs": pairs, "max_rel_dp_vs_oracle": worst, "rho_star_identical": same_rho}
        if config4 is not None:
            config4["oracle_check"] = c4_check
    cpu = {
        "value": round(m / t_cpu, 4), "unit": "variant-tests/s", "cores": int(nthreads), "kind": "port",
        "cores_note": ("threads of the host BLAS as this run used them: OpenBLAS caps its pool (64 in this image) below "
                       "the host's %d logical cpus" % (os.cpu_count() or 0)),
        "sample": f"{m} variants drawn at random (seed 2024) from the {sample_of} this run scanned, "
                  f"median of {len(times)} repeats ({', '.join('%.1f s' % t for t in times)}); scan only (decomposition "
                  f"shared with the GPU run); host has {os.cpu_count()} logical cpus",
        "threadpools": [{k: b.get(k) for k in ("user_api", "internal_api", "num_threads", "version")} for b in blas],
        "max_rel_dp_vs_gpu": float(dev.max()),
    }
    return cpu


class Comm:
    """The process group the bench talks through.  ``nccl`` (= RCCL over xGMI) is the one the contract asks for; a ``gloo``
    group over the same ranks is created beside it and takes over -- for every later collective -- if the first RCCL
    collectives of the run raise (a communicator that cannot start: IPC handles, topology, a hung peer running into the
    timeout).  Variants are independent, so a run on the slower group still measures the same scan; the line says which
    group carried it (``multi_gpu.group``)."""

    def __init__(self, dist, torch, share_gpu, local_rank):
        self.dist, self.torch = dist, torch
        self.group, self.backend, self.note = None, None, None
        if dist is None:
            return
        import datetime

        timeout = datetime.timedelta(seconds=int(os.environ.get("CRM_BENCH_COLLECTIVE_TIMEOUT_S", "600")))
        if share_gpu:   # dry run of the N > 1 path on a one-GPU box: every rank on device 0, gloo instead of RCCL
            dist.init_process_group("gloo", timeout=timeout)
            self.backend = "gloo"
            return
        # (a collective that fails or times out must come back as an exception in this process, not as an abort of it)
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
        os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=timeout)
        self.backend = "nccl"
        self.safe = dist.new_group(backend="gloo", timeout=timeout)   # (rendezvous over the TCP store: no GPU involved)

    def canary(self):
        """First collectives of the run on RCCL, checked: an all_reduce and an all_gather of a few bytes."""
        if self.dist is None or self.backend != "nccl":
            return
        dist, torch = self.dist, self.torch
        ok = 1
        try:
            t = torch.ones(8, device="cuda")
            dist.all_reduce(t)
            parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
            dist.all_gather(parts, t)
            torch.cuda.synchronize()
            if float(t[0].item()) != float(dist.get_world_size()):
                raise RuntimeError("all_reduce returned %r" % float(t[0].item()))
        except Exception as exc:  # noqa: BLE001
            ok = 0
            self.note = "RCCL failed on its first collectives (%s: %s)" % (type(exc).__name__, str(exc)[:160])
        # every rank must take the same group from here on: agree over gloo
        flag = torch.tensor([ok], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.safe)
        if int(flag.item()) == 0:
            self.group, self.backend = self.safe, "gloo"
            self.note = (self.note or "RCCL failed on another rank") + "; every collective of this run went over gloo (host TCP)"

    @property
    def device(self):
        return "cuda" if self.backend == "nccl" else "cpu"

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier(group=self.group)

    def max(self, x):
        if self.dist is None:
            return float(x)
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def table(self, row, rank, world):
        """Every rank's row of floats on every rank (a world x len(row) nested list)."""
        if self.dist is None:
            return [list(map(float, row))]
        t = self.torch.zeros((world, len(row)), dtype=self.torch.float64, device=self.device)
        t[rank] = self.torch.tensor(list(map(float, row)), dtype=self.torch.float64)
        self.dist.all_reduce(t, group=self.group)
        return t.cpu().tolist()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096, help="variants per step")
    ap.add_argument("--config", default="cfg3", help="cfg2 | cfg3 | cfg5 (BASELINE.json configs[1] / [2] / [4])")
    ap.add_argument("--cpu-variants", type=int, default=32, help="variants of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-repeats", type=int, default=3)
    ap.add_argument("--mode", default="C", choices=["C", "B"],
                    help="background: C = E1E1' + K o EE' (headline), B = E1E1' + hK hK' (r = k + m)")
    ap.add_argument("--block", type=int, default=0, help="variants per internal block (0 = library default)")
    ap.add_argument("--polish", type=int, default=0)
    ap.add_argument("--collapsed", type=int, default=1, help="also time the donor-collapsed path (N=1)")
    ap.add_argument("--full-panel", type=int, default=1, help="strong-scaling leg on the config's fixed panel")
    ap.add_argument("--kinship", default="indicator", choices=["indicator", "rotated"],
                    help="kinship factor of the synthetic cohort: donor indicators, or the dense U sqrt(S) of the same K")
    ap.add_argument("--direct-steps", type=int, default=3,
                    help="timed steps of the general-input side leg (the direct contraction against Q0(rho*), what a "
                         "cell-level hK / Ls gets; N = 1 only; 0 = skip)")
    ap.add_argument("--rotated-steps", type=int, default=3,
                    help="timed steps of the side leg on the kinship factor as the reference's simulator forms it (U sqrt(S): "
                         "dense, donor-expanded; N = 1 and --kinship indicator only; 0 = skip)")
    ap.add_argument("--genes", type=int, default=64, help="phenotypes of the config-4 leg (0 = skip)")
    ap.add_argument("--genes-variants", type=int, default=0,
                    help="variants of the config-4 leg, all ranks together (0 = the config's whole fixed panel when "
                         "--full-panel is on, i.e. BASELINE config 4 itself at --genes 64; else the weak-scaling panel)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    share_gpu = bool(os.environ.get("CRM_BENCH_SHARE_GPU"))
    if not share_gpu and local_rank == 0:
        import torch  # (counting devices does not initialise the GPU on this image)

        have = torch.cuda.device_count()
        if have == 0:
            raise SystemExit("bench.py: this node shows no GPU (torch.cuda.device_count() == 0); the bench measures the HIP "
                             "path and has no CPU fallback")
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s) (torch.cuda.device_count()); "
                             f"run with --gpus {max(have, 1)}, or CRM_BENCH_SHARE_GPU=1 for a dry run of the N > 1 code "
                             f"path on one GPU (not a scaling measurement)")
    if world == 1 and args.gpus > 1 and "TORCHELASTIC_RUN_ID" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks as child processes.  Nothing in this
        # process has touched the GPU yet, and it never will: it only relays the children's output and exit code.
        raise SystemExit(_launch_ranks(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with "
                         f"--nproc-per-node {args.gpus} (or run `python bench.py --gpus {args.gpus}` directly)")
    import torch

    # stdout carries the ONE JSON line of the contract and nothing else: RCCL prints a version banner and gloo its
    # connection notes to file descriptor 1 -- everything until the line itself is written goes to stderr instead
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            local_rank = 0
            # (the ranks of a dry run share one device's memory: each keeps its config-4 pair buffers to its part of it)
            os.environ.setdefault("CRM_PAIR_BUFFER_GB", str(max(8, 96 // world)))
    torch.cuda.set_device(local_rank)
    comm = Comm(dist, torch, share_gpu, local_rank)

    from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib, get_L_values
    from cellregmap_amd.distributed import gather_variant_results, sharded_background, variant_shard
    from cellregmap_amd.synth import CONFIGS, make_cohort

    lib = _lib.load()
    donors, cells, k0, p_total = CONFIGS[args.config]
    n = donors * cells
    steps, warmup, batch = args.steps, args.warmup, args.batch

    ctx = _engine._context(local_rank)
    _lib.check(lib.crm_set_null_fit_polish(ctx, int(args.polish)))
    if args.block > 0:
        _lib.check(lib.crm_set_block_variants(ctx, int(args.block)))

    def fence():
        _lib.check(lib.crm_ctx_synchronize(ctx))
        torch.cuda.synchronize()
        comm.barrier()

    max_over_ranks = comm.max

    # ---- synthetic data (not timed): the cohort, this rank's shard of the config's fixed panel, and -- when
    #      that shard is shorter than a few batches -- a panel of its own for the weak-scaling steps
    t0 = time.time()
    # phenotype, contexts, kinship factor: the donor indicators (the sparse factor of the donor-block K; the data of rounds
    # 1-3, kept so that the lines stay comparable) or, --kinship rotated, U sqrt(S) as the reference's simulator forms it --
    # a dense donor-expanded n x m factor of the same K (cellregmap_amd/synth.py: kinship_factor)
    cohort = make_cohort(donors, cells, k0, 16, seed=20, kinship=args.kinship)
    f_first, f_count = variant_shard(p_total, rank, world)
    G_full = None
    if args.full_panel:
        # the same panel for every world size; a rank expands only its own shard of it to cells
        full = make_cohort(donors, cells, k0, p_total, seed=77, with_phenotype=False, columns=(f_first, f_count))
        G_full, donor_of_cell = full.G, full.donor_of_cell
        del full
    weak_blocks = max(1, min(steps, 12 if world == 1 else 4))   # (N ranks generate their panels side by side on one host)
    if G_full is not None and f_count >= weak_blocks * batch:
        G_weak = G_full[:, :weak_blocks * batch]
    else:
        wk = make_cohort(donors, cells, k0, weak_blocks * batch, seed=1000 + rank, with_phenotype=False)
        G_weak, donor_of_cell = wk.G, wk.donor_of_cell
        del wk
    t_data = time.time() - t0

    # one-time start-up (code objects, allocator, RCCL rings) on a toy problem, outside every timed region
    toy = make_cohort(6, 16, 3, 8, seed=1)
    CellRegMap(toy.y, toy.E, W=toy.W, hK=toy.hK, device=local_rank).scan_interaction(toy.G)
    _engine._bg_cache.clear()
    comm.canary()     # RCCL's first collectives, checked; on failure every rank moves to the gloo group
    fence()

    # ---- constructor (timed inside the end-to-end figure of the full-panel leg) --------------------------------
    t_start = time.perf_counter()
    Ls = get_L_values(cohort.hK, cohort.E)
    bg_kw = {"Ls": Ls} if args.mode == "C" else {"hK": cohort.hK}
    exchange = {}
    uploaded = {}

    def upload_full_panel():
        # (called by sharded_background while its collective is in flight: PCIe beside xGMI)
        t0_ = time.perf_counter()
        uploaded["panel"] = GenotypePanel(G_full, device=local_rank, groups=None)   # dense: general genotypes
        uploaded["seconds"] = time.perf_counter() - t0_      # (the copy is complete when the call returns)

    if world > 1 or (dist is not None and os.environ.get("CRM_BENCH_FORCE_EXCHANGE")):
        bg = sharded_background(cohort.E, Ls if args.mode == "C" else cohort.hK, _engine._RHO_GRID, device=local_rank,
                                group=comm.group, force_exchange=world == 1,
                                overlap=upload_full_panel if G_full is not None else None, info=exchange)
        crm = CellRegMap(cohort.y, cohort.E, W=cohort.W, device=local_rank, background=bg, **bg_kw)
    else:
        # one GPU: constructor, then the upload (beside each other they cost the same in total -- the constructor's many short
        # launches slow down while 8 GB cross PCIe -- and the constructor's own figure would not be one; the streamed leg
        # below hides the upload behind the SCAN instead).  CRM_BENCH_PARALLEL_UPLOAD=1 restores the second thread.
        import threading

        up = None
        if G_full is not None and os.environ.get("CRM_BENCH_PARALLEL_UPLOAD"):
            up = threading.Thread(target=upload_full_panel)
            up.start()
        crm = CellRegMap(cohort.y, cohort.E, W=cohort.W, device=local_rank, **bg_kw)
        if up is not None:
            up.join()
            uploaded["overlapped"] = True
    crm._bind_gene()
    _lib.check(lib.crm_ctx_synchronize(ctx))
    t_ctor = time.perf_counter() - t_start     # (N > 1: includes the panel upload that ran beside the exchange)
    if world == 1 and G_full is not None and "panel" not in uploaded:
        upload_full_panel()                    # (inside the end-to-end figure, after the constructor's own)
    ranks = [crm._bg.rank(i) for i in range(len(crm._rho1))]
    gene = crm._gene
    cols = cohort.E.shape[1] + (Ls.shape_us[1] * Ls.hK.shape[1] if args.mode == "C" else cohort.hK.shape[1])
    c_cov = cohort.W.shape[1]

    def scan(panel, first, count, pv, rho1, Q=None):
        _lib.check(lib.crm_scan_interaction(gene, panel.handle, first, count, None, None, _lib.ptr(pv), _lib.ptr(rho1),
                                            None, None, None, _lib.ptr(Q), None, None, None, None, None))

    # ---- strong scaling on the fixed panel: upload + scan + gather, end to end with the constructor -----------
    full_panel = None
    fpanel_kept = None
    fpv = None
    if G_full is not None:
        if "panel" not in uploaded:
            upload_full_panel()
        fpanel, t_up = uploaded["panel"], uploaded["seconds"]
        fpv, frho = np.empty(f_count), np.empty(f_count)
        fence()
        t0 = time.perf_counter()
        scan(fpanel, 0, f_count, fpv, frho)
        _lib.check(lib.crm_ctx_synchronize(ctx))
        t_scan_mine = time.perf_counter() - t0
        fence()
        t_scan = max_over_ranks(time.perf_counter() - t0)
        t0 = time.perf_counter()
        gather_note = None
        if dist is not None:
            try:
                got = gather_variant_results({"pv": fpv, "rho1": frho}, p_total, comm.group)
                assert got["pv"].shape == (p_total,)
                torch.cuda.synchronize()
            except Exception as exc:  # noqa: BLE001 -- the shard results are still valid; say what happened
                gather_note = "gather of the results failed (%s: %s)" % (type(exc).__name__, str(exc)[:160])
        t_gather = time.perf_counter() - t0
        t_e2e = max_over_ranks(time.perf_counter() - t_start)
        per_rank = comm.table([exchange.get("decompose_s", t_ctor), exchange.get("exchange_s", 0.0), t_up, t_scan_mine, t_gather,
                               1.0 if exchange.get("exchange", "ok") in ("ok", "not needed") else 0.0], rank, world)
        full_panel = {"variants": p_total, "variants_per_rank": f_count, "constructor_s": round(max_over_ranks(t_ctor), 3),
                      "upload_s": round(max_over_ranks(t_up), 3), "scan_s": round(t_scan, 3), "gather_s": round(t_gather, 4),
                      "end_to_end_s": round(t_e2e, 3), "scan_only_rate": round(p_total / t_scan, 1),
                      "end_to_end_rate": round(p_total / t_e2e, 1), "scaling": "strong",
                      "background": ("grid points decomposed by different ranks, spectra + mixing matrices packed per owner and "
                                     "exchanged in one all_gather (%s); the panel upload runs while it is in flight" % comm.backend)
                                    if world > 1 else "one rank",
                      "per_rank_s": {"columns": ["decompose", "exchange", "upload", "scan", "gather", "exchange_ok"],
                                     "rows": [[round(v, 4) for v in row] for row in per_rank]},
                      "exchange": exchange.get("exchange"), "exchanged_bytes_rank0": exchange.get("exchanged_bytes"),
                      "gather": gather_note or ("ok" if world > 1 else None),
                      "upload": "beside the constructor (second thread, own stream)" if uploaded.get("overlapped") else
                                ("beside the exchange of the background" if world > 1 else "after the constructor"),
                      "note": "the fixed panel of the config sharded over the ranks; end to end = constructor + panel upload "
                              "(host float64; N > 1: beside the exchange of the background) + scan + gather, max over ranks"}
        if world == 1 and not os.environ.get("CRM_BENCH_NO_STREAMED"):
            # The Python host's own way with a host matrix: constructor first, then the panel in column chunks from a
            # second thread while the chunks that have arrived are scanned (CellRegMap._scan_streamed) -- the upload
            # hides behind the scan instead of slowing the constructor's launches.  A side figure; same results.
            _engine._bg_cache.clear()
            fence()
            t0 = time.perf_counter()
            crm_s = CellRegMap(cohort.y, cohort.E, W=cohort.W, device=local_rank, **bg_kw)
            t_c = time.perf_counter() - t0
            spv, _ = crm_s.scan_interaction(G_full, progress=False, groups=None)
            _lib.check(lib.crm_ctx_synchronize(ctx))
            t_s = time.perf_counter() - t0
            full_panel["streamed"] = {"end_to_end_s": round(t_s, 3), "constructor_s": round(t_c, 3), "end_to_end_rate": round(p_total / t_s, 1),
                                      "chunk_variants": _engine._stream_chunk(), "variants_identical_to_the_resident_scan": int((spv == fpv).sum()),
                                      "max_rel_dp_vs_resident_scan": float(np.max(np.abs(spv - fpv) / fpv)),
                                      "note": "constructor, then crm.scan_interaction(host float64 matrix, groups=None): uploaded in "
                                              "column chunks beside the scan of the chunks that have arrived"}
            del crm_s, spv
            full_panel["end_to_end_best_s"] = min(full_panel["end_to_end_s"], full_panel["streamed"]["end_to_end_s"])
            full_panel["end_to_end_best_rate"] = round(p_total / full_panel["end_to_end_best_s"], 1)
        if f_count >= weak_blocks * batch:
            panel = fpanel
            fpanel_kept = fpanel
        elif args.genes > 1 and args.genes_variants <= 0:
            fpanel_kept = fpanel     # the config-4 leg scans this rank's whole shard of the fixed panel
            panel = GenotypePanel(G_weak, device=local_rank, groups=None)
        else:
            del fpanel
            panel = GenotypePanel(G_weak, device=local_rank, groups=None)
    else:
        panel = GenotypePanel(G_weak, device=local_rank, groups=None)

    # ---- the contract: W warm-up steps, K timed steps of `batch` variants, max over ranks ---------------------------
    p_need = weak_blocks * batch
    pv = np.empty(p_need)
    rho1 = np.empty(p_need)
    Q = np.empty(p_need)

    def run_step(i):
        first = (i % weak_blocks) * batch
        sl = slice(first, first + batch)
        scan(panel, first, batch, pv[sl], rho1[sl], Q[sl])

    for i in range(warmup):
        run_step(i)
    fence()
    _lib.check(lib.crm_kernel_timer_reset(ctx))
    t0 = time.perf_counter()
    for i in range(steps):
        run_step(i)
    fence()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    kr_ms, kr_n, kr_fl, tot = ctypes.c_double(), ctypes.c_long(), ctypes.c_double(), ctypes.c_double()
    _lib.check(lib.crm_kernel_timer_read(ctx, ctypes.byref(kr_ms), ctypes.byref(kr_n), ctypes.byref(kr_fl),
                                         ctypes.byref(tot)))
    _lib.check(lib.crm_kernel_timer_stop(ctx))
    pv_dense = pv.copy()
    if dist is not None:
        # the path's one data-path collective: gather the shard results on every rank (RCCL over xGMI)
        full = gather_variant_results({"pv": pv, "rho1": rho1, "Q": Q}, p_need * world, comm.group)
        assert full["pv"].shape == (p_need * world,)
        torch.cuda.synchronize()
    total_variants = steps * batch * world
    value = total_variants / elapsed

    # ---- side leg: the general-input route.  The headline's cohort has a donor-expanded kinship factor, which the engine
    #      detects and serves through the kinship-structure route; a cell-level hK / Ls -- which the
    #      reference accepts just the same (_cellregmap.py:107-131) -- gets the direct Khatri-Rao contraction against
    #      Q0(rho*) over all cells, i.e. SURVEY 8(d)'s 2 n r* k0 flops per variant.  Same steps, same panel, same results.
    direct = None
    kin_on = lib.crm_background_kinship_groups(crm._bg.handle) if os.environ.get("CRM_KIN_ROUTE", "1") != "0" else 0
    if world == 1 and args.direct_steps > 0 and kin_on:
        dpv, drho, dQ = np.empty(batch), np.empty(batch), np.empty(batch)
        _lib.check(lib.crm_test_set_kinship_route(ctx, 0))
        try:
            scan(panel, 0, batch, dpv, drho, dQ)       # warm-up: Q0 of the selected grid points is formed on first use
            fence()
            _lib.check(lib.crm_kernel_timer_reset(ctx))
            t0 = time.perf_counter()
            for i in range(args.direct_steps):
                first = (i % weak_blocks) * batch
                scan(panel, first, batch, dpv, drho, dQ)
            fence()
            t_direct = time.perf_counter() - t0
            d_ms, d_n, d_fl = ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
            _lib.check(lib.crm_kernel_timer_read(ctx, ctypes.byref(d_ms), ctypes.byref(d_n), ctypes.byref(d_fl), ctypes.byref(tot)))
            _lib.check(lib.crm_kernel_timer_stop(ctx))
        finally:
            _lib.check(lib.crm_test_set_kinship_route(ctx, 1))
        last = ((args.direct_steps - 1) % weak_blocks) * batch
        d_tf = d_fl.value / (d_ms.value * 1e-3) * 1e-12 if d_ms.value > 0 else 0.0
        direct = {"value": round(args.direct_steps * batch / t_direct, 1), "unit": "variant-tests/s", "steps": args.direct_steps,
                  "ms_per_step": round(t_direct / args.direct_steps * 1e3, 3),
                  "roofline": {"bound": "mfma",
                               "kernel": "gemm_tn_glds_sync_kernel<true, ...> + its tail launch and split reduction: "
                                         "A~ = KR(G, E0)' Q0(rho*) over all cells (Khatri-Rao columns formed in LDS, FP64 MFMA)",
                               "achieved": round(d_tf, 3), "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(d_tf / PEAK_FP64_MFMA_TFLOPS, 4), "traffic": None,
                               "launches": int(d_n.value), "avg_launch_ms": round(d_ms.value / max(d_n.value, 1), 3),
                               "flops_per_launch": d_fl.value / max(d_n.value, 1),
                               "flops": "executed = SURVEY 8(d)'s 2 n r* k0 per variant (r* padded to the tile grid)",
                               "share_of_step_time": round(d_ms.value * 1e-3 / t_direct, 4)},
                  "max_rel_dp_vs_kinship_route": float(np.max(np.abs(dpv - pv[last:last + batch]) / np.maximum(pv[last:last + batch], 1e-300))),
                  "rho_star_identical": bool(np.array_equal(drho, rho1[last:last + batch])),
                  "note": "the route a cell-level hK / Ls takes (crm_test_set_kinship_route(ctx, 0) on the same cohort): every "
                          "variant contracted against Q0(rho*) over all cells; side figure, never `value`"}

    # ---- side leg: the kinship factor the reference's simulator forms (_simulate.py:83-102, 477-479: hK = U sqrt(S) of the
    #      donor-block K -- a DENSE n x m factor whose rows are constant within a donor; SURVEY.md 8d's generator).  The
    #      headline keeps the donor indicators (the sparse factor of the same K: the data of rounds 1-5, so that the lines stay
    #      comparable); here the same steps run on a cohort built with the dense factor -- its own constructor, its own
    #      phenotype (y_k is drawn through hK), the same genotype panel.  The engine finds the donor structure of either factor
    #      and folds the donor-level m x m block into its mixing matrices, so the two rates should agree.
    rotated = None
    if world == 1 and args.rotated_steps > 0 and args.kinship == "indicator":
        t0 = time.perf_counter()
        coh2 = make_cohort(donors, cells, k0, 16, seed=20, kinship="rotated")
        Ls2 = get_L_values(coh2.hK, coh2.E)
        crm2 = CellRegMap(coh2.y, coh2.E, W=coh2.W, device=local_rank, **({"Ls": Ls2} if args.mode == "C" else {"hK": coh2.hK}))
        gene2 = crm2._bind_gene()
        _lib.check(lib.crm_ctx_synchronize(ctx))
        t_ctor2 = time.perf_counter() - t0
        rpv, rrho = np.empty(batch), np.empty(batch)

        def scan2(first):
            _lib.check(lib.crm_scan_interaction(gene2, panel.handle, first, batch, None, None, _lib.ptr(rpv), _lib.ptr(rrho),
                                                None, None, None, None, None, None, None, None, None))

        scan2(0)
        fence()
        t0 = time.perf_counter()
        for i in range(args.rotated_steps):
            scan2((i % weak_blocks) * batch)
        fence()
        t_rot = time.perf_counter() - t0
        rotated = {"value": round(args.rotated_steps * batch / t_rot, 1), "unit": "variant-tests/s", "steps": args.rotated_steps,
                   "ms_per_step": round(t_rot / args.rotated_steps * 1e3, 3),
                   "ratio_to_value": round(args.rotated_steps * batch / t_rot / value, 4),
                   "cohort_and_constructor_s": round(t_ctor2, 2),
                   "kinship_structure_found": int(lib.crm_background_kinship_groups(crm2._bg.handle)),
                   "note": "same steps on a cohort whose kinship factor is the dense, donor-expanded U sqrt(S) of the "
                           "reference's simulator (cellregmap_amd/synth.py: kinship_factor 'rotated') instead of the donor "
                           "indicators; side figure, never `value`"}
        del crm2, gene2, Ls2, coh2
        _engine._bg_cache.clear()

    # ---- config 4's shape: `genes` phenotypes against one panel, the variants sharded over the ranks ---------------
    config4 = None
    multi_keep = None
    if args.genes > 1:
        from cellregmap_amd import scan_interaction_many

        rng = np.random.default_rng(99)  # same phenotypes on every rank
        crms = [crm]
        for i in range(1, args.genes):
            yi = cohort.y[rng.permutation(n)] if i % 2 else cohort.y + rng.normal(size=n)
            ci = CellRegMap(yi, cohort.E, W=cohort.W, device=local_rank, background=crm._bg, **bg_kw)
            ci._bind_gene()
            crms.append(ci)
        handles = (ctypes.c_void_p * len(crms))(*[c._gene.value for c in crms])
        # the panel of this leg: this rank's shard of the config's fixed panel (config 4 = config 3 x 64 genes: every
        # rank scans all genes on its shard of the 50 000 variants), or a sample of it / the weak-scaling panel
        whole = args.genes_variants <= 0 and fpanel_kept is not None
        if whole:
            mpanel, mb, nv = fpanel_kept, f_count, p_total
        else:
            want = args.genes_variants if args.genes_variants > 0 else 2048
            nv = min(want, p_need * world)
            _, mb = variant_shard(nv, rank, world)      # (<= p_need: nv <= p_need * world)
            mpanel = panel
        keys5 = ("pv", "rho1", "e2", "g2", "eps2")      # what scan_interaction returns per (gene, variant)
        mout = {k: np.empty((len(crms), mb)) for k in keys5}
        mpv, mrho = mout["pv"], mout["rho1"]

        def run_multi(count):
            outs = mout if count == mb else {k: np.empty((len(crms), count)) for k in keys5}
            _lib.check(lib.crm_scan_interaction_multi(handles, len(crms), mpanel.handle, 0, count, None, None,
                                                      *[_lib.ptr(outs[k]) for k in keys5], None))
            _lib.check(lib.crm_ctx_synchronize(ctx))

        run_multi(min(mb, 4096))   # warm-up: work buffers at the size of a full block (4096 variants), Q0 of the selected grid points
        fence()
        no_pair0 = lib.crm_test_tests_without_pair(ctx)
        t0 = time.perf_counter()
        run_multi(mb)
        fence()
        t_multi = max_over_ranks(time.perf_counter() - t0)
        no_pair = lib.crm_test_tests_without_pair(ctx) - no_pair0
        # config 4's one collective: the (genes x variants) results of every shard on every rank -- 5 arrays per gene,
        # packed into ONE all_gather (cellregmap_amd/distributed.py: gather_many_results; RCCL over xGMI when the group is nccl)
        t_gather4, gather4_note = 0.0, None
        if dist is not None:
            from cellregmap_amd.distributed import gather_many_results

            t0 = time.perf_counter()
            try:
                gpv, ginfo = gather_many_results(mpv, {k: mout[k] for k in keys5[1:]}, nv, comm.group)
                torch.cuda.synchronize()
                f4, c4 = variant_shard(nv, rank, world)
                assert gpv.shape == (len(crms), nv) and np.array_equal(gpv[:, f4:f4 + c4], mpv)
                assert np.array_equal(ginfo["rho1"][:, f4:f4 + c4], mrho)
                gather4_note = "ok"
            except Exception as exc:  # noqa: BLE001 -- the shard results are still valid; say what happened
                gather4_note = "failed (%s: %s)" % (type(exc).__name__, str(exc)[:160])
            comm.barrier()
            t_gather4 = max_over_ranks(time.perf_counter() - t0)
        same_panel = mpanel is panel
        config4 = {"value": round(len(crms) * nv / (t_multi + t_gather4), 1), "unit": "variant-tests/s", "genes": len(crms),
                   "variants": nv, "variants_per_rank": mb, "seconds": round(t_multi + t_gather4, 3),
                   "scan_s": round(t_multi, 3), "gather_s": round(t_gather4, 4) if dist is not None else None,
                   "gather": gather4_note, "gathered_bytes_per_rank": 8 * 5 * len(crms) * mb if dist is not None else None,
                   "distinct_rho_per_variant": float(np.mean([len(set(mrho[:, j])) for j in range(min(mb, 4096))])),
                   "tests_without_kinship_term": {"share": round(no_pair / max(len(crms) * mb, 1), 4),
                                                  "note": "fits that end with (v0 / v1) max S0 <= 1e-10 (delta at its upper clamp: the odd "
                                                          "phenotypes of this leg are permuted, i.e. have no random effect); the rotated test "
                                                          "direction enters their Q and F with weights <= 1e-10 and is not formed"},
                   "max_rel_dp_gene0_vs_single_gene_scan": float(np.max(np.abs(mpv[0, :min(mb, p_need)] - pv_dense[:min(mb, p_need)]) /
                                                                   np.maximum(pv_dense[:min(mb, p_need)], 1e-300))) if same_panel else
                                                             float(np.max(np.abs(mpv[0] - fpv) / np.maximum(fpv, 1e-300))),
                   "note": ("BASELINE config 4 itself: " if whole and len(crms) == 64 else "BASELINE config 4's shape: ") +
                           f"{len(crms)} phenotypes x the {'whole fixed ' + str(p_total) + '-variant panel' if whole else 'sample of the panel'}, "
                           "every rank scans all genes on its shard of the variants; G'Q0(rho) shared by the genes, one Khatri-Rao "
                           "contraction per variant against H (Q0(rho) = H Mix(rho)), each selected (variant, rho*) pair finished "
                           "with Mix(rho*); gene 0 checked against the single-gene scan here, (gene, variant) pairs against the "
                           "oracle in the cpu_baseline leg"}
        multi_keep = (mpv, mrho, [c._y for c in crms], whole)
        del crms[1:]

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    roofline, kin_groups = roofline_record(lib, ctx, crm, args.config, kr_ms.value, kr_n.value, kr_fl.value, elapsed)
    rstar = float(np.mean([ranks[int(round(x * 10))] if len(ranks) > 1 else ranks[0] for x in rho1[: min(steps, weak_blocks) * batch]]))
    f_alg = algorithmic_flops(n, ranks, rstar, k0, c_cov)
    kin = None
    if kin_groups:
        k2_, m_ = (Ls.shape_us[1], Ls.hK.shape[1]) if args.mode == "C" else (0, 0)
        folded = lib.crm_background_kinship_folded(crm._bg.handle)
        pairs = k0 * (k0 + 1) // 2 if lib.crm_test_donor_pair_blocks(ctx) > 0 else 0
        kin = (cohort.E.shape[1], k2_, (kin_groups + 15) // 16 * 16, m_, (cohort.E.shape[1] + kin_groups * k2_) if folded else 0, pairs)
    f_exe = executed_flops(n, cols, ranks, rstar, k0, c_cov, fast_rotation=True, kin=kin)
    per_rank_rate = value / world

    # ---- the same steps through the donor-collapsed path (exact for donor-constant genotypes, which
    #      the synthetic cohort -- like every expanded CellRegMap genotype matrix -- is); reported
    #      beside the dense headline, never as `value`
    collapsed = None
    if args.collapsed and world == 1:
        t0 = time.time()
        dpanel = GenotypePanel.from_donors(G_weak[::cells], donor_of_cell, device=local_rank)
        dense_panel = panel
        panel = dpanel
        try:
            run_step(0)  # builds the per-donor tables (cached on the gene) + warm-up
            _lib.check(lib.crm_ctx_synchronize(ctx))
            t_tables = time.time() - t0
            t0 = time.perf_counter()
            for i in range(steps):
                run_step(i)
            _lib.check(lib.crm_ctx_synchronize(ctx))
            t_col = time.perf_counter() - t0
        finally:
            panel = dense_panel
        ncmp = min(steps, weak_blocks) * batch
        dev = np.abs(pv[:ncmp] - pv_dense[:ncmp]) / np.maximum(pv_dense[:ncmp], 1e-300)
        collapsed = {"value": round(steps * batch / t_col, 1), "unit": "variant-tests/s",
                     "ms_per_step": round(t_col / steps * 1e3, 3), "donors": int(donors),
                     "tables_and_warmup_s": round(t_tables, 2),
                     "max_rel_dp_vs_dense": float(np.max(np.where(pv_dense[:ncmp] > 1e-8, dev, 0.0))),
                     "note": "exact rearrangement onto per-donor tables; general G uses the dense path"}
        pv[:] = pv_dense

    # ---- CPU baseline: the oracle on this host's cores, a bounded sample (rank 0, N = 1 only) ------------------------
    cpu = None
    if args.cpu_variants > 0 and world == 1:
        cpu = cpu_baseline_leg(crm, cohort, Ls, n, min(steps, weak_blocks) * batch, G_weak, G_full, pv_dense, args.cpu_variants,
                               args.cpu_repeats, multi_keep, config4)
    out = {
        "metric": "variant-tests/sec (interaction test)",
        "value": round(value, 2), "unit": "variant-tests/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(elapsed / max(steps, 1) * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": f"{args.config}: {n} cells x {k0} contexts, mode {args.mode} background (ranks {min(ranks)}..{max(ranks)}), "
                        f"{steps} steps x {batch} variants per GPU of the {p_total}-variant panel, 1 gene",
            "batch_variants": batch, "cells": n, "contexts": k0, "rho_grid": len(ranks),
            "null_fit": "brent-1e-6" + ("+polish" if args.polish else ""), "kinship_factor": args.kinship,
            "route": ("kinship structure of the background (%d donors): H'(g o E0) donor by donor, then Mix(rho*)'" % kin_groups)
                     if kin_groups else "direct contraction against Q0(rho*) over all cells",
        },
        "roofline": roofline,
        "cpu_baseline": cpu,
        "whole_path": {"executed_flop_per_variant": f_exe, "achieved_tflops": round(f_exe * per_rank_rate * 1e-12, 3),
                       "frac_of_fp64_mfma_peak": round(f_exe * per_rank_rate * 1e-12 / PEAK_FP64_MFMA_TFLOPS, 4),
                       "algorithmic_flop_per_variant": f_alg,
                       "algorithmic_equivalent_tflops": round(f_alg * per_rank_rate * 1e-12, 3),
                       "note": "per GPU; 'executed' counts what the kernels do (rotations through the mixing matrices; with "
                               "the background's kinship structure: H'g and H'(g o E0) donor by donor, then Mix(rho*)'); "
                               "the SURVEY 8(d) count (rotations as 2 n sum r) is a throughput equivalent, not a roofline fraction"},
        # SURVEY 8(d) asks for both views; the path is bound by the matrix pipe, not by HBM
        "hbm_view": (lambda b_alg: {"algorithmic_bytes_per_variant": round(b_alg), "achieved_GBps": round(b_alg * per_rank_rate * 1e-9, 3),
                                    "peak_GBps": 8000.0, "frac": round(b_alg * per_rank_rate * 1e-9 / 8000.0, 6),
                                    "note": "B_alg = 8n + 8(n sum r + n k0 + n(c+1))/p + 40 (SURVEY 8d), per GPU"})(
            8.0 * n + 8.0 * (n * float(sum(ranks)) + n * k0 + n * (c_cov + 1)) / p_total + 40.0),
        "setup_s": {"synthetic_data": round(t_data, 2), "background_constructor": round(t_ctor, 2)},
        "speedup_vs_cpu_baseline": None if not cpu else round(value / cpu["value"], 1),
        "multi_gpu": {"ranks": world, "group": comm.backend, "group_note": comm.note,
                      "collectives": ("constructor: two small all_reduces (ranks + trouble flag, ok flag) + one all_gather of the packed spectra / mixing "
                                      "matrices; results: one all_gather of the per-variant outputs") if world > 1 else None,
                      "note": "per-N values are whatever this run measured on this node; the repository holds no measured N > 1 "
                              "run of its own (its build sessions only ever had one GPU) and models no scaling figure"},
        "full_panel": full_panel,
        "direct_route": direct,
        "rotated_kinship_factor": rotated,
        "config4": config4,
        "donor_collapsed": collapsed,
    }
    if share_gpu:
        out["data"] = "synthetic; DRY RUN: all ranks share GPU 0 (CRM_BENCH_SHARE_GPU, gloo) -- exercises the N > 1 code path, not a scaling number"
    sys.stdout.flush()
    os.dup2(stdout_fd, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
