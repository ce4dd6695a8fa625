#!/usr/bin/env python3
"""Throughput of the interaction score test on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = the whole hot path (null fits over the rho grid, Khatri-Rao contraction, score
statistic, eigenvalues, Davies) over one batch of `--batch` synthetic variants of BASELINE
config 3 (20 000 cells, 50 contexts, mode C background K o EE' + EE', r ~ 5 000).  Inputs
(background decomposition, phenotype, genotype panel) are resident in HBM before the timed
region; the background constructor is timed separately.  N > 1: one process per GPU, variants
sharded across ranks (weak scaling: every rank runs K steps on its own shard), the only
collective is the final gather of p-values over RCCL.

Prints ONE JSON line on rank 0.
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

PEAK_FP64_MFMA_TFLOPS = 78.6  # MI355X FP64 matrix peak (vendor sheet)


def algorithmic_flops(n, r_list, r_star, k0, c):
    """F_alg per variant-test (dense general G)."""
    R = float(sum(r_list))
    return 2.0 * n * R + 2.0 * n * r_star * k0 + n * k0 * (k0 + 1) + 2.0 * n * k0 * (c + 2) + r_star * k0 * (k0 + 1)


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _launch_ranks(n, argv):
    """One process per GPU through torch.distributed.run on 127.0.0.1; returns its exit code."""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096, help="variants per step")
    ap.add_argument("--config", default="cfg3", help="cfg2 | cfg3 (BASELINE.json configs[1] / [2])")
    ap.add_argument("--cpu-variants", type=int, default=8, help="variants of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--mode", default="C", choices=["C", "B"],
                    help="background: C = E1E1' + K o EE' (headline), B = E1E1' + hK hK' (r = k + m)")
    ap.add_argument("--block", type=int, default=0, help="variants per internal block (0 = library default)")
    ap.add_argument("--polish", type=int, default=0)
    ap.add_argument("--collapsed", type=int, default=1, help="also time the donor-collapsed path (N=1)")
    ap.add_argument("--genes", type=int, default=16, help="phenotypes of the shared multi-gene leg (0 = skip, N=1)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "TORCHELASTIC_RUN_ID" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks as child processes.  Nothing in this
        # process has touched the GPU yet (torch is not even imported), and it never will: it only
        # relays the children's output and exit code.
        raise SystemExit(_launch_ranks(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with "
                         f"--nproc-per-node {args.gpus} (or run `python bench.py --gpus {args.gpus}` directly)")
    import torch

    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    from cellregmap_amd import CellRegMap, GenotypePanel, _engine, _lib, get_L_values
    from cellregmap_amd.synth import CONFIGS, make_cohort

    lib = _lib.load()
    donors, cells, k0, p_total = CONFIGS[args.config]
    n = donors * cells
    steps, warmup, batch = args.steps, args.warmup, args.batch
    p_need = batch * max(steps, 1)
    # every rank draws its own shard of variants (same cohort otherwise): seed offset on the panel
    t0 = time.time()
    cohort = make_cohort(donors, cells, k0, 16, seed=20)  # phenotype, contexts, kinship factor
    shard = make_cohort(donors, cells, k0, p_need, seed=1000 + rank, with_phenotype=False)
    G = shard.G
    t_data = time.time() - t0

    ctx = _engine._context(local_rank)
    _lib.check(lib.crm_set_null_fit_polish(ctx, int(args.polish)))
    if args.block > 0:
        _lib.check(lib.crm_set_block_variants(ctx, int(args.block)))
    t0 = time.time()
    Ls = get_L_values(cohort.hK, cohort.E)
    bg_kw = {"Ls": Ls} if args.mode == "C" else {"hK": cohort.hK}
    crm = CellRegMap(cohort.y, cohort.E, W=cohort.W, device=local_rank, **bg_kw)
    crm._bind_gene()
    _lib.check(lib.crm_ctx_synchronize(ctx))
    t_ctor = time.time() - t0
    ranks = [crm._bg.rank(i) for i in range(len(crm._rho1))]
    t0 = time.time()
    panel = GenotypePanel(G, device=local_rank, groups=None)  # dense: general genotypes
    t_upload = time.time() - t0

    gene = crm._gene
    pv = np.empty(p_need)
    rho1 = np.empty(p_need)
    Q = np.empty(p_need)

    def run_step(i):
        first = (i % max(steps, 1)) * batch
        sl = slice(first, first + batch)
        _lib.check(lib.crm_scan_interaction(
            gene, panel.handle, first, batch, None, None, _lib.ptr(pv[sl]), _lib.ptr(rho1[sl]), None, None,
            None, _lib.ptr(Q[sl]), None, None, None, None, None))

    def fence():
        _lib.check(lib.crm_ctx_synchronize(ctx))
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for i in range(warmup):
        run_step(i)
    fence()
    _lib.check(lib.crm_kernel_timer_reset(ctx))
    t0 = time.perf_counter()
    for i in range(steps):
        run_step(i)
    fence()
    elapsed = time.perf_counter() - t0
    pv_dense = pv.copy()
    kr_ms, kr_n, kr_fl, tot = ctypes.c_double(), ctypes.c_long(), ctypes.c_double(), ctypes.c_double()
    _lib.check(lib.crm_kernel_timer_read(ctx, ctypes.byref(kr_ms), ctypes.byref(kr_n), ctypes.byref(kr_fl),
                                         ctypes.byref(tot)))
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the path's one collective: gather the shard results on every rank (RCCL over xGMI)
        from cellregmap_amd.distributed import gather_variant_results

        full = gather_variant_results({"pv": pv, "rho1": rho1, "Q": Q}, p_need * world)
        assert full["pv"].shape == (p_need * world,)
        torch.cuda.synchronize()
    total_variants = steps * batch * world
    value = total_variants / elapsed

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (Khatri-Rao contraction, FP64 MFMA bound) ----------
    kr_s = kr_ms.value * 1e-3
    achieved = kr_fl.value / kr_s * 1e-12 if kr_s > 0 else 0.0
    roofline = {
        "bound": "mfma", "kernel": "gemm_tn_glds_kernel<true, KRQ, ECQ, false> (Khatri-Rao contraction A~ = KR(G,E)' Q0, LDS-DMA operand tiles; <true, 1, 2, false> at k0 = 50)",
        "achieved": round(achieved, 3), "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_FP64_MFMA_TFLOPS, 4), "traffic": None,
        "launches": int(kr_n.value), "avg_launch_ms": round(kr_ms.value / max(kr_n.value, 1), 3),
        "flops_per_launch": kr_fl.value / max(kr_n.value, 1),
        "share_of_step_time": round(kr_s / elapsed, 4),
    }
    # HBM-side traffic of that kernel comes from separate rocprofv3 --pmc passes (profiles/), valid for
    # the launch shape it was collected on
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
        shape = pmc["launch_shape"]
        if (args.config == shape["config"] and roofline["launches"] > 0
                and abs(roofline["flops_per_launch"] / shape["flops_per_launch"] - 1.0) < 0.05):
            roofline["traffic"] = pmc["traffic_bytes_per_launch"]
            roofline["traffic_unit"] = "bytes/launch (2*FETCH_SIZE + WRITE_SIZE, fabric side incl. Infinity Cache)"
            roofline["algorithmic_bytes_per_launch"] = pmc["algorithmic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    rstar = float(np.mean([ranks[int(round(x * 10))] if len(ranks) > 1 else ranks[0] for x in rho1[: steps * batch]]))
    f_alg = algorithmic_flops(n, ranks, rstar, k0, cohort.W.shape[1])
    whole_path_tflops = f_alg * (steps * batch) / elapsed * 1e-12

    # ---- the same steps through the donor-collapsed path (exact for donor-constant genotypes, which
    #      the synthetic cohort -- like every expanded CellRegMap genotype matrix -- is); reported
    #      beside the dense headline, never as `value`
    collapsed = None
    if args.collapsed and world == 1:
        t0 = time.time()
        dpanel = GenotypePanel.from_donors(G[::cells], shard.donor_of_cell, device=local_rank)
        dense_handle = panel.handle
        panel.handle = dpanel.handle
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
            panel.handle = dense_handle
        dev = np.abs(pv - pv_dense) / np.maximum(pv_dense, 1e-300)
        collapsed = {"value": round(steps * batch / t_col, 1), "unit": "variant-tests/s",
                     "ms_per_step": round(t_col / steps * 1e3, 3), "donors": int(donors),
                     "tables_and_warmup_s": round(t_tables, 2),
                     "max_rel_dp_vs_dense": float(np.max(np.where(pv_dense > 1e-8, dev, 0.0))),
                     "note": "exact rearrangement onto per-donor tables; general G uses the dense path"}
        pv[:] = pv_dense

    # ---- several phenotypes against the same (dense) panel in one pass: BASELINE config 4's shape ----
    multi = None
    if args.genes > 1 and world == 1:
        from cellregmap_amd import scan_interaction_many

        rng = np.random.default_rng(99)
        crms = [crm]
        for i in range(1, args.genes):
            yi = cohort.y[rng.permutation(n)] if i % 2 else cohort.y + rng.normal(size=n)
            ci = CellRegMap(yi, cohort.E, W=cohort.W, device=local_rank, background=crm._bg, **bg_kw)
            ci._bind_gene()
            crms.append(ci)
        handles = (ctypes.c_void_p * len(crms))(*[c._gene.value for c in crms])
        mb = min(batch, 1024)
        mpv = np.empty((len(crms), mb)); mrho = np.empty((len(crms), mb))

        def run_multi():
            _lib.check(lib.crm_scan_interaction_multi(handles, len(crms), panel.handle, 0, mb, None, None,
                                                      _lib.ptr(mpv), _lib.ptr(mrho), None, None, None, None))
            _lib.check(lib.crm_ctx_synchronize(ctx))

        run_multi()
        t0 = time.perf_counter()
        run_multi()
        t_multi = time.perf_counter() - t0
        multi = {"value": round(len(crms) * mb / t_multi, 1), "unit": "variant-tests/s", "genes": len(crms),
                 "variants": mb, "distinct_rho_per_variant": float(np.mean([len(set(mrho[:, j])) for j in range(mb)])),
                 "max_rel_dp_gene0_vs_single_gene_scan": float(np.max(np.abs(mpv[0] - pv_dense[:mb]) / pv_dense[:mb])),
                 "note": "dense path; G'Q0(rho) shared by the genes; the Khatri-Rao contraction runs once per variant "
                         "against H (Q0(rho) = H Mix(rho)) and each selected (variant, rho*) pair is finished with Mix(rho*)"}
        del crms[1:]

    # ---- CPU baseline: the oracle (reference-shaped per-variant loop) on this host -------------
    cpu = None
    if args.cpu_variants > 0 and world == 1:
        from oracle.crm import OracleCellRegMap

        t0 = time.time()
        qs = {}
        for i, rho in enumerate(crm._rho1):
            Q0, S0 = crm._bg.read(i, n)
            qs[rho] = ((Q0,), S0)
        ocrm = OracleCellRegMap.__new__(OracleCellRegMap)
        ocrm._polish = False
        ocrm._y, ocrm._E0, ocrm._W, ocrm._E1 = cohort.y, cohort.E, cohort.W, cohort.E
        ocrm._Ls, ocrm._rho, ocrm._half, ocrm._qs = Ls, list(crm._rho1), {}, qs
        t_read = time.time() - t0
        m = args.cpu_variants
        ocrm.scan_interaction(G[:, :1])  # warm-up variant, discarded
        t0 = time.time()
        opv, _ = ocrm.scan_interaction(G[:, :m])
        t_cpu = time.time() - t0
        import threadpoolctl

        blas = threadpoolctl.threadpool_info()
        nthreads = max([b.get("num_threads", 1) for b in blas] or [1])
        dev = np.abs(opv - pv[:m]) / np.maximum(opv, 1e-300)
        cpu = {
            "value": round(m / t_cpu, 4), "unit": "variant-tests/s", "cores": int(nthreads), "kind": "port",
            "sample": f"first {m} variants of the GPU shard, scan only (decomposition shared with the GPU run), "
                      f"numpy on {blas[0].get('internal_api', '?') if blas else '?'} with {nthreads} threads, "
                      f"host has {os.cpu_count()} logical cpus",
            "max_rel_dp_vs_gpu": float(dev.max()),
        }
    out = {
        "metric": "variant-tests/sec (interaction test)",
        "value": round(value, 2), "unit": "variant-tests/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(elapsed / max(steps, 1) * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": f"{args.config}: {n} cells x {k0} contexts, mode {args.mode} background (ranks {min(ranks)}..{max(ranks)}), "
                        f"{steps} steps x {batch} variants per GPU of the {p_total}-variant panel, 1 gene",
            "batch_variants": batch, "cells": n, "contexts": k0, "rho_grid": len(ranks),
            "null_fit": "brent-1e-6" + ("+polish" if args.polish else ""),
        },
        "roofline": roofline,
        "cpu_baseline": cpu,
        "whole_path": {"algorithmic_flop_per_variant": f_alg, "achieved_tflops": round(whole_path_tflops, 3),
                       "frac_of_fp64_mfma_peak": round(whole_path_tflops / PEAK_FP64_MFMA_TFLOPS, 4),
                       "note": "SURVEY 8(d) flop count (rotations as 2 n sum r) over wall time; the engine takes the "
                               "rotations through the mixing matrices and executes fewer flops, so this can exceed 1"},
        # SURVEY 8(d) asks for both views; the path is bound by the matrix pipe, not by HBM
        "hbm_view": (lambda b_alg: {"algorithmic_bytes_per_variant": round(b_alg), "achieved_GBps": round(b_alg * value / world * 1e-9, 3),
                                    "peak_GBps": 8000.0, "frac": round(b_alg * value / world * 1e-9 / 8000.0, 6),
                                    "note": "B_alg = 8n + 8(n sum r + n k0 + n(c+1))/p + 40 (SURVEY 8d), per GPU"})(
            8.0 * n + 8.0 * (n * float(sum(ranks)) + n * k0 + n * (cohort.W.shape[1] + 1)) / p_total + 40.0),
        "setup_s": {"synthetic_data": round(t_data, 2), "background_constructor": round(t_ctor, 2),
                    "panel_upload": round(t_upload, 2)},
        "speedup_vs_cpu_baseline": None if not cpu else round(value / cpu["value"], 1),
        "donor_collapsed": collapsed,
        "multi_gene": multi,
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
