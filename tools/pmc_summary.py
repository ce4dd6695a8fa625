"""Turn the CSVs of tools/pmc_bench.sh into profiles/r06_pmc_summary.json.

A block of 4096 variants is one large Khatri-Rao launch (gemm_tn_glds_sync_kernel) plus, when the spectrum is a little
longer than a multiple of the 128-column tile, a second launch of 160-column tiles for the last columns; the counters of
both are added per block.  rocprofv3 reports FETCH_SIZE / WRITE_SIZE in units of 1024 bytes; FETCH_SIZE is doubled for
16-byte-per-lane streams (the gfx950 correction of MI355X_MICROARCH.md's HBM section).

    python tools/pmc_summary.py gpurun_out/pmc_r06 > profiles/r06_pmc_summary.json"""
import csv
import json
import os
import sys


def blocks(path, counter):
    """[(counter value, seconds)] per block: dispatches in time order.  Kinship-structure route: every dispatch of the
    tagged plain product is a block.  Direct route: a block starts at each *_sync_kernel dispatch and takes the 160-column
    tail launch behind it; toy-sized launches before the first one are dropped."""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    out = []
    for r in rows:
        name = r["Kernel_Name"]
        val, sec = float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        if "128, 1>" in name:
            if sec > 0.02:                       # (full-size blocks only)
                out.append([val, sec])
            continue
        big = "sync_kernel" in name or sec > 0.2
        if big:
            out.append([val, sec])
        elif out and "160>" in name:
            out[-1][0] += val
            out[-1][1] += sec
    return out[1:] if len(out) > 1 else out      # the first block is the warm-up step


def variant(d):
    f = blocks(os.path.join(d, "FETCH_SIZE.csv"), "FETCH_SIZE")
    w = blocks(os.path.join(d, "WRITE_SIZE.csv"), "WRITE_SIZE")
    hit = blocks(os.path.join(d, "TCC_HIT_sum_TCC_MISS_sum.csv"), "TCC_HIT_sum")
    miss = blocks(os.path.join(d, "TCC_HIT_sum_TCC_MISS_sum.csv"), "TCC_MISS_sum")
    act = blocks(os.path.join(d, "GRBM_GUI_ACTIVE.csv"), "GRBM_GUI_ACTIVE")
    mean = lambda xs, i: sum(x[i] for x in xs) / len(xs)
    fetch = mean(f, 0) * 1024.0
    write = mean(w, 0) * 1024.0
    return {
        "FETCH_SIZE_bytes_raw": fetch, "FETCH_SIZE_bytes_corrected_x2": 2.0 * fetch, "WRITE_SIZE_bytes": write,
        "traffic_bytes_per_launch": 2.0 * fetch + write,
        "L2_hit_rate": mean(hit, 0) / (mean(hit, 0) + mean(miss, 0)),
        "avg_kernel_ms_under_pmc": 1e3 * mean(f, 1),
        "clock_GHz": mean(act, 0) / 8.0 / mean(act, 1) * 1e-9,
        "launches_sampled": len(f),
    }


def main():
    dirs = sys.argv[1:]
    kin = os.environ.get("CRM_KIN_ROUTE", "1") != "0"
    names = ["default", "second directory"]
    variants = {names[i]: variant(d) for i, d in enumerate(dirs)}
    first = variants[names[0]]
    if kin:
        # operands of one launch at config 3: the per-donor sums S (5050 = k1 + donors k2 rows x 204 800 doubles) read once,
        # MixK(rho*) 5050 x 4992 per selected grid point, A~ 204 800 x 4992 written (the spectrum's 5 000 columns less the
        # last 8, which go through skinny_tn_kernel: scan.hip, spectrum tail)
        alg = 8.0 * (5050 * 204800 + 5050 * 4992 + 204800 * 4992)
        shape = {"config": "cfg3", "cells": 20000, "contexts": 50, "variants_per_launch": 4096,
                 "flops_per_launch": 2.0 * 5050 * 4992 * 50 * 4096}
        what = ("rocprofv3 --pmc over bench.py's own launches (tools/pmc_bench.sh; bench.py --steps 2 --warmup 1, cfg3): the "
                "dominant launch of the kinship-structure route, gemm_tn_glds_kernel<false, 1, 0, false, 128, 1> = "
                "MixK(rho*)' S for the 4096 variants of a block over the 39 whole tile columns of the spectrum (the donor-level kinship "
                "factor folded into the mixing matrix), "
                "one pass per counter group; summary by tools/pmc_summary.py")
        note = "S 8.27 GB + MixK(rho*) 0.2 GB read once, A~ 8.18 GB written"
    else:
        alg = 9.75e9
        shape = {"config": "cfg3", "cells": 20000, "contexts": 50, "variants_per_launch": 4096, "flops_per_launch": 4.096e13}
        what = ("rocprofv3 --pmc over bench.py's own launches (tools/pmc_bench.sh with CRM_KIN_ROUTE=0; bench.py --steps 2 "
                "--warmup 1, cfg3, 4096 variants per launch = one gemm_tn_glds_sync_kernel<true,...> launch over 38 x 128 "
                "columns plus one gemm_tn_glds_kernel<true,...,160> launch over the last 136, summed), one pass per counter "
                "group; summary by tools/pmc_summary.py")
        note = ("Q0 set read once 0.82 GB x (share of the rho* groups) + genotype block 0.66 GB + A~ written 8.4 GB "
                "(SURVEY 8d per-unit figure x 4096)")
    # the kernel form of the profiled build: from the plain bench.py run that tools/pmc_bench.sh makes beside the passes
    form = {"contraction_sync": True, "tail_launch": True, "library": "0.5.0", "kinship_route": kin, "tile_band": 8}
    try:
        line = open(os.path.join(dirs[0], "bench_plain.json")).read().strip().splitlines()[-1]
        form = json.loads(line)["roofline"]["kernel_form"]
    except (OSError, KeyError, ValueError, IndexError):
        pass
    out = {
        "collected_on": what,
        "launch_shape": shape,
        # bench.py quotes this profile only for the same kernel form (bench.py: roofline["kernel_form"])
        "kernel_form": form,
        "algorithmic_bytes_per_launch": alg,
        "algorithmic_bytes_note": note,
        "traffic_bytes_per_launch": first["traffic_bytes_per_launch"],
        "traffic_over_algorithmic": first["traffic_bytes_per_launch"] / alg,
        "gfx950_corrections": "FETCH_SIZE doubled (16 B/lane streams are tallied at half their size, MI355X_MICROARCH.md HBM section); "
                              "WRITE_SIZE as read; L2-fabric side, Infinity-Cache hits included",
        "variants": variants,
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
