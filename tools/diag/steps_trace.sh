#!/bin/bash
# Kernel trace (every dispatch) of a few timed steps of the default bench: per-launch durations grouped by kernel and grid.
#   gpurun -- 'bash tools/diag/steps_trace.sh r04xx [extra bench.py flags]'   -> gpurun_out/r04xx/step_breakdown.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-steps_trace}; shift
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o t -- python3 bench.py --steps 4 --warmup 1 --cpu-variants 0 --genes 0 --full-panel 0 --collapsed 0 --direct-steps 0 "$@" \
    > $out/bench.json 2> $out/rocprof.err; echo "rocprof rc=$?"
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $out/step_breakdown.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed steps: everything after the last constructor kernel (trd_* / dc_* / bt_*)
last_ctor = max(i for i, r in enumerate(rows) if any(t in r["Kernel_Name"] for t in ("trd_", "dc_", "bt_larft")))
steps = rows[last_ctor + 1:]
tagged = [i for i, r in enumerate(steps) if "128, 1>" in r["Kernel_Name"]]
nsteps = len(tagged)
# Whole steps only.  Every step holds exactly one tagged launch and ends with the kernel the trace ends with; a step's
# last launch is therefore the last kernel of that name before the next step's tagged launch.  The warm-up step is dropped.
last_name = steps[-1]["Kernel_Name"]
ends = []
for a_, b_ in zip(tagged, tagged[1:] + [len(steps)]):
    ends.append(max(j for j in range(a_, b_) if steps[j]["Kernel_Name"] == last_name))
use = steps[ends[0] + 1: ends[-1] + 1]
n = max(nsteps - 1, 1)
agg = collections.OrderedDict()
t0, t1 = int(steps[ends[0]]["End_Timestamp"]), int(use[-1]["End_Timestamp"])
for r in use:
    name = r["Kernel_Name"].split("(")[0].replace("void crm::", "").replace("(anonymous namespace)::", "")
    key = (name[:70], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Grid_Size_Z", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += d
print("whole steps analysed: %d (warm-up dropped); %.2f ms per step from the end of one step's last kernel to the end of the next one's; "
      "kernel time per step by (kernel, grid x, grid z):" % (n, (t1 - t0) * 1e-6 / n))
tot = 0.0
for key, (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%9.3f ms  x%-5.1f %s  grid %s z %s" % (ms / n, cnt / n, key[0], key[1], key[2]))
    tot += ms
print("%9.3f ms  sum of kernel time per step; %.3f ms per step between kernels (launch gaps, host)" % (tot / n, (t1 - t0) * 1e-6 / n - tot / n))
PY
rm -rf $out/prof
head -40 $out/step_breakdown.txt
