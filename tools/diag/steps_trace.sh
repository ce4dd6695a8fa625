#!/bin/bash
# Kernel trace (every dispatch) of a few timed steps of the default bench: per-launch durations grouped by kernel and grid.
#   gpurun -- 'bash tools/diag/steps_trace.sh r04xx [extra bench.py flags]'   -> gpurun_out/r04xx/step_breakdown.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-steps_trace}; shift
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o t -- python3 bench.py --steps 4 --warmup 1 --cpu-variants 0 --genes 0 --full-panel 0 --collapsed 0 --direct-steps 0 "$@" \
    > $out/bench.json 2> $out/rocprof.err; echo "rocprof rc=$?"
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$out" > $out/step_breakdown.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed steps: everything after the last constructor kernel (trd_* / dc_* / bt_*)
last_ctor = max(i for i, r in enumerate(rows) if any(t in r["Kernel_Name"] for t in ("trd_", "dc_", "bt_larft")))
steps = rows[last_ctor + 1:]
tagged = [i for i, r in enumerate(steps) if "128, 1>" in r["Kernel_Name"]]
nsteps = len(tagged)
# Whole steps only.  Every step holds exactly one tagged launch and issues the same number of launches (the bench scans the
# same panel every step), so the steps are the trace's last nsteps * L launches cut every L, L = the distance between two
# tagged launches; the trace ends with the last step's last kernel.  The warm-up step is dropped.
gaps = [b_ - a_ for a_, b_ in zip(tagged, tagged[1:])]
L = collections.Counter(gaps).most_common(1)[0][0]
if any(g != L for g in gaps):
    print("NOTE: launches per step vary %s; cutting every %d" % (gaps, L))
ends = [len(steps) - 1 - (nsteps - 1 - k) * L for k in range(nsteps)]
assert all(a_ <= e_ for a_, e_ in zip(tagged, ends)), (tagged, ends)
use = steps[ends[0] + 1: ends[-1] + 1]
with open(sys.argv[2] + "/steps_compact.csv", "w") as fh:   # kept beside the table for checking the cut
    for j, r in enumerate(steps):
        fh.write("%d,%s,%s,%s,%s,%s\n" % (j, r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:90].replace(",", ";"), r.get("Grid_Size_X", r.get("Grid_Size", "")),
                                          r.get("Grid_Size_Z", ""), r["Start_Timestamp"], r["End_Timestamp"]))
n = max(nsteps - 1, 1)
agg = collections.OrderedDict()
t0, t1 = int(steps[ends[0]]["End_Timestamp"]), int(use[-1]["End_Timestamp"])
for r in use:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("crm::", "")
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
