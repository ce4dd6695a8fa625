#!/bin/bash
# Kernel trace (every dispatch) of a few timed steps of the default bench: per-launch durations grouped by kernel and grid.
#   gpurun -- 'bash tools/diag/steps_trace.sh r04xx [extra bench.py flags]'   -> gpurun_out/r04xx/step_breakdown.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-steps_trace}; shift
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o t -- python3 bench.py --steps 4 --warmup 1 --cpu-variants 0 --genes 0 --full-panel 0 --collapsed 0 "$@" \
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
# drop the warm-up step: start after the first tagged launch's step ends (= just before the second step's first kernel)
agg = collections.OrderedDict()
first = tagged[1] if nsteps > 1 else 0
# walk back from the second tagged launch to the start of its step: the gather_block kernel that opens a block
start = first
while start > 0 and "gather_block" not in steps[start]["Kernel_Name"]:
    start -= 1
while start > 0 and "gather_block" in steps[start - 1]["Kernel_Name"]:
    start -= 1
use = steps[start:]
n = max(nsteps - 1, 1)
t0, t1 = int(use[0]["Start_Timestamp"]), int(use[-1]["End_Timestamp"])
for r in use:
    name = r["Kernel_Name"].split("(")[0].replace("void crm::", "").replace("(anonymous namespace)::", "")
    key = (name[:70], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Grid_Size_Z", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += d
print("steps analysed: %d; wall %.2f ms per step; kernel time per step by (kernel, grid x, grid z):" % (n, (t1 - t0) * 1e-6 / n))
tot = 0.0
for key, (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%9.3f ms  x%-5.1f %s  grid %s z %s" % (ms / n, cnt / n, key[0], key[1], key[2]))
    tot += ms
print("%9.3f ms  sum of kernel time per step" % (tot / n))
PY
rm -rf $out/prof
head -40 $out/step_breakdown.txt
