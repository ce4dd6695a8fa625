#!/bin/bash
# The background constructor alone (BASELINE config 3, two runs) under rocprofv3 --kernel-trace (per-launch trace kept).
#   gpurun -- 'bash tools/diag/ctor_profile.sh r04xx'   -> gpurun_out/r04xx/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-ctor_profile}; shift
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o t -- python3 tools/ctor_timing.py cfg3 > $out/ctor.log 2> $out/rocprof.err; echo "rocprof rc=$?"
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv
t=$(find $out/prof -name "*kernel_trace.csv" | head -1)
# launches of the second run's back-transformation and divide & conquer, by grid size: name, count, total ms
[ -n "$t" ] && python3 - "$t" > $out/gemm_by_grid.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-60:]
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Z", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    a = acc.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d
for k, (n, ms) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{ms:9.2f} ms  {n:6d} x  {k}")
PY
rm -rf $out/prof
head -30 $out/gemm_by_grid.txt
