#!/bin/bash
# tools/diag/davies_phases.py under the kernel trace: the per-dispatch durations of eig_davies_kernel, in order
# (three of reduction + bisection + Davies, then three of Davies alone).
#   gpurun -- 'bash tools/diag/davies_phases.sh [k0] [count]'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/davies_phases; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o t -- python3 tools/diag/davies_phases.py "$@" > $out/run.log 2>&1; echo "rc=$?"
tail -3 $out/run.log
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "eig_davies" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("eig_davies_kernel dispatches (us):", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows])
PY
rm -rf $out/prof
