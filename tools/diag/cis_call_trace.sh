#!/bin/bash
# Kernel trace of a few per-gene scans of cis windows (tools/diag/cis_call_trace.py): what a call costs beside its kernels.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/cis2
for gen in 0 1; do
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cis2/p$gen -o t -- python3 tools/diag/cis_call_trace.py $gen 2>&1 | grep -v "it/s\|rocprofv3\|Opened" | tail -3
  f=$(find gpurun_out/cis2/p$gen -name "*kernel_trace.csv" | head -1)
  python3 - $f <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
# the eight per-gene scans + the one-pass call: everything after the last warm-up table kernel (donor_sums)
last_tab = max(i for i, r in enumerate(rows) if "donor_sums" in r["Kernel_Name"] or "context_features" in r["Kernel_Name"])
seg = rows[last_tab + 1:]
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e6
print("after the tables: %d kernels over %.1f ms, busy %.1f ms" % (len(seg), span, busy))
acc = collections.Counter(); cnt = collections.Counter()
for r in seg:
    n = r["Kernel_Name"].split("(")[0][-60:]
    acc[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[n] += 1
for n, v in acc.most_common(16):
    print("  %7.2f ms %4d  %s" % (v / 1e6, cnt[n], n))
PY
  rm -rf gpurun_out/cis2/p$gen
done
