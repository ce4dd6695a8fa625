#!/bin/bash
# Issue-side counters of the dominant kernel inside bench.py's own launches (one rocprofv3 --pmc pass per group): the tagged
# plain product of the kinship-structure route by default; CRM_PMC_KERNELS='gemm_tn_glds_(sync_)?kernel<true' with
# CRM_KIN_ROUTE=0 for the direct Khatri-Rao contraction.
#   gpurun -- 'bash tools/pmc_sq.sh'      -> gpurun_out/pmc_sq/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_sq
mkdir -p $out
BENCH="bench.py --steps 2 --warmup 1 --cpu-variants 0 --full-panel 0 --genes 0 --collapsed 0 --direct-steps 0"
i=0
for c in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
         "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "${CRM_PMC_KERNELS:-gemm_tn_glds_kernel<false, 1, 0, false, 128, 1>}" --output-format csv \
      -d $out/g$i -o pmc -- python3 $BENCH > $out/g$i.log 2>&1
  echo "group $i rc=$?"
  f=$(find $out/g$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $out/group$i.csv && rm -rf $out/g$i
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in sorted(glob.glob("gpurun_out/pmc_sq/group*.csv")):
    for row in csv.DictReader(open(f)):
        if int(row["End_Timestamp"]) - int(row["Start_Timestamp"]) < 20_000_000: continue     # (full-size blocks only)
        tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k:28s} per launch {tot[k] / max(n[k], 1):.4e}   ({n[k]} launches)")
PY
