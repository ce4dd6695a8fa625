#!/bin/bash
# L2-fabric traffic of the dominant kernel for the launch shape bench.py itself issues: separate rocprofv3 --pmc
# passes (the TCC block cannot hold FETCH_SIZE and WRITE_SIZE together), restricted to the dominant kernel: the tagged
# plain product of the kinship-structure route by default; CRM_PMC_KERNELS='gemm_tn_glds_(sync_)?kernel<true' with
# CRM_KIN_ROUTE=0 for the direct Khatri-Rao contraction.
#   gpurun -- 'bash tools/pmc_bench.sh'      -> gpurun_out/pmc_r06/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_r06
mkdir -p $out
BENCH="bench.py --steps 2 --warmup 1 --cpu-variants 0 --full-panel 0 --genes 0 --collapsed 0 --direct-steps 0"
python3 $BENCH > $out/bench_plain.json 2> $out/bench_plain.err
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $c | tr ' ' '_')
  timeout 400 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "${CRM_PMC_KERNELS:-gemm_tn_glds_kernel<false, 1, 0, false, 128, 1>}" --output-format csv \
      -d $out/$name -o pmc -- python3 $BENCH > $out/$name.log 2>&1
  echo "$name rc=$?"
  f=$(find $out/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $out/${name}.csv && rm -rf $out/$name
done
ls -la $out
