#!/bin/bash
# Kernel-level records of the side configurations on the current build: cfg2, cfg3 mode B, cfg5 and the direct route, each
# as a bench JSON and as a `rocprofv3 --kernel-trace --stats` CSV of the same command's timed steps; plus the default
# command's step, launch by launch, windowed on whole steps (tools/diag/steps_trace.sh).
#   gpurun -- 'bash tools/diag/r05_side_profiles.sh r05xx'      -> gpurun_out/r05xx/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-side_profiles}
mkdir -p $out
prof() {   # name, bench flags...
  name=$1; shift
  python3 bench.py --cpu-variants 0 --genes 0 "$@" > $out/bench_$name.json 2> $out/bench_$name.err; echo "$name rc=$?"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -o t -- python3 bench.py --cpu-variants 0 --genes 0 --full-panel 0 --collapsed 0 --direct-steps 0 "$@" \
      > $out/bench_${name}_steps_under_rocprof.json 2> $out/rocprof_$name.err; echo "$name rocprof rc=$?"
  f=$(find $out/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_$name.csv
  rm -rf $out/prof_$name
}
prof cfg2 --config cfg2
prof cfg3_modeB --mode B
prof cfg5 --config cfg5 --steps 3 --full-panel 0
# the direct route as its own run (every scan by the Khatri-Rao contraction against Q0(rho*))
CRM_KIN_ROUTE=0 python3 bench.py --cpu-variants 0 --genes 0 --steps 4 --full-panel 0 --collapsed 0 > $out/bench_cfg3_direct_route.json 2> $out/bench_direct.err; echo "direct rc=$?"
CRM_KIN_ROUTE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_direct -o t -- python3 bench.py --cpu-variants 0 --genes 0 --steps 4 --full-panel 0 --collapsed 0 \
    > $out/bench_cfg3_direct_route_under_rocprof.json 2> $out/rocprof_direct.err; echo "direct rocprof rc=$?"
f=$(find $out/prof_direct -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_cfg3_direct_route.csv; rm -rf $out/prof_direct
for f in $out/bench_*.json; do tail -1 $f | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); dr=d.get('direct_route') or {}
print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['whole_path']['frac_of_fp64_mfma_peak'], 'ctor', d['setup_s']['background_constructor'], 'direct', dr.get('value'))"; done
