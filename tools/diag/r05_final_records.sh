#!/bin/bash
# The round's records on one build: the default bench line, its timed steps under rocprofv3 (kernel stats + launch-by-launch
# table over whole steps), the side configurations with their kernel stats, config 4 and the constructor under the kernel
# trace, constructor phases, the N > 1 dry runs on one shared GPU, the cis-window bench, smoke.
#   gpurun -- 'bash tools/diag/r05_final_records.sh'      -> gpurun_out/r05final/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05final
mkdir -p $out
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "default rc=$?"
bash tools/diag/steps_profile.sh r05final_steps > $out/steps_profile.log 2>&1
cp gpurun_out/r05final_steps/bench_steps_under_rocprof.json $out/bench_timed_steps_only_under_rocprof.json
cp gpurun_out/r05final_steps/kernel_stats.csv $out/rocprofv3_kernel_stats_timed_steps_only.csv
bash tools/diag/steps_trace.sh r05final_trace > $out/steps_trace.log 2>&1
cp gpurun_out/r05final_trace/step_breakdown.txt $out/step_breakdown_by_launch.txt
bash tools/diag/r05_side_profiles.sh r05final_side > $out/side_profiles.log 2>&1
for f in bench_cfg2 bench_cfg3_modeB bench_cfg5 bench_cfg3_direct_route; do tail -1 gpurun_out/r05final_side/$f.json > $out/$f.json; done
cp gpurun_out/r05final_side/kernel_stats_*.csv $out/
bash tools/diag/cfg4_profile.sh r05final_cfg4 > $out/cfg4_profile.log 2>&1
cp gpurun_out/r05final_cfg4/kernel_stats.csv $out/rocprofv3_kernel_stats_cfg4.csv
tail -1 gpurun_out/r05final_cfg4/bench_cfg4_under_rocprof.json > $out/bench_cfg4_under_rocprof.json
for c in cfg3 cfg5 cfg2; do python3 tools/ctor_timing.py $c 2>&1 | grep -v "defect over\|pass 0"; done > $out/constructor_phases.log
bash tools/diag/ctor_profile.sh r05final_ctor > $out/ctor_profile.log 2>&1
cp gpurun_out/r05final_ctor/kernel_stats.csv $out/rocprofv3_kernel_stats_constructor.csv
for gen in 0 1; do python3 tools/bench_cis.py cfg3 64 1024 256 $gen 2>&1 | sed "s/.*it\/s\]//" | grep "pass\|gene by\|bound\|resident"; done > $out/bench_cis.txt
python3 tools/diag/flat_flag_study.py 150 7 > $out/flat_flag_study.json 2> $out/flat_flag_study.err
bash tools/diag/r05_dry_runs.sh r05final_dry > $out/dry_runs.log 2>&1
for n in 4 8; do tail -1 gpurun_out/r05final_dry/bench_dry_run_world$n.json > $out/bench_dry_run_world${n}_one_gpu_shared.json; done
tail -1 $out/bench_default.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); fp=d['full_panel']
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), 'ctor', d['setup_s'], 'e2e', fp['end_to_end_s'], fp['streamed']['end_to_end_s'], fp['streamed']['constructor_s'], 'scan_only', fp['scan_only_rate'], 'cfg4', d['config4']['value'], 'direct', d['direct_route']['value'], d['direct_route']['roofline']['frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
cat $out/bench_cis.txt; tail -4 $out/constructor_phases.log; head -3 $out/step_breakdown_by_launch.txt | cut -c1-200
