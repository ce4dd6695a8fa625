#!/bin/bash
# Round 6's records on one build: smoke, the default bench line, the same bench under the nccl group at world 1 with the
# forced exchange, its timed steps under rocprofv3 (kernel stats + launch-by-launch table over whole steps), the side
# configurations with their kernel stats, config 4 and the constructor under the kernel trace, constructor phases, the
# permutation driver, the cis-window bench, the bit-identity of the null fits against round 5's build.
#   gpurun -- 'bash tools/diag/r06_final_records.sh'      -> gpurun_out/r06final/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06final
mkdir -p $out
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "default rc=$?"
CRM_BENCH_FORCE_EXCHANGE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 \
    > $out/bench_nccl_world1.json 2> $out/bench_nccl_world1.err; echo "nccl world 1 rc=$?"
bash tools/diag/steps_profile.sh r06final_steps > $out/steps_profile.log 2>&1
cp gpurun_out/r06final_steps/bench_steps_under_rocprof.json $out/bench_timed_steps_only_under_rocprof.json
cp gpurun_out/r06final_steps/kernel_stats.csv $out/rocprofv3_kernel_stats_timed_steps_only.csv
bash tools/diag/steps_trace.sh r06final_trace > $out/steps_trace.log 2>&1
cp gpurun_out/r06final_trace/step_breakdown.txt $out/step_breakdown_by_launch.txt
bash tools/diag/steps_trace.sh r06final_trace_cfg2 --config cfg2 > $out/steps_trace_cfg2.log 2>&1
cp gpurun_out/r06final_trace_cfg2/step_breakdown.txt $out/step_breakdown_cfg2.txt
bash tools/diag/steps_trace.sh r06final_trace_modeB --mode B > $out/steps_trace_modeB.log 2>&1
cp gpurun_out/r06final_trace_modeB/step_breakdown.txt $out/step_breakdown_cfg3_modeB.txt
bash tools/diag/r05_side_profiles.sh r06final_side > $out/side_profiles.log 2>&1
for f in bench_cfg2 bench_cfg3_modeB bench_cfg5 bench_cfg3_direct_route; do tail -1 gpurun_out/r06final_side/$f.json > $out/$f.json; done
cp gpurun_out/r06final_side/kernel_stats_*.csv $out/
bash tools/diag/cfg4_profile.sh r06final_cfg4 > $out/cfg4_profile.log 2>&1
cp gpurun_out/r06final_cfg4/kernel_stats.csv $out/rocprofv3_kernel_stats_cfg4.csv
tail -1 gpurun_out/r06final_cfg4/bench_cfg4_under_rocprof.json > $out/bench_cfg4_under_rocprof.json
for c in cfg3 cfg5 cfg2; do python3 tools/ctor_timing.py $c 2>&1 | grep -v "defect over\|pass 0"; done > $out/constructor_phases.log
bash tools/diag/ctor_profile.sh r06final_ctor > $out/ctor_profile.log 2>&1
cp gpurun_out/r06final_ctor/kernel_stats.csv $out/rocprofv3_kernel_stats_constructor.csv
python3 tools/bench_permutations.py cfg2 C 16 4096 2>/dev/null | grep "^{" > $out/bench_permutations_cfg2.json
python3 tools/bench_permutations.py cfg3 B 16 4096 2>/dev/null | grep "^{" > $out/bench_permutations_cfg3_modeB.json
python3 tools/bench_permutations.py cfg3 C 4 4096 2>/dev/null | grep "^{" > $out/bench_permutations_cfg3.json
for gen in 0 1; do python3 tools/bench_cis.py cfg3 64 1024 256 $gen 2>&1 | sed "s/.*it\/s\]//" | grep "pass\|gene by\|bound\|resident"; done > $out/bench_cis.txt
python3 tools/diag/compare_builds.py 150 2026 > $out/compare_builds.log 2>&1; cp gpurun_out/compare_builds_seed2026_package.json $out/null_fits_bit_identical_to_round5_seed2026.json
python3 tools/diag/compare_builds.py 150 4242 > $out/compare_builds_4242.log 2>&1; cp gpurun_out/compare_builds_seed4242_package.json $out/null_fits_bit_identical_to_round5_seed4242.json
tail -1 $out/bench_default.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); fp=d['full_panel']
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), 'ctor', d['setup_s'], 'e2e', fp['end_to_end_s'], fp['streamed']['end_to_end_s'], fp['streamed']['constructor_s'], 'scan_only', fp['scan_only_rate'], 'cfg4', d['config4']['value'], 'direct', d['direct_route']['value'], d['direct_route']['roofline']['frac'], 'rotated', d['rotated_kinship_factor'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
tail -1 $out/bench_nccl_world1.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('nccl', d['value'], d['multi_gpu']['group'], d['full_panel']['exchange'])"
cat $out/bench_permutations_*.json | cut -c1-330; cat $out/bench_cis.txt; tail -4 $out/constructor_phases.log; head -3 $out/step_breakdown_by_launch.txt | cut -c1-200
