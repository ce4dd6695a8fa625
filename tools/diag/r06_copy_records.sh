#!/bin/bash
# gpurun_out/r06final (tools/diag/r06_final_records.sh) + the suite logs -> profiles/r06_* (the bench lines as one JSON document each)
cd "$(dirname "$0")/../.."
src=gpurun_out/r06final
for f in bench_default bench_nccl_world1 bench_timed_steps_only_under_rocprof bench_cfg2 bench_cfg3_modeB bench_cfg5 bench_cfg3_direct_route bench_cfg4_under_rocprof; do
python3 - "$src/$f.json" "profiles/r06_$f.json" <<'PY'
import json,sys
s=open(sys.argv[1]).read()
i=s.rindex('{"metric')
json.dump(json.loads(s[i:]), open(sys.argv[2],'w'), indent=1)
PY
done
cp $src/rocprofv3_kernel_stats_timed_steps_only.csv profiles/r06_rocprofv3_kernel_stats_timed_steps_only.csv
cp $src/step_breakdown_by_launch.txt profiles/r06_step_breakdown_by_launch.txt
cp $src/step_breakdown_cfg2.txt profiles/r06_step_breakdown_cfg2.txt
cp $src/step_breakdown_cfg3_modeB.txt profiles/r06_step_breakdown_cfg3_modeB.txt
for k in cfg2 cfg3_modeB cfg5 cfg3_direct_route; do cp $src/kernel_stats_$k.csv profiles/r06_kernel_stats_$k.csv; done
cp $src/rocprofv3_kernel_stats_cfg4.csv profiles/r06_rocprofv3_kernel_stats_cfg4.csv
cp $src/constructor_phases.log profiles/r06_constructor_phases.log
cp $src/rocprofv3_kernel_stats_constructor.csv profiles/r06_rocprofv3_kernel_stats_constructor.csv
for k in cfg2 cfg3_modeB cfg3; do cp $src/bench_permutations_$k.json profiles/r06_bench_permutations_$k.json; done
cp $src/bench_cis.txt profiles/r06_bench_cis.txt
cp $src/null_fits_bit_identical_to_round5_seed2026.json profiles/r06_null_fits_bit_identical_to_round5_seed2026.json
cp $src/null_fits_bit_identical_to_round5_seed4242.json profiles/r06_null_fits_bit_identical_to_round5_seed4242.json
cp $src/bench_nccl_world1.err profiles/r06_bench_nccl_world1.err
cp gpurun_out/fuzz_verbatim.json profiles/r06_fuzz_verbatim.json
cp gpurun_out/fuzz_polished.json profiles/r06_fuzz_polished.json
for s in "" _direct_route _poison_fill; do [ -f gpurun_out/r06/gpu_suite_final$s.log ] && grep -v "it/s\]" gpurun_out/r06/gpu_suite_final$s.log > profiles/r06_gpu_suite_final$s.log; done
python3 - <<'PY'
import json
d=json.load(open("profiles/r06_bench_default.json")); fp=d["full_panel"]
print("value", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "ctor", d["setup_s"], "e2e", fp["end_to_end_s"], fp["streamed"]["end_to_end_s"],
      "cfg4", d["config4"]["value"], d["config4"]["seconds"], "direct", d["direct_route"]["value"], "rotated", d["rotated_kinship_factor"]["value"], "collapsed", d["donor_collapsed"]["value"], "cpu", d["cpu_baseline"]["value"])
for k in ("cfg2","cfg3_modeB","cfg5"):
    e=json.load(open("profiles/r06_bench_%s.json"%k)); print(k, e["value"], e["ms_per_step"], e["whole_path"]["frac_of_fp64_mfma_peak"], e["setup_s"]["background_constructor"])
e=json.load(open("profiles/r06_bench_nccl_world1.json")); print("nccl", e["value"], e["multi_gpu"]["group"], e["full_panel"]["exchange"])
e=json.load(open("profiles/r06_bench_cfg4_under_rocprof.json")); print("cfg4 under rocprof", e["config4"]["value"], e["config4"]["seconds"])
PY
