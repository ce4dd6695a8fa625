#!/bin/bash
mkdir -p gpurun_out/r06
for tag in package old_sum; do
  if [ $tag = package ]; then unset CRM_THIS_LIB; else export CRM_THIS_LIB=$PWD/tools/_r05/libcrm_hip_$tag.so; fi
  timeout 600 python tools/diag/compare_builds.py 150 2026 > gpurun_out/r06/compare_builds_$tag.log 2>&1; echo "compare $tag rc=$?"
  python - <<PY
import json
d=json.load(open("gpurun_out/compare_builds_seed2026_$tag.json"))
print("$tag", {k:(v["different"], v["worst_rel_difference"]) for k,v in d.items() if isinstance(v,dict)})
PY
done
unset CRM_THIS_LIB
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06/gpu_suite.log 2>&1; tail -6 gpurun_out/r06/gpu_suite.log
bash tools/diag/steps_trace.sh r06/cfg2_trace --config cfg2 > gpurun_out/r06/cfg2_trace.log 2>&1; head -8 gpurun_out/r06/cfg2_trace/step_breakdown.txt
bash tools/diag/steps_trace.sh r06/cfg3_trace > gpurun_out/r06/cfg3_trace.log 2>&1; head -6 gpurun_out/r06/cfg3_trace/step_breakdown.txt
for seed in 2026 4242; do
  timeout 1500 python tools/diag/flat_flag_study.py 1000 $seed > gpurun_out/r06/flat_flag_study_$seed.log 2>&1; echo "study $seed rc=$?"; head -22 gpurun_out/r06/flat_flag_study_$seed.log | tail -18
done
