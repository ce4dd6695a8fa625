#!/bin/bash
mkdir -p gpurun_out/r06
for tag in package old_sum_old_logs old_sum old_logs; do
  if [ $tag = package ]; then unset CRM_THIS_LIB; else export CRM_THIS_LIB=$PWD/tools/_r05/libcrm_hip_$tag.so; fi
  timeout 600 python tools/diag/compare_builds.py 150 2026 > gpurun_out/r06/compare_builds_$tag.log 2>&1; echo "compare $tag rc=$?"
  python - <<PY
import json
d=json.load(open("gpurun_out/compare_builds_seed2026_$tag.json"))
print("$tag", {k:(v["different"], v["worst_rel_difference"]) for k,v in d.items() if isinstance(v,dict)})
PY
done
unset CRM_THIS_LIB
timeout 900 python -m pytest tests/test_gpu_permutations.py tests/test_gpu_effects.py tests/test_gpu_eigh2.py tests/test_gpu_edges.py -x -q -m gpu > gpurun_out/r06/tests_a.log 2>&1; tail -4 gpurun_out/r06/tests_a.log
for seed in 2026 4242; do
  timeout 1500 python tools/diag/flat_flag_study.py 1000 $seed > gpurun_out/r06/flat_flag_study_$seed.log 2>&1; echo "study $seed rc=$?"
done
