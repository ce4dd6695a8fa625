#!/bin/bash
mkdir -p gpurun_out/r06
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r06/gpu_suite.log 2>&1; tail -12 gpurun_out/r06/gpu_suite.log
timeout 300 python tools/diag/step_forms.py cfg2 C 8 > gpurun_out/r06/step_forms_cfg2.log 2>&1; cat gpurun_out/r06/step_forms_cfg2.log | grep "^{"
timeout 300 python tools/diag/step_forms.py cfg3 B 8 > gpurun_out/r06/step_forms_cfg3B.log 2>&1; cat gpurun_out/r06/step_forms_cfg3B.log | grep "^{"
timeout 600 python tools/bench_permutations.py cfg2 C 16 4096 > gpurun_out/r06/perm_cfg2.log 2>&1; grep "^{" gpurun_out/r06/perm_cfg2.log
timeout 600 python tools/bench_permutations.py cfg3 B 16 4096 > gpurun_out/r06/perm_cfg3B.log 2>&1; grep "^{" gpurun_out/r06/perm_cfg3B.log
for seed in 9001 31337; do
  timeout 1500 python tools/diag/flat_flag_study.py 1000 $seed > gpurun_out/r06/flat_flag_study_$seed.log 2>&1; echo "study $seed rc=$?"; head -16 gpurun_out/r06/flat_flag_study_$seed.log | tail -12
done
CRM_FUZZ_MANY_CONTEXTS=1 timeout 900 python tools/diag/flat_flag_study.py 120 99 > gpurun_out/r06/flat_flag_study_many.log 2>&1; echo "study many rc=$?"; head -16 gpurun_out/r06/flat_flag_study_many.log | tail -12
