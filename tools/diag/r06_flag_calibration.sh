#!/bin/bash
# round 6, calibration of the CRM_MODEL_FLAT_OPTIMUM rule: bit-identity of the refactored null fits against round 5's build,
# then the decision-distance records of the fuzz streams (rows -> gpurun_out/flat_flag_study_<seed>.npy)
mkdir -p gpurun_out/r06
timeout 900 python tools/diag/compare_builds.py 150 2026 > gpurun_out/r06/compare_builds_2026.log 2>&1; echo "compare rc=$?"
tail -30 gpurun_out/r06/compare_builds_2026.log
for seed in 2026 4242; do
  timeout 1500 python tools/diag/flat_flag_study.py 1000 $seed > gpurun_out/r06/flat_flag_study_$seed.log 2>&1; echo "study $seed rc=$?"
  tail -5 gpurun_out/r06/flat_flag_study_$seed.log
done
CRM_FUZZ_MANY_CONTEXTS=1 timeout 900 python tools/diag/flat_flag_study.py 120 99 > gpurun_out/r06/flat_flag_study_many.log 2>&1; echo "study many rc=$?"
timeout 900 python -m pytest tests/test_gpu_effects.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r06/effects_fuzz.log 2>&1; tail -5 gpurun_out/r06/effects_fuzz.log
