cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/bt2r
timeout 900 python3 -m pytest tests/test_gpu_eigh2.py -x -q > gpurun_out/bt2r/eigh2_tests.log 2>&1; tail -5 gpurun_out/bt2r/eigh2_tests.log
timeout 300 python3 tools/diag/eigh2_timing.py 5064 10 > gpurun_out/bt2r/timing_5064.log 2>&1; grep -v "^\[crm back" gpurun_out/bt2r/timing_5064.log | tail -14
timeout 600 python3 tools/diag/eigh2_timing.py 10064 10 > gpurun_out/bt2r/timing_10064.log 2>&1; tail -9 gpurun_out/bt2r/timing_10064.log
bash tools/diag/steps_trace.sh bt2r_trace > gpurun_out/bt2r/steps_trace.log 2>&1; head -45 gpurun_out/bt2r_trace/step_breakdown.txt
