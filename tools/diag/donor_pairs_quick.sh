#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/dpairs; mkdir -p $out
timeout 600 python3 -m pytest tests/test_gpu_edges.py -x -q -k "symmetric_pair" > $out/edges.log 2>&1; tail -2 $out/edges.log
bash tools/diag/steps_trace.sh dpairs_trace > $out/steps_trace.log 2>&1; head -9 gpurun_out/dpairs_trace/step_breakdown.txt; tail -1 gpurun_out/dpairs_trace/bench.json | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
