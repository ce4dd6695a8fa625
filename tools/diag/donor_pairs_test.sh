#!/bin/bash
# The donor-pairs form: its test, the parity tests that run through it, a short bench and the step table.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/dpairs; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_edges.py -x -q -k "symmetric_pair or many_contexts or gram_kernel" > $out/edges.log 2>&1; tail -4 $out/edges.log
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_interaction.py -x -q > $out/parity.log 2>&1; tail -4 $out/parity.log
python3 bench.py --steps 6 --cpu-variants 0 --genes 0 --full-panel 0 --collapsed 0 --direct-steps 0 > $out/bench_short.json 2> $out/bench_short.err; tail -1 $out/bench_short.json | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('whole_path',{}).get('achieved_tflops'))"
bash tools/diag/steps_trace.sh dpairs_trace > $out/steps_trace.log 2>&1; head -16 gpurun_out/dpairs_trace/step_breakdown.txt
