#!/bin/bash
# Validation of a build: constructor phases at cfg3 / cfg5 / cfg2, the GPU suite, the default bench line.
#   gpurun -- 'bash tools/diag/r05_validate.sh r05xx'      -> gpurun_out/r05xx/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-validate}
mkdir -p $out
for c in cfg3 cfg5 cfg2; do timeout 900 python3 tools/ctor_timing.py $c > $out/ctor_$c.log 2>&1; grep "constructor run\|eigh2\|orthonormality\|eigen-decompositions" $out/ctor_$c.log | tail -9; done
timeout 2400 python3 -m pytest tests -m gpu -x -q > $out/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 $out/gpu_suite.log
timeout 900 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
tail -1 $out/bench_default.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); fp=d['full_panel']
print(d['value'], d['ms_per_step'], d['roofline']['frac'], 'ctor', d['setup_s'], 'e2e', fp['end_to_end_s'], fp['streamed']['end_to_end_s'], fp['streamed']['constructor_s'], 'scan_only', fp['scan_only_rate'], 'cfg4', d['config4']['value'], 'direct', d['direct_route']['value'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
