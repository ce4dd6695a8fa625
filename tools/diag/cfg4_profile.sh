#!/bin/bash
# BASELINE config 4 on one GPU (64 phenotypes x the fixed 50 000-variant panel) under rocprofv3 --kernel-trace --stats.
#   gpurun -- 'bash tools/diag/cfg4_profile.sh r04xx'   -> gpurun_out/r04xx/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-cfg4_profile}; shift
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o t -- python3 bench.py --steps 1 --warmup 0 --cpu-variants 0 --collapsed 0 "$@" \
    > $out/bench_cfg4_under_rocprof.json 2> $out/rocprof.err; echo "rocprof rc=$?"
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv && rm -rf $out/prof
tail -1 $out/bench_cfg4_under_rocprof.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config4']['value'], d['config4']['seconds'])"
head -14 $out/kernel_stats.csv | cut -c1-200
