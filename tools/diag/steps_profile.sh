#!/bin/bash
# The default command's timed steps under rocprofv3 --kernel-trace --stats (1 warm-up + 12 timed steps of 4096 variants,
# nothing else scanned): the kernel-stats CSV whose averages must agree with bench.py's own HIP events.
#   gpurun -- 'bash tools/diag/steps_profile.sh r04xx [extra bench.py flags]'   -> gpurun_out/r04xx/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-steps_profile}; shift
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o t -- python3 bench.py --cpu-variants 0 --genes 0 --full-panel 0 --collapsed 0 --direct-steps 0 "$@" \
    > $out/bench_steps_under_rocprof.json 2> $out/rocprof.err; echo "rocprof rc=$?"
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv && rm -rf $out/prof
tail -1 $out/bench_steps_under_rocprof.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
head -16 $out/kernel_stats.csv | cut -c1-220
