#!/bin/bash
# The round's bench records beside the default line: cfg2, cfg3 mode B, cfg5, and the default command's timed steps under
# rocprofv3 --kernel-trace --stats (the kernel-stats CSV whose averages must agree with bench.py's own HIP events).
#   gpurun -- 'bash tools/diag/bench_set.sh r03xx'      -> gpurun_out/r03xx/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-bench_set}
mkdir -p $out
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "default rc=$?"
python3 bench.py --config cfg2 --cpu-variants 0 --genes 0 > $out/bench_cfg2.json 2> $out/bench_cfg2.err; echo "cfg2 rc=$?"
python3 bench.py --mode B --cpu-variants 0 --genes 0 > $out/bench_cfg3_modeB.json 2> $out/bench_modeB.err; echo "modeB rc=$?"
python3 bench.py --config cfg5 --steps 3 --cpu-variants 0 --genes 0 --full-panel 0 > $out/bench_cfg5.json 2> $out/bench_cfg5.err; echo "cfg5 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o t -- python3 bench.py --cpu-variants 0 --genes 0 --full-panel 0 --collapsed 0 \
    > $out/bench_steps_under_rocprof.json 2> $out/rocprof.err; echo "rocprof rc=$?"
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv && rm -rf $out/prof
for f in $out/bench_*.json; do tail -1 $f | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"; done
head -8 $out/kernel_stats.csv | cut -c1-200
