#!/bin/bash
# tools/diag/many_contexts_at_size.py under rocprofv3 --kernel-trace --stats: where the slower kernel forms spend their time.
#   gpurun -- 'bash tools/diag/many_contexts_profile.sh r04xx [contexts] [variants]'   -> gpurun_out/r04xx/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-many_contexts_profile}; shift
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o t -- python3 tools/diag/many_contexts_at_size.py "$@" \
    > $out/result.json 2> $out/rocprof.err; echo "rocprof rc=$?"
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv && rm -rf $out/prof
head -24 $out/kernel_stats.csv | cut -c1-230
