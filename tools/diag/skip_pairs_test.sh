#!/bin/bash
# Fits without a kinship term: the new test, the multi-phenotype and cis tests, a short bench with the config-4 leg.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/skipp; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_edges.py -x -q -k "kinship_term or symmetric_pair" > $out/edges.log 2>&1; tail -3 $out/edges.log
timeout 1200 python3 -m pytest tests/test_gpu_interaction.py tests/test_gpu_fullsize.py -x -q -k "phenotypes or cis or config4 or many" > $out/multi.log 2>&1; tail -3 $out/multi.log
python3 bench.py --steps 4 --cpu-variants 0 --full-panel 1 --collapsed 0 --direct-steps 0 > $out/bench_cfg4.json 2> $out/bench_cfg4.err; tail -1 $out/bench_cfg4.json | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config4']; print(d['value'], d['ms_per_step'], d['roofline']['frac'], 'cfg4', c['value'], c['seconds'], c['distinct_rho_per_variant'], c.get('oracle_check'))"
