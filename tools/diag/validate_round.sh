#!/bin/bash
# End-of-round validation on a GPU box:   gpurun -- 'bash tools/diag/validate_round.sh r03xx [part]'
#   part 1 (default): the GPU suite twice in fresh processes, once under the poison fill, smoke(), the default bench line
#   part 2: every test_gpu_*.py in a process of its own, the two-rank dry run of bench.py on the one GPU, the PMC passes
out=gpurun_out/${1:-validate}
mkdir -p $out; ulimit -c 0
if [ "${2:-1}" = "1" ]; then
  for i in 1 2; do python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $out/suite_run$i.log 2>&1; echo "suite$i rc=$?"; tail -1 $out/suite_run$i.log; done
  CRM_POISON=1 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $out/suite_poison.log 2>&1; echo "poison rc=$?"; tail -1 $out/suite_poison.log
  python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; echo "smoke rc=$?"
  python3 bench.py > $out/bench_default.json 2> $out/bench.err; echo "bench rc=$?"
else
  : > $out/per_file.txt
  for f in tests/test_gpu_*.py tests/test_c_example.py; do
    python3 -m pytest $f -x -q -m gpu -p no:cacheprovider > $out/per_file_tmp.log 2>&1; rc=$?
    echo "$f rc=$rc $(tail -1 $out/per_file_tmp.log)" >> $out/per_file.txt
  done
  cat $out/per_file.txt
  CRM_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 4 --cpu-variants 0 --genes 0 --collapsed 0 > $out/bench_two_ranks_dry_run.json 2> $out/bench_two_ranks.err; echo "two-rank dry run rc=$?"
  bash tools/pmc_bench.sh > $out/pmc_log.txt 2>&1; echo "pmc rc=$?"
  bash tools/pmc_sq.sh > $out/pmc_sq.txt 2>&1; echo "sq rc=$?"
fi
