mkdir -p gpurun_out/r03ap; ulimit -c 0
for i in 1 2; do python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/r03ap/suite_run$i.log 2>&1; echo "suite$i rc=$?"; tail -1 gpurun_out/r03ap/suite_run$i.log; done
CRM_POISON=1 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/r03ap/suite_poison.log 2>&1; echo "poison rc=$?"; tail -1 gpurun_out/r03ap/suite_poison.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03ap/smoke.log 2>&1; echo "smoke rc=$?"
python3 bench.py > gpurun_out/r03ap/bench_default.json 2> gpurun_out/r03ap/bench.err; echo "bench rc=$?"
