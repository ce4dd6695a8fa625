#!/bin/bash
# N > 1 code path at the world sizes the driver uses, on ONE GPU (all ranks share device 0, gloo): the branches of bench.py
# that only world >= 4 takes (a rank's shard of the fixed panel shorter than the weak-scaling panel; the config-4 leg on a
# 6 250-variant shard with its gather).  Launched the way the driver launches N > 1.  Not a scaling measurement.
#   gpurun -- 'bash tools/diag/r05_dry_runs.sh r05xx'      -> gpurun_out/r05xx/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-dry_runs}
mkdir -p $out
free -g | tee $out/host_memory.txt
mem=$(free -g | awk '/^Mem:/{print $7}')
python3 bench.py --cpu-variants 0 > $out/bench_n1.json 2> $out/bench_n1.err; echo "n1 rc=$?"
for n in 4 8; do
  if [ "$mem" -lt $((n * 8 + 16)) ]; then echo "skip world $n: only $mem GiB of host memory available"; continue; fi
  # internal blocks of 1024 variants and small pair buffers: eight ranks' work buffers beside each other in one GPU's HBM
  CRM_BENCH_SHARE_GPU=1 CRM_PAIR_BUFFER_GB=6 CRM_BENCH_COLLECTIVE_TIMEOUT_S=300 timeout 1500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
      --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 4 --warmup 1 --block 1024 \
      > $out/bench_dry_run_world$n.json 2> $out/bench_dry_run_world$n.err; echo "world $n rc=$?"
  tail -3 $out/bench_dry_run_world$n.err
done
for f in $out/bench_*.json; do tail -1 $f | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); fp=d.get('full_panel') or {}; c4=d.get('config4') or {}; dr=d.get('direct_route') or {}
print('$f', d['n_gpus'], d['value'], d['ms_per_step'], d['roofline']['frac'], 'e2e', fp.get('end_to_end_s'), (fp.get('streamed') or {}).get('end_to_end_s'), fp.get('exchange'), fp.get('gather'),
      'cfg4', c4.get('value'), c4.get('variants_per_rank'), c4.get('gather'), c4.get('gather_s'), 'direct', dr.get('value'), (dr.get('roofline') or {}).get('frac'))"; done
