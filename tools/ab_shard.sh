# world-1 RCCL rehearsal of the row-sharded C3 step, inside ONE gpurun call: front-end pipelining on / off
set -e
cd $GRAFT_REPO_ROOT
for cfg in 0 1 0 1; do
  echo "TFR_SHARD_NO_PIPELINE=$cfg"
  if [ "$cfg" = "1" ]; then export TFR_SHARD_NO_PIPELINE=1; else unset TFR_SHARD_NO_PIPELINE; fi
  TFR_FORCE_DP=1 python bench.py --workload c3 --steps 20 --warmup 5 --no-single-gpu-reference 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('  ms_per_step %.4f  value %.3e  host enqueue us/step %.0f' % (d['ms_per_step'], d['value'], d['roofline']['host_enqueue_us_per_step']), {k: round(v,1) for k,v in d['roofline']['phases_us'].items()})
"
done
