# world-1 RCCL rehearsal of the row-sharded C3 step, inside ONE gpurun call: TFR_FAST=0/1 (the fused kernels' load rounds)
set -e
cd $GRAFT_REPO_ROOT
for cfg in 1 0 1 0; do
  echo "TFR_FAST=$cfg"
  TFR_FAST=$cfg TFR_FORCE_DP=1 python bench.py --workload c3 --steps 20 --warmup 5 --no-single-gpu-reference 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('  ms_per_step %.4f  value %.3e  host enqueue us/step %.0f' % (d['ms_per_step'], d['value'], d['roofline']['host_enqueue_us_per_step']), {k: round(v,1) for k,v in d['roofline']['phases_us'].items()})
"
done
