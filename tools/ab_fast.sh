# A/B of the big-table step's fused kernels: TFR_FAST=0 (loads in program order, eight dependent rounds) against 1 (three rounds)
set -e
cd $GRAFT_REPO_ROOT
for cfg in 1 0 1 0 1 0; do
  echo "TFR_FAST=$cfg"
  TFR_FAST=$cfg python bench.py --workload ${WL:-c3} --steps 60 --warmup 10 --no-cpu-baseline --no-north-star ${EXTRA:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['roofline']['kernels']
print('  ms_per_step %.4f  value %.3e' % (d['ms_per_step'], d['value']), {s: round(v['us_per_step'],1) for s,v in k.items()})
"
done
