# cache-policy bits of the fused big-table kernels (RedArgs::nt: 1 partner rows, 2 own/m/v loads, 4 w/m/v stores), one gpurun call
set -e
cd $GRAFT_REPO_ROOT
for nt in 23 6 22 7 4 0 23 6; do
  echo "TFR_NT=$nt"
  TFR_NT=$nt python bench.py --workload c3 --steps 60 --warmup 10 --no-cpu-baseline --no-north-star 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['roofline']['kernels']
print('  ms_per_step %.4f ' % d['ms_per_step'], {s: round(v['us_per_step'],1) for s,v in k.items() if s in ('reduce_item','reduce_user','sort','apply')})
"
done
