# the wide-tile radix pass for batch-sized sorts too (TFR_RSORT_WIDE_MIN), one gpurun call
set -e
cd $GRAFT_REPO_ROOT
for lo in 65536 1048576 65536 1048576; do
  echo "TFR_RSORT_WIDE_MIN=$lo"
  TFR_RSORT_WIDE_MIN=$lo python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline --no-north-star 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['roofline']['kernels']
print('  ms_per_step %.4f ' % d['ms_per_step'], {s: round(v['us_per_step'],1) for s,v in k.items() if s in ('reduce_item','reduce_user','sort','apply')})
"
done
