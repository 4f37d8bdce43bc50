# experiments on the big-table step's fused kernels: TFR_SEG_PAD (unused dynamic LDS caps the blocks per CU), TFR_SEG_NTHR (threads per block, timing only)
set -e
cd $GRAFT_REPO_ROOT
for cfg in "0 1024" "0 256" "0 512" "0 1024" "0 256" "0 512"; do
  set -- $cfg
  echo "TFR_SEG_PAD=$1 TFR_SEG_NTHR=$2"
  TFR_SEG_PAD=$1 TFR_SEG_NTHR=$2 python bench.py --workload c3 --steps 60 --warmup 10 --no-cpu-baseline --no-north-star 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['roofline']['kernels']
print('  ms_per_step %.4f  value %.3e' % (d['ms_per_step'], d['value']), {s: round(v['us_per_step'],1) for s,v in k.items()})
"
done
