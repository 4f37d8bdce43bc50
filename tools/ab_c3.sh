# A/B of the big-table step inside ONE gpurun call: the two-table item form (TFR_DUALQ) and the look-ahead sort (TFR_NO_LOOKAHEAD)
set -e
cd $GRAFT_REPO_ROOT
for cfg in "1 0" "0 0" "1 1" "1 0" "0 0"; do
  set -- $cfg
  echo "TFR_DUALQ=$1 TFR_NO_LOOKAHEAD=$2"
  TFR_DUALQ=$1 TFR_NO_LOOKAHEAD=$2 python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline --no-north-star 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['roofline']['kernels']
print('  ms_per_step %.4f  value %.3e  prestaged %.3e' % (d['ms_per_step'], d['value'], d['feeds']['prestaged_ids']['ratings_per_s']), {s: round(v['us_per_step'],1) for s,v in k.items()})
"
done
