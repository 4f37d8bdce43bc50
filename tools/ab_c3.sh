# A/B of the big-table step inside ONE gpurun call: the one-launch-per-pass radix sort (k_psort_pass: 0 off, 1 look-ahead sorts only, 2 all) and the two-table item form
set -e
cd $GRAFT_REPO_ROOT
for cfg in "2 1" "1 1" "0 1" "2 0" "0 0" "2 1" "0 1"; do
  set -- $cfg
  echo "TFR_PSORT=$1 TFR_DUALQ=$2"
  TFR_PSORT=$1 TFR_DUALQ=$2 python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline --no-north-star 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['roofline']['kernels']
print('  ms_per_step %.4f  value %.3e ' % (d['ms_per_step'], d['value']), {s: round(v['us_per_step'],1) for s,v in k.items()})
"
done
