# where the look-ahead sort chain starts: beside the user side (TFR_SORT_LATE=1, default) or at the step's start (0); one gpurun call
set -e
cd $GRAFT_REPO_ROOT
for f in 1 0 1 0 1 0; do
  echo "TFR_SORT_LATE=$f"
  TFR_SORT_LATE=$f python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline --no-north-star 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('  ms_per_step %.4f  prestaged %.4f ms' % (d['ms_per_step'], 262144/d['feeds']['prestaged_ids']['ratings_per_s']*1e3))
"
done
