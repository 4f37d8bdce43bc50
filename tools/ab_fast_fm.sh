# FM training step (c5's `train` object) with TFR_FAST=0/1, one gpurun call
set -e
cd $GRAFT_REPO_ROOT
for cfg in 1 0 1 0; do
  echo "TFR_FAST=$cfg"
  TFR_FAST=$cfg python bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('  fm forward ms %.4f frac %.3f   train ms_per_step %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['frac'], d['train']['ms_per_step'], d['train']['frac']))
"
done
