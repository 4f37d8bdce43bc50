# headline step with / without the store records left beside the drawn ids (TFR_RECS=0/1), one gpurun call
set -e
cd $GRAFT_REPO_ROOT
for v in 1 0 1 0 1 0; do
  echo "TFR_RECS=$v"
  TFR_RECS=$v python bench.py --steps 900 --warmup 50 --no-configs --no-north-star --no-convergence --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('  900 steps: %.3f us/step' % (d['ms_per_step']*1e3), {k: round(v['us_per_step'],2) for k,v in d['roofline']['kernels'].items()}, d['val_rmse_after_timed_steps'])"
  TFR_RECS=$v python bench.py --steps 20 --warmup 5 --no-configs --no-north-star --no-convergence --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   20 steps: %.3f us/step' % (d['ms_per_step']*1e3), d['val_rmse_after_timed_steps'])"
done
