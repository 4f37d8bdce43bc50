# A/B of two builds of the library (tools/ab/libtfrecomm_hip_{base,mv}.so copied over the in-tree one) on several workloads
set -e
cd $GRAFT_REPO_ROOT
for v in base mv base mv; do
  cp tools/ab/libtfrecomm_hip_$v.so tf-recomm_amd/csrc/libtfrecomm_hip.so
  echo "build=$v"
  python bench.py --steps 900 --warmup 50 --no-configs --no-north-star --no-convergence --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('  c2 900 steps: %.3f us/step' % (d['ms_per_step']*1e3), {k: round(v['us_per_step'],2) for k,v in d['roofline']['kernels'].items()})"
  python bench.py --steps 20 --warmup 5 --no-configs --no-north-star --no-convergence --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('  c2  20 steps: %.3f us/step' % (d['ms_per_step']*1e3))"
  python bench.py --workload c3 --steps 60 --warmup 10 --no-cpu-baseline --no-north-star 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('  c3 ms_per_step %.4f' % d['ms_per_step'], {s: round(v['us_per_step'],1) for s,v in k.items()})"
  python bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('  c5 fm forward ms %.4f   train ms %.4f' % (d['ms_per_step'], d['train']['ms_per_step']))"
  python bench.py --only-north-star --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); n=d.get('north_star_forward') or d; print('  forward frac', round(n.get('frac', n.get('roofline',{}).get('frac',0)),4))"
done
cp tools/ab/libtfrecomm_hip_mv.so tf-recomm_amd/csrc/libtfrecomm_hip.so
