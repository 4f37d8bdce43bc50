# A/B of two builds of the library on the headline workload inside one gpurun call (tools/ab/*.so copied over the in-tree one)
set -e
cd $GRAFT_REPO_ROOT
for v in base mv base mv base mv; do
  cp tools/ab/libtfrecomm_hip_$v.so tf-recomm_amd/csrc/libtfrecomm_hip.so
  echo "build=$v"
  python bench.py --steps 900 --warmup 50 --no-configs --no-north-star --no-convergence --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('  900 steps: %.3f us/step' % (d['ms_per_step']*1e3), {k: round(v['us_per_step'],2) for k,v in d['roofline']['kernels'].items()})"
  python bench.py --steps 20 --warmup 5 --no-configs --no-north-star --no-convergence --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   20 steps: %.3f us/step' % (d['ms_per_step']*1e3))"
done
cp tools/ab/libtfrecomm_hip_base.so tf-recomm_amd/csrc/libtfrecomm_hip.so
