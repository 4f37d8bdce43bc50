# A/B of two builds of the library inside one gpurun call (tools/ab/*.so copied over the in-tree one in the box's scratch copy)
set -e
cd $GRAFT_REPO_ROOT
for v in base mv base mv; do
  cp tools/ab/libtfrecomm_hip_$v.so tf-recomm_amd/csrc/libtfrecomm_hip.so
  for z in "" "--zipf 1.05"; do
    echo "build=$v $z"
    python bench.py --workload c3 $z --steps 60 --warmup 10 --no-cpu-baseline --no-north-star 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['roofline']['kernels']
print('  ms_per_step %.4f  value %.3e' % (d['ms_per_step'], d['value']), {s: round(v['us_per_step'],1) for s,v in k.items()})
"
  done
done
cp tools/ab/libtfrecomm_hip_base.so tf-recomm_amd/csrc/libtfrecomm_hip.so
