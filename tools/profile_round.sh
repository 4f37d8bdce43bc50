set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r01b
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.log
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  N=$(echo $C | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$N -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/pmc_$N.log 2>&1
  echo done $N
done
ls $O
