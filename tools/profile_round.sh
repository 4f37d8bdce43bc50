# One gpurun call that produces every rocprofv3 artefact of a round (reduced into profiles/ afterwards by tools/collect_profiles.py):
#   bash tools/profile_round.sh r03 [only=<workload>]
# Per workload: one --kernel-trace --stats pass (durations) and four separate --pmc passes (FETCH_SIZE; WRITE_SIZE;
# TCC_HIT/MISS; TCC_EA0_RDREQ/WRREQ) - never combined with other trace domains (MI355X_MICROARCH.md, rocprofv3 PMC slots).
# The program after `--` is python3 itself (no env / bash -c hop: the profiler's preloaded library initialises the GPU).
set -e
R=${1:-r03}
ONLY=${2:-}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$R
mkdir -p $O
run() {   # name, bench args...
  local name=$1; shift
  if [ -n "$ONLY" ] && [ "$ONLY" != "$name" ]; then return; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$name -- python3 bench.py "$@" > $O/bench_$name.json 2> $O/stats_$name.log || echo "stats $name failed"
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    N=$(echo $C | tr ' ' '_')
    timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${name}_$N -- python3 bench.py "$@" > $O/pmc_${name}_$N.log 2>&1 || echo "pmc $name $N failed"
    echo done $name $N
  done
  # the raw dumps stay on the box: only the per-kernel reductions travel back (gpurun merges <= 64 MiB)
  python3 tools/summarize_pmc.py $O/pmc_${name}_* > $O/pmc_$name.csv
  cp $(ls $O/stats_$name/*/*kernel_stats.csv | head -1) $O/kernel_stats_$name.csv
  rm -rf $O/pmc_${name}_*/ $O/stats_$name/
}
run c2 --steps 200 --warmup 20 --no-cpu-baseline --no-north-star --no-convergence --no-configs
run forward_uniform --only-north-star --steps 100 --warmup 10
run forward_zipf --only-north-star --steps 100 --warmup 10 --ns-zipf 1.05
run forward_8x_batch --only-north-star --steps 30 --warmup 4 --ns-batch 2097152
run c3 --workload c3 --steps 40 --warmup 8 --no-cpu-baseline --no-north-star
run c4 --workload c4 --steps 40 --warmup 8 --no-cpu-baseline --no-north-star
run c5 --workload c5 --steps 30 --warmup 5 --no-cpu-baseline
ls $O
