# a few SQ-side counters of one workload's kernels, one pass per counter set (never with other trace domains):
#   bash tools/pmc_probe.sh <name> <bench args...>
set -e
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcp_$NAME
mkdir -p $O
for C in "OccupancyPercent" "MemUnitStalled" "LDSBankConflict" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" "SQ_INSTS_VALU SQ_INSTS_LDS"; do
  N=$(echo $C | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p_$N -- python3 bench.py "$@" > $O/p_$N.log 2>&1 || echo "pass $N failed"
done
python3 tools/summarize_pmc.py $O/p_* > $O/summary.csv
rm -rf $O/p_*/
grep -E "k_seg_reduce|k_fm_forward|k_tile_step|k_dense_tiles|k_forward|k_als_fit" $O/summary.csv | cut -c1-140
