# ~150 distinct sequences per cluster (2 000 clusters x 1 000 samples; SURVEY's "star" alleles and related "tree" alleles):
# rocprofv3 kernel stats + the three PMC passes (HBM read, HBM write, SQ) of tools/tree_time.py, one summary per model.
# usage (GPU box): bash tools/d150_profile.sh [ROUND] -> gpurun_out/ROUND_d150/, to be copied into profiles/ROUND/
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r05}; OUT=gpurun_out/${R}_d150
rm -rf $OUT && mkdir -p $OUT
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"
for M in star tree; do
  T="python tools/tree_time.py 2000 150 $M"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${M}_stats -o run -- $T > $OUT/${M}_stats.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${M}_fetch -o run -- $T > $OUT/${M}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${M}_write -o run -- $T > $OUT/${M}_write.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/${M}_sq -o run -- $T > $OUT/${M}_sq.log 2>&1
  python tools/pmc_summary.py 2 $OUT/d150_${M}_pmc_summary.json $OUT/${M}_fetch $OUT/${M}_write $OUT/${M}_sq
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
ls -la $OUT
