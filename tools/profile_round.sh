# bench line + rocprofv3 kernel stats + PMC passes (HBM bytes, SQ counters) of the default bench command.
# usage (on a GPU box): PF_COMMIT=<git rev-parse --short HEAD, taken where .git is> bash tools/profile_round.sh [--prof-only] [ROUND] ;
# results under gpurun_out/ROUND/ (default r05), to be
# copied into profiles/ROUND/
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PROF_ONLY=0; if [ "$1" = "--prof-only" ]; then PROF_ONLY=1; shift; fi
OUT=gpurun_out/${1:-r05}
rm -rf $OUT && mkdir -p $OUT
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-every-copy-leg --no-n-leg --no-e2e-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $B > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- $B > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- $B > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -o run -- $B > $OUT/sq.log 2>&1
python tools/pmc_summary.py 4 $OUT/final_pmc_summary.json $OUT/fetch $OUT/write $OUT/sq
# the bench line last, with this build's counter summary where bench.py looks for it (profiles/ROUND/): the line's
# roofline.traffic then comes from the passes above (traffic_commit, traffic_matches_this_build: true)
R=$(basename $OUT); mkdir -p profiles/$R && cp $OUT/final_pmc_summary.json profiles/$R/final_pmc_summary.json
if [ "$PROF_ONLY" = 0 ]; then
timeout -k 10 500 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
fi
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
ls -la $OUT/*
