# rocprofv3 kernel stats of round 5's new kernels: kt_len / kt_text (kmers.tsv on the device: the --targets pass with every strain a
# target, 4 clusters x 5 000 samples) and genome_pack_text_kernel (one-pass ingest: files -> files at 2 000 clusters).
# usage (GPU box): bash tools/new_kernels_stats.sh [ROUND] -> gpurun_out/ROUND_newk/
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r05}; OUT=gpurun_out/${R}_newk
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/targets -o run -- python bench.py --samples 5000 --k 21 --clusters 200 --steps 1 --warmup 1 --no-n-leg --no-e2e-leg --no-every-copy-leg --no-cpu-baseline --targets-clusters 4 > $OUT/targets.json 2> $OUT/targets.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ingest -o run -- python tools/e2e_ab.py 2000 1 16 > $OUT/ingest.txt 2> $OUT/ingest.err
find $OUT -name "*kernel_trace.csv" -delete
grep -h "kt_\|genome_pack\|gather_segments" $OUT/targets/run_kernel_stats.csv $OUT/ingest/run_kernel_stats.csv
