set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r01b && mkdir -p gpurun_out/r01b
if [ "$1" != "--prof-only" ]; then
timeout -k 10 500 python bench.py --steps 5 --warmup 2 > gpurun_out/r01b/bench.json 2> gpurun_out/r01b/bench.err
fi
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01b/stats -o run -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-every-copy-leg > gpurun_out/r01b/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r01b/fetch -o run -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-every-copy-leg > gpurun_out/r01b/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r01b/write -o run -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-every-copy-leg > gpurun_out/r01b/write.log 2>&1
python tools/pmc_summary.py 4 gpurun_out/r01b/pmc_traffic.json gpurun_out/r01b/fetch gpurun_out/r01b/write
find gpurun_out/r01b -name "*kernel_trace.csv" -delete
ls -la gpurun_out/r01b/*
