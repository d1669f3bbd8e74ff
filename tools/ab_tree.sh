# star / tree timings (tools/tree_time.py 2000 150) of library variants built by tools/ab_bench.sh, alternated on one box:
#   bash tools/ab_tree.sh ab/A.so ab/B.so ...
set -e
mkdir -p gpurun_out/ab
cp panfeed_amd/libpanfeed_hip.so gpurun_out/ab/.orig.so
trap 'cp gpurun_out/ab/.orig.so panfeed_amd/libpanfeed_hip.so' EXIT
for v in "$@"; do
  cp "$v" panfeed_amd/libpanfeed_hip.so
  for m in star tree; do
    timeout -k 10 300 python tools/tree_time.py 2000 150 $m 2> gpurun_out/ab/err | python -c "
import ast, sys
d = ast.literal_eval(sys.stdin.read().strip().splitlines()[-1])
print('$v', '$m', {k: round(d[k], 3) for k in ('total_ms', 'scan_ms', 'rows_ms', 'emit_ms', 'patrows_ms', 'md5_ms', 'dedup_ms', 'finish_ms')}, flush=True)"
  done
done
