# A/B timing of library variants on one GPU box (the one script for kernel experiments).
#   bash tools/ab_bench.sh build NAME ["EXTRA CXXFLAGS"]     build the tree as it is into ab/NAME.so (gfx950)
#   bash tools/ab_bench.sh run ab/A.so ab/B.so [ROUNDS] [CMD...]   alternate the two libraries ROUNDS times over CMD
#       (default CMD: the headline bench without its extra legs); prints value, ms/step and the per-kernel device times
# ab/ is git-ignored and travels to the GPU box with the snapshot.
set -e
BENCH_DEFAULT="python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-every-copy-leg --no-n-leg --no-e2e-leg"
case "$1" in
build)
  mkdir -p ab
  cp panfeed_amd/libpanfeed_hip.so ab/.shipped.so 2>/dev/null || true
  trap 'if [ -f ab/.shipped.so ]; then mv ab/.shipped.so panfeed_amd/libpanfeed_hip.so; fi' EXIT
  PF_CXXFLAGS="$3" python -c 'import __graft_entry__ as g; g.build(force=True)'
  cp panfeed_amd/libpanfeed_hip.so "ab/$2.so"
  ;;
run)
  A=$2; B=$3; R=${4:-2}; shift; shift; shift; [ $# -gt 0 ] && shift
  CMD=${*:-$BENCH_DEFAULT}
  mkdir -p gpurun_out/ab
  cp panfeed_amd/libpanfeed_hip.so gpurun_out/ab/.orig.so
  # whatever way this ends (a failed or timed-out step under set -e included), the shipped library is back in its place
  trap 'cp gpurun_out/ab/.orig.so panfeed_amd/libpanfeed_hip.so' EXIT
  for r in $(seq 1 $R); do
    for v in "$A" "$B"; do
      cp "$v" panfeed_amd/libpanfeed_hip.so
      timeout -k 10 300 $CMD > gpurun_out/ab/out.json 2> gpurun_out/ab/err
      python - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab/out.json").read().strip().splitlines()[-1])
print(sys.argv[1], "%.3e" % d["value"], round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d.get("device_ms_per_step", {}).items() if v}, flush=True)
PY
    done
  done
  ;;
*) echo "usage: ab_bench.sh build NAME [FLAGS] | run A.so B.so [ROUNDS] [CMD...]"; exit 2;;
esac
