# A/B of two prebuilt libraries on the same box: bash tools/ab_bench.sh ab/base.so ab/new.so [rounds]
set -e
mkdir -p gpurun_out/ab
for r in $(seq 1 ${3:-2}); do
  for v in "$1" "$2"; do
    cp "$v" panfeed_amd/libpanfeed_hip.so
    timeout -k 10 200 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-every-copy-leg > gpurun_out/ab/out.json 2> gpurun_out/ab/err
    python - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab/out.json").read().strip().splitlines()[-1])
print(sys.argv[1], "%.3e" % d["value"], round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in d["device_ms_per_step"].items() if v})
PY
  done
done
