# build the library with extra compile flags and run the default bench: bash tools/variant_bench.sh NAME "FLAGS" [bench args]
# (on a GPU box; results under gpurun_out/variants/)
set -e
NAME=$1; FLAGS=$2; shift 2
mkdir -p gpurun_out/variants
PF_CXXFLAGS="$FLAGS" python -c 'import __graft_entry__ as g; g.build(force=True)' > gpurun_out/variants/$NAME.build.log 2>&1
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-every-copy-leg "$@" > gpurun_out/variants/$NAME.json 2> gpurun_out/variants/$NAME.err
python - "$NAME" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/variants/{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], "%.3e" % d["value"], round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in d["device_ms_per_step"].items() if v})
PY
