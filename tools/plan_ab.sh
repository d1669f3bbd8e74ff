set -e
OUT=gpurun_out/r04f; mkdir -p $OUT
F="--no-every-copy-leg --no-n-leg --no-e2e-leg --no-cpu-baseline"
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/test.log 2>&1 || true
tail -3 $OUT/test.log
for V in "--device-plan" ""; do
  T=noplan; [ -n "$V" ] && T=plan
  timeout -k 10 300 python bench.py --clusters 5000 --samples 200 --flank 0 --steps 40 --warmup 5 $F $V > $OUT/cfg1_$T.json 2> $OUT/cfg1_$T.err
  timeout -k 10 300 python bench.py --clusters 6250 --steps 40 --warmup 5 $F $V > $OUT/shard_$T.json 2> $OUT/shard_$T.err
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 $F $V > $OUT/head_$T.json 2> $OUT/head_$T.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04f/*.json")):
    try:
        d=json.loads(open(f).read().strip().split("\n")[-1])
        pk=d["roofline"].get("per_kernel",{})
        print(f.split("/")[-1], "ms_per_step", round(d["ms_per_step"],3), "device", round(d.get("device_ms_per_step",0),3), "kernels", round(sum(v.get("ms",0) for v in pk.values()),3), "value %.3e"%d["value"])
    except Exception as e: print(f, "ERR", e)
PY
