cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/s4
cp panfeed_amd/libpanfeed_hip.so gpurun_out/s4/.orig.so
for v in split ins_ko1 ins_ko2; do
  cp ab/$v.so panfeed_amd/libpanfeed_hip.so
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s4/$v -o run -- python tools/tree_time.py 2000 150 star > gpurun_out/s4/$v.log 2>&1
  find gpurun_out/s4 -name "*kernel_trace.csv" -delete
  python - $v <<PY
import csv,glob,sys
f=glob.glob("gpurun_out/s4/%s/**/run_kernel_stats.csv" % sys.argv[1],recursive=True)[0]
print(sys.argv[1], {r["Name"].split("(")[0][-28:]: round(float(r["TotalDurationNs"])/2e6,3) for r in csv.DictReader(open(f)) if "emit" in r["Name"] or "pattern_list" in r["Name"] or "pattern_rows" in r["Name"]})
PY
done
cp gpurun_out/s4/.orig.so panfeed_amd/libpanfeed_hip.so
