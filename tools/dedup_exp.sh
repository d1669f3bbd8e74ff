# timing experiments on the dedup pass (GPU box): every cluster ends in mode 0, only dedup_ms means anything
set -e
mkdir -p gpurun_out/dedupexp
for v in ${1:-1 2 3 0}; do
  PF_CXXFLAGS="-DPF_DEDUP_EXP=$v" python -c 'import __graft_entry__ as g; g.build(force=True)' > gpurun_out/dedupexp/build$v.log 2>&1
  echo "PF_DEDUP_EXP=$v" >> gpurun_out/dedupexp/times.txt
  timeout -k 10 300 python tools/scan_time.py ${2:-20000} 2>/dev/null | tail -1 >> gpurun_out/dedupexp/times.txt
done
cat gpurun_out/dedupexp/times.txt
