# timing experiments on the scan's window loop (GPU box): bash tools/scan_exp.sh "4 0"
set -e
mkdir -p gpurun_out/scanexp
for v in ${1:-3 2 1 0}; do
  PF_CXXFLAGS="-DPF_SCAN_EXP=$v" python -c 'import __graft_entry__ as g; g.build(force=True)' > gpurun_out/scanexp/build$v.log 2>&1
  echo "PF_SCAN_EXP=$v" >> gpurun_out/scanexp/times.txt
  timeout -k 10 200 python tools/scan_time.py ${2:-20000} 2>/dev/null | tail -1 >> gpurun_out/scanexp/times.txt
done
cat gpurun_out/scanexp/times.txt
