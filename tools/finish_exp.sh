# timing experiments on finish_kernel (GPU box): the kernel cut off after each of its phases; only finish_ms means anything
set -e
mkdir -p gpurun_out/finexp
for v in ${1:-1 2 3 4 5 6 7 0}; do
  PF_CXXFLAGS="-DPF_FIN_EXP=$v" python -c 'import __graft_entry__ as g; g.build(force=True)' > gpurun_out/finexp/build$v.log 2>&1
  echo "PF_FIN_EXP=$v" >> gpurun_out/finexp/times.txt
  timeout -k 10 200 python tools/scan_time.py ${2:-20000} 2>/dev/null | tail -1 >> gpurun_out/finexp/times.txt
done
cat gpurun_out/finexp/times.txt
