# timing experiments on rows_kernel's wide path (GPU box): the kernel cut off after each of its phases; only rows_ms means anything
set -e
mkdir -p gpurun_out/rowsexp
for v in ${1:-1 2 3 4 0}; do
  PF_CXXFLAGS="-DPF_ROWS_EXP=$v" python -c 'import __graft_entry__ as g; g.build(force=True)' > gpurun_out/rowsexp/build$v.log 2>&1
  echo "PF_ROWS_EXP=$v" >> gpurun_out/rowsexp/times.txt
  timeout -k 10 200 python tools/tree_time.py ${2:-2000} ${3:-150} ${4:-tree} 2>/dev/null | tail -1 >> gpurun_out/rowsexp/times.txt
done
cat gpurun_out/rowsexp/times.txt
