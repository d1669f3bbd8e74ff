# emit_kernel's item-local pattern table: 4096 slots (one workgroup per CU) against 2048 (two)  (GPU box)
set -e
mkdir -p gpurun_out/ltexp
for v in 4096 2048 1024; do
  PF_CXXFLAGS="-DPF_LT_SLOTS=$v" python -c 'import __graft_entry__ as g; g.build(force=True)' > gpurun_out/ltexp/build.log 2>&1
  for a in "150 tree" "60 star" "60 tree" "500 tree"; do
    echo "LT_SLOTS=$v $a" >> gpurun_out/ltexp/times.txt
    timeout -k 10 200 python tools/tree_time.py 2000 $a 2>/dev/null | tail -1 >> gpurun_out/ltexp/times.txt
  done
done
cat gpurun_out/ltexp/times.txt
