set -e
mkdir -p gpurun_out/prexp
for v in 1024 512; do
  PF_CXXFLAGS="-DPF_PR_THREADS=$v" python -c 'import __graft_entry__ as g; g.build(force=True)' > gpurun_out/prexp/build.log 2>&1
  for a in "150 tree" "60 star" "60 tree"; do
    echo "PR_THREADS=$v $a" >> gpurun_out/prexp/times.txt
    timeout -k 10 200 python tools/tree_time.py 2000 $a 2>/dev/null | tail -1 >> gpurun_out/prexp/times.txt
  done
done
cat gpurun_out/prexp/times.txt
