# the round's secondary profile artifacts (GPU box): bash tools/collect_profiles.sh [ROUND] [a|b|all]; results under gpurun_out/ROUND_extra/
# (a: small configurations, launcher checks, allele sweeps; b: fuzz runs, phase profile, configs[4], long end-to-end -- two calls fit gpurun's limit),
# to be copied into profiles/ROUND/.  (The headline's bench line, kernel stats and PMC passes: tools/profile_round.sh.)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r05}; OUT=gpurun_out/${R}_extra
if [ "${2:-all}" != "b" ]; then
rm -rf $OUT && mkdir -p $OUT
F="--no-every-copy-leg --no-n-leg --no-e2e-leg"
# BASELINE configs[1]: 5 000 clusters x 200 samples, no flanks
timeout -k 10 300 python bench.py --clusters 5000 --samples 200 --flank 0 --steps 40 --warmup 5 --no-e2e-leg > $OUT/bench_cfg1_5000x200.json 2> $OUT/cfg1.err
# the N = 8 shard of configs[3] on one GPU (what one rank of the strong-scaling run does per step, without the merge)
timeout -k 10 300 python bench.py --clusters 6250 --steps 40 --warmup 5 $F --no-cpu-baseline > $OUT/bench_shard_6250.json 2> $OUT/shard.err
# --gpus N: refusal on a one-GPU box, and the rehearsal (two ranks sharing the GPU, gloo: control flow only)
(python bench.py --gpus 2; echo "exit code $?") > $OUT/gpus2_on_one_gpu_refusal.txt 2>&1 || true
PANFEED_BENCH_SHARED_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 2> $OUT/rehearsal.err | grep '^{"metric"' > $OUT/bench_gpus2_shared_gpu_rehearsal.json
timeout -k 10 120 python tools/nccl_selftest.py > $OUT/nccl_selftest_world1.txt 2>&1
# distinct sequences per cluster: both sweeps
timeout -k 10 600 python bench.py --clusters 2000 --steps 2 --warmup 1 $F --no-cpu-baseline --sweep-alleles > $OUT/allele_sweep_star_and_tree.json 2> $OUT/sweep.err
# (kernel stats and PMC passes at ~150 distinct sequences per cluster: tools/d150_profile.sh)
if [ "${2:-all}" = "a" ]; then ls -la $OUT; exit 0; fi
fi
if [ "${2:-all}" != "a" ]; then
mkdir -p $OUT
# randomised differential runs against the oracle (tests/fuzz_parity.py): small and BASELINE-sized clusters
(timeout -k 10 400 python tests/fuzz_parity.py 1000 301 | tail -1; timeout -k 10 400 python tests/fuzz_parity.py 200 302 big | tail -1) > $OUT/fuzz_parity_runs.txt 2>&1
# cycles per phase inside rows / emit / scan / finish at ~140 distinct sequences per cluster (a -DPF_PROF build, then the shipped one again)
cp panfeed_amd/libpanfeed_hip.so $OUT/.shipped.so
trap 'cp '"$OUT"'/.shipped.so panfeed_amd/libpanfeed_hip.so' EXIT     # the shipped library back, whatever fails below (set -e)
PF_PROF=1 python -c "import __graft_entry__ as g; g.build(force=True)" > /dev/null 2>&1
for M in tree star; do (echo "== 2000 clusters, mean 150 alleles, $M"; timeout -k 10 200 python tools/phase_prof.py 2000 150 $M 2>/dev/null) >> $OUT/phase_profile_150_alleles.txt; done
cp $OUT/.shipped.so panfeed_amd/libpanfeed_hip.so
# BASELINE configs[4] shape: 5 000 samples, k = 21 and 51, --targets second pass
timeout -k 10 400 python bench.py --samples 5000 --k 21 --clusters 6000 --steps 5 --warmup 2 --no-n-leg --no-e2e-leg --targets-clusters 4 > $OUT/bench_cfg4_6000x5000_k21_targets.json 2> $OUT/cfg4a.err
timeout -k 10 400 python bench.py --samples 5000 --k 51 --clusters 6000 --steps 5 --warmup 2 --no-n-leg --no-e2e-leg > $OUT/bench_cfg4_6000x5000_k51.json 2> $OUT/cfg4b.err
# end to end at 2 000 clusters (2.4 GB of GFF input)
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-every-copy-leg --no-n-leg --e2e-clusters 2000 --e2e-long-clusters 0 > $OUT/bench_e2e_2000.json 2> $OUT/e2e.err
ls -la $OUT
fi
