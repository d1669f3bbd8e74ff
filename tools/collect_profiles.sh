# the round's secondary profile artifacts (GPU box): bash tools/collect_profiles.sh [ROUND] [a|b|c|all]; results under gpurun_out/ROUND_extra/
# (a: small configurations, launcher checks, allele sweeps; b: fuzz runs, end to end at 2 000 clusters; c: configs[4] at one rank's real
# share, 25 000 x 5 000 -- each call fits gpurun's limit),
# to be copied into profiles/ROUND/.  (The headline's bench line, kernel stats and PMC passes: tools/profile_round.sh.)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r05}; OUT=gpurun_out/${R}_extra
if [ "${2:-all}" = "a" ] || [ "${2:-all}" = "all" ]; then
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
if [ "${2:-all}" = "b" ] || [ "${2:-all}" = "all" ]; then
mkdir -p $OUT
# randomised differential runs against the oracle (tests/fuzz_parity.py): small and BASELINE-sized clusters
(timeout -k 10 400 python tests/fuzz_parity.py 1000 301 | tail -1; timeout -k 10 400 python tests/fuzz_parity.py 200 302 big | tail -1) > $OUT/fuzz_parity_runs.txt 2>&1
# (the -DPF_PROF phase-stamp build: tools/phase_prof.py, by hand -- finish_kernel's stamps are off in it since round 5, where they made
# the kernel fault on large batches; its phases have the knock-out builds of tools/ab_bench.sh, -DPF_KO_FINISH, and the dedup pass
# -DPF_KO_DEDUP: profiles/r05/experiment_*_knockout.txt)
# end to end at 2 000 clusters (2.4 GB of GFF input)
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-every-copy-leg --no-n-leg --e2e-clusters 2000 --e2e-long-clusters 0 > $OUT/bench_e2e_2000.json 2> $OUT/e2e.err
ls -la $OUT
fi
if [ "${2:-all}" = "c" ] || [ "${2:-all}" = "all" ]; then
mkdir -p $OUT
# BASELINE configs[4] at one rank's real share of 200 000 x 5 000 over 8 GPUs: 25 000 clusters x 5 000 samples, k = 21 (with the --targets
# second pass on 64 clusters, every strain a target) and k = 51
timeout -k 10 560 python bench.py --samples 5000 --k 21 --clusters 25000 --steps 3 --warmup 1 --no-n-leg --no-e2e-leg --targets-clusters 64 > $OUT/bench_cfg4_25000x5000_k21_targets64.json 2> $OUT/cfg4a.err
timeout -k 10 560 python bench.py --samples 5000 --k 51 --clusters 25000 --steps 3 --warmup 1 --no-n-leg --no-e2e-leg > $OUT/bench_cfg4_25000x5000_k51.json 2> $OUT/cfg4b.err
ls -la $OUT
fi
