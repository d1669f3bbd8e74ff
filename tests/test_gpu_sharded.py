"""The N>1 end-to-end path on the GPU: world 2 and 3, one process per rank, each with its own Engine (all on cuda:0
on a one-GPU box; collectives over gloo on host tensors).  The files rank 0 assembles must equal the reference's
files (golden case) and the single-rank files (seeded 1 000-sample case) byte for byte: BASELINE configs[3]'s
"sharded + pattern dedup" with the drop-in outputs, not just counts."""
import os

import pytest

from test_sharded_cpu import CASES, FILES, read_out, run_world

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", ["rand70_shuffled", "rand12_missing", "rand12_k51_noncanon"])
def test_sharded_engine_equals_reference_files(tmp_path, name, world):
    out = str(tmp_path / "panfeed")
    os.mkdir(out)
    stats = run_world("gpu", world, out, name)
    exp = CASES[name]["expect"]
    for f in FILES:
        assert read_out(out, f, False) == exp[f], f
    assert stats[0]["patterns"] == exp["n_patterns"]
    assert sum(s["pattern_rows"] for s in stats.values()) == exp["n_patterns"]
    # some rank held a pattern an earlier rank had seen first and did not write it
    assert sum(s["local_patterns"] for s in stats.values()) >= exp["n_patterns"]


def test_sharded_1000_samples_equals_single_rank_and_oracle(tmp_path):
    """seeded 14 clusters x 1 000 samples, k = 31, +-60 bp, paralogs, Ns, two target strains: world 1, 2 and 3 write
    the same bytes, and those are the oracle's"""
    spec = "seeded:14:1000:31:60:4242"
    outs = {}
    shared = 0
    for world in (1, 2, 3):
        out = str(tmp_path / f"w{world}")
        os.mkdir(out)
        stats = run_world("gpu", world, out, spec, timeout=900)
        outs[world] = {f: read_out(out, f, False) for f in FILES}
        if world > 1:
            shared += sum(s["local_patterns"] for s in stats.values()) - stats[0]["patterns"]
    for f in FILES:
        assert outs[2][f] == outs[1][f], f
        assert outs[3][f] == outs[1][f], f
    assert shared > 0, "no pattern was shared between ranks: the test does not exercise the merge"
    from sharded_worker import load_case
    from oracle import oracle as po
    records, strains, opts = load_case(spec)
    run = po.OracleRun(klength=opts["klength"], stroi=set(opts["stroi"]), threads=8)
    run.feed(records)
    ek, ekh, ehp = run.texts()
    from panfeed_amd.engine import KMERS_TO_HASHES_HEADER, KMERS_TSV_HEADER, hashes_to_patterns_header
    assert outs[2]["kmers_to_hashes.tsv"] == KMERS_TO_HASHES_HEADER + ekh
    assert outs[2]["hashes_to_patterns.tsv"] == hashes_to_patterns_header(strains) + ehp
    assert outs[2]["kmers.tsv"] == KMERS_TSV_HEADER + ek


def test_sharded_gzip_and_multiple_files(tmp_path):
    out = str(tmp_path / "gz")
    os.mkdir(out)
    run_world("gpu", 2, out, "rand40_shuffled", compress=True)
    exp = CASES["rand40_shuffled"]["expect"]
    for f in FILES:
        assert read_out(out, f, True) == exp[f], f
    out = str(tmp_path / "mf")
    os.mkdir(out)
    run_world("gpu", 2, out, "rand12_mf")
    exp = CASES["rand12_mf"]["expect"]["dirs"]
    assert sorted(os.listdir(out)) == sorted(exp)
    for d in exp:
        for f in exp[d]:
            assert read_out(os.path.join(out, d), f, False) == exp[d][f], (d, f)


def test_single_rank_with_device_tensors(tmp_path):
    """world 1 through the sharded driver with the digests kept on the GPU (the tensors an RCCL run exchanges; the marks
    come from the library's hash-table kernels, pf_merge_patterns): the files are the reference's"""
    import torch

    from conftest import case_records
    from panfeed_amd import sharded
    case = CASES["rand70_shuffled"]
    out = str(tmp_path / "one")
    os.mkdir(out)
    o = case["opts"]
    stats = sharded.run_records_sharded(case_records(case), out, case["all_strains"], 0, 1, None, torch.device("cuda", 0),
                                        klength=o["klength"], canon=o["canon"], consider_missing=o["consider_missing"],
                                        patfilt=o["patfilt"], maf=o["maf"], targets=o["stroi"] or (), batch_clusters=2)
    for f in FILES:
        assert read_out(out, f, False) == case["expect"][f], f
    assert stats["patterns"] == case["expect"]["n_patterns"] == stats["pattern_rows"]


def test_rccl_calls_through_a_real_nccl_group():
    """tools/nccl_selftest.py as its own process: a `torch.distributed` group of the nccl backend (= RCCL) on the one GPU
    of this box -- the owner form's all-to-alls and all-reduce with their split sizes and dtypes, the ordinal-range
    all-gather and the id selection of `sharded.finish_shard`, on device tensors of the rank's own GPU.  (World 2 over
    nccl needs two devices; the multi-rank control flow is covered over gloo above.  No scaling curve exists.)"""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "nccl_selftest.py")], cwd=repo, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "owner: n_global 3000 (expected 3000), keep marks ok" in r.stdout
    assert "finish_shard path on cuda:0" in r.stdout and "WRONG" not in r.stdout


def test_collective_refuses_a_tensor_of_another_device():
    """under nccl a host tensor must not reach a collective (it would fail inside RCCL, or hang the group)"""
    import torch
    from panfeed_amd import distributed

    class FakeDist:
        @staticmethod
        def is_initialized():
            return True

        @staticmethod
        def get_backend():
            return "nccl"
    with pytest.raises(ValueError, match="RCCL collective"):
        distributed._check_collective_device(FakeDist, torch.zeros(4, 16, dtype=torch.uint8))
    distributed._check_collective_device(FakeDist, torch.zeros(4, device="cuda:0"))
