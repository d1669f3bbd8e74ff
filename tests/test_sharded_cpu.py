"""The N>1 end-to-end path without a GPU: world 2 and 3 under gloo.  Every rank runs the product's shard code
(panfeed_amd/sharded.py: part files, digest exchange, keep marks -> which pattern rows a rank writes, rank-ordered
assembly) with the CPU oracle standing in for the engine; the assembled files must equal the reference's own files
(golden cases) / the single-run oracle files byte for byte -- the single-writer semantics of
/root/reference/panfeed/__main__.py:67-81 and the first-seen rule of panfeed.py:179-187, 210-223."""
import gzip
import json
import os
import subprocess
import sys

import pytest

from conftest import all_cases

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {c["name"]: c for c in all_cases()}
FILES = ("kmers.tsv", "kmers_to_hashes.tsv", "hashes_to_patterns.tsv")


def run_world(mode, world, outdir, spec, compress=False, method="owner", timeout=600):
    port = str(29600 + (os.getpid() * 7 + world * 13 + hash((spec, compress, method)) % 500) % 3000)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "sharded_worker.py"), mode, str(r), str(world),
                               port, outdir, spec, "1" if compress else "0", method],
                              cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    stats = {}
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=timeout)
        assert p.returncode == 0, f"rank {r} failed:\n{out[-2000:]}\n{err[-4000:]}"
        line = [x for x in out.splitlines() if x.startswith("STATS ")][-1]
        stats[r] = json.loads(line[6:])
    return stats


def read_out(outdir, name, compress):
    if compress:
        with gzip.open(os.path.join(outdir, name + ".gz"), "rt") as fh:
            return fh.read()
    with open(os.path.join(outdir, name)) as fh:
        return fh.read()


@pytest.mark.parametrize("method", ["owner", "allgather"])
@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", ["rand12_basic", "rand40_shuffled_missing", "rand70_shuffled"])
def test_sharded_files_equal_reference_files(tmp_path, name, world, method):
    if method == "allgather" and name != "rand12_basic":
        pytest.skip("one case is enough for the second form of the exchange")
    out = str(tmp_path / "panfeed")
    os.mkdir(out)
    stats = run_world("oracle", world, out, name, method=method)
    exp = CASES[name]["expect"]
    for f in FILES:
        assert read_out(out, f, False) == exp[f], f
    assert sorted(os.listdir(out)) == sorted(FILES)          # the parts are gone
    assert stats[0]["patterns"] == exp["n_patterns"]
    assert sum(s["pattern_rows"] for s in stats.values()) == exp["n_patterns"]
    # patterns shared between ranks were dropped by the later rank: some rank wrote fewer rows than it holds
    assert [s["range"] for s in stats.values()] == sorted(s["range"] for s in stats.values())


def test_sharded_gzip_parts_concatenate(tmp_path):
    out = str(tmp_path / "panfeed")
    os.mkdir(out)
    run_world("oracle", 2, out, "rand12_basic", compress=True)
    exp = CASES["rand12_basic"]["expect"]
    for f in FILES:
        assert read_out(out, f, True) == exp[f], f
        assert subprocess.run(["gzip", "-t", os.path.join(out, f + ".gz")]).returncode == 0


def test_world_8_as_on_a_full_node(tmp_path):
    """eight ranks (the node the multi-GPU runs are for), 12 clusters: ranges of one or two clusters each, every rank but
    the first dropping the patterns an earlier one saw; gzip parts of eight ranks concatenate"""
    out = str(tmp_path / "w8")
    os.mkdir(out)
    stats = run_world("oracle", 8, out, "rand12_basic", timeout=900)
    exp = CASES["rand12_basic"]["expect"]
    for f in FILES:
        assert read_out(out, f, False) == exp[f], f
    assert sum(s["pattern_rows"] for s in stats.values()) == exp["n_patterns"]
    out = str(tmp_path / "w8gz")
    os.mkdir(out)
    run_world("oracle", 8, out, "rand12_missing", compress=True, timeout=900)
    for f in FILES:
        assert read_out(out, f, True) == CASES["rand12_missing"]["expect"][f], f


def test_more_ranks_than_clusters(tmp_path):
    """edge_k5 has 4 clusters: with world 3 the ranges are 2/1/1, with a 1-cluster case some ranks are empty"""
    out = str(tmp_path / "a")
    os.mkdir(out)
    run_world("oracle", 3, out, "toy_canon")
    for f in FILES:
        assert read_out(out, f, False) == CASES["toy_canon"]["expect"][f], f


def test_overlapping_ordinals_are_refused():
    """ranks that number their clusters from 0 each (the bug the check exists for) must not pass silently"""
    import numpy as np

    from panfeed_amd.sharded import check_first_seen_disjoint

    class FakeDist:
        def __init__(self, ranges):
            self.ranges = ranges

        def is_initialized(self):
            return True

        def get_world_size(self):
            return len(self.ranges)

        def get_backend(self):
            return "gloo"

        def all_gather(self, out, mine):
            import torch
            for t, r in zip(out, self.ranges):
                t.copy_(torch.tensor(r, dtype=torch.int64))

    fs = (np.arange(5, dtype=np.uint64) << np.uint64(32))
    check_first_seen_disjoint(fs, FakeDist([[0, 4], [5, 9]]))
    with pytest.raises(RuntimeError):
        check_first_seen_disjoint(fs, FakeDist([[0, 4], [0, 4]]))
