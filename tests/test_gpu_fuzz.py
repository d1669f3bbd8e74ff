"""A seeded slice of the randomised differential test (tests/fuzz_parity.py: random options and cluster shapes, HIP path
vs the oracle, text for text; 4 300 cases were run clean when this was written)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_fuzz_slice():
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "tests", "fuzz_parity.py"), "80", "7"], cwd=repo,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DONE 80 cases 0 failures" in r.stdout


def test_sibling_items_insert_the_same_patterns():
    """Case 8 of the slice on its own, with a short fuse: a cluster that overflows and is re-run in key partitions, whose
    workgroups then insert the same patterns into the run-global table at the same time.  The first version of
    pattern_insert_block never came back from it (the compiler had put the waiting lanes' loop in front of the claiming lanes'
    publish: profiles/r05/experiment_pattern_id_counter.txt); a hang here is that class of bug."""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "tests", "fuzz_parity.py"), "80", "7"], cwd=repo,
                       capture_output=True, text=True, timeout=120, env=dict(os.environ, PF_FUZZ_ONLY="8"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 failures" in r.stdout and "'retried': 1" in r.stdout
