"""A tripwire for one compiler-dependent property of the kernels (no GPU needed: hipcc cross-compiles to gfx950 assembly).

Three kernels have "some lanes publish, other lanes of possibly the same wave wait for a publish" written as consecutive
branches over disjoint sets of lanes.  The order of such branches in the emitted code is the compiler's choice; it once put
a waiting loop in front of the store it waits for and two waves deadlocked (profiles/r05/experiment_pattern_id_counter.txt).
pattern_insert_block and finish_kernel now have a workgroup barrier between the two; cluster_dedup_kernel does not (a
convergent point there cost 2 - 3 % of the kernel).  This test reads the assembly: in each of those kernels the first
waiting loop (`s_sleep`) must come after the claim and the stores that publish it."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc is not here")
    out = tmp_path_factory.mktemp("isa") / "pf_api.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out),
                           os.path.join(REPO, "panfeed_amd", "csrc", "pf_api.hip")], stderr=subprocess.DEVNULL)
    bodies, name, lines = {}, None, []
    for line in open(out):
        m = re.match(r"^(_ZN2pf\w+):", line)
        if m:
            name, lines = m.group(1), []
            bodies[name] = lines
        elif name is not None:
            lines.append(line)
            if line.startswith(".Lfunc_end"):
                name = None
    return bodies


def _first(lines, pattern, start=0):
    rx = re.compile(pattern)
    for i in range(start, len(lines)):
        if rx.search(lines[i]):
            return i
    return None


@pytest.mark.parametrize("cfg", ["10DedupSmall", "9DedupWide"])
def test_dedup_publishes_before_it_waits(kernels, cfg):
    body = kernels[f"_ZN2pf20cluster_dedup_kernelINS_{cfg}EEEvNS_11DedupParamsE"]
    claim = _first(body, r"ds_cmpst_rtn_b64")                 # the group table's compare-and-swap
    wait = _first(body, r"\bs_sleep\b")                       # the others' wait for the pool offset
    assert claim is not None and wait is not None and claim < wait
    stores = [i for i in range(claim, wait) if re.search(r"\bds_write", body[i])]
    assert len(stores) >= 3, "the registrar's pool / length / offset stores are not in front of the waiting loop any more"


@pytest.mark.parametrize("name", ["_ZN2pf11emit_kernelENS_10EmitParamsE",
                                  "_ZN2pf13finish_kernelINS_8FinSmallELb0EEEvNS_12FinishParamsE",
                                  "_ZN2pf13finish_kernelINS_8FinLargeELb0EEEvNS_12FinishParamsE",
                                  "_ZN2pf13finish_kernelINS_9FinLargeMELb1EEEvNS_12FinishParamsE",
                                  "_ZN2pf13finish_kernelINS_7FinHugeELb1EEEvNS_12FinishParamsE"])
def test_workgroup_insert_publishes_before_it_waits(kernels, name):
    """the workgroup's ONE add on the id counter (a scalar-addressed global_atomic_add: thread 0's), then the publish store
    (an 8-byte sc1 store), a barrier, and only then a waiting loop"""
    body = kernels[name]
    # emit_kernel's thread 0 inserts the cluster's own row first (the waiting form, with its own add): skip to the barrier
    # that closes pass 1 -- finish_kernel has no such prologue
    at = 0
    if "emit_kernel" in name:
        for _ in range(3):                                    # start-up, pass 1, pattern_insert_block's first barrier
            at = _first(body, r"\bs_barrier\b", at) + 1
        at -= 1
    add = _first(body, r"global_atomic_add\b", at)
    assert add is not None
    publish = _first(body, r"global_store_dwordx2 .* sc1", add)
    barrier = _first(body, r"\bs_barrier\b", publish)
    wait = _first(body, r"\bs_sleep\b", add)
    assert publish is not None and barrier is not None and wait is not None
    assert add < publish < barrier < wait
