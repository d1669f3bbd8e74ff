"""Key binning (bin_kernel): a cluster whose distinct k-mers take three or more passes of the on-chip table has its windows
sorted by key partition first; its scan items read their own entries.  Clusters built to need that many partitions -- alleles
that each carry their own substitutions -- with one- and two-word keys, both strands kept (two entries per window), 'N's and
paralogs on the slow path beside them, plain views (unit view off) and unit views: HIP path with binning on and off against
the oracle, text for text, and the counter that says the path was taken."""
import numpy as np
import pytest

from test_gpu_parity import _oracle_texts

pytestmark = pytest.mark.gpu


def _records(n, S, **gen):
    from panfeed_amd import synth
    return [c.record() for c in synth.generate(n, S, **gen)]


@pytest.mark.parametrize("k,canon,unit_dedup", [(31, True, True), (31, False, True), (47, True, True), (63, False, False), (21, True, False)],
                         ids=["k31", "k31_both_strands", "k47_two_words", "k63_both_strands_plain_view", "k21_plain_view"])
def test_binned_clusters_match_the_oracle(k, canon, unit_dedup):
    from panfeed_amd.engine import Engine
    S = 320
    recs = _records(5, S, first=700 + k, flank=60, mean_len=900, min_len=400, max_len=1600, n_rate=0.02, paralog_rate=0.05,
                    sub_rate=0.02, mean_alleles=48.0, allele_decay=1.0, allele_model="star")
    (ek, ekh, ehp), _ = _oracle_texts(recs, stroi=(), klength=k, canon=canon)
    seen = {}
    for binning in (True, False):
        eng = Engine(klength=k, canon=canon, max_strains=S, unit_dedup=unit_dedup, key_binning=binning, max_items=4096)
        out = eng.run(recs)
        assert out.kmers_to_hashes == ekh, binning
        assert out.hashes_to_patterns == ehp, binning
        assert out.kmers_tsv == ek, binning
        seen[binning] = out.timing["n_binned_clusters"]
        eng.close()
    assert seen[True] > 0 and seen[False] == 0


def test_binned_clusters_in_several_sub_batches_and_batches():
    """few work items per sub-batch (the entry arrays are per sub-batch), two submits on one context (pattern ids carry
    over), a target strain (positional rows)"""
    from panfeed_amd.engine import Engine
    S, k = 200, 31
    recs = _records(9, S, first=4242, flank=30, mean_len=700, min_len=300, max_len=1200, n_rate=0.0, paralog_rate=0.02,
                    sub_rate=0.03, mean_alleles=40.0, allele_decay=1.0, allele_model="star")
    names = sorted(recs[0][0].keys())
    stroi = {names[3]}
    (ek, ekh, ehp), _ = _oracle_texts(recs, stroi=stroi, klength=k, canon=True)
    eng = Engine(klength=k, max_strains=S + 24, stroi=stroi, max_items=24)
    outs = [eng.run(recs[:4]), eng.run(recs[4:])]
    assert sum(o.timing["n_binned_clusters"] for o in outs) > 0
    assert "".join(o.kmers_to_hashes for o in outs) == ekh
    assert "".join(o.hashes_to_patterns for o in outs) == ehp
    assert "".join(o.kmers_tsv for o in outs) == ek
    eng.close()


def test_scratch_grows_for_a_cluster_of_more_items_than_max_items():
    """a cluster that needs more work items than the context was created for used to end the run (PF_ERR_CAPACITY "raise
    max_items"); the scratch is re-made instead, once, and the results are the oracle's"""
    from panfeed_amd.engine import Engine
    S, k = 200, 31
    recs = _records(5, S, first=777, flank=30, mean_len=900, min_len=600, max_len=1200, n_rate=0.0, paralog_rate=0.02,
                    sub_rate=0.03, mean_alleles=60.0, allele_decay=1.0, allele_model="star")
    (ek, ekh, ehp), _ = _oracle_texts(recs, klength=k, canon=True)
    eng = Engine(klength=k, max_strains=S + 24, max_items=3)
    outs = [eng.run(recs[:2]), eng.run(recs[2:])]
    t = eng.timing()
    assert t["n_scratch_grown"] >= 1 and max(o.timing["n_items"] for o in outs) > 3
    assert "".join(o.kmers_to_hashes for o in outs) == ekh
    assert "".join(o.hashes_to_patterns for o in outs) == ehp
    eng.close()


def test_fused_finish_beside_the_general_path():
    """a launch with a few simple clusters (fused finish) and a few of many distinct sequences (general path): the finish
    kernels run on the context's second stream beside rows / emit / pattern rows; output room is claimed atomically by both
    sides -- the files are the oracle's, in cluster order, whatever order the two sides finish in"""
    from panfeed_amd.engine import Engine
    S, k = 200, 31
    few = _records(20, S, first=31000, flank=30, mean_len=500, min_len=200, max_len=900, n_rate=0.01, paralog_rate=0.03)
    many = _records(5, S, first=32000, flank=30, mean_len=700, min_len=500, max_len=900, n_rate=0.0, paralog_rate=0.02,
                    mean_alleles=90.0, allele_decay=1.0, allele_model="tree")
    recs = few[:10] + many[:3] + few[10:] + many[3:]
    (ek, ekh, ehp), _ = _oracle_texts(recs, klength=k, canon=True)
    eng = Engine(klength=k, max_strains=S + 24)
    outs = [eng.run(recs[:14]), eng.run(recs[14:])]
    assert sum(o.timing["n_side_launches"] for o in outs) >= 2
    assert "".join(o.kmers_to_hashes for o in outs) == ekh
    assert "".join(o.hashes_to_patterns for o in outs) == ehp
    eng.close()
