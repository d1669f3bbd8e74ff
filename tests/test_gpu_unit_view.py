"""The unit view (unit_class_kernel): distinct sequences of a cluster that share a 64-window unit -- the same 64 + k - 1
bases at the same place -- have it scanned once.  Cases built around the unit grid: a substitution in a unit's halo (the
k - 1 bases that belong to the next unit's windows too), at a unit's first and last base, halos of k > 64 that span two
units, alleles of different lengths (the last unit shorter in some), units that exist in some alleles only.  Always:
HIP path (unit view on, off, no dedup at all) against the oracle, text for text."""
import numpy as np
import pytest

from test_gpu_parity import _oracle_texts

pytestmark = pytest.mark.gpu


def _cluster(idx, names, alleles, copies=3):
    """samples carry the alleles round robin (`copies` samples per allele at least, so that the identical-sequence
    shortcut -- and with it the unit view -- applies)"""
    from panfeed_amd.classes import Seqinfo
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    gs, presab = {}, np.zeros(len(names), dtype=np.int64)
    col = {x: i for i, x in enumerate(sorted(names))}
    assert len(names) >= copies * len(alleles)
    for i, nm in enumerate(names):
        sq = alleles[i % len(alleles)]
        gs[nm] = [Seqinfo(sq.decode(), sq.translate(comp).decode(), f"{nm}_{idx}", f"{nm}_c", 50 + i, 50 + i + len(sq) - 1,
                          1 if i % 2 else -1, 0)]
        presab[col[nm]] = 1
    return gs, idx, presab


def _mut(base, pos, rng=None):
    b = bytearray(base)
    for p in ([pos] if np.isscalar(pos) else pos):
        b[p] = ord("ACGT"[("ACGT".index(chr(b[p])) + 1) % 4])
    return bytes(b)


def _run_all(recs, k, S, stroi=(), canon=True):
    from panfeed_amd.engine import Engine
    (ek, ekh, ehp), st = _oracle_texts(recs, stroi=stroi, klength=k, canon=canon)
    timings = []
    for kw in (dict(), dict(unit_dedup=False), dict(dedup=False)):
        eng = Engine(klength=k, canon=canon, max_strains=(S + 31) // 32 * 32, stroi=stroi, **kw)
        out = eng.run(recs)
        assert out.kmers_to_hashes == ekh, kw
        assert out.hashes_to_patterns == ehp, kw
        assert out.kmers_tsv == ek, kw
        timings.append(out.timing)
        eng.close()
    return timings


@pytest.mark.parametrize("k", [5, 31, 33, 64, 65, 100, 126])
def test_substitutions_on_the_unit_grid(k):
    """one ancestral sequence of 5 units and a bit; alleles with ONE substitution each, at the positions where the unit
    view could go wrong: first / last base of a unit, first / last base of its halo, the sequence's first and last base"""
    rng = np.random.default_rng(k)
    L = 64 * 5 + k + 17
    anc = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].tobytes()
    spots = sorted({0, 1, 63, 64, 65, 64 + k - 2, 64 + k - 1, 64 + k, 127, 128, 128 + k - 2, 191, 192, 2 * 64 + k - 1,
                    L - k, L - k - 1, L - 1, L - 2, 256, 255, 319, 320} & set(range(L)))
    alleles = [anc] + [_mut(anc, p) for p in spots]
    S = 3 * len(alleles) + 5
    names = [f"u{i:03d}" for i in range(S)]
    recs = [_cluster("g_grid", names, alleles), _cluster("g_grid_rev", names, alleles[::-1])]
    _run_all(recs, k, S, stroi={names[1]})


@pytest.mark.parametrize("k,canon", [(31, True), (31, False), (77, True)])
def test_alleles_of_different_lengths_and_short_tails(k, canon):
    """truncated alleles: the last unit of some alleles is shorter (or missing) while its first bases equal the longer
    alleles' -- a unit's class takes its number of windows into account; alleles shorter than k contribute nothing"""
    rng = np.random.default_rng(100 + k)
    L = 64 * 4 + k + 40
    anc = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].tobytes()
    cuts = [L, L - 1, L - 39, L - 40, L - 41, 64 * 3 + k - 1, 64 * 3 + k, 64 * 2 + k + 5, k, k - 1, 64 + k - 1]
    alleles = []
    for c in cuts:
        alleles.append(anc[:c])
        if c > 70:
            alleles.append(_mut(anc[:c], 70))
    alleles = list(dict.fromkeys(alleles))
    S = 3 * len(alleles) + 2
    names = [f"t{i:03d}" for i in range(S)]
    _run_all([_cluster("g_tails", names, alleles)], k, S, stroi={names[0], names[7]}, canon=canon)


@pytest.mark.parametrize("D,S,k", [(40, 200, 31), (150, 600, 31), (150, 600, 51), (400, 1300, 21)])
def test_many_related_alleles_unit_view(D, S, k):
    """many alleles that descend from one another by one or two substitutions (mode 1 and the wide class, several
    column chunks per class, several key partitions for the larger ones): unit view on / off / no dedup == oracle"""
    rng = np.random.default_rng(D + k)
    L = 900
    alleles = [np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].tobytes()]
    seen = set(alleles)
    while len(alleles) < D:
        a = _mut(alleles[int(rng.integers(0, len(alleles)))], [int(x) for x in rng.integers(0, L, int(rng.integers(1, 3)))])
        if a not in seen:
            seen.add(a)
            alleles.append(a)
    names = [f"r{i:04d}" for i in range(S)]
    tms = _run_all([_cluster("g_rel", names, alleles), _cluster("g_rel2", names, alleles[::2])], k, S)
    assert tms[0]["n_dedup_clusters"] == 2


def test_more_distinct_sequences_than_samples():
    """17 samples, paralogs and sequences cut at an 'N': more than 32 distinct sequences with one 32-column word per
    sample row (the case the randomised test found when every cluster with a copy started to take the view of distinct
    sequences: the view's second chunk of columns has no room in the per-slot words, the cluster stays on the
    every-copy path); non-canonical, two-word keys, several key partitions"""
    from panfeed_amd import synth
    from panfeed_amd.engine import Engine
    cl = synth.generate(3, 17, first=946990, flank=60, mean_len=1500, min_len=5, max_len=2500, n_rate=0.2, paralog_rate=0.3,
                        sub_rate=0.0, mean_alleles=40.0, allele_decay=1.0, allele_model="tree", shuffle_columns=94)
    recs = [c.record() for c in cl]
    for k, canon in ((64, False), (31, True)):
        (ek, ekh, ehp), st = _oracle_texts(recs, klength=k, canon=canon, maf=0.0)
        for kw in (dict(), dict(unit_dedup=False)):
            eng = Engine(klength=k, canon=canon, maf=0.0, max_strains=32, **kw)
            out = eng.run(recs)
            assert out.kmers_to_hashes == ekh and out.hashes_to_patterns == ehp, (k, canon, kw)
            eng.close()


@pytest.mark.parametrize("D,L,S", [(260, 1250, 800), (500, 900, 1600), (600, 980, 1800), (640, 1700, 1920)], ids=["dense_325k", "dense_450k", "dense_570k_two_stretches", "dense_1070k_sorted"])
def test_wide_cluster_beyond_two_lds_bitmaps(D, L, S):
    """a wide cluster whose distinct sequences carry more than 262 144 windows (two LDS ordinal bitmaps' worth) and at
    most 1 048 576: ranks from ONE LDS bitmap walked over the ordinal space in stretches of 524 288, used twice per stretch
    (round 3), instead of sorted pairs and a search in every sibling item; several key partitions; against the oracle"""
    rng = np.random.default_rng(D + L)
    alleles = [np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].tobytes()]
    seen = set(alleles)
    while len(alleles) < D:
        a = _mut(alleles[int(rng.integers(0, len(alleles)))], [int(x) for x in rng.integers(0, L, 3)])
        if a not in seen:
            seen.add(a)
            alleles.append(a)
    assert 262144 < D * (L - 30) and (D * (L - 30) <= 1048576) == (D < 640)      # the last case: beyond the bitmaps, sorted pairs
    names = [f"b{i:04d}" for i in range(S)]
    tms = _run_all([_cluster("g_big", names, alleles)], 31, S)
    assert tms[0]["n_wide_clusters"] == 1 and tms[0]["n_dedup_clusters"] == 1
