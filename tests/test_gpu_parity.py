"""Parity of the HIP path (through the C ABI) with (a) outputs of the reference itself
(tests/golden/) and (b) the CPU oracle on seeded inputs.  Bit-exact: all of this is integer /
byte work (k-mer keys, presence bits, MD5 digests, text)."""
import io

import numpy as np
import pytest

from conftest import all_cases, case_ids, case_records

pytestmark = pytest.mark.gpu

CASES = all_cases()


def _engine(case_opts, max_strains, **kw):
    from panfeed_amd.engine import Engine
    o = case_opts
    return Engine(klength=o["klength"], canon=o["canon"], consider_missing=o["consider_missing"],
                  patfilt=o["patfilt"], maf=o["maf"], multiple_files=o["multiple_files"],
                  max_strains=max_strains, stroi=set(o["stroi"]) if o["stroi"] else (), **kw)


def _headers(case):
    from panfeed_amd.engine import KMERS_TSV_HEADER, KMERS_TO_HASHES_HEADER, hashes_to_patterns_header
    return KMERS_TSV_HEADER, KMERS_TO_HASHES_HEADER, hashes_to_patterns_header(case["all_strains"])


@pytest.mark.parametrize("dedup", [True, False], ids=["dedup", "nodedup"])
@pytest.mark.parametrize("case", CASES, ids=case_ids(CASES))
def test_golden_engine(case, dedup):
    """Engine.run over the whole case as ONE batch vs the reference's files (with and without the
    identical-segment shortcut: both must give the reference's bytes)."""
    from panfeed_amd._lib import PanfeedHipError
    o = case["opts"]
    ms = max(32, (len(case["all_strains"]) + 31) // 32 * 32)
    assert o["klength"] <= 126              # rand12_k64 runs with three key words
    eng = _engine(o, ms, dedup=dedup)
    out = eng.run(case_records(case))
    hk, hkh, hhp = _headers(case)
    exp = case["expect"]
    if o["multiple_files"]:
        for idx, kt, kh, hp in out.per_cluster:
            assert hkh + kh == exp["dirs"][idx]["kmers_to_hashes.tsv"], idx
            assert hhp + hp == exp["dirs"][idx]["hashes_to_patterns.tsv"], idx
            assert hk + kt == exp["dirs"][idx]["kmers.tsv"], idx
    else:
        assert hkh + out.kmers_to_hashes == exp["kmers_to_hashes.tsv"]
        assert hhp + out.hashes_to_patterns == exp["hashes_to_patterns.tsv"]
        assert hk + out.kmers_tsv == exp["kmers.tsv"]
        assert out.stats["patterns"] == exp["n_patterns"]
    eng.close()


@pytest.mark.parametrize("case", [c for c in CASES if not c["opts"]["multiple_files"]][::3],
                         ids=lambda c: c["name"])
def test_golden_mirror_one_cluster_per_call(case):
    """The reference's own driver loop (__main__.py:350-356) over the mirror API, one cluster per call."""
    import pandas as pd
    from panfeed_amd.panfeed import cluster_cutter, pattern_hasher, write_headers
    from panfeed_amd.engine import KMERS_TSV_HEADER
    o = case["opts"]
    stroi = set(o["stroi"]) if o["stroi"] is not None else ""
    ks, hp, kh = io.StringIO(), io.StringIO(), io.StringIO()
    ks.write(KMERS_TSV_HEADER)
    genepres = pd.DataFrame(columns=case["all_strains"])
    write_headers(hp, kh, genepres)
    patterns = None
    for x in case_records(case):
        ret = cluster_cutter(x, o["klength"], stroi, False, o["canon"], o["consider_missing"], "unused")
        patterns = pattern_hasher((ret,), ks, hp, kh, genepres, o["patfilt"], o["maf"], "unused",
                                  patterns=patterns, consider_missing_cluster=o["consider_missing"])
    exp = case["expect"]
    assert kh.getvalue() == exp["kmers_to_hashes.tsv"]
    assert hp.getvalue() == exp["hashes_to_patterns.tsv"]
    assert ks.getvalue() == exp["kmers.tsv"]
    assert len(patterns) == exp["n_patterns"]


def _oracle_texts(records, stroi=(), **kw):
    from oracle import oracle as po
    run = po.OracleRun(stroi=stroi, **kw)
    run.feed(records)
    return run.texts(), run.stats()


@pytest.mark.parametrize("dedup", [True, False], ids=["dedup", "nodedup"])
@pytest.mark.parametrize("k,canon,S,flank", [(31, True, 200, 0), (31, False, 96, 10), (51, True, 130, 0), (21, True, 333, 25),
                                             (64, True, 70, 0), (77, False, 70, 20), (94, True, 90, 10), (95, True, 90, 10),
                                             (126, False, 70, 40)])
def test_seeded_vs_oracle(k, canon, S, flank, dedup):
    """mid-size seeded clusters (multi-word rows, several sample chunks, paralogs, Ns); k up to 126 = one to four
    63-bit key words (the reference slices any length, panfeed.py:65-67)"""
    from panfeed_amd import synth
    from panfeed_amd.engine import Engine
    cl = synth.generate(10, S, first=1234, flank=flank, mean_len=400, min_len=80, max_len=1500, n_rate=0.01,
                        paralog_rate=0.03, shuffle_columns=5)
    recs = [c.record() for c in cl]
    stroi = {cl[0].names[3], cl[0].names[S // 2]}
    eng = Engine(klength=k, canon=canon, max_strains=(S + 31) // 32 * 32, stroi=stroi, dedup=dedup)
    out = eng.run(recs)
    if dedup:
        assert out.timing["n_dedup_clusters"] > 0
    (ek, ekh, ehp), st = _oracle_texts(recs, stroi=stroi, klength=k, canon=canon)
    assert out.kmers_to_hashes == ekh
    assert out.hashes_to_patterns == ehp
    assert out.kmers_tsv == ek
    assert out.stats["unique_kmers"] == st["unique_kmers"]
    assert out.stats["instances"] == st["instances"]
    eng.close()


def test_dedup_single_pass_paths():
    """cluster_dedup_kernel's one-pass grouping: segments longer than its registers hold, more distinct bytes than
    its LDS pool holds (compare against the first copy's global words), sequences that differ only in the last
    base or only by trailing A's (identical packed words, different lengths)"""
    from panfeed_amd.classes import Seqinfo
    from panfeed_amd.engine import Engine
    rng = np.random.default_rng(11)
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    names = [f"q{i:03d}" for i in range(90)]

    def mk(seq, i, j=0):
        return Seqinfo(seq.decode(), seq.translate(comp).decode(), f"g{i}_{j}", "ctg", 5, 5 + len(seq) - 1, 1, 0)

    def rnd(L):
        return np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].tobytes()
    ones = np.ones(90, dtype=np.int64)
    recs = []
    a = [rnd(5500), rnd(5500), rnd(4100)]
    a.append(a[0][:-1] + (b"C" if a[0][-1:] != b"C" else b"G"))         # differs in the very last base
    recs.append(({nm: [mk(a[i % 4], i)] for i, nm in enumerate(names)}, "long", ones))
    b = [rnd(5000) for _ in range(40)]                                   # 40 x 1250 B > the 48 KiB pool
    recs.append(({nm: [mk(b[i % 40], i)] for i, nm in enumerate(names)}, "poolover", ones))
    x = rnd(250)[:-1] + b"C"
    c = [x, x + b"A", x + b"AA", x + b"A" * 70, x[:-1] + b"A"]          # same words, other lengths
    recs.append(({nm: [mk(c[i % 5], i)] for i, nm in enumerate(names)}, "trailingA", ones))
    anc = np.frombuffer(rnd(1500), np.uint8)
    d = []
    for _ in range(30):                                                  # far more distinct allele masks than the
        m = anc.copy()                                                   # fused kernel's mask table holds
        hit = rng.random(len(m)) < 0.02
        m[hit] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(hit.sum()))]
        d.append(m.tobytes())
    recs.append(({nm: [mk(d[i % 30], i)] for i, nm in enumerate(names)}, "manymasks", ones))
    eng = Engine(klength=31, max_strains=96)
    out = eng.run(recs)
    assert out.timing["n_dedup_clusters"] == 4
    (ek, ekh, ehp), st = _oracle_texts(recs, klength=31)
    assert out.kmers_to_hashes == ekh
    assert out.hashes_to_patterns == ehp
    assert out.stats["unique_kmers"] == st["unique_kmers"]
    eng.close()


def test_dedup_thousands_of_segments():
    """clusters with more segments than samples fit one LDS tile of the old kernel (2 600 samples, paralogs): the
    one-pass grouping keeps one byte per segment and reads the metadata from global memory one trip ahead"""
    from panfeed_amd.classes import Seqinfo
    from panfeed_amd.engine import Engine
    rng = np.random.default_rng(21)
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    S = 2600
    names = [f"z{i:05d}" for i in range(S)]

    def mk(seq, i, j=0):
        return Seqinfo(seq.decode(), seq.translate(comp).decode(), f"g{i}_{j}", "ctg", 5, 5 + len(seq) - 1, 1, 0)

    def rnd(L):
        return np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].tobytes()
    recs = []
    for ci, (nall, L) in enumerate(((5, 90), (11, 140))):
        al = [rnd(L) for _ in range(nall)]
        al[1] = al[0][:L // 2] + (b"C" if al[0][L // 2:L // 2 + 1] != b"C" else b"G") + al[0][L // 2 + 1:]   # near copy
        pres = np.ones(S, dtype=np.int64)
        pres[ci::17] = 0
        gs = {}
        for i, nm in enumerate(names):
            if pres[i]:
                gs[nm] = [mk(al[(i * 7 + ci) % nall], i)] + ([mk(al[(i + 1) % nall], i, 1)] if i % 9 == 0 else [])
            else:
                gs[nm] = []
        recs.append((gs, f"wide{ci}", pres))
    eng = Engine(klength=15, max_strains=2624)
    out = eng.run(recs)
    assert out.timing["n_dedup_clusters"] == 2
    (ek, ekh, ehp), st = _oracle_texts(recs, klength=15)
    assert out.kmers_to_hashes == ekh
    assert out.hashes_to_patterns == ehp
    eng.close()


def test_large_batch_two_part_first_pass():
    """>= 8192 clusters: the first pass is launched in two parts (a quarter of the clusters while the GPU is still on
    the dedup of the rest); same texts as the oracle, overflow retries from both parts included"""
    from panfeed_amd import synth
    from panfeed_amd.engine import Engine
    cl = synth.generate(9000, 12, first=40000, flank=5, mean_len=70, min_len=40, max_len=160, n_rate=0.01, paralog_rate=0.05)
    recs = [c.record() for c in cl]
    # two clusters whose tables overflow, one in each part
    for pos, seed in ((100, 1), (8000, 2)):
        recs[pos] = _diverse_records(24, 900, seed, n_clusters=1)[0][0]
    (ek, ekh, ehp), st = _oracle_texts(recs, klength=21)
    # max_items=4096: several sub-batches per part as well -- and plan_kernel (the simple clusters' work items laid out
    # on the device) takes the first 4096 of a part, the host the rest; with it off the host builds every item
    for device_plan in (True, False):
        eng = Engine(klength=21, max_strains=32, max_items=4096, device_plan=device_plan)
        out = eng.run(recs)
        assert out.timing["n_retried"] >= 2 and out.timing["n_dedup_clusters"] > 4000
        assert (out.timing["n_device_planned"] >= 4096) == device_plan
        assert out.kmers_to_hashes == ekh
        assert out.hashes_to_patterns == ehp
        assert out.stats["unique_kmers"] == st["unique_kmers"]
        eng.close()


@pytest.mark.parametrize("k,canon,missing", [(31, True, False), (51, False, False), (21, True, True)])
def test_device_planned_equals_host_planned(k, canon, missing):
    """plan_kernel against the host's item builder on a batch that has a bit of everything -- simple clusters of all three
    fused classes, clusters of several key partitions, slow-path rows ('N's), paralogs, a target strain: the same files as
    the oracle either way, two submits on one context (the learned key-partition line comes into play in the second)"""
    from panfeed_amd import synth
    from panfeed_amd.engine import Engine
    S = 70
    cl = synth.generate(260, S, first=9100, flank=20, mean_len=500, min_len=60, max_len=2500, n_rate=0.01, paralog_rate=0.05)
    cl += synth.generate(12, S, first=9500, flank=20, mean_len=900, min_len=600, max_len=1500, n_rate=0.0, paralog_rate=0.02,
                         mean_alleles=30.0, allele_decay=1.0)
    recs = [c.record() for c in cl]
    names = sorted(recs[0][0].keys())
    stroi = {names[4]}
    (ek, ekh, ehp), st = _oracle_texts(recs, stroi=stroi, klength=k, canon=canon, consider_missing=missing)
    planned = {}
    for device_plan in (True, False):
        eng = Engine(klength=k, canon=canon, consider_missing=missing, max_strains=96, stroi=stroi, device_plan=device_plan)
        outs = [eng.run(recs[:120]), eng.run(recs[120:])]
        planned[device_plan] = sum(o.timing["n_device_planned"] for o in outs)
        assert "".join(o.kmers_to_hashes for o in outs) == ekh
        assert "".join(o.hashes_to_patterns for o in outs) == ehp
        assert "".join(o.kmers_tsv for o in outs) == ek
        eng.close()
    assert planned[False] == 0 and planned[True] > 100


def _diverse_records(n_samples, length, seed, n_clusters=2):
    """every sample carries its own random sequence: unique k-mers ~ instances (table overflow path)"""
    from panfeed_amd.classes import Seqinfo
    rng = np.random.default_rng(seed)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    names = [f"d{i:03d}" for i in range(n_samples)]
    recs = []
    for c in range(n_clusters):
        gs = {}
        for i, nm in enumerate(names):
            s = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, length)].tobytes()
            gs[nm] = [Seqinfo(s.decode(), s.translate(comp).decode(), f"g{c}_{i}", "ctg", 10, 10 + length - 1, 1, 0)]
        recs.append((gs, f"div{c}", np.ones(n_samples, dtype=np.int64)))
    return recs, names


@pytest.mark.parametrize("k,canon", [(31, True), (31, False), (47, True)])
def test_table_overflow_repartitions(k, canon):
    """one LDS table cannot hold these clusters: they are re-run with 4x / 16x key partitions and the
    merged order must still be the reference's dict insertion order"""
    from panfeed_amd.engine import Engine
    recs, names = _diverse_records(48, 1500, seed=99)
    eng = Engine(klength=k, canon=canon, max_strains=64, maf=0.0)
    out = eng.run(recs)
    assert out.timing["n_retried"] >= 2
    (ek, ekh, ehp), st = _oracle_texts(recs, klength=k, canon=canon, maf=0.0)
    assert out.kmers_to_hashes == ekh
    assert out.hashes_to_patterns == ehp
    assert out.stats["unique_kmers"] == st["unique_kmers"]
    eng.close()


def _mix64(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(30); x *= np.uint64(0xbf58476d1ce4e5b9); x ^= x >> np.uint64(27)
    x *= np.uint64(0x94d049bb133111eb); x ^= x >> np.uint64(31)
    return x


@pytest.mark.parametrize("k", [31, 51])
def test_result_checksum_is_the_checksum_of_the_fetched_rows(k):
    """pf_result_checksum (the full-size parity property of bench.py) recomputed in numpy from pf_fetch: every
    (cluster, position, key words, digest) and every cluster row enters it; the identical-sequence shortcut does not
    change it"""
    import ctypes as C
    from panfeed_amd import synth
    from panfeed_amd.engine import Engine
    cl = synth.generate(40, 130, first=5, flank=30, mean_len=300, min_len=60, max_len=900, n_rate=0.02, paralog_rate=0.05)
    recs = [c.record() for c in cl]
    sums = {}
    for dedup in (True, False):
        eng = Engine(klength=k, max_strains=160, dedup=dedup)
        eng.run(recs)
        got = eng.result_checksum()
        f = eng.fetch()
        n = len(recs)
        KW = 1 if k <= 31 else 2
        off = np.ctypeslib.as_array(f.cluster_kmer_off, shape=(n,)).astype(np.int64)
        cnt = np.ctypeslib.as_array(f.cluster_kmer_cnt, shape=(n,)).astype(np.int64)
        uniq = np.ctypeslib.as_array(f.cluster_unique, shape=(n,)).astype(np.uint64)
        cpat = np.ctypeslib.as_array(f.cluster_pattern, shape=(n,)).copy()
        tot = int(cnt.sum())
        keys = np.ctypeslib.as_array(f.kmer_key, shape=(tot * KW,)).reshape(tot, KW)
        pids = np.ctypeslib.as_array(f.kmer_pattern, shape=(tot,))
        P = int(f.n_patterns)
        md5 = np.ctypeslib.as_array(f.pattern_md5, shape=(P * 16,)).copy().view(np.uint64).reshape(P, 2)
        with np.errstate(over="ignore"):
            a0 = np.uint64(0)
            a1 = np.uint64(0)
            for c in range(n):
                j = np.arange(cnt[c], dtype=np.uint64)
                h = _mix64((np.uint64(c) << np.uint64(32)) ^ j ^ np.uint64(0x9E3779B97F4A7C15))
                rows = slice(int(off[c]), int(off[c] + cnt[c]))
                for w in range(KW):
                    h = _mix64(h ^ keys[rows, w])
                d = md5[pids[rows]]
                a0 += _mix64(_mix64(h ^ d[:, 0]) + d[:, 1]).sum(dtype=np.uint64)
                hc = _mix64(np.array([(np.uint64(c) << np.uint64(32)) ^ np.uint64(cnt[c])], np.uint64)) ^ \
                    _mix64(np.array([(uniq[c] << np.uint64(32)) | np.uint64(0x5bd1e995)], np.uint64))
                if cpat[c] != 0xFFFFFFFF:
                    hc = _mix64(_mix64(hc ^ md5[cpat[c], 0]) + md5[cpat[c], 1])
                a1 += _mix64(hc)[0]
        assert got == (int(a0), int(a1), tot)
        sums[dedup] = got
        eng.close()
    assert sums[True] == sums[False] and sums[True][2] > 0


def test_sub_batches_and_pattern_carry_over():
    """max_items=3 forces many internal sub-batches; two run() calls share the run-global patterns"""
    from panfeed_amd import synth
    from panfeed_amd.engine import Engine
    cl = synth.generate(14, 24, first=500, mean_len=150, min_len=40, max_len=400, n_rate=0.02)
    recs = [c.record() for c in cl]
    eng = Engine(klength=15, max_strains=32, max_items=3)
    o1 = eng.run(recs[:9])
    o2 = eng.run(recs[9:])
    (ek, ekh, ehp), st = _oracle_texts(recs, klength=15)
    assert o1.kmers_to_hashes + o2.kmers_to_hashes == ekh
    assert o1.hashes_to_patterns + o2.hashes_to_patterns == ehp
    assert o2.stats["patterns"] == st["patterns"]
    eng.close()


def test_device_resident_batch_matches_host_batch():
    """pf_submit on device pointers (the bench path) == on host pointers; synth expansion == host packing"""
    from panfeed_amd import devbatch, synth
    from panfeed_amd.engine import Engine
    from numpy_packer import build_batch
    cl = synth.generate(9, 64, first=42, flank=15, mean_len=300, min_len=50, max_len=800, n_rate=0.0)
    recs = [c.record() for c in cl]
    eng = Engine(klength=31, max_strains=64)
    ref = eng.run(recs)
    eng2 = Engine(klength=31, max_strains=64)
    db = devbatch.from_synth(eng2, cl, 31)
    res = db.submit()
    assert int(res.n_instances) == ref.stats["device_instances"] == db.n_instances
    assert int(res.n_unique) == ref.stats["unique_kmers"]
    assert int(res.n_kept) == ref.stats["kept_kmers"]
    assert int(res.n_new_patterns) == ref.stats["new_patterns"]
    hb = build_batch(recs, 31, True, eng2.W)
    got = eng2._render(hb, eng2.fetch())
    assert got.kmers_to_hashes == ref.kmers_to_hashes
    assert got.hashes_to_patterns == ref.hashes_to_patterns
    # the expanded device buffer is byte-identical to the host packer's
    packed = db.download("packed", np.uint64, len(hb.packed))
    assert np.array_equal(packed, hb.packed)
    db.free()
    eng.close()
    eng2.close()


@pytest.mark.parametrize("S,flank,n_rate,missing", [(64, 15, 0.05, False), (300, 40, 0.01, False), (300, 40, 0.01, True),
                                                    (1000, 100, 0.002, False)])
def test_device_resident_batch_with_N(S, flank, n_rate, missing):
    """synthetic batches built on the device from allele pools, some sequences with an 'N' (split into their A/C/G/T
    runs, the windows around the 'N' as slow-path rows): the texts are the oracle's.  The fused finish kernel takes
    the slow-path rows of its clusters along."""
    from panfeed_amd import devbatch, synth
    from panfeed_amd.engine import Engine
    from numpy_packer import build_batch
    cl = synth.generate(14, S, first=77, flank=flank, mean_len=350, min_len=60, max_len=900, n_rate=n_rate, paralog_rate=0.02)
    assert sum(int((c.seq_npos >= 0).sum()) for c in cl) > 3
    recs = [c.record() for c in cl]
    eng = Engine(klength=31, max_strains=(S + 31) // 32 * 32, consider_missing=missing)
    db = devbatch.from_synth(eng, cl, 31)
    assert db.n_extra > 0
    res = db.submit()
    tm = eng.timing()
    assert tm["finish_ms"] > 0                       # clusters with slow-path rows stay on the fused path
    hb = build_batch(recs, 31, True, eng.W)
    assert int(res.n_instances) + 0 <= hb.n_instances
    got = eng._render(hb, eng.fetch())
    (ek, ekh, ehp), st = _oracle_texts(recs, klength=31, consider_missing=missing)
    assert got.kmers_to_hashes == ekh
    assert got.hashes_to_patterns == ehp
    assert got.stats["unique_kmers"] == st["unique_kmers"]
    db.free()
    eng.close()


@pytest.mark.parametrize("flags", [dict(), dict(consider_missing=True), dict(patfilt=False, maf=0.0), dict(canon=False)])
def test_dedup_big_alleles_repartition(flags):
    """few distinct but long alleles x many samples: mode 1 with a table overflow (key partitions) and,
    in a second cluster, more distinct sequences than the dedup path takes (falls back to mode 0)"""
    from panfeed_amd.classes import Seqinfo
    from panfeed_amd.engine import Engine
    rng = np.random.default_rng(5)
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    names = [f"q{i:03d}" for i in range(90)]

    def mk(seq, i, j):
        return Seqinfo(seq.decode(), seq.translate(comp).decode(), f"g{i}_{j}", "ctg", 5, 5 + len(seq) - 1, -1 if i % 3 else 1, 2)

    def alleles(n, L):
        return [np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].tobytes() for _ in range(n)]
    recs = []
    a1 = alleles(5, 4000)                      # 5 x 4000 distinct windows > one table
    a1[2] = a1[2][:1500] + b"N" + a1[2][1501:]  # a slow-path row inside a deduplicated cluster
    gs = {nm: [mk(a1[(i * 7) % 5], i, 0)] + ([mk(a1[0], i, 1)] if i % 11 == 0 else []) for i, nm in enumerate(names)}
    recs.append((gs, "big5", np.ones(90, dtype=np.int64)))
    a2 = alleles(70, 120)                      # 70 distinct sequences > DEDUP_MAX_D
    pres = np.ones(90, dtype=np.int64)
    pres[::9] = 0
    gs = {nm: ([mk(a2[i % 70], i, 0)] if pres[sorted(names).index(nm)] else []) for i, nm in enumerate(names)}
    recs.append((gs, "many", pres))
    a3 = alleles(3, 200)
    gs = {nm: [mk(a3[i % 3], i, 0)] for i, nm in enumerate(names)}
    recs.append((gs, "small3", np.ones(90, dtype=np.int64)))
    a4 = alleles(6, 3500)                      # pure ACGT, several key partitions, finished by the fused kernel
    gs = {nm: [mk(a4[(i * 5) % 6], i, 0)] for i, nm in enumerate(names)}
    recs.append((gs, "big6", np.ones(90, dtype=np.int64)))
    a5 = alleles(4, 1800)                      # single partition, large fused class (dense ordinals > 32k? no: 7k) 
    gs = {nm: [mk(a5[i % 4], i, 0), mk(a5[(i + 1) % 4], i, 1)] for i, nm in enumerate(names)}
    recs.append((gs, "para4", np.ones(90, dtype=np.int64)))
    kw = dict(klength=31, canon=True, consider_missing=False, patfilt=True, maf=0.01)
    kw.update(flags)
    eng = Engine(max_strains=96, stroi={names[4]}, **kw)
    out = eng.run(recs)
    # (all five: since round 3 the view of distinct sequences is taken whenever it fits -- "many", 70 distinct sequences
    # among 80, goes through the wide class instead of staying on the every-copy path)
    assert out.timing["n_dedup_clusters"] == 5 and out.timing["n_retried"] >= 1
    (ek, ekh, ehp), st = _oracle_texts(recs, stroi={names[4]}, **kw)
    assert out.kmers_to_hashes == ekh
    assert out.hashes_to_patterns == ehp
    assert out.kmers_tsv == ek
    eng.close()


def test_pattern_export_device_path_and_merge():
    """the tensors the RCCL all-gather starts from: device-to-device export == host export, and the
    single-process merge keeps every pattern exactly once"""
    import torch
    from panfeed_amd import synth
    from panfeed_amd.distributed import export_patterns, merge_pattern_tensors
    from panfeed_amd.engine import Engine
    cl = synth.generate(6, 48, first=77, mean_len=200, min_len=60, max_len=500, n_rate=0.0)
    eng = Engine(klength=21, max_strains=64)
    out = eng.run([c.record() for c in cl])
    md5_d, fs_d = export_patterns(eng, torch.device("cuda", 0))
    md5_h, fs_h = export_patterns(eng, torch.device("cpu"))
    assert md5_d.shape[0] == out.stats["patterns"] == md5_h.shape[0]
    assert torch.equal(md5_d.cpu(), md5_h) and torch.equal(fs_d.cpu(), fs_h)
    keep, n_global = merge_pattern_tensors(md5_d, fs_d)
    assert n_global == out.stats["patterns"] and bool(keep.all())
    # first_seen order == order of hashes_to_patterns.tsv
    order = torch.argsort(fs_h).tolist()
    import ctypes as C
    buf = C.create_string_buffer(24)
    hashes = []
    for i in order:
        eng.L.pf_b64_digest(md5_h[i].numpy().tobytes(), buf)
        hashes.append(buf.raw[:24].decode())
    assert hashes == [ln.split("\t")[0] for ln in out.hashes_to_patterns.splitlines()]
    eng.close()


def test_run_stream_equals_one_batch():
    """the pipelined driver (pack ahead on a host thread, GPU, render) == one big batch, any batch size"""
    from panfeed_amd import synth
    from panfeed_amd.engine import Engine
    cl = synth.generate(23, 40, first=900, flank=10, mean_len=150, min_len=40, max_len=400, n_rate=0.03, paralog_rate=0.05)
    recs = [c.record() for c in cl]
    st = {cl[0].names[7]}
    ref = Engine(klength=25, max_strains=64, stroi=st).run(recs)
    for bc in (1, 5, 64):
        eng = Engine(klength=25, max_strains=64, stroi=st)
        outs = list(eng.run_stream(iter(recs), batch_clusters=bc))
        assert "".join(o.kmers_to_hashes for o in outs) == ref.kmers_to_hashes
        assert "".join(o.hashes_to_patterns for o in outs) == ref.hashes_to_patterns
        assert "".join(o.kmers_tsv for o in outs) == ref.kmers_tsv
        assert outs[-1].stats["patterns"] == ref.stats["patterns"]
        eng.close()


@pytest.mark.parametrize("S,n,flank,k,head", [(200, 1500, 0, 31, 25), (1000, 600, 100, 31, 3), (1000, 300, 100, 51, 2),
                                              (5000, 48, 100, 21, 2), (5000, 32, 100, 51, 2)],
                         ids=["cfg1_200s", "cfg2_1000s_flank", "cfg2_k51", "cfg4_5000s_k21", "cfg4_5000s_k51"])
def test_config1_scale_properties(S, n, flank, k, head):
    """BASELINE configs[1] shape (200 samples, no flanks), configs[2] shape (1 000 samples, +-100 bp: 32-word rows,
    both finish classes, key partitions; k = 51: two-word keys) and configs[4] shape (5 000 samples = 157-word rows,
    k = 21 and 51, paralogs, +-100 bp) at sizes the oracle cannot cover in seconds:
    the whole pipeline with and without the identical-sequence shortcut must agree k-mer for k-mer and digest
    for digest; totals must add up; first_seen must be a strict order; the head of the run equals the oracle."""
    import ctypes as C
    from panfeed_amd import devbatch, synth
    from panfeed_amd.engine import Engine
    ms = (S + 31) // 32 * 32
    cl = synth.generate(n, S, first=0, flank=flank, n_rate=0.0)
    res = {}
    for dedup in (True, False):
        eng = Engine(klength=k, max_strains=ms, dedup=dedup, max_items=512)
        db = devbatch.from_synth(eng, cl, k)
        r = db.submit()
        f = eng.fetch()
        C_ = n
        off = np.ctypeslib.as_array(f.cluster_kmer_off, shape=(C_,)).copy()
        cnt = np.ctypeslib.as_array(f.cluster_kmer_cnt, shape=(C_,)).copy()
        uniq = np.ctypeslib.as_array(f.cluster_unique, shape=(C_,)).copy()
        cpat = np.ctypeslib.as_array(f.cluster_pattern, shape=(C_,)).copy()
        tot = int(cnt.sum())
        KW = 1 if k <= 31 else 2
        keys = np.ctypeslib.as_array(f.kmer_key, shape=(tot * KW,)).reshape(tot, KW).copy()
        pids = np.ctypeslib.as_array(f.kmer_pattern, shape=(tot,)).copy()
        P = int(f.n_patterns)
        md5 = np.ctypeslib.as_array(f.pattern_md5, shape=(P * 16,)).reshape(P, 16).copy()
        fs = np.ctypeslib.as_array(f.pattern_first_seen, shape=(P,)).copy()
        assert int(r.n_kept) == tot and int(r.n_unique) == int(uniq.sum()) and int(r.n_instances) == db.n_instances
        assert len(np.unique(fs)) == P                       # first_seen is a strict order over patterns
        assert len(np.unique(md5.view([("a", "u8"), ("b", "u8")]))) == P      # one pool entry per digest
        # per cluster: keys in order + digest per k-mer, independent of arena placement
        order = np.concatenate([np.arange(int(o), int(o) + int(c), dtype=np.int64) for o, c in zip(off, cnt)]) if tot else np.zeros(0, np.int64)
        res[dedup] = (cnt, uniq, keys[order], md5[pids[order]], md5[cpat], np.sort(fs), np.array(eng.result_checksum(), np.uint64))
        if dedup:
            assert eng.timing()["n_dedup_clusters"] > n * 0.9
            hb_head = [c.record() for c in cl[:head]]
        db.free()
        eng.close()
    for a, b in zip(res[True], res[False]):
        assert np.array_equal(a, b)
    # the first clusters against the oracle, text for text, with target strains (kmers.tsv: the positional rows of
    # configs[4]'s --targets pass)
    names = cl[0].names
    stroi = {names[1], names[S // 2], names[S - 1]}
    eng = Engine(klength=k, max_strains=ms, stroi=stroi)
    out = eng.run(hb_head)
    (ek, ekh, ehp), st = _oracle_texts(hb_head, stroi=stroi, klength=k)
    assert out.kmers_to_hashes == ekh and out.hashes_to_patterns == ehp
    assert out.kmers_tsv == ek and len(ek) > 0
    eng.close()


def _allele_cluster(rng, idx, names, D, L, sub=0.02, flank=0, paralogs=0, n_every=0, absent=()):
    """one cluster whose present samples carry exactly D distinct sequences (allele d = ancestral + its own
    substitutions + its own flanks), round-robin so that every allele has copies"""
    from panfeed_amd.classes import Seqinfo
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    acgt = np.frombuffer(b"ACGT", np.uint8)
    anc = rng.integers(0, 4, L)
    alleles = []
    seen = set()
    for d in range(D):
        while True:
            a = anc.copy()
            if d:
                m = rng.random(L) < sub
                m[int(rng.integers(0, L))] = True           # at least one substitution
                a[m] = (a[m] + rng.integers(1, 4, int(m.sum()))) & 3
            if flank:
                a = np.concatenate([rng.integers(0, 4, flank), a, rng.integers(0, 4, flank)])
            b = acgt[a].tobytes()
            if b not in seen:
                break
        seen.add(b)
        alleles.append(b)
    assert len(set(alleles)) == D
    gs, presab = {}, np.zeros(len(names), dtype=np.int64)
    col = {x: i for i, x in enumerate(sorted(names))}
    q = 0
    order = [nm for nm in names if nm not in absent]
    for i, nm in enumerate(order):
        seqs = [alleles[i % D]]
        if paralogs and i % paralogs == 1:
            seqs.append(alleles[(i * 7 + 3) % D])
        lst = []
        for j, sq in enumerate(seqs):
            if n_every and q % n_every == 5:
                pos = int(rng.integers(0, len(sq)))
                sq = sq[:pos] + b"N" + sq[pos + 1:]
            q += 1
            lst.append(Seqinfo(sq.decode(), sq.translate(comp).decode(), f"{nm}_{idx}_{j}", f"{nm}_c", 100 + 3 * i,
                               100 + 3 * i + len(sq) - 1, 1 if i % 3 else -1, flank))
        gs[nm] = lst
        presab[col[nm]] = 1
    for nm in sorted(absent):
        gs[nm] = []
    return gs, idx, presab


@pytest.mark.parametrize("D,S,L,k,canon,kw", [
    (64, 200, 260, 31, True, {}),                                   # the last size of the small class
    (65, 200, 260, 31, True, {}),                                   # one more: the wide class
    (65, 200, 260, 31, True, {"paralogs": 9, "n_every": 40}),       # + paralogs of other alleles, slow-path rows
    (200, 600, 300, 31, True, {"paralogs": 11}),
    (150, 500, 300, 31, True, {"paralogs": 7, "n_every": 25}),      # key partitions + slow-path rows: one merged bitmap
    (60, 300, 1500, 31, True, {}),                                  # 88 000 windows over 60 sequences, several key
                                                                    # partitions: the general path with bitmaps
    (40, 300, 1800, 31, True, {"n_every": 30, "sub": 0.001}),       # 70 000+ windows, one partition: the fused kernel's
                                                                    # third class, with slow-path rows
    (25, 300, 1500, 21, False, {"sub": 0.001}),                     # the same class, both strands (twice the ordinals)
    (200, 600, 300, 51, False, {"flank": 30}),                      # two-word keys, both strands, key partitions
    (300, 1000, 150, 21, True, {"absent": 37}),                     # 32-word rows; some samples without the cluster
    (1024, 2100, 90, 15, True, {}),                                 # the largest wide cluster
    (1025, 2100, 90, 15, True, {}),                                 # one too many: every copy is scanned
    (300, 700, 1000, 31, True, {}),                                 # 291 000 windows over the distinct sequences: more than
                                                                    # the ordinal bitmaps hold -> ranks by sorting (the
                                                                    # shorter wide cases above rank from bitmaps)
], ids=["D64", "D65", "D65_paralogs_N", "D200", "D150_partitions_N", "D60_long", "D40_long_fused_N", "D25_long_noncanon", "D200_k51_noncanon", "D300_1000s", "D1024", "D1025", "D300_long_sorted"])
@pytest.mark.parametrize("missing", [False, True], ids=["", "consider_missing"])
def test_many_distinct_sequences_vs_oracle(D, S, L, k, canon, kw, missing):
    """clusters with more distinct sequences than the 64 an allele-mask pair of words holds: representatives are
    still scanned once each (wide dedup class, rows gathered through the segment list, ranks from ordinal bitmaps or,
    for the largest, by sorting) and the files are the oracle's"""
    from panfeed_amd.engine import Engine
    if missing and (D not in (65, 300) or kw.get("n_every")):
        pytest.skip("consider_missing: two shapes are enough")
    rng = np.random.default_rng(D * 7 + S)
    names = [f"w{i:04d}" for i in range(S)]
    kw = dict(kw)
    nabs = kw.pop("absent", 0)
    recs = []
    for ci in range(3):
        absent = set(names[5::max(2, S // nabs)][:nabs]) if nabs and ci != 1 else ()
        recs.append(_allele_cluster(rng, f"grp{ci}", names, D if ci < 2 else max(3, D // 9), L, absent=absent, **kw))
    stroi = {names[2], names[S - 3]}
    eng = Engine(klength=k, canon=canon, consider_missing=missing, max_strains=(S + 31) // 32 * 32, stroi=stroi)
    out = eng.run(recs)
    tm = out.timing
    third = max(3, D // 9)
    assert tm["n_wide_clusters"] == sum(1 for x in (D, D, third) if x > 64)     # clusters the wide class was tried on
    assert tm["n_dedup_clusters"] == (3 if D <= 1024 else 1)
    (ek, ekh, ehp), st = _oracle_texts(recs, stroi=stroi, klength=k, canon=canon, consider_missing=missing)
    assert out.kmers_to_hashes == ekh
    assert out.hashes_to_patterns == ehp
    assert out.kmers_tsv == ek
    eng.close()
    # the same through the no-dedup path
    eng = Engine(klength=k, canon=canon, consider_missing=missing, max_strains=(S + 31) // 32 * 32, stroi=stroi, dedup=False)
    out2 = eng.run(recs)
    assert (out2.kmers_to_hashes, out2.hashes_to_patterns) == (ekh, ehp)
    eng.close()


@pytest.mark.parametrize("mean_alleles,S", [(40, 300), (70, 500), (250, 800)], ids=["D40", "D60_fused_third_class", "D250"])
def test_alleles_descending_from_one_another_vs_oracle(mean_alleles, S):
    """many distinct sequences that share most of their k-mers (synth's "tree" alleles, the population-like case of
    bench.py's second allele sweep): long allele masks over few k-mers"""
    from panfeed_amd import synth
    from panfeed_amd.engine import Engine
    cl = synth.generate(4, S, first=11, flank=100, n_rate=0.002, mean_alleles=mean_alleles, allele_decay=1.0,
                        allele_model="tree")
    assert max(len(np.unique(c.seq_allele)) for c in cl) > (64 if mean_alleles > 64 else 20)
    recs = [c.record() for c in cl]
    stroi = {cl[0].names[1], cl[0].names[S - 2]}
    eng = Engine(klength=31, max_strains=(S + 31) // 32 * 32, stroi=stroi)
    out = eng.run(recs)
    (ek, ekh, ehp), st = _oracle_texts(recs, stroi=stroi, klength=31)
    assert out.kmers_to_hashes == ekh
    assert out.hashes_to_patterns == ehp
    assert out.kmers_tsv == ek
    eng.close()


def test_device_batch_with_a_bad_slow_path_list_is_refused():
    """the slow-path rows of a batch in device memory are checked where they are (extra_csr_kernel): a list out of order
    or naming a cluster that does not exist is PF_ERR_ARG, as for a host batch, and the context stays usable"""
    from panfeed_amd import _lib, devbatch, synth
    from panfeed_amd.engine import Engine
    import ctypes as C
    cl = synth.generate(10, 64, first=5, flank=10, mean_len=300, min_len=60, max_len=600, n_rate=0.1)
    eng = Engine(klength=31, max_strains=64)
    db = devbatch.from_synth(eng, cl, 31)
    assert db.n_extra > 3
    good = db.download("extra_cluster", np.uint32, db.n_extra)
    ref = db.submit()
    for bad in (good[::-1].copy(), np.where(np.arange(db.n_extra) == db.n_extra - 1, 10, good).astype(np.uint32)):
        if (bad == good).all():
            continue
        _lib.check(eng.L.pf_dev_upload(eng.ctx, db.ptrs["extra_cluster"], bad.ctypes.data_as(C.c_void_p), bad.nbytes))
        with pytest.raises(_lib.PanfeedHipError) as e:
            db.submit()
        assert "extra_cluster" in str(e.value)
    _lib.check(eng.L.pf_dev_upload(eng.ctx, db.ptrs["extra_cluster"], good.ctypes.data_as(C.c_void_p), good.nbytes))
    _lib.check(eng.L.pf_reset_patterns(eng.ctx))
    again = db.submit()
    assert (int(again.n_kept), int(again.n_unique), int(again.n_new_patterns)) == (int(ref.n_kept), int(ref.n_unique), int(ref.n_new_patterns))
    db.free()
    eng.close()


def test_key_partitions_learned_from_earlier_batches():
    """a context that has scanned 16 clusters of many related alleles sizes the next ones' key partitions by what those
    brought (first attempts of several partitions, few failures); the files stay the oracle's"""
    from panfeed_amd import synth
    from panfeed_amd.engine import Engine
    S = 400
    cl = synth.generate(60, S, first=500, flank=100, mean_len=300, min_len=200, max_len=500, n_rate=0.0,
                        mean_alleles=170, allele_decay=1.0, allele_model="tree")
    recs = [c.record() for c in cl]
    eng = Engine(klength=31, max_strains=(S + 31) // 32 * 32)
    out1 = eng.run(recs[:44])
    out2 = eng.run(recs[44:])
    (ek, ekh, ehp), st = _oracle_texts(recs, klength=31)
    assert out1.kmers_to_hashes + out2.kmers_to_hashes == ekh
    assert out1.hashes_to_patterns + out2.hashes_to_patterns == ehp
    assert out1.timing["n_retried"] >= 16           # the first batch had to find out
    assert out2.timing["n_retried"] <= 3 and out2.timing["n_items"] >= 24     # the second knew
    eng.close()


def test_wide_dedup_with_5000_samples():
    """configs[4] shape: at 5 000 samples the sample-set matrix only fits 25 distinct sequences: clusters with more
    go through the wide class instead of scanning all 5 000 copies"""
    from panfeed_amd.engine import Engine
    rng = np.random.default_rng(99)
    S = 5000
    names = [f"v{i:04d}" for i in range(S)]
    recs = [_allele_cluster(rng, f"big{ci}", names, D, 400, paralogs=50) for ci, D in enumerate((40, 12, 90))]
    eng = Engine(klength=21, max_strains=(S + 31) // 32 * 32)
    out = eng.run(recs)
    assert out.timing["n_dedup_clusters"] == 3 and out.timing["n_wide_clusters"] == 2
    (ek, ekh, ehp), st = _oracle_texts(recs, klength=21)
    assert out.kmers_to_hashes == ekh and out.hashes_to_patterns == ehp
    eng.close()


def test_pattern_table_grows_instead_of_failing():
    """the reference's `patterns` is an unbounded set (panfeed.py:146-150): a table that starts far too small is
    enlarged and the batch re-run -- in the middle of a run too, with the earlier batches' patterns kept -- and the
    files are the oracle's"""
    from panfeed_amd import synth
    from panfeed_amd.engine import Engine
    cl = synth.generate(150, 40, first=5, mean_len=200, min_len=60, max_len=500, n_rate=0.0)
    recs = [c.record() for c in cl]
    (ek, ekh, ehp), st = _oracle_texts(recs, klength=21)
    assert st["patterns"] > 1024
    eng = Engine(klength=21, max_strains=64, pattern_capacity=1024)      # pool of 512 patterns
    out = eng.run(recs)
    assert out.kmers_to_hashes == ekh and out.hashes_to_patterns == ehp
    assert out.stats["patterns"] == st["patterns"]
    eng.close()
    # three calls: growth ahead of need, growth after a failed attempt, ids of earlier batches still valid
    eng = Engine(klength=21, max_strains=64, pattern_capacity=1024)
    kh, hp = "", ""
    for part in (recs[:10], recs[10:60], recs[60:]):
        o = eng.run(part)
        kh += o.kmers_to_hashes
        hp += o.hashes_to_patterns
    assert kh == ekh and hp == ehp
    # device-written text after growth (base64 table re-made)
    eng.close()
    eng = Engine(klength=21, max_strains=64, pattern_capacity=1024)
    outs = list(eng.run_stream(recs, batch_clusters=40))
    assert "".join(o.kmers_to_hashes for o in outs) == ekh and "".join(o.hashes_to_patterns for o in outs) == ehp
    eng.close()
    got_kh, got_hp = b"", b""
    eng = Engine(klength=21, max_strains=64, pattern_capacity=1024)
    from panfeed_amd.packing import build_batch_native
    ordinal = 0
    for part in (recs[:10], recs[10:60], recs[60:]):
        hb = build_batch_native(part, 21, True, eng.W, first_ordinal=ordinal)
        ordinal += len(part)
        eng.submit_host_batch(hb)
        a, b = eng.render_device(hb)
        got_kh += bytes(a)
        got_hp += bytes(b)
    assert got_kh.decode() == ekh and got_hp.decode() == ehp
    eng.close()


def test_multiple_files_resets_the_pool_per_batch():
    """--multiple-files: the pattern set restarts in every cluster (panfeed.py:165), so the pool must not grow with
    the number of clusters: many batches through a small table"""
    from panfeed_amd import synth
    from panfeed_amd.engine import Engine
    cl = synth.generate(60, 40, first=77, mean_len=200, min_len=60, max_len=500, n_rate=0.01)
    recs = [c.record() for c in cl]
    from oracle import oracle as po
    eng = Engine(klength=21, max_strains=64, multiple_files=True, pattern_capacity=4096)
    for a in range(0, 60, 6):
        out = eng.run(recs[a:a + 6])
        assert eng.pattern_count() < 2048
        for (idx, kt, kh, hp), rec in zip(out.per_cluster, recs[a:a + 6]):
            run = po.OracleRun(klength=21, multiple_files=True)
            run.feed([rec])
            ek, ekh, ehp = run.texts()
            assert (kh, hp) == (ekh, ehp), idx
    eng.close()


def test_error_paths():
    """bad arguments are reported, not silently accepted"""
    from panfeed_amd import synth
    from panfeed_amd._lib import PanfeedHipError
    from panfeed_amd.engine import Engine
    cl = synth.generate(20, 40, first=5, mean_len=200, min_len=60, max_len=500, n_rate=0.0)
    recs = [c.record() for c in cl]
    with pytest.raises(PanfeedHipError):
        Engine(klength=0, max_strains=64)
    with pytest.raises(PanfeedHipError):
        Engine(klength=127, max_strains=64)                              # more than four 63-bit key words
    with pytest.raises(PanfeedHipError):
        Engine(klength=31, max_strains=9000)
    eng = Engine(klength=21, max_strains=32)
    with pytest.raises(ValueError):
        eng.run(recs)                                                    # 40 strains > max_strains 32
    with pytest.raises(PanfeedHipError):
        eng.result_checksum()                                            # nothing has been submitted yet
    eng.close()


def test_merge_kernel_matches_sort_merge():
    """pf_merge_patterns (device hash table) == the torch sort-based merge on a simulated 3-rank gather"""
    import torch
    from panfeed_amd import synth
    from panfeed_amd.distributed import _merge_on_device, export_patterns, merge_pattern_tensors
    from panfeed_amd.engine import Engine
    cl = synth.generate(40, 48, first=3, mean_len=200, min_len=60, max_len=500, n_rate=0.0)
    eng = Engine(klength=21, max_strains=64)
    eng.run([c.record() for c in cl])
    dev = torch.device("cuda", 0)
    md5, fs = export_patterns(eng, dev)
    P = md5.shape[0]
    rows = torch.empty((P, 3), dtype=torch.int64, device=dev)
    rows[:, :2] = md5.contiguous().view(torch.int64).view(P, 2)
    rows[:, 2] = fs
    g = torch.Generator(device="cpu").manual_seed(1)
    dup = torch.randperm(P, generator=g)[: P // 2].to(dev)
    rank1 = rows[dup].clone()
    rank1[:, 2] += (1 << 40)                                   # a later rank saw half of them again
    # ... and a few of them earlier than rank 0 (first_seen is unsigned on the device: leave the row with 0 alone)
    early = rows[dup[rows[dup, 2] > 0][: P // 8]].clone()
    early[:, 2] -= 1
    rank2 = torch.cat([early, torch.randint(-2**62, 2**62, (P // 3, 3), generator=g).to(dev)])
    allpay = torch.cat([rows, rank1, rank2])
    # single-process torch merge over the concatenation: which rows are the global firsts
    allmd5 = allpay[:, :2].contiguous().view(torch.uint8).view(-1, 16)
    keep_all, n_ref = merge_pattern_tensors(allmd5.cpu(), allpay[:, 2].cpu())
    for first, cnt in ((0, P), (P, rank1.shape[0]), (P + rank1.shape[0], rank2.shape[0])):
        keep, n_glob = _merge_on_device(eng, allpay, first, cnt)
        assert n_glob == n_ref
        assert torch.equal(keep.cpu(), keep_all[first:first + cnt])
    # what the owner rank of the all-to-all form computes: a mark for every row it received
    from panfeed_amd.distributed import _first_per_digest
    k_dev, n_dev = _first_per_digest(allpay, eng)
    k_cpu, n_cpu = _first_per_digest(allpay.cpu())
    assert n_dev == n_cpu == n_ref and torch.equal(k_dev.cpu(), k_cpu) and torch.equal(k_cpu, keep_all)
    # the padded layout all_gather_into_tensor leaves (equal slots per rank, garbage behind the real rows)
    import ctypes as C
    from panfeed_amd import _lib
    parts = [rows, rank1, rank2]
    counts = torch.tensor([x.shape[0] for x in parts], dtype=torch.int64, device=dev)
    nmax = int(counts.max())
    padded = torch.randint(-2**62, 2**62, (3 * nmax, 3), generator=g).to(dev)
    for r, x in enumerate(parts):
        padded[r * nmax:r * nmax + x.shape[0]] = x
    first = 0
    for r, x in enumerate(parts):
        keep = torch.zeros(x.shape[0], dtype=torch.uint8, device=dev)
        ng = C.c_uint64()
        torch.cuda.synchronize()
        _lib.check(eng.L.pf_merge_patterns_padded(eng.ctx, C.c_void_p(padded.data_ptr()), 3, nmax,
                                                  C.c_void_p(counts.data_ptr()), r, x.shape[0],
                                                  C.c_void_p(keep.data_ptr()), C.byref(ng)))
        assert ng.value == n_ref
        assert torch.equal(keep.bool().cpu(), keep_all[first:first + x.shape[0]])
        first += x.shape[0]
    eng.close()


@pytest.mark.parametrize("kw", [dict(klength=31), dict(klength=47, canon=False), dict(klength=21, consider_missing=True),
                                dict(klength=15, patfilt=False, maf=0.0)])
def test_device_rendered_text_equals_host_rendered(kw):
    """row N2 on the device: kh_text_kernel / hp_text_kernel write the same bytes as the host renderers (slow-path
    rows with 'N' k-mers, NaN cells, two-word keys, several calls with the pattern pool growing)"""
    from panfeed_amd import synth
    from panfeed_amd.engine import Engine
    from panfeed_amd.packing import build_batch_native
    cm = kw.get("consider_missing", False)
    cl = synth.generate(30, 70, first=500, flank=10, mean_len=260, min_len=50, max_len=900, n_rate=0.05, paralog_rate=0.05)
    recs = [c.record() for c in cl]
    eng = Engine(max_strains=96, **kw)
    for lo in (0, 11, 19):
        part = recs[lo:lo + 11]
        hb = build_batch_native(part, eng.k, eng.canon, eng.W, first_ordinal=lo)
        eng.submit_host_batch(hb)
        kh, hp = eng.render_device(hb)
        kh, hp = bytes(kh), bytes(hp)
        out = eng._render(hb, eng.fetch())
        assert kh.decode() == out.kmers_to_hashes
        assert hp.decode() == out.hashes_to_patterns
        assert (b"N" in kh) or not hb.extra_keys
        if cm:
            assert b"\t\t" in hp or b"\t\n" in hp
    eng.close()


def test_pattern_table_growth_that_fails():
    """the failure branches of the growing pattern table (pf_debug_limit_pattern_slots stands in for a full device):
    growing AHEAD of need fails -> the run carries on with the table it has and does not try again at that size; a batch
    that really runs out of ids and cannot grow fails with PF_ERR_OOM, the context then refuses further batches until
    pf_reset_patterns; with the limit lifted the same context gives the oracle's files again"""
    from panfeed_amd import _lib, synth
    from panfeed_amd.engine import Engine
    cl = synth.generate(150, 40, first=5, mean_len=200, min_len=60, max_len=500, n_rate=0.0)
    recs = [c.record() for c in cl]
    (ek, ekh, ehp), st = _oracle_texts(recs, klength=21)
    assert st["patterns"] > 1024
    eng = Engine(klength=21, max_strains=64, pattern_capacity=1024)      # pool of 512 patterns
    _lib.check(eng.L.pf_debug_limit_pattern_slots(eng.ctx, 1024))
    parts = (recs[:6], recs[6:12], recs[12:18], recs[18:])
    kh = hp = ""
    failed_at = None
    for i, part in enumerate(parts):
        try:
            o = eng.run(part)
        except _lib.PanfeedHipError as e:
            assert e.status == _lib.ERR_OOM and "limit" in str(e)
            failed_at = i
            break
        kh += o.kmers_to_hashes
        hp += o.hashes_to_patterns
    assert failed_at is not None and failed_at >= 1       # the first batches fit; growth ahead of need failed silently
    assert ekh.startswith(kh) and ehp.startswith(hp)      # what was written before the failure is the oracle's
    with pytest.raises(_lib.PanfeedHipError, match="pf_reset_patterns"):
        eng.run(parts[failed_at])
    _lib.check(eng.L.pf_debug_limit_pattern_slots(eng.ctx, 0))
    _lib.check(eng.L.pf_reset_patterns(eng.ctx))
    eng.next_ordinal = 0
    outs = [eng.run(part) for part in parts]
    assert "".join(o.kmers_to_hashes for o in outs) == ekh and "".join(o.hashes_to_patterns for o in outs) == ehp
    eng.close()


def test_buffer_slack_is_not_required():
    """a growable device buffer asks for its size plus slack; when that does not fit it must fall back to the exact size
    instead of failing (round 4's allele sweep ran out of HBM on the slack of a 123 GB scratch).  pf_debug_limit_alloc
    stands in for a nearly full device: the limit is the largest buffer a first run asked for, so that buffer's slack is
    refused and its exact size is not."""
    import ctypes as C
    from panfeed_amd import _lib, synth
    from panfeed_amd.engine import Engine
    cl = synth.generate(40, 60, first=11, flank=20, mean_len=400, min_len=60, max_len=900, n_rate=0.01, paralog_rate=0.03)
    recs = [c.record() for c in cl]
    (ek, ekh, ehp), st = _oracle_texts(recs, klength=31)
    L = _lib.load()
    stats = (C.c_uint64 * 2)()
    _lib.check(L.pf_debug_limit_alloc(0, stats))                  # clears the counters
    eng = Engine(klength=31, max_strains=64)
    o = eng.run(recs)
    assert o.kmers_to_hashes == ekh and o.hashes_to_patterns == ehp
    eng.close()
    _lib.check(L.pf_debug_limit_alloc(0, stats))
    largest = int(stats[0])
    assert largest > 0 and int(stats[1]) == 0
    try:
        _lib.check(L.pf_debug_limit_alloc(largest, None))
        eng = Engine(klength=31, max_strains=64)
        o = eng.run(recs)
        assert o.kmers_to_hashes == ekh and o.hashes_to_patterns == ehp
        # a buffer that is re-made (a second, larger batch) takes the same way
        o2 = eng.run(recs + recs)
        eng.close()
        _lib.check(L.pf_debug_limit_alloc(largest - 1, stats))
        assert int(stats[1]) >= 1, "no allocation took the exact-size retry"
        # below the largest buffer's own size the run has to fail as out of memory, loudly
        with pytest.raises(_lib.PanfeedHipError) as ei:
            eng = Engine(klength=31, max_strains=64)
            eng.run(recs)
        assert ei.value.status == _lib.ERR_OOM
    finally:
        _lib.check(L.pf_debug_limit_alloc(0, None))


def _device_kmers_tsv(eng, records, chunk=None):
    """submit `records` as one batch and have the GPU write the kmers.tsv rows (pf_render_kmers_tsv_device)"""
    from panfeed_amd.packing import build_batch_native
    hb = build_batch_native(records, eng.k, eng.canon, eng.W, stroi=eng.stroi, first_ordinal=eng.next_ordinal)
    eng.next_ordinal += len(records)
    eng.submit_host_batch(hb)
    dt = eng.render_targets_device(hb)
    if chunk is None:
        return bytes(dt).decode(), hb
    parts = [bytes(b) for b in dt.chunks(chunk)]          # (a block is only valid until the next one is asked for)
    assert all(len(p) <= chunk for p in parts) and sum(map(len, parts)) == len(dt)
    return b"".join(parts).decode(), hb


@pytest.mark.parametrize("case", [c for c in CASES if c["opts"]["stroi"] and not c["opts"]["multiple_files"]],
                         ids=lambda c: c["name"])
def test_golden_kmers_tsv_written_on_the_device(case):
    """H6 (panfeed.py:90-107): the positional rows written by kt_text_kernel -- pure-ACGT target sequences on the GPU,
    sequences with another letter by the host renderer, spliced in place -- against the reference's kmers.tsv"""
    o = case["opts"]
    ms = max(32, (len(case["all_strains"]) + 31) // 32 * 32)
    eng = _engine(o, ms)
    text, hb = _device_kmers_tsv(eng, case_records(case))
    hk, _, _ = _headers(case)
    assert hk + text == case["expect"]["kmers.tsv"]
    eng.close()


@pytest.mark.parametrize("k,canon,n_rate", [(31, True, 0.0), (31, True, 0.05), (21, False, 0.02), (65, True, 0.01), (126, False, 0.0)])
def test_kmers_tsv_device_equals_host_renderer(k, canon, n_rate):
    """every strain a target (BASELINE configs[4]'s second pass in small): both strands, paralogs, flanks that make
    gene_start negative, 'N's (host-rendered sequences between device-rendered ones), two rows per window in
    non-canonical mode, blocks handed out in small pieces; device text == host renderer's text == the oracle's"""
    from panfeed_amd import synth
    from panfeed_amd.engine import Engine
    S = 70
    cl = synth.generate(6, S, first=77, flank=40, mean_len=300, min_len=40, max_len=900, n_rate=n_rate,
                        paralog_rate=0.05, shuffle_columns=3)
    recs = [c.record() for c in cl]
    stroi = set(cl[0].names)
    eng = Engine(klength=k, canon=canon, max_strains=96, stroi=stroi)
    text, hb = _device_kmers_tsv(eng, recs, chunk=100_000)
    host = eng._render_targets(hb, hb.targets)
    assert text == host
    (ek, _, _), _ = _oracle_texts(recs, stroi=stroi, klength=k, canon=canon)
    assert text == ek
    # through the batched driver (run_batches renders target rows on the device)
    eng2 = Engine(klength=k, canon=canon, max_strains=96, stroi=stroi)
    outs = list(eng2.run_stream(recs, batch_clusters=4, device_text=True))
    assert b"".join(bytes(o.kmers_tsv) if not isinstance(o.kmers_tsv, str) else o.kmers_tsv.encode() for o in outs).decode() == ek
    eng.close(); eng2.close()


def test_kmers_tsv_device_long_names_and_large_coordinates():
    """rows too long for the kernel's tile (a 300-character contig name) fall to the host renderer; 13-digit coordinates
    and negative ones go through the device's integer formatting"""
    from panfeed_amd.classes import Seqinfo
    from panfeed_amd.engine import Engine
    rng = np.random.default_rng(5)
    comp = bytes.maketrans(b"ACGT", b"TGCA")

    def seq(n):
        s = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n).tobytes().decode()
        return s, s.encode().translate(comp).decode()
    gs = {}
    s1, c1 = seq(120); s2, c2 = seq(90); s3, c3 = seq(150); s4, c4 = seq(64)
    gs["strainA"] = [Seqinfo(s1, c1, "geneA", "contig" + "x" * 300, 10, 130, 1, 0)]
    gs["strainB"] = [Seqinfo(s2, c2, "geneB", "ctg2", 9_999_999_999_950, 10_000_000_000_040, -1, 7)]
    gs["strainC"] = [Seqinfo(s3, c3, "geneC", "ctg3", -40, 110, 1, 135), Seqinfo(s4, c4, "geneC2", "ctg3", 3, 67, -1, 2)]
    rec = (gs, "grpX", np.array([1, 1, 1], dtype=np.int64))
    for canon in (True, False):
        eng = Engine(klength=31, canon=canon, max_strains=32, stroi={"strainA", "strainB", "strainC"})
        text, hb = _device_kmers_tsv(eng, [rec])
        assert text == eng._render_targets(hb, hb.targets)
        (ek, _, _), _ = _oracle_texts([rec], stroi={"strainA", "strainB", "strainC"}, klength=31, canon=canon)
        assert text == ek
        eng.close()
