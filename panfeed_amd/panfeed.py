"""Drop-in mirror of the reference's two hot-path callables, same names, argument order and
return shapes, running on the MI355X through libpanfeed_hip:

    cluster_cutter  /root/reference/panfeed/panfeed.py:23-113
    pattern_hasher  /root/reference/panfeed/panfeed.py:132-235

so the driver loop of /root/reference/panfeed/__main__.py:277-297,350-356 works unchanged:

    iter_o = partial(cluster_cutter, klength, stroi, multiple_files, canon, consider_missing, output, compress)
    func_w = partial(pattern_hasher, kmer_stroi, hash_pat, kmer_hash, genepres, patfilt, maf, output, ...)
    for x in iter_i:
        ret = iter_o(x)
        patterns = func_w((ret,), patterns=patterns)

Differences a caller can see: the second element of cluster_cutter's result is a `CutCluster`
handle (the packed cluster, not a {kmer: vector} dict) and the GPU work happens when
pattern_hasher consumes it -- feed it many results per call (any iterable) to batch; and
`patterns` is a `PatternSet` that keeps the run-global pattern table on the device.
There is no CPU fallback: without the HIP library or a GPU the calls raise.
"""
import os

import numpy as np

from .engine import Engine
from .output import create_hash_files, create_kmer_stroi, write_headers  # noqa: F401  (re-exported)


def init_presabs_vector(n_strains, clusterpresab, missing_nan=False):
    """panfeed.py:16-20 (kept for callers that import it; the device builds the same image from bits)."""
    v = np.zeros(n_strains, dtype=np.float64)
    if missing_nan:
        v[clusterpresab == 0] = np.nan
    return v


class CutCluster:
    """What cluster_cutter hands to pattern_hasher in place of the reference's cluster_dict."""

    def __init__(self, record, klength, stroi, canon, consider_missing, multiple_files, output, compress):
        self.record = record
        self.klength = klength
        self.stroi = stroi
        self.canon = canon
        self.consider_missing = consider_missing
        self.multiple_files = multiple_files
        self.output = output
        self.compress = compress


class _DeferredChunk:
    """memchunk placeholder: the kmers.tsv rows of one cluster, rendered by pattern_hasher."""

    def __init__(self):
        self.text = ""

    def __str__(self):
        return self.text


class PatternSet:
    """The run-global `patterns` (panfeed.py:146-150): lives on the device inside an Engine."""

    def __init__(self):
        self.engine = None
        self.key = None
        self.count = 0

    def __len__(self):
        return self.count


def cluster_cutter(cluster_gen, klength, stroi, multiple_files, canon, consider_missing_cluster,
                   output, compress=False):
    """Same signature and result tuple as panfeed.py:23-25, 109-113."""
    cluster, idx, clusterpresab = cluster_gen
    handle = CutCluster((cluster, idx, clusterpresab), klength, stroi, bool(canon == True),  # noqa: E712
                        bool(consider_missing_cluster), bool(multiple_files), output, compress)
    if multiple_files:
        return idx, handle, clusterpresab, None
    return idx, handle, clusterpresab, _DeferredChunk()


def pattern_hasher(cluster_dict_iter, kmer_stroi, hash_pat, kmer_hash, genepres, patfilt, maf, output,
                   patterns=None, consider_missing_cluster=False, compress=False):
    """Same signature and return value as panfeed.py:132-135, 235."""
    multiple_files = hash_pat is None or kmer_hash is None            # panfeed.py:142-144
    results = list(cluster_dict_iter)
    if patterns is None:
        patterns = PatternSet()
    if not results:
        return patterns
    h0 = results[0][1]
    if not isinstance(h0, CutCluster):
        raise TypeError("pattern_hasher expects the results of panfeed_amd.panfeed.cluster_cutter")
    n_cols = len(genepres.columns) if genepres is not None else 0
    need = max([n_cols] + [max(len(r[1].record[0]), len(r[1].record[2])) for r in results] + [1])
    key = (h0.klength, h0.canon, bool(consider_missing_cluster), bool(patfilt), float(maf), multiple_files)
    if patterns.engine is None or patterns.key != key or patterns.engine.max_strains < need:
        if patterns.engine is not None and patterns.count:
            raise ValueError("pattern_hasher options changed in the middle of a run")
        patterns.engine = Engine(klength=h0.klength, canon=h0.canon, consider_missing=consider_missing_cluster,
                                 patfilt=patfilt, maf=maf, multiple_files=multiple_files,
                                 max_strains=(need + 31) // 32 * 32, stroi=h0.stroi)
        patterns.key = key
    eng = patterns.engine
    eng.stroi = h0.stroi if h0.stroi else ()
    out = eng.run([r[1].record for r in results])
    patterns.count = out.stats["patterns"] if not multiple_files else out.stats["new_patterns"]

    if multiple_files:
        for (idx, _h, _p, _m), (cidx, kt, kh, hp) in zip(results, out.per_cluster):
            path = os.path.join(output, idx)                           # panfeed.py:38-43, 159-167
            if not os.path.exists(path):
                os.mkdir(path)
            ks = create_kmer_stroi(path, compress)
            ks.write(kt)
            ks.close()
            f_hp, f_kh = create_hash_files(path, compress)
            write_headers(f_hp, f_kh, genepres)
            f_hp.write(hp)
            f_kh.write(kh)
            f_hp.close()
            f_kh.close()
        return patterns
    if results[0][3] is not None:
        kmer_stroi.write(out.kmers_tsv)                                # panfeed.py:171
    hash_pat.write(out.hashes_to_patterns)                             # panfeed.py:225
    kmer_hash.write(out.kmers_to_hashes)                               # panfeed.py:226
    hash_pat.flush()                                                   # panfeed.py:232-233
    kmer_hash.flush()
    return patterns
