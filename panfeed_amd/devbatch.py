"""Device-resident batches: keep a packed batch in HBM across pf_submit calls (bench.py, large tests).

`from_host_batch` uploads a packing.HostBatch once; `from_synth` builds the full-size synthetic
workload without ever materialising per-sample strings: allele pools are packed on the host
(a few hundred MB), per-sample segments are expanded from them by a copy kernel on the device
(pf_synth_expand).
"""
import ctypes as C

import numpy as np

from . import _lib
from .packing import pack_codes


class DeviceBatch:
    def __init__(self, engine):
        self.engine = engine
        self.L = engine.L
        self.ptrs = {}
        self.batch = _lib.Batch()
        self.n_instances = 0
        self.packed_bytes = 0
        self.n_clusters = 0
        self.n_segs = 0
        self.n_extra = 0
        self.idx = []            # cluster names (batch order) and the k-mer text of the slow-path rows: what the
        self.extra_keys = []     # text renderers need beside the device results (Engine.render_device)

    def _alloc(self, name, nbytes):
        p = C.c_void_p()
        _lib.check(self.L.pf_dev_alloc(self.engine.ctx, int(nbytes), C.byref(p)))
        self.ptrs[name] = p
        return p

    def _put(self, name, arr):
        arr = np.ascontiguousarray(arr)
        p = self._alloc(name, max(arr.nbytes, 16))
        if arr.nbytes:
            _lib.check(self.L.pf_dev_upload(self.engine.ctx, p, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return p

    def free(self):
        for p in self.ptrs.values():
            self.L.pf_dev_free(self.engine.ctx, p)
        self.ptrs = {}

    def download(self, name, dtype, count):
        out = np.empty(count, dtype=dtype)
        _lib.check(self.L.pf_dev_download(self.engine.ctx, out.ctypes.data_as(C.c_void_p), self.ptrs[name], out.nbytes))
        return out

    def submit(self, engine=None):
        """run the batch on `engine` (default: the engine that owns the buffers; any context on the same
        device can read them)"""
        res = _lib.Result()
        _lib.check(self.L.pf_submit((engine or self.engine).ctx, C.byref(self.batch), C.byref(res)))
        return res


def from_host_batch(engine, hb):
    db = DeviceBatch(engine)
    b = db.batch
    b.n_clusters, b.n_segs, b.n_words, b.on_device = hb.n_clusters, len(hb.seg_len), len(hb.packed), 1
    b.packed = db._put("packed", hb.packed)
    b.seg_word_off = db._put("seg_word_off", hb.seg_word_off)
    b.seg_len = db._put("seg_len", hb.seg_len)
    b.seg_sample = db._put("seg_sample", hb.seg_sample)
    b.seg_ord_base = db._put("seg_ord_base", hb.seg_ord_base)
    b.cluster_seg_off = db._put("cluster_seg_off", hb.cluster_seg_off)
    b.cluster_nstrains = db._put("cluster_nstrains", hb.cluster_nstrains)
    b.cluster_npresab = db._put("cluster_npresab", hb.cluster_npresab)
    b.cluster_presab = db._put("cluster_presab", hb.cluster_presab)
    b.cluster_ordinal = db._put("cluster_ordinal", hb.cluster_ordinal)
    b.n_extra = len(hb.extra_ord)
    if b.n_extra:
        b.extra_cluster = db._put("extra_cluster", hb.extra_cluster)
        b.extra_ord = db._put("extra_ord", hb.extra_ord)
        b.extra_bits = db._put("extra_bits", hb.extra_bits)
    db.n_instances = hb.n_instances
    db.packed_bytes = int(((hb.seg_len.astype(np.int64) + 3) // 4).sum())
    db.n_clusters, db.n_segs = hb.n_clusters, len(hb.seg_len)
    db.idx, db.extra_keys = list(hb.idx), list(hb.extra_keys)
    return db


_N_COMP = bytes.maketrans(b"ACGTN", b"TGCAN")
_ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)


def from_synth(engine, clusters, k, canon=True, first_ordinal=0):
    """clusters: list of synth.SynthCluster with sample names in sorted order (iteration order == column order).
    Sequences with an 'N' (synth's n_rate; canonical mode only) are split at it into their A/C/G/T runs, packed on the
    host as literals, and the windows that contain the 'N' become slow-path rows with the reference's string
    semantics (panfeed.py:65-79) -- what `packing` does for records, without ever making the other sequences' strings."""
    W = engine.W
    allele_words, allele_off, allele_len = [], [], []
    woff = 0
    seg_allele, seg_len, seg_sample, seg_ord = [], [], [], []
    cl_seg_off = np.zeros(len(clusters) + 1, dtype=np.int64)
    cl_nstr = np.zeros(len(clusters), dtype=np.uint32)
    cl_presab = np.zeros((len(clusters), W), dtype=np.uint32)
    ex_cluster, ex_ord, ex_bits, ex_keys = [], [], [], []
    n_inst = 0

    def add_allele(codes):
        nonlocal woff
        w = pack_codes(codes)
        allele_words.append(w)
        allele_off.append(woff)
        allele_len.append(len(codes))
        woff += len(w)
        return len(allele_off) - 1

    for ci, cl in enumerate(clusters):
        has_n = cl.seq_npos >= 0
        if has_n.any() and not canon:
            raise ValueError("from_synth: sequences with 'N' are supported in canonical mode only")
        if cl.names != sorted(cl.names):
            raise ValueError("from_synth needs sample names in sorted order")
        a0 = len(allele_off)
        for a in cl.alleles:
            add_allele(a)
        lens = np.array([len(a) for a in cl.alleles], dtype=np.int64)[cl.seq_allele]
        ninst = np.maximum(lens - k + 1, 0)
        ob = np.concatenate(([0], np.cumsum(ninst)[:-1])) if len(ninst) else np.zeros(0, np.int64)
        keep = (lens >= k) & ~has_n
        s_allele = (a0 + cl.seq_allele).astype(np.int64)
        s_len, s_smp, s_ord, s_seq = lens.copy(), cl.seq_sample.astype(np.int64), ob.copy(), np.arange(len(lens))
        rows = [(s_allele[keep], s_len[keep], s_smp[keep], s_ord[keep], s_seq[keep] * 2)]
        if has_n.any():
            amb = {}
            for q in np.flatnonzero(has_n):
                q = int(q)
                codes = cl.alleles[int(cl.seq_allele[q])]
                L, p, col = len(codes), int(cl.seq_npos[q]), int(cl.seq_sample[q])
                for part, (a, b) in enumerate(((0, p), (p + 1, L))):     # the A/C/G/T runs on both sides of the N
                    if b - a >= k:
                        lit = add_allele(codes[a:b])
                        rows.append((np.array([lit]), np.array([b - a]), np.array([col]), np.array([ob[q] + a]),
                                     np.array([q * 2 + part])))
                if L >= k:
                    seq = bytearray(_ASCII[codes].tobytes())
                    seq[p] = ord("N")
                    seq = bytes(seq)
                    comp = seq.translate(_N_COMP)
                    for pos in range(max(0, p - k + 1), min(L - k, p) + 1):
                        spec = seq[pos:pos + k]                           # panfeed.py:65
                        rev = comp[pos:pos + k][::-1]                     # panfeed.py:67
                        key = spec if spec <= rev else rev                # panfeed.py:70-75
                        ent = amb.setdefault(key, [int(ob[q]) + pos, set()])
                        ent[0] = min(ent[0], int(ob[q]) + pos)
                        ent[1].add(col)
            for key, (o, cols) in amb.items():
                row = np.zeros(W, dtype=np.uint32)
                for c in cols:
                    row[c >> 5] |= np.uint32(1 << (c & 31))
                ex_cluster.append(ci)
                ex_ord.append(o)
                ex_bits.append(row)
                ex_keys.append(key.decode())
        al = np.concatenate([r[0] for r in rows])
        ln = np.concatenate([r[1] for r in rows])
        sm = np.concatenate([r[2] for r in rows])
        od = np.concatenate([r[3] for r in rows])
        sq = np.concatenate([r[4] for r in rows])
        order = np.lexsort((sq, sm))                  # by sample column, then the reference's iteration order
        seg_allele.append(al[order].astype(np.uint32))
        seg_len.append(ln[order].astype(np.uint32))
        seg_sample.append(sm[order].astype(np.uint32))
        seg_ord.append(od[order].astype(np.uint32))
        cl_seg_off[ci + 1] = cl_seg_off[ci] + len(order)
        cl_nstr[ci] = len(cl.names)
        pres = np.flatnonzero(cl.present)
        np.bitwise_or.at(cl_presab[ci], pres >> 5, (np.uint32(1) << (pres & 31).astype(np.uint32)))
        n_inst += int(ninst.sum())
    seg_allele = np.concatenate(seg_allele) if seg_allele else np.zeros(0, np.uint32)
    seg_len = np.concatenate(seg_len) if seg_len else np.zeros(0, np.uint32)
    seg_sample = np.concatenate(seg_sample) if seg_sample else np.zeros(0, np.uint32)
    seg_ord = np.concatenate(seg_ord) if seg_ord else np.zeros(0, np.uint32)
    seg_words = 2 * ((seg_len.astype(np.int64) + 63) // 64)
    seg_word_off = np.concatenate(([0], np.cumsum(seg_words)[:-1])).astype(np.uint64) if len(seg_len) else np.zeros(0, np.uint64)
    n_words = int(seg_words.sum()) + 4

    db = DeviceBatch(engine)
    b = db.batch
    b.n_clusters, b.n_segs, b.n_words, b.on_device = len(clusters), len(seg_len), n_words, 1
    aw = np.concatenate(allele_words + [np.zeros(4, np.uint64)])
    p_aw = db._put("allele_words", aw)
    p_ao = db._put("allele_off", np.asarray(allele_off, dtype=np.uint64))
    p_sa = db._put("seg_allele", seg_allele)
    b.seg_word_off = db._put("seg_word_off", seg_word_off)
    b.seg_len = db._put("seg_len", seg_len)
    b.packed = db._alloc("packed", n_words * 8)
    # the four words of tail padding are not covered by the expansion kernel: zero them (fresh device memory is not)
    zeros = np.zeros(4, dtype=np.uint64)
    _lib.check(db.L.pf_dev_upload(engine.ctx, C.c_void_p(b.packed + (n_words - 4) * 8), zeros.ctypes.data_as(C.c_void_p), 32))
    _lib.check(db.L.pf_synth_expand(engine.ctx, p_aw, p_ao, p_sa, b.seg_word_off, b.seg_len, len(seg_len), b.packed))
    b.seg_sample = db._put("seg_sample", seg_sample)
    b.seg_ord_base = db._put("seg_ord_base", seg_ord)
    b.cluster_seg_off = db._put("cluster_seg_off", cl_seg_off.astype(np.uint32))
    b.cluster_nstrains = db._put("cluster_nstrains", cl_nstr)
    b.cluster_npresab = db._put("cluster_npresab", cl_nstr)
    b.cluster_presab = db._put("cluster_presab", cl_presab)
    b.cluster_ordinal = db._put("cluster_ordinal", (first_ordinal + np.arange(len(clusters))).astype(np.uint64))
    b.n_extra = len(ex_ord)
    if b.n_extra:
        b.extra_cluster = db._put("extra_cluster", np.asarray(ex_cluster, dtype=np.uint32))
        b.extra_ord = db._put("extra_ord", np.asarray(ex_ord, dtype=np.uint32))
        b.extra_bits = db._put("extra_bits", np.stack(ex_bits).astype(np.uint32))
    for name in ("allele_words", "allele_off", "seg_allele"):
        db.L.pf_dev_free(engine.ctx, db.ptrs.pop(name))
    db.n_instances = n_inst * (1 if canon else 2)
    db.n_extra = int(b.n_extra)
    db.packed_bytes = int(((seg_len.astype(np.int64) + 3) // 4).sum())
    db.n_clusters, db.n_segs = len(clusters), len(seg_len)
    db.idx, db.extra_keys = [cl.idx for cl in clusters], ex_keys
    return db
