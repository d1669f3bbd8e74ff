"""Host side of the device boundary: reference-shaped cluster records -> the packed batch of
include/panfeed_hip.h, and the slow path for windows that contain a non-ACGT base.

A record is what /root/reference/panfeed/input.py:468 yields:
``(gene_sequences, idx, clusterpresab)`` with ``gene_sequences = {strain: [Seqinfo, ...]}``.

Everything here is bookkeeping (ordering, 2-bit packing, coordinates); the k-mer work itself
runs on the GPU, except for windows with a base outside A/C/G/T -- they cannot be 2-bit
packed, are rare, and are grouped here exactly as cluster_cutter does
(/root/reference/panfeed/panfeed.py:64-88), then handed to the device as pre-grouped rows.
"""
from dataclasses import dataclass, field

import numpy as np
from itertools import chain
from sys import intern as sys_intern

_LUT = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _LUT[_c] = _i
_COMP_CODE = np.array([3, 2, 1, 0], dtype=np.uint8)
_ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)


def maf_tables(maf, max_strains):
    """keep-interval per denominator n: a k-mer with `count` ones among n strains survives
    panfeed.py:190-200 iff lo[n] <= count <= hi[n].  Same float64 arithmetic as the reference
    (sum of 0/1 float64 == count exactly; count / n; 1 - af; af < maf)."""
    lo = np.zeros(max_strains + 1, dtype=np.uint32)
    hi = np.zeros(max_strains + 1, dtype=np.uint32)
    maf = float(maf)
    for n in range(1, max_strains + 1):
        cnt = np.arange(0, n + 1, dtype=np.float64)
        af = cnt / np.float64(n)
        af = np.where(af >= 0.5, 1 - af, af)
        keep = ~(af < maf)
        idx = np.flatnonzero(keep)
        if len(idx) == 0:
            lo[n], hi[n] = 1, 0
            continue
        if idx[-1] - idx[0] + 1 != len(idx):
            raise ValueError(f"MAF keep set is not an interval for n={n}")
        lo[n], hi[n] = idx[0], idx[-1]
    return lo, hi


def pack_codes(codes):
    """uint8 codes (0..3) -> uint64 words: 32 bases per word, first base in bits 63:62,
    padded with A to a multiple of 64 bases (16 bytes)."""
    n = len(codes)
    npad = (n + 63) // 64 * 64
    buf = np.zeros(npad, dtype=np.uint8)
    buf[:n] = codes
    bits = np.empty(npad * 2, dtype=np.uint8)
    bits[0::2] = buf >> 1
    bits[1::2] = buf & 1
    return np.packbits(bits).view(">u8").astype(np.uint64)


@dataclass
class SeqMeta:
    """what the positional rows (panfeed.py:90-107) need about one Seqinfo"""
    cluster: int
    strain: str
    seq: object            # the Seqinfo
    ord_base: int
    num_kmer: int
    segs: list             # [(seg_index_in_batch, start_pos, n_windows)]
    ambig: dict            # pos -> (canonseq, used_strand) for slow-path windows (canonical mode)


class TargetTable:
    """The target-strain sequences of a batch as the packer's flat arrays (pf_packed's target_* arrays): which input
    sequence each is, its A/C/G/T runs (segment index in the batch, first window, windows) and its windows with another
    letter (position, strand used, canonical text).  `metas()` gives the per-sequence SeqMeta objects the host renderer's
    marshalling and the tests read; the device renderer's marshalling (Engine._marshal_table) goes by the arrays, with
    `resolve(t_seq) -> (cluster indices, strain names, Seqinfos)` for what only the records know."""

    def __init__(self, k, t_seq, t_so, t_si, t_ss, t_sn, t_ao, t_ap, t_au, akeys, seq_ref, resolve=None):
        self.k = k
        self.t_seq, self.t_so, self.t_si, self.t_ss, self.t_sn = t_seq, t_so, t_si, t_ss, t_sn
        self.t_ao, self.t_ap, self.t_au, self.akeys = t_ao, t_ap, t_au, akeys
        self.seq_ref, self.resolve = seq_ref, resolve
        self._metas = None

    def __len__(self):
        return len(self.t_seq)

    def metas(self):
        if self._metas is None:
            k, out = self.k, []
            t_so, t_ao, t_si, t_ss, t_sn, t_ap, t_au, akeys = (self.t_so, self.t_ao, self.t_si, self.t_ss, self.t_sn,
                                                               self.t_ap, self.t_au, self.akeys)
            for ti in range(len(self.t_seq)):
                ci, strain, s = self.seq_ref(int(self.t_seq[ti]))
                segs = [(int(t_si[j]), int(t_ss[j]), int(t_sn[j])) for j in range(t_so[ti], t_so[ti + 1])]
                ambig = {int(t_ap[j]): (akeys[j * k:(j + 1) * k].decode("latin-1"), int(t_au[j]))
                         for j in range(t_ao[ti], t_ao[ti + 1])}
                out.append(SeqMeta(ci, strain, s, 0, max(len(s.sequence) - k + 1, 0), segs, ambig))
            self._metas = out
            self.seq_ref = None
        return self._metas


@dataclass
class HostBatch:
    k: int
    canon: bool
    W: int
    idx: list = field(default_factory=list)               # cluster names
    sorted_strains: list = field(default_factory=list)    # per cluster: sorted(cluster.keys())
    presab: list = field(default_factory=list)            # per cluster: the int vector as given
    packed: np.ndarray = None
    seg_word_off: np.ndarray = None
    seg_len: np.ndarray = None
    seg_sample: np.ndarray = None
    seg_ord_base: np.ndarray = None
    seg_strand_off: np.ndarray = None
    n_strand_words: int = 0
    cluster_seg_off: np.ndarray = None
    cluster_nstrains: np.ndarray = None
    cluster_npresab: np.ndarray = None
    cluster_presab: np.ndarray = None
    cluster_ordinal: np.ndarray = None
    extra_cluster: np.ndarray = None
    extra_ord: np.ndarray = None
    extra_bits: np.ndarray = None
    extra_keys: list = field(default_factory=list)        # k-mer strings of the slow-path rows
    target_table: object = None                           # TargetTable: the target strains' sequences, flat
    _targets: list = None                                 # their SeqMeta (made on first use)
    n_instances: int = 0                                  # trip count of panfeed.py:64 (x2 non-canonical)
    # with genomes resident in HBM (pf_submit_gather): `packed` holds only the host-packed ("literal") segments,
    # seg_word_off are offsets in the device buffer of n_words_dev words the gather fills
    n_words_dev: int = 0
    gather_src_off: np.ndarray = None
    gather_src_start: np.ndarray = None
    gather_src_flags: np.ndarray = None

    @property
    def n_clusters(self):
        return len(self.idx)

    @property
    def targets(self):
        """SeqMeta of the sequences in target strains (panfeed.py:90), in iteration order"""
        if self._targets is None:
            self._targets = self.target_table.metas() if self.target_table is not None else []
        return self._targets

    @property
    def n_targets(self):
        if self._targets is not None:
            return len(self._targets)
        return len(self.target_table) if self.target_table is not None else 0


def decode_keys(keys, k, key_words):
    """device k-mer keys -> list of str.  A key is the k-mer's 2k-bit value (first base most significant) in
    `key_words` words of 63 bits, most significant word first: bit b lives in word key_words - 1 - b // 63, bit b % 63."""
    keys = np.asarray(keys, dtype=np.uint64).reshape(-1, key_words)
    n = len(keys)
    if n == 0:
        return []
    out = np.empty((n, k), dtype=np.uint8)

    def bit(b):
        return (keys[:, key_words - 1 - b // 63] >> np.uint64(b % 63)) & np.uint64(1)
    for i in range(k):
        b0 = 2 * (k - 1 - i)
        code = bit(b0) | (bit(b0 + 1) << np.uint64(1))
        out[:, i] = _ASCII[code.astype(np.intp)]
    return [row.tobytes().decode() for row in out]


def _fill_from_packed(L, hb, handle, n_clusters, seq_ref, resolve=None):
    """Copy a pf_packed result into `hb`; seq_ref(q) -> (cluster index, strain, Seqinfo) of input sequence q.  With
    `resolve` (TargetTable) the records behind seq_ref outlive the call and the per-target objects are made on first use;
    without it they are made here."""
    import ctypes as C

    from . import _lib
    k, W = hb.k, hb.W
    v = _lib.PackedView()
    _lib.check(L.pf_packed_view(handle, C.byref(v)))

    def arr(ptr, n, dtype):
        return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).copy() if n else np.zeros(0, dtype=dtype)
    hb.packed = arr(v.packed, v.n_words, np.uint64)
    hb.seg_word_off = arr(v.seg_word_off, v.n_segs, np.uint64)
    hb.seg_len = arr(v.seg_len, v.n_segs, np.uint32)
    hb.seg_sample = arr(v.seg_sample, v.n_segs, np.uint32)
    hb.seg_ord_base = arr(v.seg_ord_base, v.n_segs, np.uint32)
    hb.seg_strand_off = arr(v.seg_strand_off, v.n_segs, np.uint32)
    hb.n_strand_words = int(v.n_strand_words)
    hb.cluster_seg_off = arr(v.cluster_seg_off, n_clusters + 1, np.uint32)
    hb.extra_cluster = arr(v.extra_cluster, v.n_extra, np.uint32)
    hb.extra_ord = arr(v.extra_ord, v.n_extra, np.uint32)
    hb.extra_bits = arr(v.extra_bits, v.n_extra * W, np.uint32).reshape(-1, W)
    keys = C.string_at(v.extra_keys, v.n_extra * k) if v.n_extra else b""
    hb.extra_keys = [keys[i * k:(i + 1) * k].decode("latin-1") for i in range(v.n_extra)]
    hb.n_instances = int(v.n_instances)
    if v.n_words_dev:
        hb.n_words_dev = int(v.n_words_dev)
        hb.gather_src_off = arr(v.gather_src_off, v.n_segs, np.uint64)
        hb.gather_src_start = arr(v.gather_src_start, v.n_segs, np.uint32)
        hb.gather_src_flags = arr(v.gather_src_flags, v.n_segs, np.uint32)
    nt = v.n_targets
    t_seq = arr(v.target_seq, nt, np.uint32)
    t_so = arr(v.target_seg_off, nt + 1, np.uint32)
    t_ao = arr(v.target_ambig_off, nt + 1, np.uint32)
    nts, nta = (int(t_so[-1]), int(t_ao[-1])) if nt else (0, 0)
    t_si = arr(v.target_seg_index, nts, np.uint32)
    t_ss = arr(v.target_seg_start, nts, np.uint32)
    t_sn = arr(v.target_seg_nwin, nts, np.uint32)
    t_ap = arr(v.target_ambig_pos, nta, np.uint32)
    t_au = arr(v.target_ambig_used, nta, np.int8)
    akeys = C.string_at(v.target_ambig_keys, nta * k) if nta else b""
    hb.target_table = TargetTable(k, t_seq, t_so, t_si, t_ss, t_sn, t_ao, t_ap, t_au, akeys, seq_ref, resolve)
    hb._targets = None
    if resolve is None:
        hb.targets                     # (the caller's records go away with the call: the objects are made now)


def _ascii_addresses(strings):
    """addresses (uint64 array) of the character data of ASCII `str` objects, which must stay alive while they are used.
    PyUnicode_AsUTF8 per string through ctypes costs 0.4 us a string -- at 2 000 strings per cluster more than everything
    else in the packer's Python side -- so the library walks the object array itself and calls the interpreter's
    PyUnicode_AsUTF8 (handed over as a function pointer) for EVERY string: no assumption about how a str lays out its
    bytes, nothing sampled.  The call goes through a GIL-holding handle (ctypes.PyDLL)."""
    import ctypes as C
    import sys

    from . import _lib
    n = len(strings)
    if n == 0:
        return np.zeros(1, dtype=np.uint64)
    if sys.implementation.name != "cpython":
        raise RuntimeError("handing str data over by address needs CPython")
    objs = np.empty(n, dtype=object)
    objs[:] = strings                              # the array holds the objects' addresses (and a reference each)
    out = np.zeros(n, dtype=np.uint64)
    as_utf8 = C.cast(C.pythonapi.PyUnicode_AsUTF8, C.c_void_p)
    _lib.check(_lib.load_pydll().pf_py_str_addresses(C.c_void_p(objs.ctypes.data), n, as_utf8, C.c_void_p(out.ctypes.data)))
    if not out.all():
        raise ValueError("a sequence that is not a str")
    return out


_PY_API = None


def _py_api():
    """the six interpreter functions pf_py_seqinfo_columns calls, as the library's pf_py_api struct"""
    global _PY_API
    if _PY_API is None:
        import ctypes as C
        fn = [C.pythonapi.PyList_GetItem, C.pythonapi.PyObject_GetAttr, C.pythonapi.PyUnicode_AsUTF8AndSize,
              C.pythonapi.PyUnicode_GetLength, C.pythonapi.Py_DecRef, C.pythonapi.PyErr_Clear]
        _PY_API = (C.c_void_p * 6)(*[C.cast(f, C.c_void_p).value for f in fn])
    return _PY_API


def _seqinfo_columns(flat_s):
    """(addresses of .sequence bytes, of .compsequence bytes, lengths, held references) for a list of Seqinfo-like objects
    whose two attributes are plain ASCII str of equal length -- or None when any of them is not (the caller's general path
    then looks at them one by one and says what is wrong)"""
    import ctypes as C
    import sys

    from . import _lib
    n = len(flat_s)
    if n == 0 or sys.implementation.name != "cpython" or type(flat_s) is not list:
        return None
    a_seq = np.empty(n, dtype=np.uint64); a_comp = np.empty(n, dtype=np.uint64)
    a_len = np.empty(n, dtype=np.uint32); flags = np.empty(n, dtype=np.uint8)
    held = np.zeros(2 * n, dtype=np.uint64)
    api = _py_api()
    _lib.check(_lib.load_pydll().pf_py_seqinfo_columns(
        C.c_void_p(id(flat_s)), n, C.c_void_p(id(_ATTR_SEQ)), C.c_void_p(id(_ATTR_COMP)), C.cast(api, C.c_void_p),
        C.c_void_p(a_seq.ctypes.data), C.c_void_p(a_comp.ctypes.data), C.c_void_p(a_len.ctypes.data),
        C.c_void_p(flags.ctypes.data), C.c_void_p(held.ctypes.data)))
    if not flags.all():
        _release_held(held)
        return None
    return a_seq, a_comp, a_len.astype(np.int64), held


def _release_held(held):
    import ctypes as C

    from . import _lib
    _lib.check(_lib.load_pydll().pf_py_release(C.c_void_p(held.ctypes.data), len(held), C.cast(_py_api(), C.c_void_p)))


_ATTR_SEQ = sys_intern("sequence")
_ATTR_COMP = sys_intern("compsequence")


def build_batch_native(records, klength, canon, W, stroi=(), first_ordinal=0, want_strand=True):
    """Same result as build_batch, with the per-base work (packing, non-ACGT splitting, slow-path grouping)
    done by the library's host threads (pf_pack_records, csrc/pf_pack.cpp)."""
    import ctypes as C

    from . import _lib
    L = _lib.load()
    k = int(klength)
    hb = HostBatch(k=k, canon=bool(canon), W=W)
    # The sequences are handed over by address -- of the str objects' own bytes when they are ASCII, else of two NUL-joined
    # latin-1 blocks -- with the per-sequence columns built by numpy, not as one bytes object and one pointer per Seqinfo:
    # the per-sequence Python work (two encodes, five appends) was what bounded the host-strings path.
    flat_s = []                                   # Seqinfo per sequence, iteration order (panfeed.py:54-55)
    col_parts, tgt_parts, strain_parts = [], [], []
    cl_names = []
    cl_off = [0]
    cl_nstr, cl_npres, cl_presab, cl_ord = [], [], [], []
    any_target = False
    stroi_set = stroi if isinstance(stroi, (set, frozenset, dict)) else frozenset(stroi or ())
    for ci, (gs, idx, presab) in enumerate(records):
        names = list(gs.keys())
        n = len(names)
        if n > W * 32:
            raise ValueError(f"cluster {idx}: {n} strains exceed the context's max_strains")
        sorted_names = sorted(names)
        col = dict(zip(sorted_names, range(n)))                    # panfeed.py:47-49
        presab = np.asarray(presab)
        if len(presab) > W * 32:
            raise ValueError(f"cluster {idx}: clusterpresab longer than max_strains")
        if presab.size and not np.isin(presab, (0, 1)).all():
            raise ValueError(f"cluster {idx}: clusterpresab must hold 0/1")
        pb = np.zeros(W, dtype=np.uint32)
        nz = np.flatnonzero(presab)
        np.bitwise_or.at(pb, nz >> 5, (np.uint32(1) << (nz & 31).astype(np.uint32)))
        hb.idx.append(str(idx))
        hb.sorted_strains.append(sorted_names)
        hb.presab.append(presab)
        cl_nstr.append(n)
        cl_npres.append(len(presab))
        cl_presab.append(pb)
        cl_ord.append(first_ordinal + ci)
        # (per strain: whole-container calls, no Python-level loop with a dict look-up per name -- four of those were a
        # third of the packer's Python side)
        counts = np.fromiter(map(len, gs.values()), dtype=np.int64, count=n)
        flat_s.extend(chain.from_iterable(gs.values()))           # panfeed.py:54-55
        if names == sorted_names:
            colv = np.arange(n, dtype=np.uint32)
        else:
            colv = np.fromiter(map(col.__getitem__, names), dtype=np.uint32, count=n)
        if stroi:                                                 # panfeed.py:90 (`strain in stroi`, one look-up per strain)
            tg = np.fromiter(map(stroi_set.__contains__, names), dtype=np.uint8, count=n)
        else:
            tg = np.zeros(n, dtype=np.uint8)
        any_target = any_target or bool(tg.any())
        if int(counts.min(initial=1)) == 1 and int(counts.max(initial=1)) == 1:
            col_parts.append(colv); tgt_parts.append(tg); strain_parts.append(np.arange(n, dtype=np.uint32))
        else:
            col_parts.append(np.repeat(colv, counts))
            tgt_parts.append(np.repeat(tg, counts))
            strain_parts.append(np.repeat(np.arange(n, dtype=np.uint32), counts))
        cl_names.append(names)
        cl_off.append(len(flat_s))
    nseq = len(flat_s)
    # the two attributes of every Seqinfo: one pass inside the library (addresses of the strings' own bytes, lengths, "plain
    # ASCII of equal length" flags) instead of two list comprehensions, four maps and two address passes up here
    fast = _seqinfo_columns(flat_s)
    if fast is not None:
        a_seq, a_comp, a_len, held = fast
    else:
        held = None
    if fast is None:
        seq_strs = [s.sequence for s in flat_s]
        comp_strs = [s.compsequence for s in flat_s]
        a_len = np.fromiter(map(len, seq_strs), dtype=np.int64, count=nseq)
        if not np.array_equal(a_len, np.fromiter(map(len, comp_strs), dtype=np.int64, count=nseq)):
            q = int(np.flatnonzero(a_len != np.fromiter(map(len, comp_strs), dtype=np.int64, count=nseq))[0])
            ci = int(np.searchsorted(np.asarray(cl_off), q, side="right") - 1)
            raise ValueError(f"{hb.idx[ci]}: sequence and compsequence differ in length")
    if fast is not None:
        pass
    elif all(map(str.isascii, seq_strs)) and all(map(str.isascii, comp_strs)):
        # an ASCII str keeps its bytes in the object itself: their address is all the packer needs (the strings stay alive
        # in flat_s for the duration of the call) -- no copy of the batch's 300 MB of text at all
        a_seq = _ascii_addresses(seq_strs)
        a_comp = _ascii_addresses(comp_strs)
    else:
        seq_blob = "\0".join(seq_strs).encode("latin-1")
        comp_blob = "\0".join(comp_strs).encode("latin-1")
        if len(seq_blob) != int(a_len.sum()) + max(nseq - 1, 0) or len(comp_blob) != len(seq_blob):
            raise ValueError("a sequence with characters outside latin-1")
        starts = np.zeros(max(nseq, 1), dtype=np.uint64)
        if nseq > 1:
            starts[1:nseq] = np.cumsum(a_len[:-1] + 1).astype(np.uint64)
        a_seq = np.ascontiguousarray(starts + np.uint64(C.cast(C.c_char_p(seq_blob), C.c_void_p).value or 0))
        a_comp = np.ascontiguousarray(starts + np.uint64(C.cast(C.c_char_p(comp_blob), C.c_void_p).value or 0))
    a_len = a_len.astype(np.uint32)
    a_col = np.concatenate(col_parts).astype(np.uint32) if col_parts else np.zeros(0, dtype=np.uint32)
    a_tgt = np.concatenate(tgt_parts).astype(np.uint8) if tgt_parts else np.zeros(0, dtype=np.uint8)
    a_strain = np.concatenate(strain_parts) if strain_parts else np.zeros(0, dtype=np.uint32)
    a_off = np.asarray(cl_off, dtype=np.uint32)
    seq_cluster = np.repeat(np.arange(len(cl_nstr)), np.diff(a_off.astype(np.int64)))

    def seq_ref(q):
        ci = int(seq_cluster[q])
        return ci, cl_names[ci][int(a_strain[q])], flat_s[q]

    def resolve(t_seq):
        ci = seq_cluster[t_seq]
        st = a_strain[t_seq]
        return ci, [cl_names[c][a] for c, a in zip(ci.tolist(), st.tolist())], [flat_s[q] for q in t_seq.tolist()]

    pin = _lib.PackIn(len(cl_nstr), nseq, C.cast(a_seq.ctypes.data, C.POINTER(C.c_char_p)),
                      C.cast(a_comp.ctypes.data, C.POINTER(C.c_char_p)), a_len.ctypes.data, a_col.ctypes.data,
                      a_tgt.ctypes.data if any_target else None, a_off.ctypes.data, k, int(bool(canon)), W,
                      int(bool(want_strand)))
    handle = C.c_void_p()
    try:
        _lib.check(L.pf_pack_records(C.byref(pin), C.byref(handle)))
        try:
            _fill_from_packed(L, hb, handle, len(cl_nstr), seq_ref, resolve)
        finally:
            L.pf_packed_free(handle)
    finally:
        if held is not None:
            _release_held(held)
    hb.cluster_nstrains = np.asarray(cl_nstr, dtype=np.uint32)
    hb.cluster_npresab = np.asarray(cl_npres, dtype=np.uint32)
    hb.cluster_presab = (np.stack(cl_presab) if cl_presab else np.zeros((0, W), dtype=np.uint32)).astype(np.uint32)
    hb.cluster_ordinal = np.asarray(cl_ord, dtype=np.uint64)
    return hb
