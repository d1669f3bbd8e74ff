"""ctypes front-end of the CPU oracle (oracle/panfeed_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package (panfeed_amd/) never does.

Clusters use the reference's own record shape
(/root/reference/panfeed/input.py:455-468): ``(gene_sequences, idx, clusterpresab)`` with
``gene_sequences = {strain: [Seqinfo, ...]}`` in dict insertion order.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Seq(C.Structure):
    _fields_ = [("sequence", C.c_char_p), ("compsequence", C.c_char_p), ("len", C.c_uint32),
                ("strain", C.c_uint32), ("id", C.c_char_p), ("chromosome", C.c_char_p),
                ("start", C.c_int64), ("end", C.c_int64), ("strand", C.c_int32), ("_pad", C.c_int32),
                ("offset", C.c_int64)]


class _Cluster(C.Structure):
    _fields_ = [("idx", C.c_char_p), ("n_strains", C.c_uint32), ("n_seqs", C.c_uint32),
                ("strain_names", C.POINTER(C.c_char_p)), ("strain_is_target", C.POINTER(C.c_uint8)),
                ("seqs", C.POINTER(_Seq)), ("n_presab", C.c_uint32), ("_pad", C.c_uint32),
                ("presab", C.POINTER(C.c_int64))]


class _Opts(C.Structure):
    _fields_ = [("klength", C.c_int32), ("canon", C.c_int32), ("consider_missing", C.c_int32),
                ("patfilt", C.c_int32), ("maf", C.c_double), ("multiple_files", C.c_int32),
                ("want_kmers_tsv", C.c_int32)]


def build(force=False):
    so = os.path.join(_HERE, "libpanfeed_oracle.so")
    src = os.path.join(_HERE, "panfeed_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libpanfeed_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.po_run_new.restype = C.c_void_p
        L.po_run_free.argtypes = [C.c_void_p]
        L.po_run_clear_text.argtypes = [C.c_void_p]
        L.po_run_text.restype = C.c_void_p
        L.po_run_text.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
        L.po_run_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.po_run_clusters.restype = C.c_int
        L.po_run_clusters.argtypes = [C.c_void_p, C.POINTER(_Cluster), C.c_uint32, C.POINTER(_Opts), C.c_int]
        L.po_md5.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.po_b64_16.argtypes = [C.c_void_p, C.c_char_p]
        L.po_kmers_tsv_header.restype = C.c_char_p
        L.po_kmers_to_hashes_header.restype = C.c_char_p
        _LIB = L
    return _LIB


def md5_b64(data: bytes) -> str:
    """base64(md5(data))[:24] as panfeed.py:175-176 computes it."""
    L = lib()
    dig = C.create_string_buffer(16)
    buf = C.create_string_buffer(data, len(data)) if len(data) else C.create_string_buffer(1)
    L.po_md5(buf, len(data), dig)
    out = C.create_string_buffer(25)
    L.po_b64_16(dig, out)
    return out.value.decode()


def md5_hex(data: bytes) -> str:
    L = lib()
    dig = C.create_string_buffer(16)
    buf = C.create_string_buffer(data, len(data)) if len(data) else C.create_string_buffer(1)
    L.po_md5(buf, len(data), dig)
    return dig.raw.hex()


def _b(s):
    return s if isinstance(s, bytes) else str(s).encode()


class OracleRun:
    """One panfeed run: keeps the run-global ``patterns`` set across ``feed`` calls."""

    def __init__(self, klength=31, stroi=(), canon=True, consider_missing=False, patfilt=True,
                 maf=0.01, multiple_files=False, want_kmers_tsv=True, threads=1):
        self.L = lib()
        self.h = C.c_void_p(self.L.po_run_new())
        self.opts = _Opts(int(klength), int(bool(canon)), int(bool(consider_missing)), int(bool(patfilt)),
                          float(maf), int(bool(multiple_files)), int(bool(want_kmers_tsv)))
        self.stroi = stroi
        self.threads = threads

    def close(self):
        if self.h:
            self.L.po_run_free(self.h)
            self.h = None

    __del__ = close

    def feed(self, clusters):
        """clusters: iterable of (gene_sequences, idx, clusterpresab)."""
        self.run_prepared(self.prepare(clusters))

    def prepare(self, clusters):
        """marshal records into C structs (kept out of timed regions)"""
        clusters = list(clusters)
        keep = []  # keep ctypes buffers alive
        arr = (_Cluster * max(1, len(clusters)))()
        for ci, (gs, idx, presab) in enumerate(clusters):
            names = list(gs.keys())
            n = len(names)
            c_names = (C.c_char_p * max(1, n))(*[_b(x) for x in names])
            # `strain in stroi` (panfeed.py:90); stroi == "" when no targets (input.py:200-202)
            tgt = (C.c_uint8 * max(1, n))(*[1 if (x in self.stroi) else 0 for x in names])
            nseq = sum(len(v) for v in gs.values())
            seqs = (_Seq * max(1, nseq))()
            j = 0
            for si, name in enumerate(names):
                for s in gs[name]:
                    seq, comp = _b(s.sequence), _b(s.compsequence)
                    assert len(seq) == len(comp)
                    seqs[j] = _Seq(seq, comp, len(seq), si, _b(s.id), _b(s.chromosome),
                                   int(s.start), int(s.end), int(s.strand), 0, int(s.offset))
                    j += 1
            pa = np.ascontiguousarray(np.asarray(presab), dtype=np.int64)
            if self.opts.consider_missing and len(pa) != n:
                # init_presabs_vector's `v[clusterpresab == 0] = np.nan` (panfeed.py:19): numpy refuses a boolean
                # mask of another length -- e.g. a strain of the table without a GFF under --consider-missing-cluster
                raise IndexError(f"boolean index did not match indexed array along axis 0; size of axis is {n} "
                                 f"but size of corresponding boolean axis is {len(pa)}")
            keep += [c_names, tgt, seqs, pa]
            arr[ci] = _Cluster(_b(idx), n, nseq, c_names, tgt, seqs, len(pa), 0,
                               pa.ctypes.data_as(C.POINTER(C.c_int64)))
        return arr, len(clusters), keep

    def run_prepared(self, prepared):
        arr, n, keep = prepared
        rc = self.L.po_run_clusters(self.h, arr, n, C.byref(self.opts), int(self.threads))
        if rc != 0:
            raise RuntimeError("oracle failed")
        del keep

    def text(self, which):
        n = C.c_uint64()
        p = self.L.po_run_text(self.h, which, C.byref(n))
        return C.string_at(p, n.value).decode()

    def texts(self):
        """(kmers.tsv body, kmers_to_hashes.tsv body, hashes_to_patterns.tsv body) -- no headers."""
        return self.text(0), self.text(1), self.text(2)

    def clear_text(self):
        self.L.po_run_clear_text(self.h)

    def stats(self):
        a = (C.c_uint64 * 4)()
        self.L.po_run_stats(self.h, a)
        return {"instances": a[0], "unique_kmers": a[1], "kept_kmers": a[2], "patterns": a[3]}


def kmers_tsv_header():
    return lib().po_kmers_tsv_header().decode()


def kmers_to_hashes_header():
    return lib().po_kmers_to_hashes_header().decode()


def hashes_to_patterns_header(strains):
    # panfeed.py:120-123
    return "hashed_pattern" + "".join("\t" + s for s in sorted(strains)) + "\n"
