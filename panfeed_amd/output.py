"""Output file surface of the hot path (names, headers, gzip mode), as the reference fixes it in
/root/reference/panfeed/input.py:235-259 and /root/reference/panfeed/panfeed.py:116-129."""
import ctypes as C
import os

from . import _lib
from .engine import KMERS_TSV_HEADER, KMERS_TO_HASHES_HEADER, hashes_to_patterns_header


class ParallelGzipWriter:
    """Text handle over a .gz file, like gzip.open(path, "wt", compresslevel=9) (input.py:239-241, 255-258), whose
    deflate work is done by the library's host threads: what is written is buffered, cut at line ends into chunks
    and compressed as independent gzip members (pf_gzip_members).  Reads back as the same text with any gzip reader."""

    def __init__(self, path, compresslevel=9, buffer_bytes=64 << 20, chunk_bytes=4 << 20):
        self.L = _lib.load()
        self.fh = open(path, "wb")
        self.level, self.buffer_bytes, self.chunk_bytes = compresslevel, buffer_bytes, chunk_bytes
        self.parts, self.pending = [], 0
        self.wrote = False
        self.closed = False

    def write(self, text):
        b = text.encode() if isinstance(text, str) else bytes(text)
        if not b:
            return 0
        self.parts.append(b)
        self.pending += len(b)
        if self.pending >= self.buffer_bytes:
            self._emit()
        return len(text)

    def _emit(self):
        data = b"".join(self.parts)
        self.parts, self.pending = [], 0
        if not data:
            return
        out, n = C.c_void_p(), C.c_uint64()
        _lib.check(self.L.pf_gzip_members(data, len(data), self.level, self.chunk_bytes, C.byref(out), C.byref(n)))
        try:
            self.fh.write(memoryview((C.c_char * n.value).from_address(out.value)) if n.value else b"")
        finally:
            self.L.pf_free_text(out)
        self.wrote = True

    def flush(self):
        # a flush of the reference's GzipFile only empties Python-side buffers into the deflate stream; members are
        # cut when enough text has gathered, so nothing is forced out here except at close
        self.fh.flush()

    def close(self):
        if self.closed:
            return
        self._emit()
        if not self.wrote:                      # an empty file is still a valid (empty) gzip stream
            import gzip
            self.fh.write(gzip.compress(b"", compresslevel=self.level))
        self.fh.close()
        self.closed = True

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def create_kmer_stroi(output, compress=False):
    """kmers.tsv[.gz] with its header written (input.py:235-247)."""
    if not compress:
        fh = open(os.path.join(output, "kmers.tsv"), "w")
    else:
        fh = ParallelGzipWriter(os.path.join(output, "kmers.tsv.gz"), compresslevel=9)
    fh.write(KMERS_TSV_HEADER)
    fh.flush()
    return fh


def create_hash_files(output, compress=False):
    """(hashes_to_patterns, kmers_to_hashes) handles, no headers yet (input.py:249-259)."""
    if not compress:
        hash_pat = open(os.path.join(output, "hashes_to_patterns.tsv"), "w")
        kmer_hash = open(os.path.join(output, "kmers_to_hashes.tsv"), "w")
    else:
        hash_pat = ParallelGzipWriter(os.path.join(output, "hashes_to_patterns.tsv.gz"), compresslevel=9)
        kmer_hash = ParallelGzipWriter(os.path.join(output, "kmers_to_hashes.tsv.gz"), compresslevel=9)
    return hash_pat, kmer_hash


def write_headers(hash_pat, kmer_hash, genepres):
    """panfeed.py:116-129; `genepres` only needs `.columns` (the strain names)."""
    hash_pat.write(hashes_to_patterns_header(list(genepres.columns)))
    hash_pat.flush()
    kmer_hash.write(KMERS_TO_HASHES_HEADER)
    kmer_hash.flush()
