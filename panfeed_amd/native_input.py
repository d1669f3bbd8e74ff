"""Native reader in front of the hot path: panaroo table + GFF3 + FASTA -> packed batches.

Host-side mirror of /root/reference/panfeed/input.py: `what_are_my_inputfiles` (:16-64) is restated here (it only
lists files); `parse_gff` (:274-332), `iter_gene_clusters` (:335-468) and the table load (:188-191) run in the
library (csrc/pf_input.cpp), and their records go to `pf_pack_records` by pointer -- no Python string is made
for a sequence unless its strain is a `--targets` strain.

PARITY UNPINNED: the reference reads sequences through pyfaidx, which is absent here; see DESIGN.md.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from .classes import Seqinfo
from .packing import HostBatch, _fill_from_packed


def what_are_my_inputfiles(gffdir, fastadir=None):
    """(sorted genome names, sorted genomes that have their own nucleotide fasta, {genome: gff path},
    {genome: fasta path}); names are file names minus the last extension (input.py:28-33, 47-52)."""
    def listing(path):
        if os.path.isfile(path):                                   # a file of files (input.py:20-23)
            return [x.rstrip() for x in open(path)], True
        return sorted(os.listdir(path)), False

    gffs, fastas = {}, {}
    files, fof = listing(gffdir)
    for f in files:
        if f.endswith(".gff"):
            genome = ".".join(os.path.split(f)[-1].split(".")[:-1])
            gffs[genome] = f if fof else os.path.join(gffdir, f)
    if fastadir is not None:
        files, fof = listing(fastadir)
        for f in files:
            if f.endswith(".fasta") or f.endswith(".fna"):
                genome = ".".join(os.path.split(f)[-1].split(".")[:-1])
                if genome not in gffs:
                    continue
                path = f if fof else os.path.join(fastadir, f)
                # a directory holding both spellings: .fna wins (input.py:118-124)
                if genome not in fastas or path.endswith(".fna"):
                    fastas[genome] = path
    if not gffs:
        raise FileNotFoundError(f"No GFF files found in the inputs provided ({gffdir})")
    return sorted(gffs), sorted(fastas), gffs, fastas


def _cstr_array(strings):
    arr = (C.c_char_p * max(len(strings), 1))()
    for i, s in enumerate(strings):
        arr[i] = None if s is None else os.fsencode(s)
    return arr


class Pangenome:
    """An opened pangenome: table, features and contigs resident in host memory (the library owns them)."""

    def __init__(self, presence_absence, gffdir, fastadir=None, upstream=0, downstream=0,
                 downstream_start_codon=False, targets=(), genes=None, genome_names=None, gff_paths=None,
                 fasta_paths=None, raise_missing=False, engine=None, debug_hostsink=False):
        """engine: open with the genomes going to that engine's GPU as the files are read (pf_pangenome_open_device: one
        pass over the input; the reader is resident from the start and keeps contig text only where text is needed).
        When the device store's size estimate does not hold (PF_ERR_CAPACITY: contigs of a few letters each) the reader
        is opened the two-step way and made resident.  `engine` may be a callable that returns the engine: it is called
        when the first genome needs the context (the caller may be creating it while the table is parsed)."""
        self.L = _lib.load()
        if genome_names is None:
            names, _with_fa, gffs, fastas = what_are_my_inputfiles(gffdir, fastadir)
            gff_paths = [gffs[n] for n in names]
            fasta_paths = [fastas.get(n) for n in names]
        else:
            names = list(genome_names)
            fasta_paths = list(fasta_paths) if fasta_paths else [None] * len(names)
        self.targets = tuple(targets or ())
        o = _lib.PangenomeOpts()
        self._keep = (_cstr_array(names), _cstr_array(gff_paths), _cstr_array(fasta_paths),
                      _cstr_array(list(self.targets)), _cstr_array(list(genes)) if genes is not None else None)
        o.presence_absence_csv = os.fsencode(presence_absence)
        o.n_genomes = len(names)
        o.genome_names, o.gff_paths = self._keep[0], self._keep[1]
        o.fasta_paths = self._keep[2] if any(p is not None for p in fasta_paths) else None
        o.upstream, o.downstream = int(upstream), int(downstream)
        o.downstream_start_codon = int(bool(downstream_start_codon))
        o.raise_missing = int(bool(raise_missing))      # input.py:345, 399, 409: a KeyError instead of a warning
        o.target_strains, o.n_targets = self._keep[3], len(self.targets)
        if genes is not None:
            o.gene_list, o.n_genes = self._keep[4], len(genes)
        self.h = C.c_void_p()
        self.one_pass = False
        got = {}

        def engine_now():
            if "eng" not in got:
                got["eng"] = engine() if callable(engine) else engine
            return got["eng"]
        if engine is not None:
            def get_ctx(_user):
                try:
                    return engine_now().ctx.value
                except BaseException as e:       # noqa: BLE001  (no exception crosses the C frame: the open fails instead)
                    got["err"] = e
                    return None
            cb = _lib.GET_CTX(get_ctx)
            try:
                _lib.check(self.L.pf_pangenome_open_device_cb(C.byref(o), cb, None, C.byref(self.h)))
                self.one_pass = True
            except _lib.PanfeedHipError as e:
                if "err" in got:
                    raise got["err"] from e
                if e.status != _lib.ERR_CAPACITY:
                    raise
        self.store_words = None
        if debug_hostsink:
            # tests: the one-pass reader without a device; the store it would have filled on the GPU comes back as an array
            st, nw = C.POINTER(C.c_uint64)(), C.c_uint64()
            _lib.check(self.L.pf_debug_open_hostsink(C.byref(o), C.byref(self.h), C.byref(st), C.byref(nw)))
            self.store_words = np.ctypeslib.as_array(st, shape=(max(int(nw.value), 1),))[:int(nw.value)].copy()
            self.L.pf_free_text(C.cast(st, C.c_void_p))
            self.one_pass = True
        if not self.one_pass:
            _lib.check(self.L.pf_pangenome_open(C.byref(o), C.byref(self.h)))
        info = _lib.PangenomeInfo()
        _lib.check(self.L.pf_pangenome_info(self.h, C.byref(info)))
        self.n_clusters, self.n_strains = int(info.n_clusters), int(info.n_strains)
        self.strains = [self.L.pf_pangenome_strain(self.h, i, 0).decode() for i in range(self.n_strains)]
        self.sorted_strains = [self.L.pf_pangenome_strain(self.h, i, 1).decode() for i in range(self.n_strains)]
        self.resident = self.one_pass
        if engine is not None and not self.one_pass:
            self.make_resident(engine_now())

    def make_resident(self, engine):
        """Upload every contig to `engine`'s GPU (2 bits per base, pf_genomes_upload) and switch the reader to
        by-reference records: a sequence that is pure A/C/G/T and not a target strain's is handed on as (contig,
        start, length, strand) and cut out -- or reverse-complemented -- by the device (pf_submit_gather); the host
        then does no per-base work for it.  Target strains and sequences with other letters stay text."""
        n = C.c_uint32()
        ptrs = C.POINTER(C.c_char_p)()
        lens = C.POINTER(C.c_uint64)()
        _lib.check(self.L.pf_pangenome_contigs(self.h, C.byref(n), C.byref(ptrs), C.byref(lens)))
        off = (C.c_uint64 * max(n.value, 1))()
        _lib.check(self.L.pf_genomes_upload(engine.ctx, n.value, ptrs, lens, off))
        _lib.check(self.L.pf_pangenome_set_store(self.h, off, n.value))
        self.resident = True
        self.n_contigs = int(n.value)
        self.genome_bases = int(sum(lens[i] for i in range(n.value)))
        return self

    def contig_layout(self):
        """(n, ptrs, lens, off): the contigs and the word offsets pf_genomes_upload gives them in the genome store --
        2 * ceil(len / 64) + 4 words each, one after the other (csrc/pf_api.hip: pf_genomes_upload)"""
        n = C.c_uint32()
        ptrs = C.POINTER(C.c_char_p)()
        lens = C.POINTER(C.c_uint64)()
        _lib.check(self.L.pf_pangenome_contigs(self.h, C.byref(n), C.byref(ptrs), C.byref(lens)))
        ln = np.ctypeslib.as_array(lens, shape=(n.value,)).astype(np.uint64) if n.value else np.zeros(0, np.uint64)
        words = 2 * ((ln + np.uint64(63)) // np.uint64(64)) + np.uint64(4)
        off = np.zeros(max(n.value, 1), dtype=np.uint64)
        off[1:n.value] = np.cumsum(words)[:-1] if n.value > 1 else off[1:n.value]
        return n, ptrs, lens, off

    def assign_store(self):
        """Switch the reader to by-reference records BEFORE the genomes are in HBM: the store's layout follows from the
        contig lengths alone, so the packer thread can go ahead while `upload_store` runs on another thread."""
        n, _ptrs, _lens, off = self.contig_layout()
        self._store_off = off
        _lib.check(self.L.pf_pangenome_set_store(self.h, off.ctypes.data_as(C.POINTER(C.c_uint64)), n.value))
        self.resident = True
        self.n_contigs = int(n.value)
        return self

    def upload_store(self, engine):
        """the upload that belongs to `assign_store` (blocking; the library packs 2 bits per base on the device)"""
        n, ptrs, lens, off = self.contig_layout()
        got = (C.c_uint64 * max(n.value, 1))()
        _lib.check(self.L.pf_genomes_upload(engine.ctx, n.value, ptrs, lens, got))
        if n.value and not np.array_equal(np.ctypeslib.as_array(got)[:n.value], off[:n.value]):
            raise RuntimeError("pf_genomes_upload laid the contigs out differently from Pangenome.contig_layout")
        self.genome_bases = int(sum(lens[i] for i in range(n.value)))
        return self

    def weights(self):
        """per processed cluster (table rows that pass --genes, table order): its number of gene entries, paralogs
        counted -- what a sharded run balances its contiguous ranges by (`distributed.shard_range`)"""
        n = C.c_uint32()
        _lib.check(self.L.pf_pangenome_weights(self.h, 0, None, C.byref(n)))
        w = np.zeros(max(n.value, 1), dtype=np.uint32)
        _lib.check(self.L.pf_pangenome_weights(self.h, n.value, w.ctypes.data_as(C.c_void_p), C.byref(n)))
        return w[:n.value]

    def set_range(self, first, count):
        """restrict the reader to processed clusters [first, first + count) and rewind it to `first`"""
        _lib.check(self.L.pf_pangenome_set_range(self.h, int(first), int(count)))

    def close(self, wait=True):
        """give the reader back; wait=False: on a thread of the library's own (the call returns at once)"""
        if self.h:
            (self.L.pf_pangenome_close if wait else self.L.pf_pangenome_close_async)(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def take_log(self):
        """the warnings iter_gene_clusters logs (input.py:396-397 and parse_gff's :326-329), one per line"""
        s = self.L.pf_pangenome_take_log(self.h)
        return s.decode() if s else ""

    def next_records(self, max_clusters):
        """(handle, view) of the next rows of the table, or None at the end; free with `free_records`."""
        h = C.c_void_p()
        v = _lib.RecordsView()
        _lib.check(self.L.pf_pangenome_next(self.h, int(max_clusters), C.byref(h), C.byref(v)))
        if v.n_clusters == 0:
            self.L.pf_records_free(h)
            return None
        return h, v

    def free_records(self, h):
        self.L.pf_records_free(h)

    def records(self, max_clusters=64):
        """Reference-shaped records `(gene_sequences, idx, clusterpresab)` (input.py:468) -- for tests and for
        callers that want the mirror API; the batch path below never builds them."""
        if self.resident:
            raise RuntimeError("records() yields text; after make_resident() use batches()")
        while True:
            got = self.next_records(max_clusters)
            if got is None:
                return
            h, v = got
            try:
                for ci in range(v.n_clusters):
                    yield _record_from_view(v, ci, self.n_strains)
            finally:
                self.free_records(h)

    def batches(self, klength, canon, W, max_clusters=256, want_strand=True, first_ordinal=0, with_names=False):
        """HostBatch per `max_clusters` rows of the table, packed by the library straight from the reader's buffers.
        with_names: also fill hb.sorted_strains / hb.presab (Python lists per cluster; nothing on the GPU path reads them).
        (The reader runs up to three blocks of rows ahead of what has been yielded: a generator that is dropped before its
        end leaves the reader further down the table than its last batch -- `set_range` rewinds it.)"""
        k = int(klength)
        ordinal = int(first_ordinal)
        # Two stages on two threads: the reader cuts the next rows' records (pf_pangenome_next: 8 M gene look-ups per
        # 8 000 x 1 000 pangenome, bound by memory latency, a serial merge at its end) while this thread packs the rows
        # before them (pf_pack_records + the arrays' way into the HostBatch).  Both stages belong to the run's critical
        # path -- the GPU waits for the packer -- and neither fills the host on its own.
        import queue
        import threading
        q = queue.Queue(maxsize=2)
        stop = threading.Event()

        def reader():
            try:
                while not stop.is_set():
                    got = self.next_records(max_clusters)
                    sent = False
                    while not stop.is_set() and not sent:
                        try:
                            q.put(got, timeout=0.1)
                            sent = True
                        except queue.Full:
                            pass
                    if not sent:                            # (the consumer went away: the block in hand goes back)
                        if got is not None:
                            self.free_records(got[0])
                        return
                    if got is None:
                        return
            except BaseException as e:      # noqa: BLE001  (handed to the consumer)
                q.put(e)
        rt = threading.Thread(target=reader, name="panfeed-reader")
        rt.start()
        try:
            yield from self._batches_from(q, k, canon, W, want_strand, ordinal, with_names)
        finally:
            stop.set()
            while rt.is_alive():                                # drain what the reader still hands over
                try:
                    got = q.get(timeout=0.05)
                    if isinstance(got, tuple):
                        self.free_records(got[0])
                except queue.Empty:
                    pass
            rt.join()
            while True:
                try:
                    got = q.get_nowait()
                    if isinstance(got, tuple):
                        self.free_records(got[0])
                except queue.Empty:
                    break

    def _batches_from(self, q, k, canon, W, want_strand, ordinal, with_names):
        L = self.L
        while True:
            got = q.get()
            if got is None:
                return
            if isinstance(got, BaseException):
                raise got
            h, v = got
            try:
                nc = int(v.n_clusters)
                if v.W > W:
                    raise ValueError(f"{self.n_strains} strains exceed the context's max_strains")
                hb = HostBatch(k=k, canon=bool(canon), W=W)
                pin = _lib.PackIn(nc, v.n_seqs, v.seq, v.comp, C.cast(v.seq_len, C.c_void_p),
                                  C.cast(v.seq_col, C.c_void_p),
                                  C.cast(v.seq_target, C.c_void_p) if self.targets else None,
                                  C.cast(v.cluster_seq_off, C.c_void_p), k, int(bool(canon)), W,
                                  int(bool(want_strand)),
                                  C.cast(v.seq_src_off, C.c_void_p) if self.resident else None,
                                  C.cast(v.seq_src_start, C.c_void_p) if self.resident else None,
                                  C.cast(v.seq_flags, C.c_void_p) if self.resident else None)
                ph = C.c_void_p()
                _lib.check(L.pf_pack_records(C.byref(pin), C.byref(ph)))
                try:
                    seq_cluster = np.repeat(np.arange(nc), np.diff(_np(v.cluster_seq_off, nc + 1, np.uint32)))

                    def seq_ref(q, v=v, seq_cluster=seq_cluster):
                        ci = int(seq_cluster[q])
                        strain = v.cluster_strain[int(v.cluster_strain_off[ci]) + int(v.seq_strain[q])].decode()
                        return ci, strain, _seqinfo(v, q)
                    _fill_from_packed(L, hb, ph, nc, seq_ref)
                finally:
                    L.pf_packed_free(ph)
                pres_words = _np(v.cluster_presab, nc * v.W, np.uint32).reshape(nc, v.W)
                hb.cluster_presab = np.zeros((nc, W), dtype=np.uint32)
                hb.cluster_presab[:, :v.W] = pres_words
                hb.cluster_nstrains = _np(v.cluster_nstrains, nc, np.uint32).copy()
                hb.cluster_npresab = _np(v.cluster_npresab, nc, np.uint32).copy()
                hb.cluster_ordinal = np.arange(ordinal, ordinal + nc, dtype=np.uint64)
                ordinal += nc
                for ci in range(nc):
                    hb.idx.append(v.cluster_name[ci].decode())
                    if not with_names:
                        continue
                    a, b = int(v.cluster_strain_off[ci]), int(v.cluster_strain_off[ci + 1])
                    hb.sorted_strains.append(sorted(v.cluster_strain[j].decode() for j in range(a, b)))
                    npres = int(v.cluster_npresab[ci])
                    bits = np.unpackbits(pres_words[ci].view(np.uint8), bitorder="little")[:npres]
                    hb.presab.append(bits.astype(np.int64))
            finally:
                self.free_records(h)
            yield hb


def _np(ptr, n, dtype):
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype) if n else np.zeros(0, dtype=dtype)


def _seqinfo(v, q):
    n = int(v.seq_len[q])
    return Seqinfo(C.string_at(v.seq[q], n).decode("latin-1"), C.string_at(v.comp[q], n).decode("latin-1"),
                   v.id[q].decode(), v.chromosome[q].decode(), int(v.seq_start[q]), int(v.seq_end[q]),
                   int(v.seq_strand[q]), int(v.seq_offset[q]))


def _record_from_view(v, ci, n_strains):
    a, b = int(v.cluster_strain_off[ci]), int(v.cluster_strain_off[ci + 1])
    gs = {v.cluster_strain[j].decode(): [] for j in range(a, b)}
    keys = list(gs)
    for q in range(int(v.cluster_seq_off[ci]), int(v.cluster_seq_off[ci + 1])):
        gs[keys[int(v.seq_strain[q])]].append(_seqinfo(v, q))
    npres = int(v.cluster_npresab[ci])
    words = _np(v.cluster_presab, (ci + 1) * v.W, np.uint32)[ci * v.W:]
    presab = np.unpackbits(words.view(np.uint8), bitorder="little")[:npres].astype(np.int64)
    return gs, v.cluster_name[ci].decode(), presab
