"""Batched host driver of the HIP hot path: packs cluster records, calls libpanfeed_hip through
ctypes, and renders the three TSV bodies exactly as the reference writes them
(/root/reference/panfeed/panfeed.py:104-107, 177, 187, 208, 223).

One ``Engine`` = one panfeed run on one GPU: it owns the run-global pattern set
(panfeed.py:149-150) in device memory, so clusters must be fed in processing order.
"""
import ctypes as C

import numpy as np

from . import _lib
from .packing import build_batch_native, decode_keys, maf_tables

KMERS_TSV_HEADER = ("cluster\tstrain\tfeature_id\tcontig\tfeature_strand\tcontig_start\tcontig_end\t"
                    "gene_start\tgene_end\tstrand\tk-mer\n")           # input.py:243
KMERS_TO_HASHES_HEADER = "cluster\tk-mer\thashed_pattern\n"            # panfeed.py:127


def hashes_to_patterns_header(strains):
    return "hashed_pattern" + "".join(f"\t{s}" for s in sorted(strains)) + "\n"   # panfeed.py:120-123


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def _take(ptr, n):
    """n bytes at `ptr` as `bytes` (ctypes.string_at takes a C int: texts beyond 2 GiB need this)"""
    p = ptr.value if isinstance(ptr, C.c_void_p) else ptr
    if not n:
        return b""
    return bytes(memoryview((C.c_char * int(n)).from_address(p)))


class OwnedText:
    """a block of text the library malloc'd (pf_free_text), handed on without a copy: `view` for the writer, `release()`
    (or garbage collection) to give it back"""

    def __init__(self, L, ptr, n):
        self._L, self._p, self._n = L, (ptr.value if isinstance(ptr, C.c_void_p) else ptr), int(n)
        self.view = memoryview((C.c_char * self._n).from_address(self._p)).cast("B") if self._n else memoryview(b"")

    def __len__(self):
        return self._n

    def __bytes__(self):
        return bytes(self.view)

    def release(self):
        if self._p:
            self.view = memoryview(b"")
            self._L.pf_free_text(C.c_void_p(self._p))
            self._p = None

    def __del__(self):
        try:
            self.release()
        except Exception:   # noqa: BLE001  (interpreter shutdown)
            pass


class DeviceText:
    """Text the GPU wrote and still holds (pf_render_kmers_tsv_device): handed out block by block through pinned memory
    (`chunks`, each block valid until the next is asked for), or as one `bytes` (`bytes(x)`).  Belongs to the engine's
    last submit: use it before the next one."""

    def __init__(self, engine, nbytes):
        self._eng, self._n = engine, int(nbytes)

    def __len__(self):
        return self._n

    def chunks(self, max_bytes=64 << 20):
        eng, off = self._eng, 0
        ptr, nb = C.c_void_p(), C.c_uint64()
        while off < self._n:
            _lib.check(eng.L.pf_device_text_chunk(eng.ctx, off, int(max_bytes), C.byref(ptr), C.byref(nb)))
            n = int(nb.value)
            yield memoryview((C.c_char * n).from_address(ptr.value)).cast("B")
            off += n

    def __bytes__(self):
        out = bytearray(self._n)
        at = 0
        for blk in self.chunks():
            out[at:at + len(blk)] = blk
            at += len(blk)
        return bytes(out)


def _view(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype) if ptr else np.zeros(0, dtype=dtype)


class BatchOutput:
    """Texts of one batch (bodies only, no headers) plus counters."""

    def __init__(self):
        self.kmers_tsv = ""
        self.kmers_to_hashes = ""
        self.hashes_to_patterns = ""
        self.per_cluster = []   # multiple_files: [(idx, kmers_tsv, kmers_to_hashes, hashes_to_patterns)]
        self.stats = {}
        self.timing = {}


class Engine:
    def __init__(self, klength=31, canon=True, consider_missing=False, patfilt=True, maf=0.01,
                 multiple_files=False, max_strains=1024, stroi=(), device=0, pattern_capacity=0,
                 max_items=0, dedup=True, unit_dedup=True, key_binning=True, device_plan=False):
        self.L = _lib.load()
        self.k = int(klength)
        self.canon = bool(canon)
        self.consider_missing = bool(consider_missing)
        self.patfilt = bool(patfilt)
        self.maf = float(maf)
        self.multiple_files = bool(multiple_files)
        self.max_strains = int(max_strains)
        self.W = (self.max_strains + 31) // 32
        self.stroi = stroi if stroi else ()
        self._lo, self._hi = maf_tables(self.maf, self.max_strains)
        o = _lib.Opts(self.k, int(self.canon), int(self.consider_missing), int(self.patfilt),
                      int(self.multiple_files), self.max_strains,
                      self._lo.ctypes.data_as(C.POINTER(C.c_uint32)), self._hi.ctypes.data_as(C.POINTER(C.c_uint32)),
                      int(pattern_capacity), int(max_items),
                      (0 if dedup else _lib.FLAG_NO_DEDUP) | (0 if unit_dedup else _lib.FLAG_NO_UNIT_DEDUP) |
                      (0 if key_binning else _lib.FLAG_NO_KEY_BINNING) | (_lib.FLAG_DEVICE_PLAN if device_plan else 0))
        self.ctx = C.c_void_p()
        _lib.check(self.L.pf_create(C.byref(self.ctx), int(device), C.byref(o)))
        self.next_ordinal = 0
        self._md5_b64 = {}      # pattern id -> 24-char hash string (patterns seen so far)
        self.n_patterns = 0

    def close(self):
        if getattr(self, "ctx", None):
            self.L.pf_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ device calls
    def submit_host_batch(self, hb):
        if self.consider_missing and not np.array_equal(hb.cluster_nstrains, hb.cluster_npresab):
            # the reference stops in init_presabs_vector (panfeed.py:19): a boolean mask of another length
            ci = int(np.flatnonzero(np.asarray(hb.cluster_nstrains) != np.asarray(hb.cluster_npresab))[0])
            raise IndexError(f"cluster {hb.idx[ci]}: boolean index did not match indexed array along axis 0; size of "
                             f"axis is {int(hb.cluster_nstrains[ci])} but size of corresponding boolean axis is "
                             f"{int(hb.cluster_npresab[ci])}")
        if self.multiple_files:
            # the library starts every batch with an empty pattern pool (ids are cluster-local, panfeed.py:165)
            self._md5_b64 = {}
            self.n_patterns = 0
        b = _lib.Batch()
        b.n_clusters = hb.n_clusters
        b.n_segs = len(hb.seg_len)
        b.n_words = len(hb.packed)
        b.on_device = 0
        b.packed = _ptr(hb.packed)
        b.seg_word_off = _ptr(hb.seg_word_off)
        b.seg_len = _ptr(hb.seg_len)
        b.seg_sample = _ptr(hb.seg_sample)
        b.seg_ord_base = _ptr(hb.seg_ord_base)
        b.cluster_seg_off = _ptr(hb.cluster_seg_off)
        b.cluster_nstrains = _ptr(hb.cluster_nstrains)
        b.cluster_npresab = _ptr(hb.cluster_npresab)
        b.cluster_presab = _ptr(hb.cluster_presab)
        b.cluster_ordinal = _ptr(hb.cluster_ordinal)
        b.n_extra = len(hb.extra_ord)
        b.extra_cluster = _ptr(hb.extra_cluster)
        b.extra_ord = _ptr(hb.extra_ord)
        b.extra_bits = _ptr(hb.extra_bits)
        if hb.n_strand_words:
            b.seg_strand_off = _ptr(hb.seg_strand_off)
            b.n_strand_words = hb.n_strand_words
        res = _lib.Result()
        if hb.gather_src_off is not None:
            # segments are ranges of the genomes resident in HBM; b.packed = the few the host packed itself
            g = _lib.Gather(int(hb.n_words_dev), hb.gather_src_off.ctypes.data, hb.gather_src_start.ctypes.data,
                            hb.gather_src_flags.ctypes.data)
            _lib.check(self.L.pf_submit_gather(self.ctx, C.byref(b), C.byref(g), C.byref(res)))
        else:
            _lib.check(self.L.pf_submit(self.ctx, C.byref(b), C.byref(res)))
        return res

    def fetch(self):
        res = _lib.Result()
        _lib.check(self.L.pf_fetch(self.ctx, C.byref(res)))
        return res

    def timing(self):
        t = _lib.Timing()
        _lib.check(self.L.pf_get_timing(self.ctx, C.byref(t)))
        return {f: getattr(t, f) for f, _ in _lib.Timing._fields_ if f != "reserved"}

    # ------------------------------------------------------------------ one batch, text out
    def run(self, records):
        """cluster_cutter + pattern_hasher over `records` (in processing order)."""
        records = list(records)
        hb = build_batch_native(records, self.k, self.canon, self.W, stroi=self.stroi,
                                first_ordinal=self.next_ordinal)
        self.next_ordinal += len(records)
        self.submit_host_batch(hb)
        res = self.fetch()
        return self._render(hb, res)

    def run_stream(self, records, batch_clusters=256, prefetch=2, defer_patterns=False, device_text=False):
        """Generator over BatchOutput: packs batch i+1 (host threads, pf_pack_records) while the GPU works on
        batch i and the caller writes batch i-1 -- the reference's reader / workers / writer pipeline
        (__main__.py:39-81,299-344) with the GPU in the workers' place and a deterministic order
        (always the --cores 1 order, whatever finishes first).  device_text: the two big files' text written by the GPU
        and handed over as bytes-like blocks (see run_batches) instead of `str`."""
        import itertools
        it = iter(records)

        def host_batches():
            ordinal = self.next_ordinal
            while True:
                chunk = list(itertools.islice(it, batch_clusters))
                if not chunk:
                    return
                yield build_batch_native(chunk, self.k, self.canon, self.W, stroi=self.stroi, first_ordinal=ordinal)
                ordinal += len(chunk)
        return self.run_batches(host_batches(), prefetch, device_text, defer_patterns=defer_patterns)

    def run_pangenome(self, pangenome, batch_clusters=256, prefetch=2, device_text=False, defer_patterns=False,
                      before_first_submit=None, targets_sink=None):
        """run_stream fed by the native reader (native_input.Pangenome): table rows -> records -> packed batches
        without leaving the library, then the GPU; yields BatchOutput in table order."""
        if tuple(sorted(pangenome.targets)) != tuple(sorted(self.stroi or ())):
            raise ValueError("the pangenome reader and the engine were given different target strains")
        return self.run_batches(pangenome.batches(self.k, self.canon, self.W, max_clusters=batch_clusters,
                                                  first_ordinal=self.next_ordinal), prefetch, device_text,
                                defer_patterns=defer_patterns, before_first_submit=before_first_submit,
                                targets_sink=targets_sink)

    def run_batches(self, host_batches, prefetch=2, device_text=False, defer_patterns=False, before_first_submit=None,
                    targets_sink=None):
        """GPU over an iterator of HostBatch, the next ones being prepared on one host thread meanwhile.
        device_text: kmers_to_hashes / hashes_to_patterns of a batch without target-strain rows come back as
        memoryviews of text the GPU wrote (render_device; valid until the batch after the next one has been rendered)
        instead of str; batches with kmers.tsv rows and --multiple-files runs keep the host renderers.
        defer_patterns: leave hashes_to_patterns empty -- a rank of a sharded run renders its pattern rows after the
        run-global merge (`render_pattern_rows`).
        before_first_submit: called once, right before the first pf_submit (pipeline.run_files joins the thread that
        uploads the genomes there: the packer has been at work on the first batches meanwhile).
        targets_sink (with device_text): called with every block of a batch's kmers.tsv rows as it leaves the device
        (a bytes-like view, valid during the call) instead of the rows being gathered into `out.kmers_tsv` -- with every
        strain a target a batch's rows are tens of gigabytes (BASELINE configs[4]'s second pass: 256 clusters x 5 000
        samples = 150 GB), which no host buffer should hold; `out.stats["kmers_tsv_streamed"]` says how many bytes went."""
        from concurrent.futures import ThreadPoolExecutor
        import time as _time
        it = iter(host_batches)
        # where the wall time of this run goes (seconds; bench.py's end-to-end leg): the packer thread's busy time
        # (read + pack, overlapped with the GPU), the time this thread waited for it, pf_submit (upload + kernels),
        # the text stage (device text + its copy to the host, or pf_fetch + the host renderers)
        st = self.stages = {"pack_busy_s": 0.0, "pack_wait_s": 0.0, "submit_s": 0.0, "device_ms": 0.0, "text_s": 0.0,
                            "batches": 0}

        def pack_next():
            t0 = _time.perf_counter()
            hb = next(it, None)
            st["pack_busy_s"] += _time.perf_counter() - t0
            return hb

        with ThreadPoolExecutor(max_workers=1) as pool:      # one packer thread keeps the record order
            pending = [pool.submit(pack_next) for _ in range(max(1, prefetch))]
            while pending:
                t0 = _time.perf_counter()
                hb = pending.pop(0).result()
                st["pack_wait_s"] += _time.perf_counter() - t0
                if hb is None:
                    break
                pending.append(pool.submit(pack_next))
                self.next_ordinal = int(hb.cluster_ordinal[-1]) + 1 if hb.n_clusters else self.next_ordinal
                if before_first_submit is not None:
                    t0 = _time.perf_counter()
                    before_first_submit()
                    before_first_submit = None
                    st["first_submit_wait_s"] = _time.perf_counter() - t0
                t0 = _time.perf_counter()
                res = self.submit_host_batch(hb)
                t1 = _time.perf_counter()
                st["submit_s"] += t1 - t0
                st["batches"] += 1
                texts = None
                if device_text and not self.multiple_files:
                    try:
                        texts = self.render_device(hb, defer_patterns)
                    except _lib.PanfeedHipError as e:
                        # a batch the text kernels cannot lay out (too many passes, an over-long cluster name)
                        # goes through the host renderers instead
                        if e.status not in (_lib.ERR_CAPACITY, _lib.ERR_ARG):
                            raise
                if texts is not None:
                    out = BatchOutput()
                    out.kmers_to_hashes, out.hashes_to_patterns = texts
                    # target strains: their rows written by the GPU too (the next submit reuses the device's text
                    # block, so the rows come over now)
                    streamed = 0
                    if hb.n_targets and targets_sink is not None:
                        for blk in self.render_targets_device(hb).chunks():
                            targets_sink(blk)
                            streamed += len(blk)
                        out.kmers_tsv = b""
                    else:
                        out.kmers_tsv = bytes(self.render_targets_device(hb)) if hb.n_targets else b""
                    out.stats = {"clusters": int(hb.n_clusters), "instances": int(hb.n_instances), "kmers_tsv_streamed": streamed,
                                 "device_instances": int(res.n_instances), "unique_kmers": int(res.n_unique),
                                 "kept_kmers": int(res.n_kept), "new_patterns": int(res.n_new_patterns),
                                 "patterns": self.pattern_count()}
                    out.timing = self.timing()
                else:
                    out = self._render(hb, self.fetch(), defer_patterns)
                    if targets_sink is not None and out.kmers_tsv and not self.multiple_files:
                        # (the rows of this batch must not overtake, or be overtaken by, the streamed rows of its neighbours)
                        out.stats["kmers_tsv_streamed"] = len(out.kmers_tsv)
                        targets_sink(out.kmers_tsv)
                        out.kmers_tsv = ""
                st["text_s"] += _time.perf_counter() - t1
                st["device_ms"] += out.timing.get("total_ms", 0.0)
                yield out

    def result_checksum(self):
        """(k-mer rows, cluster rows, kept) checksums of the last submit, computed on the device (pf_result_checksum)"""
        out = (C.c_uint64 * 3)()
        _lib.check(self.L.pf_result_checksum(self.ctx, out))
        return tuple(int(x) for x in out)

    def pattern_count(self):
        n = C.c_uint64()
        _lib.check(self.L.pf_pattern_count(self.ctx, C.byref(n)))
        return int(n.value)

    def _hash_strings(self, res):
        p0, p1 = self.n_patterns, int(res.n_patterns)
        if p1 > p0:
            md5 = _view(res.pattern_md5, p1 * 16, np.uint8).reshape(p1, 16)
            buf = C.create_string_buffer(24)
            for pid in range(p0, p1):
                d = md5[pid].tobytes()
                self.L.pf_b64_digest(d, buf)
                self._md5_b64[pid] = buf.raw[:24].decode()
            self.n_patterns = p1
        return self._md5_b64

    def _pattern_row(self, res, pid):
        """'<hash>\\t<patterntup>\\n'  (panfeed.py:181-187, 217-223)"""
        W = self.W
        nk = int(res.pattern_n[pid])
        n = nk & 0x7FFFFFFF
        bits = _view(res.pattern_bits, (pid + 1) * W, np.uint32)[pid * W:(pid + 1) * W]
        b = (bits[np.arange(n) >> 5] >> (np.arange(n, dtype=np.uint32) & 31)) & 1 if n else np.zeros(0, np.uint32)
        cells = [str(int(x)) for x in b]
        if self.consider_missing and not (nk >> 31) and res.pattern_nan:
            nan = _view(res.pattern_nan, (pid + 1) * W, np.uint32)[pid * W:(pid + 1) * W]
            isn = (nan[np.arange(n) >> 5] >> (np.arange(n, dtype=np.uint32) & 31)) & 1
            cells = ["" if isn[i] else cells[i] for i in range(n)]
        return self._md5_b64[pid] + "\t" + "\t".join(cells) + "\n"

    def _render(self, hb, res, defer_patterns=False):
        out = BatchOutput()
        C_ = hb.n_clusters
        first_seen = _view(res.pattern_first_seen, int(res.n_patterns), np.uint64)
        new_ids = _view(res.new_pattern_id, int(res.n_new_patterns), np.uint32)

        # kmers_to_hashes.tsv body (panfeed.py:177, 208): rendered by the library's host threads
        names = (C.c_char_p * max(C_, 1))(*[s.encode() for s in hb.idx])
        extra = (C.c_char_p * len(hb.extra_keys))(*[k.encode() for k in hb.extra_keys]) if hb.extra_keys else None
        buf, nb = C.c_void_p(), C.c_uint64()
        ends = (C.c_uint64 * max(C_, 1))()
        _lib.check(self.L.pf_render_kmers_to_hashes(self.ctx, names, extra, C.byref(buf), C.byref(nb), ends))
        kh_bytes = _take(buf, nb.value)
        self.L.pf_free_text(buf)
        kh_all = kh_bytes.decode()
        kh_parts = []
        if self.multiple_files:
            prev = 0
            for ci in range(C_):
                kh_parts.append(kh_bytes[prev:ends[ci]].decode())
                prev = ends[ci]
        # hashes_to_patterns.tsv body: new patterns in first-seen order (panfeed.py:179-187, 210-223)
        hp_by_cluster = [[] for _ in range(C_)]
        if self.multiple_files:
            hashes = self._hash_strings(res)
            ord0 = int(hb.cluster_ordinal[0]) if C_ else 0
            for pid in new_ids:
                pid = int(pid)
                ci = int(first_seen[pid] >> np.uint64(32)) - ord0
                hp_by_cluster[ci].append(self._pattern_row(res, pid))
            hp_all = "".join("".join(x) for x in hp_by_cluster)
        elif defer_patterns:
            hp_all = ""
        else:
            buf, nb = C.c_void_p(), C.c_uint64()
            _lib.check(self.L.pf_render_hashes_to_patterns(self.ctx, C.byref(buf), C.byref(nb)))
            hp_all = _take(buf, nb.value).decode()
            self.L.pf_free_text(buf)
        # kmers.tsv body (panfeed.py:90-107): one row per window of the target strains' sequences
        kt_by_cluster = [[] for _ in range(C_)]
        if hb.n_targets:
            if self.multiple_files:
                by_cl = {}
                for meta in hb.targets:
                    by_cl.setdefault(meta.cluster, []).append(meta)
                for ci, metas in by_cl.items():
                    kt_by_cluster[ci].append(self._render_targets(hb, metas))
            else:
                # written by the GPU (kt_text_kernel); sequences with a letter outside A/C/G/T by the host renderer, in place
                kt_by_cluster[0].append(bytes(self.render_targets_device(hb)).decode())
        if self.multiple_files:
            for ci in range(C_):
                out.per_cluster.append((hb.idx[ci], "".join(kt_by_cluster[ci]), kh_parts[ci],
                                        "".join(hp_by_cluster[ci])))
        out.kmers_to_hashes = kh_all
        out.hashes_to_patterns = hp_all
        out.kmers_tsv = "".join("".join(x) for x in kt_by_cluster)
        out.stats = {"clusters": int(hb.n_clusters), "instances": int(hb.n_instances),
                     "device_instances": int(res.n_instances),
                     "unique_kmers": int(res.n_unique), "kept_kmers": int(res.n_kept),
                     "new_patterns": int(res.n_new_patterns), "patterns": int(res.n_patterns)}
        out.timing = self.timing()
        return out

    def render_device(self, hb, defer_patterns=False):
        """(kmers_to_hashes body, hashes_to_patterns body) of the last submit as memoryviews over pinned host memory,
        the text having been written by the GPU (pf_render_device): no pf_fetch, no bytes -> str; valid until the
        render after the next one.  kmers.tsv rows (target strains) still come from `_render` / `_render_targets`."""
        C_ = hb.n_clusters
        names = (C.c_char_p * max(C_, 1))(*[s.encode() for s in hb.idx])
        extra = "".join(hb.extra_keys).encode("latin-1") if hb.extra_keys else None
        kh, kn, hp, hn = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_uint64()
        _lib.check(self.L.pf_render_device_ex(self.ctx, names, extra, len(hb.extra_keys),
                                              _lib.RENDER_NO_PATTERN_ROWS if defer_patterns else 0,
                                              C.byref(kh), C.byref(kn), C.byref(hp), C.byref(hn)))
        mk = (lambda p, n: memoryview((C.c_char * n).from_address(p)).cast("B") if n else memoryview(b""))
        return mk(kh.value, kn.value), mk(hp.value, hn.value)

    # ------------------------------------------------------------------ run-global patterns (multi-GPU)
    def export_patterns(self):
        """(md5 uint8 [n,16], first_seen uint64 [n]) of every pattern this engine holds, as numpy arrays"""
        n = C.c_uint64()
        p_md5 = C.POINTER(C.c_uint8)()
        p_fs = C.POINTER(C.c_uint64)()
        _lib.check(self.L.pf_export_patterns(self.ctx, C.byref(n), C.byref(p_md5), C.byref(p_fs)))
        n = int(n.value)
        if n == 0:
            return np.zeros((0, 16), dtype=np.uint8), np.zeros(0, dtype=np.uint64)
        md5 = np.ctypeslib.as_array(p_md5, shape=(n * 16,)).reshape(n, 16).copy()
        return md5, np.ctypeslib.as_array(p_fs, shape=(n,)).copy()

    def render_pattern_rows(self, pids, chunk_bytes=256 << 20):
        """hashes_to_patterns.tsv rows of the patterns `pids` (any ids of the pool, in this order), written on the
        device (pf_render_pattern_rows); yields `bytes` blocks of about chunk_bytes."""
        pids = np.ascontiguousarray(pids, dtype=np.uint32)
        per_row = 24 + 2 * self.max_strains + 1
        step = max(1, int(chunk_bytes // per_row))
        for a in range(0, len(pids), step):
            part = pids[a:a + step]
            txt, nb = C.c_void_p(), C.c_uint64()
            _lib.check(self.L.pf_render_pattern_rows(self.ctx, part.ctypes.data_as(C.c_void_p), len(part),
                                                     C.byref(txt), C.byref(nb)))
            yield _take(txt, nb.value)

    def render_targets_device(self, hb):
        """kmers.tsv rows of every target sequence of `hb` (the last submit), written on the device
        (pf_render_kmers_tsv_device): a DeviceText.  Same bytes as `_render_targets(hb, hb.targets)`."""
        import time as _time
        t0 = _time.time()
        if hb.target_table is not None and hb.target_table.resolve is not None and hb._targets is None:
            rec, n, keep, held = self._marshal_table(hb, hb.target_table)
        else:
            rec, n, keep = self._marshal_targets(hb, hb.targets)
            held = None
        try:
            arr = C.cast(rec.ctypes.data, C.POINTER(_lib.TargetSeq))
            nb = C.c_uint64()
            t1 = _time.time()
            _lib.check(self.L.pf_render_kmers_tsv_device(self.ctx, arr, n, C.byref(nb)))
        finally:
            if held is not None:
                from .packing import _release_held
                _release_held(held)
        del keep
        self.render_targets_timing = {"marshal_s": t1 - t0, "render_s": _time.time() - t1, "copy_s": 0.0}
        return DeviceText(self, nb.value)

    _TS = np.dtype([("cluster", "u8"), ("strain", "u8"), ("id", "u8"), ("chromosome", "u8"), ("sequence", "u8"),
                    ("compsequence", "u8"), ("len", "u4"), ("strand", "i4"), ("start", "i8"), ("end", "i8"),
                    ("offset", "i8"), ("n_segs", "u4"), ("n_ambig", "u4"), ("seg_index", "u8"), ("seg_start", "u8"),
                    ("seg_nwin", "u8"), ("ambig_pos", "u8"), ("ambig_used", "u8"), ("ambig_key", "u8")])

    def _marshal_table(self, hb, tt):
        """pf_target_seq records straight from the packer's flat target arrays (packing.TargetTable): no per-sequence
        Python objects, numpy columns only.  Returns (records, n, what must stay alive, held string references)."""
        from .packing import _seqinfo_columns
        TS = self._TS
        assert TS.itemsize == C.sizeof(_lib.TargetSeq)
        n = len(tt)
        rec = np.zeros(max(n, 1), dtype=TS)
        keep = [rec]
        held = None
        if not n:
            return rec, 0, keep, None

        def block(strings):
            blob = ("\0".join(strings) + "\0").encode()
            ends = np.flatnonzero(np.frombuffer(blob, dtype=np.uint8) == 0)
            if len(ends) != len(strings):
                raise ValueError("a NUL byte inside a name or sequence")
            starts = np.concatenate(([0], ends[:-1] + 1)).astype(np.uint64)
            keep.append(blob)
            return starts + np.uint64(C.cast(C.c_char_p(blob), C.c_void_p).value), (ends - starts.astype(np.int64)).astype(np.uint32)

        ci, strains, seqs = tt.resolve(tt.t_seq)
        rec["cluster"][:n] = block(list(hb.idx))[0][ci]
        for field, values in (("strain", [str(x) for x in strains]), ("id", [str(s.id) for s in seqs]),
                              ("chromosome", [str(s.chromosome) for s in seqs])):
            uniq = {}
            idx = np.fromiter((uniq.setdefault(v, len(uniq)) for v in values), dtype=np.int64, count=n)
            rec[field][:n] = block(list(uniq))[0][idx]
        fast = _seqinfo_columns(seqs)
        if fast is not None:
            a_seq, a_comp, a_len, held = fast
            rec["sequence"][:n], rec["compsequence"][:n], rec["len"][:n] = a_seq, a_comp, a_len
        else:
            addr, lens = block([s.sequence for s in seqs])
            rec["sequence"][:n], rec["len"][:n] = addr, lens
            addr, lens2 = block([s.compsequence for s in seqs])
            rec["compsequence"][:n] = addr
            if not np.array_equal(lens, lens2):
                raise ValueError("sequence and compsequence of different lengths")
        rec["strand"][:n] = np.fromiter((int(s.strand) for s in seqs), dtype=np.int32, count=n)
        rec["start"][:n] = np.fromiter((int(s.start) for s in seqs), dtype=np.int64, count=n)
        rec["end"][:n] = np.fromiter((int(s.end) for s in seqs), dtype=np.int64, count=n)
        rec["offset"][:n] = np.fromiter((int(s.offset) for s in seqs), dtype=np.int64, count=n)
        so, ao = tt.t_so.astype(np.uint64), tt.t_ao.astype(np.uint64)
        cols = [np.ascontiguousarray(x, dtype=np.uint32) if len(x) else np.zeros(1, np.uint32) for x in (tt.t_si, tt.t_ss, tt.t_sn)]
        keep += cols
        rec["n_segs"][:n] = np.diff(tt.t_so.astype(np.int64))
        for field, col in zip(("seg_index", "seg_start", "seg_nwin"), cols):
            rec[field][:n] = np.uint64(col.ctypes.data) + so[:-1] * np.uint64(4)
        rec["n_ambig"][:n] = np.diff(tt.t_ao.astype(np.int64))
        ap = np.ascontiguousarray(tt.t_ap, dtype=np.uint32) if len(tt.t_ap) else np.zeros(1, np.uint32)
        au = np.ascontiguousarray(tt.t_au, dtype=np.int8) if len(tt.t_au) else np.zeros(1, np.int8)
        akeys = bytes(tt.akeys) + b"\0"
        kaddr = np.uint64(C.cast(C.c_char_p(akeys), C.c_void_p).value) + np.arange(max(len(tt.t_ap), 1), dtype=np.uint64) * np.uint64(tt.k)
        kaddr = np.ascontiguousarray(kaddr)
        keep += [ap, au, akeys, kaddr]
        rec["ambig_pos"][:n] = np.uint64(ap.ctypes.data) + ao[:-1] * np.uint64(4)
        rec["ambig_used"][:n] = np.uint64(au.ctypes.data) + ao[:-1]
        rec["ambig_key"][:n] = np.uint64(kaddr.ctypes.data) + ao[:-1] * np.uint64(8)
        return rec, n, keep, held

    def _render_targets(self, hb, metas, as_bytes=False, owned=False):
        """kmers.tsv rows of `metas` (packing.SeqMeta, in order) through pf_render_kmers_tsv: str, `bytes` (as_bytes) or
        the library's own block without a copy (owned: an OwnedText)"""
        import time as _time
        t0 = _time.time()
        rec, n, keep = self._marshal_targets(hb, metas)
        arr = C.cast(rec.ctypes.data, C.POINTER(_lib.TargetSeq))
        buf, nb = C.c_void_p(), C.c_uint64()
        sso = hb.seg_strand_off.ctypes.data_as(C.c_void_p) if hb.n_strand_words else None
        t1 = _time.time()
        _lib.check(self.L.pf_render_kmers_tsv(self.ctx, arr, n, sso, C.byref(buf), C.byref(nb)))
        t2 = _time.time()
        del keep
        if owned:
            self.render_targets_timing = {"marshal_s": t1 - t0, "render_s": t2 - t1, "copy_s": 0.0}
            return OwnedText(self.L, buf, nb.value)
        text = _take(buf, nb.value)
        self.L.pf_free_text(buf)
        # where the time of the last call went: Python marshalling of the records, the library's renderer, the copy out
        self.render_targets_timing = {"marshal_s": t1 - t0, "render_s": t2 - t1, "copy_s": _time.time() - t2}
        return text if as_bytes else text.decode()

    def _marshal_targets(self, hb, metas):
        """pf_target_seq records of `metas` (packing.SeqMeta): (records, n, what must stay alive while they are used)"""
        n = len(metas)
        # The C structs are filled column by column (numpy) instead of sequence by sequence (ctypes): every kind of
        # string goes into ONE NUL-separated block whose pieces are addressed by offset, the few per-sequence lists
        # (ACGT runs, non-ACGT windows) into flat arrays.  23 -> ~4 us per target sequence.
        TS = self._TS
        assert TS.itemsize == C.sizeof(_lib.TargetSeq)
        rec = np.zeros(max(n, 1), dtype=TS)
        keep = [rec]

        def block(strings):
            """addresses of `strings` laid out NUL-terminated in one bytes object"""
            blob = ("\0".join(strings) + "\0").encode()
            ends = np.flatnonzero(np.frombuffer(blob, dtype=np.uint8) == 0)
            if len(ends) != len(strings):
                raise ValueError("a NUL byte inside a name or sequence")
            starts = np.concatenate(([0], ends[:-1] + 1)).astype(np.uint64)
            keep.append(blob)
            base = C.cast(C.c_char_p(blob), C.c_void_p).value
            return starts + np.uint64(base), (ends - starts.astype(np.int64)).astype(np.uint32)

        if n:
            seqs = [m.seq for m in metas]
            # names repeat (a cluster, a strain, a contig): one copy each
            for field, values in (("cluster", [hb.idx[m.cluster] for m in metas]), ("strain", [str(m.strain) for m in metas]),
                                  ("id", [str(s.id) for s in seqs]), ("chromosome", [str(s.chromosome) for s in seqs])):
                uniq = {}
                idx = np.fromiter((uniq.setdefault(v, len(uniq)) for v in values), dtype=np.int64, count=n)
                addr, _ = block(list(uniq))
                rec[field][:n] = addr[idx]
            addr, lens = block([s.sequence for s in seqs])
            rec["sequence"][:n], rec["len"][:n] = addr, lens
            addr, lens2 = block([s.compsequence for s in seqs])
            rec["compsequence"][:n] = addr
            if not np.array_equal(lens, lens2):
                raise ValueError("sequence and compsequence of different lengths")
            rec["strand"][:n] = np.fromiter((int(s.strand) for s in seqs), dtype=np.int32, count=n)
            rec["start"][:n] = np.fromiter((int(s.start) for s in seqs), dtype=np.int64, count=n)
            rec["end"][:n] = np.fromiter((int(s.end) for s in seqs), dtype=np.int64, count=n)
            rec["offset"][:n] = np.fromiter((int(s.offset) for s in seqs), dtype=np.int64, count=n)
            # ACGT runs: (segment index, first window, windows) per run, flat
            nseg = np.fromiter((len(m.segs) for m in metas), dtype=np.int64, count=n)
            flat = np.array([x for m in metas for t in m.segs for x in t], dtype=np.uint32).reshape(-1, 3)
            cols = [np.ascontiguousarray(flat[:, j]) for j in range(3)] if len(flat) else [np.zeros(1, np.uint32)] * 3
            keep += cols
            soff = (np.concatenate(([0], np.cumsum(nseg)[:-1])) * 4).astype(np.uint64)
            rec["n_segs"][:n] = nseg
            for field, col in zip(("seg_index", "seg_start", "seg_nwin"), cols):
                rec[field][:n] = np.uint64(col.ctypes.data) + soff
            # windows with a non-ACGT letter (rare): position, strand used and text, as the packer worked them out
            namb = np.fromiter((len(m.ambig) for m in metas), dtype=np.int64, count=n)
            rec["n_ambig"][:n] = namb
            tot = int(namb.sum())
            ap = np.zeros(max(tot, 1), dtype=np.uint32)
            au = np.zeros(max(tot, 1), dtype=np.int8)
            akeys = []
            at = 0
            for i in np.flatnonzero(namb):
                m = metas[int(i)]
                for q in sorted(m.ambig):
                    ap[at], au[at] = q, m.ambig[q][1]
                    akeys.append(m.ambig[q][0])
                    at += 1
            kaddr = block(akeys)[0] if akeys else np.zeros(1, dtype=np.uint64)
            kaddr = np.ascontiguousarray(kaddr)
            keep += [ap, au, kaddr]
            aoff = np.concatenate(([0], np.cumsum(namb)[:-1])).astype(np.uint64)
            rec["ambig_pos"][:n] = np.uint64(ap.ctypes.data) + aoff * np.uint64(4)
            rec["ambig_used"][:n] = np.uint64(au.ctypes.data) + aoff
            rec["ambig_key"][:n] = np.uint64(kaddr.ctypes.data) + aoff * np.uint64(8)
        return rec, n, keep
