"""Host pipeline of a whole run: files on disk -> the three output files.

Stands where the reference's reader / worker / writer processes do (/root/reference/panfeed/__main__.py:39-81 and the
serial loop :350-356): the native reader hands out table rows (host threads), the next batch is packed while the GPU
works on the current one (`Engine.run_batches`), the texts of a finished batch are written -- and, under `compress`,
deflated on all host threads -- by a writer thread while the GPU has the next batch.  Output order is the table order
(the reference's --cores 1 order) whatever finishes first."""
import os
import queue
import threading

from .engine import Engine, OwnedText
from .native_input import Pangenome
from .output import create_hash_files, create_kmer_stroi, write_headers


class _Columns:
    """what write_headers needs of the reference's `genepres` (panfeed.py:116-129)"""

    def __init__(self, columns):
        self.columns = list(columns)


def _peek_n_strains(presence_absence):
    """strain columns of the panaroo table, from its header record alone (input.py:188-191: index_col=0, the columns
    'Non-unique Gene name' and 'Annotation' dropped); 0 when the file cannot be read that way"""
    import csv
    try:
        with open(presence_absence, newline="") as fh:
            header = next(csv.reader(fh))
        return sum(1 for h in header[1:] if h not in ("Non-unique Gene name", "Annotation"))
    except Exception:       # noqa: BLE001
        return 0


def run_files(presence_absence, gffdir, output, fastadir=None, klength=31, canon=True, consider_missing=False,
              patfilt=True, maf=0.01, upstream=0, downstream=0, downstream_start_codon=False, targets=(), genes=None,
              compress=False, multiple_files=False, batch_clusters=256, resident=True, device_text=True, device=0,
              max_items=0, pattern_capacity=0, overlap=True, one_pass=True):
    """One directory of outputs (`kmers.tsv`, `kmers_to_hashes.tsv`, `hashes_to_patterns.tsv`, `.gz` under
    `compress`; under `multiple_files` one such directory per gene cluster, `<output>/<cluster>/`, the pattern set
    starting empty in each: `panfeed.py:35-43,153-167`) from a panaroo table and a directory (or file of files) of GFFs.  Option names and meaning follow
    the reference's (`__main__.py:86-186`); `patfilt` is what `pattern_hasher` receives (`--no-filter` inverted,
    `__main__.py:283-297`).  one_pass (with resident): the genomes go to the GPU as their files are read
    (pf_pangenome_open_device) instead of being read into host strings first and uploaded afterwards.
    Returns a dict of counters."""
    import time as _time
    t_start = _time.perf_counter()
    if os.path.isdir(output):                       # the reference refuses an existing directory (input.py:213-216)
        raise FileExistsError(f"Output directory {output} already exists; remove it or change the output path")
    os.makedirs(output)
    targets = tuple(targets or ())

    def make_engine(n_strains):
        return Engine(klength=klength, canon=canon, consider_missing=consider_missing, patfilt=patfilt, maf=maf,
                      multiple_files=multiple_files, max_strains=max(32, (n_strains + 31) // 32 * 32),
                      stroi=set(targets), device=device,
                      # work items in flight = scratch slices (1.9 MB each at 1 000 strains): sized for the batches of this
                      # run, not the library's default of 2 048 -- creating and freeing 4 GB of scratch was a third of a
                      # one-second run's time
                      max_items=max_items or max(512, 2 * int(batch_clusters)), pattern_capacity=pattern_capacity)

    # the context (stream, tables, ~1 GB of scratch: 7 ms) is made on a thread of its own while the reader opens the
    # pangenome; it needs the number of strains, which the table's header line says
    early = {}
    n_peek = _peek_n_strains(presence_absence) if overlap else 0
    from . import _lib
    _lib.load()        # once, on this thread, before two threads could both make the process's first call to it

    def early_engine():
        t0 = _time.perf_counter()
        try:
            early["eng"] = make_engine(n_peek)
        except Exception as e:       # noqa: BLE001  (made again below, where the error belongs)
            early["err"] = e
        early["s"] = _time.perf_counter() - t0
    et = None
    pg = None
    if n_peek and resident and one_pass:
        # one pass over the input: the genomes go into the context's store as their files are read; the context is made on
        # its thread while the reader parses the table and is asked for when the first genome needs it
        et = threading.Thread(target=early_engine, name="panfeed-context")
        et.start()

        def engine_when_needed():
            et.join()
            if "err" in early:
                raise early["err"]
            return early["eng"]
        try:
            pg = Pangenome(presence_absence, gffdir, fastadir, upstream, downstream, downstream_start_codon,
                           targets=targets, genes=genes, engine=engine_when_needed)
        except BaseException:
            et.join()
            if "eng" in early:
                early["eng"].close()
            raise
        et.join()
        if "eng" in early and early["eng"].max_strains < pg.n_strains:       # (the header was not what the reader made of it: the classic way)
            pg.close()
            early.pop("eng").close()
            pg = None
        et = None
    if pg is None:
        if n_peek:
            et = threading.Thread(target=early_engine, name="panfeed-context")
            et.start()
        try:
            pg = Pangenome(presence_absence, gffdir, fastadir, upstream, downstream, downstream_start_codon, targets=targets,
                           genes=genes)
        except BaseException:
            if et is not None:
                et.join()
                if "eng" in early:
                    early["eng"].close()
            raise
    eng = None
    uploader = None
    stats = {"clusters": 0, "instances": 0, "kept_kmers": 0, "patterns": 0, "device_ms": 0.0, "bytes": 0}
    stages = {"open_parse_s": _time.perf_counter() - t_start, "write_busy_s": 0.0}
    try:
        t0 = _time.perf_counter()
        if et is not None:
            et.join()
        if et is not None or pg.resident:
            eng = early.get("eng")
            if eng is not None and eng.max_strains < pg.n_strains:        # (the header was not what the reader made of it)
                eng.close()
                eng = None
        if eng is None:
            eng = make_engine(pg.n_strains)
        stages["context_s"] = early.get("s", 0.0) if "eng" in early and eng is early["eng"] else _time.perf_counter() - t0
        stages["context_wait_s"] = _time.perf_counter() - t0
        t0 = _time.perf_counter()
        upload_err = []
        stages["genome_upload_s"] = 0.0
        stages["one_pass_ingest"] = bool(pg.one_pass)
        if resident and not pg.resident:
            # The genome store's layout follows from the contig lengths: the reader switches to by-reference records at
            # once and the packer thread starts on the first batches while the contigs go up on a thread of their own
            # (the library packs them to 2 bits per base on the device); the first pf_submit waits for that thread.
            if not overlap:
                pg.make_resident(eng)
                stages["genome_upload_s"] = _time.perf_counter() - t0
        if resident and overlap and not pg.resident:
            pg.assign_store()

            def upload():
                t1 = _time.perf_counter()
                try:
                    pg.upload_store(eng)
                except Exception as e:       # noqa: BLE001
                    upload_err.append(e)
                stages["genome_upload_s"] = _time.perf_counter() - t1
            uploader = threading.Thread(target=upload, name="panfeed-genomes")
            uploader.start()

        def wait_for_genomes():
            if uploader is not None:
                uploader.join()
            if upload_err:
                raise upload_err[0]
        cols = _Columns(pg.strains)
        if multiple_files:
            kmer_stroi = hash_pat = kmer_hash = None
        else:
            kmer_stroi = create_kmer_stroi(output, compress)
            hash_pat, kmer_hash = create_hash_files(output, compress)
            write_headers(hash_pat, kmer_hash, cols)
        q = queue.Queue(maxsize=4)
        # text the GPU wrote lives in two pinned blocks used alternately: a batch may only be rendered once the batch
        # before the previous one has been written out
        slots = threading.Semaphore(2)
        failed = []

        def put(fh, data):
            # text the GPU wrote arrives as bytes: straight into the file's binary layer (or the gzip writer)
            if isinstance(data, str):
                fh.write(data)
            elif isinstance(data, OwnedText):        # engine.OwnedText: the library's block, written where it lies
                try:
                    put(fh, data.view)
                finally:
                    data.release()
            elif len(data):
                raw = getattr(fh, "buffer", None)
                if raw is not None:
                    fh.flush()
                    raw.write(data)
                else:
                    fh.write(data)

        def write_one(o):
            if not multiple_files:
                put(kmer_stroi, o.kmers_tsv)
                put(kmer_hash, o.kmers_to_hashes)
                put(hash_pat, o.hashes_to_patterns)
                return
            for idx, kt, kh, hp in o.per_cluster:
                path = os.path.join(output, idx)
                os.makedirs(path, exist_ok=True)
                ks = create_kmer_stroi(path, compress)
                ks.write(kt)
                ks.close()
                f_hp, f_kh = create_hash_files(path, compress)
                write_headers(f_hp, f_kh, cols)
                f_hp.write(hp)
                f_kh.write(kh)
                f_hp.close()
                f_kh.close()

        def writer():
            while True:
                o = q.get()
                if o is None:
                    return
                try:
                    if not failed:
                        tw = _time.perf_counter()
                        write_one(o)
                        stages["write_busy_s"] += _time.perf_counter() - tw
                except Exception as e:          # keep draining so that the producer never blocks on a dead writer
                    failed.append(e)
                finally:
                    slots.release()

        wt = threading.Thread(target=writer, name="panfeed-writer")
        wt.start()
        try:
            # (target strains' rows go to kmers.tsv block by block as they leave the device, from this thread: the file is
            # the writer thread's only when a batch's rows come as one object -- the host renderers' path)
            sink = (lambda blk: put(kmer_stroi, blk)) if (device_text and not multiple_files) else None
            batches = eng.run_pangenome(pg, batch_clusters=batch_clusters, device_text=device_text,
                                        before_first_submit=wait_for_genomes, targets_sink=sink)
            while True:
                slots.acquire()
                o = next(batches, None)
                if o is None:
                    break
                stats["clusters"] += o.stats.get("clusters", 0)
                stats["instances"] += o.stats.get("instances", 0)
                stats["kept_kmers"] += o.stats.get("kept_kmers", 0)
                stats["patterns"] = o.stats.get("patterns", stats["patterns"])
                stats["device_ms"] += o.timing.get("total_ms", 0.0)
                stats["bytes"] += (len(o.kmers_tsv) + len(o.kmers_to_hashes) + len(o.hashes_to_patterns) +
                                   o.stats.get("kmers_tsv_streamed", 0))
                q.put(o)
        finally:
            q.put(None)
            wt.join()
            for fh in (kmer_stroi, kmer_hash, hash_pat):
                if fh is not None:
                    fh.close()
        if failed:
            raise failed[0]
        wait_for_genomes()      # a run that submitted nothing (empty table, --genes matching nothing) still reports a failed upload
        stats["log"] = pg.take_log()
        # where the wall time went: opening + parsing the inputs, creating the context, uploading the genomes, then the
        # overlapped stages of the batches (Engine.run_batches: read + pack on its thread, pf_submit = upload + kernels,
        # text = device text + D2H or fetch + host renderers) and the writer thread's busy time
        stages.update(getattr(eng, "stages", {}))
        stages["total_s"] = _time.perf_counter() - t_start
        stats["stages"] = stages
        return stats
    finally:
        if uploader is not None:
            uploader.join()                  # (an error on the way: the upload reads the reader's contigs)
        t0 = _time.perf_counter()
        pg.close(wait=False)                 # the run is over: the reader's memory goes back in the background
        t1 = _time.perf_counter()
        if eng is not None:
            eng.close()
        if "stages" in stats:                                             # reader and context given back
            stats["stages"]["close_reader_s"] = t1 - t0
            stats["stages"]["close_context_s"] = _time.perf_counter() - t1
