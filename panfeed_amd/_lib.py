"""ctypes binding of libpanfeed_hip.so (include/panfeed_hip.h).

The library is the product: there is no CPU fallback.  Loading fails loudly when the shared
object is missing (build it with ``python -c 'import __graft_entry__ as g; g.build()'``), and
every entry point raises ``PanfeedHipError`` with the library's message on a non-zero status.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpanfeed_hip.so")


class PanfeedHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"libpanfeed_hip: status {status}: {message}")
        self.status = status


class Opts(C.Structure):
    _fields_ = [("klength", C.c_uint32), ("canon", C.c_uint32), ("consider_missing", C.c_uint32),
                ("patfilt", C.c_uint32), ("multiple_files", C.c_uint32), ("max_strains", C.c_uint32),
                ("maf_lo", C.POINTER(C.c_uint32)), ("maf_hi", C.POINTER(C.c_uint32)),
                ("pattern_capacity", C.c_uint64), ("max_items", C.c_uint32), ("flags", C.c_uint32)]


class Batch(C.Structure):
    _fields_ = [("n_clusters", C.c_uint32), ("n_segs", C.c_uint32), ("n_words", C.c_uint64),
                ("on_device", C.c_uint32), ("reserved", C.c_uint32),
                ("packed", C.c_void_p), ("seg_word_off", C.c_void_p), ("seg_len", C.c_void_p),
                ("seg_sample", C.c_void_p), ("seg_ord_base", C.c_void_p), ("cluster_seg_off", C.c_void_p),
                ("cluster_nstrains", C.c_void_p), ("cluster_npresab", C.c_void_p), ("cluster_presab", C.c_void_p),
                ("cluster_ordinal", C.c_void_p),
                ("n_extra", C.c_uint32), ("reserved2", C.c_uint32),
                ("extra_cluster", C.c_void_p), ("extra_ord", C.c_void_p), ("extra_bits", C.c_void_p),
                ("seg_strand_off", C.c_void_p), ("n_strand_words", C.c_uint64)]


class Result(C.Structure):
    _fields_ = [("n_instances", C.c_uint64), ("n_unique", C.c_uint64), ("n_kept", C.c_uint64),
                ("n_new_patterns", C.c_uint64), ("W", C.c_uint32), ("key_words", C.c_uint32),
                ("cluster_kmer_off", C.POINTER(C.c_uint64)), ("cluster_kmer_cnt", C.POINTER(C.c_uint32)),
                ("cluster_pattern", C.POINTER(C.c_uint32)), ("cluster_unique", C.POINTER(C.c_uint32)),
                ("kmer_key", C.POINTER(C.c_uint64)), ("kmer_pattern", C.POINTER(C.c_uint32)),
                ("new_pattern_id", C.POINTER(C.c_uint32)), ("n_patterns", C.c_uint64),
                ("pattern_md5", C.POINTER(C.c_uint8)), ("pattern_bits", C.POINTER(C.c_uint32)),
                ("pattern_nan", C.POINTER(C.c_uint32)), ("pattern_n", C.POINTER(C.c_uint32)),
                ("pattern_first_seen", C.POINTER(C.c_uint64)), ("strand_bits", C.POINTER(C.c_uint64))]


class TargetSeq(C.Structure):
    _fields_ = [("cluster", C.c_char_p), ("strain", C.c_char_p), ("id", C.c_char_p), ("chromosome", C.c_char_p),
                ("sequence", C.c_char_p), ("compsequence", C.c_char_p), ("len", C.c_uint32), ("strand", C.c_int32),
                ("start", C.c_int64), ("end", C.c_int64), ("offset", C.c_int64),
                ("n_segs", C.c_uint32), ("n_ambig", C.c_uint32),
                ("seg_index", C.POINTER(C.c_uint32)), ("seg_start", C.POINTER(C.c_uint32)),
                ("seg_nwin", C.POINTER(C.c_uint32)), ("ambig_pos", C.POINTER(C.c_uint32)),
                ("ambig_used", C.POINTER(C.c_int8)), ("ambig_key", C.POINTER(C.c_char_p))]


class PackIn(C.Structure):
    _fields_ = [("n_clusters", C.c_uint32), ("n_seqs", C.c_uint32),
                ("seq", C.POINTER(C.c_char_p)), ("comp", C.POINTER(C.c_char_p)), ("seq_len", C.c_void_p),
                ("seq_col", C.c_void_p), ("seq_target", C.c_void_p), ("cluster_seq_off", C.c_void_p),
                ("klength", C.c_uint32), ("canon", C.c_uint32), ("W", C.c_uint32), ("want_strand", C.c_uint32),
                ("seq_src_off", C.c_void_p), ("seq_src_start", C.c_void_p), ("seq_flags", C.c_void_p)]


class Gather(C.Structure):
    _fields_ = [("n_words", C.c_uint64), ("src_off", C.c_void_p), ("src_start", C.c_void_p), ("src_flags", C.c_void_p)]


class PackedView(C.Structure):
    _fields_ = [("n_segs", C.c_uint32), ("n_extra", C.c_uint32), ("n_targets", C.c_uint32), ("reserved", C.c_uint32),
                ("n_words", C.c_uint64), ("n_strand_words", C.c_uint64), ("n_instances", C.c_uint64),
                ("packed", C.POINTER(C.c_uint64)), ("seg_word_off", C.POINTER(C.c_uint64)),
                ("seg_len", C.POINTER(C.c_uint32)), ("seg_sample", C.POINTER(C.c_uint32)),
                ("seg_ord_base", C.POINTER(C.c_uint32)), ("seg_strand_off", C.POINTER(C.c_uint32)),
                ("cluster_seg_off", C.POINTER(C.c_uint32)), ("cluster_ninst", C.POINTER(C.c_uint64)),
                ("extra_cluster", C.POINTER(C.c_uint32)), ("extra_ord", C.POINTER(C.c_uint32)),
                ("extra_bits", C.POINTER(C.c_uint32)), ("extra_keys", C.POINTER(C.c_char)),
                ("target_seq", C.POINTER(C.c_uint32)), ("target_seg_off", C.POINTER(C.c_uint32)),
                ("target_seg_index", C.POINTER(C.c_uint32)), ("target_seg_start", C.POINTER(C.c_uint32)),
                ("target_seg_nwin", C.POINTER(C.c_uint32)), ("target_ambig_off", C.POINTER(C.c_uint32)),
                ("target_ambig_pos", C.POINTER(C.c_uint32)), ("target_ambig_used", C.POINTER(C.c_int8)),
                ("target_ambig_keys", C.POINTER(C.c_char)),
                ("n_words_dev", C.c_uint64), ("gather_src_off", C.POINTER(C.c_uint64)),
                ("gather_src_start", C.POINTER(C.c_uint32)), ("gather_src_flags", C.POINTER(C.c_uint32))]


class PangenomeOpts(C.Structure):
    _fields_ = [("presence_absence_csv", C.c_char_p), ("n_genomes", C.c_uint32), ("reserved", C.c_uint32),
                ("genome_names", C.POINTER(C.c_char_p)), ("gff_paths", C.POINTER(C.c_char_p)),
                ("fasta_paths", C.POINTER(C.c_char_p)),
                ("upstream", C.c_uint32), ("downstream", C.c_uint32), ("downstream_start_codon", C.c_uint32),
                ("raise_missing", C.c_uint32),
                ("target_strains", C.POINTER(C.c_char_p)), ("n_targets", C.c_uint32), ("n_genes", C.c_uint32),
                ("gene_list", C.POINTER(C.c_char_p))]


class PangenomeInfo(C.Structure):
    _fields_ = [("n_clusters", C.c_uint32), ("n_strains", C.c_uint32), ("next_cluster", C.c_uint32), ("reserved", C.c_uint32)]


class RecordsView(C.Structure):
    _fields_ = [("n_clusters", C.c_uint32), ("n_seqs", C.c_uint32), ("W", C.c_uint32), ("reserved", C.c_uint32),
                ("seq", C.POINTER(C.c_char_p)), ("comp", C.POINTER(C.c_char_p)), ("id", C.POINTER(C.c_char_p)),
                ("chromosome", C.POINTER(C.c_char_p)),
                ("seq_len", C.POINTER(C.c_uint32)), ("seq_col", C.POINTER(C.c_uint32)), ("seq_strain", C.POINTER(C.c_uint32)),
                ("seq_target", C.POINTER(C.c_uint8)), ("seq_strand", C.POINTER(C.c_int32)),
                ("seq_start", C.POINTER(C.c_int64)), ("seq_end", C.POINTER(C.c_int64)), ("seq_offset", C.POINTER(C.c_int64)),
                ("cluster_seq_off", C.POINTER(C.c_uint32)), ("cluster_name", C.POINTER(C.c_char_p)),
                ("cluster_nstrains", C.POINTER(C.c_uint32)), ("cluster_npresab", C.POINTER(C.c_uint32)),
                ("cluster_presab", C.POINTER(C.c_uint32)), ("cluster_strain_off", C.POINTER(C.c_uint32)),
                ("cluster_strain", C.POINTER(C.c_char_p)),
                ("seq_src_off", C.POINTER(C.c_uint64)), ("seq_src_start", C.POINTER(C.c_uint32)),
                ("seq_flags", C.POINTER(C.c_uint32))]


class Timing(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("scan_ms", C.c_float), ("rows_ms", C.c_float), ("emit_ms", C.c_float),
                ("scan_launches", C.c_uint32), ("n_items", C.c_uint32), ("n_retried", C.c_uint32),
                ("n_dedup_clusters", C.c_uint32), ("scan_packed_bytes", C.c_uint64),
                ("dedup_ms", C.c_float), ("patrows_ms", C.c_float), ("md5_ms", C.c_float), ("finish_ms", C.c_float),
                ("n_wide_clusters", C.c_uint32), ("n_binned_clusters", C.c_uint32),
                ("n_scratch_grown", C.c_uint32), ("n_device_planned", C.c_uint32),
                ("n_side_launches", C.c_uint32), ("reserved", C.c_uint32)]

FLAG_NO_DEDUP = 1
FLAG_NO_UNIT_DEDUP = 2
FLAG_NO_KEY_BINNING = 4
FLAG_DEVICE_PLAN = 8


# every symbol include/panfeed_hip.h declares
EXPORTS = ["pf_last_error", "pf_version", "pf_device_count", "pf_create", "pf_destroy", "pf_reset_patterns",
           "pf_submit", "pf_fetch", "pf_get_timing", "pf_export_patterns", "pf_export_patterns_dev",
           "pf_merge_patterns", "pf_merge_patterns_padded", "pf_pattern_count", "pf_debug_limit_pattern_slots", "pf_debug_limit_alloc", "pf_result_checksum", "pf_dev_alloc", "pf_dev_free",
           "pf_dev_upload", "pf_dev_download", "pf_synth_expand", "pf_pack_acgt", "pf_b64_digest",
           "pf_render_kmers_to_hashes", "pf_render_hashes_to_patterns", "pf_render_kmers_tsv", "pf_render_kmers_tsv_device", "pf_device_text_chunk", "pf_free_text",
           "pf_pack_records", "pf_packed_view", "pf_packed_free",
           "pf_pangenome_open", "pf_pangenome_open_device", "pf_pangenome_open_device_cb", "pf_debug_open_hostsink", "pf_pangenome_close", "pf_pangenome_info", "pf_pangenome_strain",
           "pf_pangenome_take_log", "pf_pangenome_next", "pf_records_free", "pf_pangenome_contigs",
           "pf_pangenome_set_store", "pf_genomes_upload", "pf_genomes_clear", "pf_submit_gather", "pf_gzip_members", "pf_render_device",
           "pf_render_device_ex", "pf_render_pattern_rows", "pf_pangenome_weights", "pf_pangenome_set_range",
           "pf_rowfilter_create", "pf_rowfilter_scan", "pf_rowfilter_stats", "pf_rowfilter_destroy",
           "pf_py_str_addresses", "pf_pangenome_close_async", "pf_py_seqinfo_columns", "pf_py_release"]

RENDER_NO_PATTERN_ROWS = 1
GET_CTX = C.CFUNCTYPE(C.c_void_p, C.c_void_p)     # pf_pangenome_open_device_cb: the context, when the first genome needs it
ERR_ARG, ERR_OOM, ERR_HIP, ERR_CAPACITY, ERR_STATE = -1, -2, -3, -4, -5

_lib = None
_load_lock = threading.Lock()


def load():
    """dlopen the HIP library; raises if it is not built -- never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    with _load_lock:
        return _load_locked()


def _load_locked():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build the HIP extension first "
                          "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    try:
        # one HIP runtime per process: PyTorch ships its own libamdhip64; if this library pulled in the system copy
        # first, a later `import torch` (device tensors for the RCCL merge) would bring a second runtime that finds
        # no GPU.  Importing torch first makes both resolve to the same loaded runtime.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        getattr(L, name)  # AttributeError here means header and library disagree
    L.pf_last_error.restype = C.c_char_p
    L.pf_version.restype = C.c_char_p
    L.pf_device_count.restype = C.c_int
    L.pf_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(Opts)]
    L.pf_destroy.argtypes = [C.c_void_p]
    L.pf_destroy.restype = None
    L.pf_reset_patterns.argtypes = [C.c_void_p]
    L.pf_submit.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(Result)]
    L.pf_fetch.argtypes = [C.c_void_p, C.POINTER(Result)]
    L.pf_get_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
    L.pf_export_patterns.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.POINTER(C.c_uint8)),
                                     C.POINTER(C.POINTER(C.c_uint64))]
    L.pf_export_patterns_dev.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    L.pf_pattern_count.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.pf_debug_limit_pattern_slots.argtypes = [C.c_void_p, C.c_uint64]
    L.pf_debug_limit_alloc.argtypes = [C.c_uint64, C.POINTER(C.c_uint64)]
    L.pf_result_checksum.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.pf_merge_patterns_padded.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64,
                                           C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]
    L.pf_merge_patterns.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p,
                                    C.POINTER(C.c_uint64)]
    L.pf_dev_alloc.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
    L.pf_dev_free.argtypes = [C.c_void_p, C.c_void_p]
    L.pf_dev_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.pf_dev_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.pf_synth_expand.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_uint32, C.c_void_p]
    L.pf_pack_acgt.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p]
    L.pf_pack_acgt.restype = C.c_uint64
    L.pf_b64_digest.argtypes = [C.c_void_p, C.c_char_p]
    L.pf_b64_digest.restype = None
    L.pf_render_kmers_to_hashes.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.pf_render_hashes_to_patterns.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.pf_render_kmers_tsv_device.argtypes = [C.c_void_p, C.POINTER(TargetSeq), C.c_uint32, C.POINTER(C.c_uint64)]
    L.pf_device_text_chunk.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.pf_render_kmers_tsv.argtypes = [C.c_void_p, C.POINTER(TargetSeq), C.c_uint32, C.c_void_p,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.pf_pack_records.argtypes = [C.POINTER(PackIn), C.POINTER(C.c_void_p)]
    L.pf_packed_view.argtypes = [C.c_void_p, C.POINTER(PackedView)]
    L.pf_packed_free.argtypes = [C.c_void_p]
    L.pf_packed_free.restype = None
    L.pf_pangenome_open.argtypes = [C.POINTER(PangenomeOpts), C.POINTER(C.c_void_p)]
    L.pf_pangenome_open_device.argtypes = [C.POINTER(PangenomeOpts), C.c_void_p, C.POINTER(C.c_void_p)]
    L.pf_pangenome_open_device_cb.argtypes = [C.POINTER(PangenomeOpts), GET_CTX, C.c_void_p, C.POINTER(C.c_void_p)]
    L.pf_debug_open_hostsink.argtypes = [C.POINTER(PangenomeOpts), C.POINTER(C.c_void_p), C.POINTER(C.POINTER(C.c_uint64)),
                                         C.POINTER(C.c_uint64)]
    L.pf_pangenome_close.argtypes = [C.c_void_p]
    L.pf_pangenome_close.restype = None
    L.pf_pangenome_close_async.argtypes = [C.c_void_p]
    L.pf_pangenome_close_async.restype = None
    L.pf_pangenome_info.argtypes = [C.c_void_p, C.POINTER(PangenomeInfo)]
    L.pf_pangenome_strain.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
    L.pf_pangenome_strain.restype = C.c_char_p
    L.pf_pangenome_take_log.argtypes = [C.c_void_p]
    L.pf_pangenome_take_log.restype = C.c_char_p
    L.pf_pangenome_next.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(RecordsView)]
    L.pf_pangenome_contigs.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_char_p)),
                                       C.POINTER(C.POINTER(C.c_uint64))]
    L.pf_pangenome_set_store.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32]
    L.pf_genomes_upload.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64),
                                    C.POINTER(C.c_uint64)]
    L.pf_genomes_clear.argtypes = [C.c_void_p]
    L.pf_render_device.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.pf_render_device_ex.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_char_p, C.c_uint64, C.c_uint32,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_uint64)]
    L.pf_render_pattern_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.pf_pangenome_weights.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
    L.pf_pangenome_set_range.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.pf_rowfilter_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
    L.pf_rowfilter_scan.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.POINTER(C.POINTER(C.c_uint64)),
                                    C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.pf_rowfilter_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
    L.pf_rowfilter_destroy.argtypes = [C.c_void_p]
    L.pf_rowfilter_destroy.restype = None
    L.pf_gzip_members.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.pf_submit_gather.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(Gather), C.POINTER(Result)]
    L.pf_records_free.argtypes = [C.c_void_p]
    L.pf_records_free.restype = None
    L.pf_free_text.argtypes = [C.c_void_p]
    L.pf_free_text.restype = None
    _lib = L
    return L


_pydll = None


def load_pydll():
    """the same library through a handle whose calls keep the GIL (ctypes.PyDLL): for pf_py_str_addresses, which calls
    back into the interpreter's C API"""
    global _pydll
    if _pydll is None:
        load()
        P = C.PyDLL(LIB_PATH)
        P.pf_py_str_addresses.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        P.pf_py_str_addresses.restype = C.c_int
        P.pf_py_seqinfo_columns.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_void_p] * 5
        P.pf_py_seqinfo_columns.restype = C.c_int
        P.pf_py_release.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        P.pf_py_release.restype = C.c_int
        _pydll = P
    return _pydll


def check(status):
    if status != 0:
        raise PanfeedHipError(status, load().pf_last_error().decode(errors="replace"))
