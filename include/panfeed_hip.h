/*
 * panfeed_hip.h -- C ABI of libpanfeed_hip.so: panfeed's per-gene-cluster k-mer extraction and
 * presence/absence pattern hashing hot path on MI355X (gfx950).
 *
 * The reference has no FFI: the path sits behind two Python callables,
 *   cluster_cutter  /root/reference/panfeed/panfeed.py:23-113
 *   pattern_hasher  /root/reference/panfeed/panfeed.py:132-235
 * bound with functools.partial at /root/reference/panfeed/__main__.py:277-297 and called at
 * :352-356 (serial) / :47,:81 (worker, writer).  This header is what a binding for those two call
 * sites talks to (ctypes stub: INTEGRATION.md; in-repo mirror: panfeed_amd/panfeed.py).
 *
 * Conventions: every function returns 0 on success or a negative pf_status; the message of the
 * last failure on the calling thread is pf_last_error().  No C++ exception crosses the ABI.  All
 * pointers are plain host pointers unless a field says "device".  One pf_ctx per GPU, used from
 * one host thread at a time.
 */
#ifndef PANFEED_HIP_H
#define PANFEED_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pf_ctx pf_ctx;

enum pf_status {
    PF_OK = 0,
    PF_ERR_ARG = -1,      /* bad argument / unsupported option */
    PF_ERR_OOM = -2,      /* host or device allocation failed */
    PF_ERR_HIP = -3,      /* a HIP runtime call failed */
    PF_ERR_CAPACITY = -4, /* a capacity that cannot grow was exceeded (work items per cluster, 2^31 patterns) */
    PF_ERR_STATE = -5     /* call order violated */
};

#define PF_MAX_K 126 /* 2 bits/base in at most four 63-bit key words */

/* Options of one run = the arguments functools.partial freezes at __main__.py:277-297. */
typedef struct {
    uint32_t klength;          /* -k ; 1..PF_MAX_K */
    uint32_t canon;            /* canon == True  (panfeed.py:69); 0 = --non-canonical */
    uint32_t consider_missing; /* consider_missing_cluster (panfeed.py:17-19, 193-196, 220-222) */
    uint32_t patfilt;          /* the value handed to pattern_hasher: the same-as-cluster filter
                                  runs when it is 0 (panfeed.py:202-204) */
    uint32_t multiple_files;   /* patterns set reset per cluster (panfeed.py:165) */
    uint32_t max_strains;      /* upper bound of len(cluster.keys()) and len(clusterpresab) */
    /* MAF filter (panfeed.py:190-200) as exact float64 semantics precomputed by the caller:
       a k-mer with `count` ones among `n` non-missing strains is kept iff
       maf_lo[n] <= count <= maf_hi[n]; arrays of max_strains+1 entries. */
    const uint32_t* maf_lo;
    const uint32_t* maf_hi;
    uint64_t pattern_capacity; /* initial slots of the run-global pattern table; 0 = default (2^24).  The table and
                                  its pool grow on demand (the reference's `patterns` is an unbounded set,
                                  panfeed.py:146-150): a batch that runs out of ids is re-run with a larger table */
    uint32_t max_items;        /* work items (cluster x key partition) in flight per internal sub-batch; 0 = default;
                                  clamped so that their scratch slices take at most half of the free device memory */
    uint32_t flags;            /* PF_FLAG_* */
} pf_opts;

/* Scan every segment even when a cluster holds identical copies of a sequence (same output, slower:
   by default only one representative per distinct sequence is scanned). */
#define PF_FLAG_NO_DEDUP 1u
/* With the shortcut above on: scan every 64-window unit of every distinct sequence even when distinct sequences of a
   cluster share it (same output, slower on clusters of many alleles: by default a unit that several alleles of a gene
   hold unchanged at the same place is scanned once for all of them). */
#define PF_FLAG_NO_UNIT_DEDUP 2u
/* A cluster whose distinct k-mers take three or more passes of the on-chip table (key partitions) has its windows sorted
   by partition first, so that every pass reads its own windows only.  With this flag every pass walks the whole cluster
   and keeps its share (same output, slower on clusters of many alleles). */
#define PF_FLAG_NO_KEY_BINNING 4u
/* Opt-in: the work items of a batch's simple clusters (at most 64 distinct sequences, one key partition, a fused finish
   class) are laid out by three small kernels right behind the dedup pass, and the host reads a 40-byte summary before it
   launches their scan (the reference's hand-off is a queue, __main__.py:39-52), building only the rest of the batch from
   the per-cluster records.  Same output.  What it is for: batches of a few thousand SIMPLE clusters that go through in one
   part, where the host's work lies on the critical path -- configs[1] (5 000 x 200) 1.337 -> 1.30 ms a pass, a rank's
   share of configs[3] (6 250 clusters) 2.53 -> 2.51 (profiles/r05/device_plan_ab_small_batches.txt).  Not the default: a
   50 000-cluster batch hides the host's work behind its other part's kernels (14.98 -> 15.18 ms with the plan), and in a
   batch that mixes simple clusters with many-allele ones the planned clusters' finish kernels run in front of the general
   path instead of beside it (the second stream). */
#define PF_FLAG_DEVICE_PLAN 8u

/*
 * One batch of gene clusters = the records iter_gene_clusters yields
 * (/root/reference/panfeed/input.py:455-468), already packed by the caller.
 *
 * A "segment" is a maximal run of A/C/G/T inside one Seqinfo.sequence (non-ACGT bases split a
 * sequence; the windows that contain one are the caller's slow path and come back in as
 * `extra_*` rows).  Packing: 2 bits per base (A=0 C=1 G=2 T=3), 32 bases per uint64 word, the first
 * base in bits 63:62; every segment starts on a 16-byte boundary; the buffer carries 16 bytes
 * of padding after the last segment (32 bytes when klength > 63: a lane reads up to four words past its window's
 * first one).
 *
 * Padding bits inside a segment's last 16 bytes are zero.  cluster_seg_off[0] == 0.
 * Inside a cluster the segments are sorted by seg_sample (stable).  A window starting at base
 * `pos` of a segment is instance number  seg_ord_base + pos  of the cluster in the reference's
 * iteration order (panfeed.py:54-64); in non-canonical mode the forward k-mer is instance
 * 2*(seg_ord_base+pos) and the reverse complement 2*(seg_ord_base+pos)+1 (panfeed.py:82-88).
 */
typedef struct {
    uint32_t n_clusters;
    uint32_t n_segs;
    uint64_t n_words;               /* uint64 words in `packed`, padding included */
    uint32_t on_device;             /* 1: every array below is a device pointer (bench / tests) */
    uint32_t reserved;
    const uint64_t* packed;
    const uint64_t* seg_word_off;   /* [n_segs]   word offset of the segment (even) */
    const uint32_t* seg_len;        /* [n_segs]   bases */
    const uint32_t* seg_sample;     /* [n_segs]   column in sorted(cluster.keys()) order (panfeed.py:47-49) */
    const uint32_t* seg_ord_base;   /* [n_segs] */
    const uint32_t* cluster_seg_off;  /* [n_clusters+1] */
    const uint32_t* cluster_nstrains; /* [n_clusters] len(cluster.keys()) */
    const uint32_t* cluster_npresab;  /* [n_clusters] len(clusterpresab) */
    const uint32_t* cluster_presab;   /* [n_clusters * W] clusterpresab as bits, W = ceil(max_strains/32) */
    const uint64_t* cluster_ordinal;  /* [n_clusters] position in the run's processing order */
    /* slow-path rows (k-mers containing a non-ACGT base), grouped by cluster, any order inside: */
    uint32_t n_extra;
    uint32_t reserved2;
    const uint32_t* extra_cluster;  /* [n_extra] batch-local cluster index, non-decreasing */
    const uint32_t* extra_ord;      /* [n_extra] first-occurrence instance number */
    const uint32_t* extra_bits;     /* [n_extra * W] presence bits */
    /* strand bits for positional rows (panfeed.py:69-75 used_strand; canonical mode only):
       seg_strand_off[s] = index of the first uint64 of segment s in the strand-bit stream, or
       0xFFFFFFFF to skip the segment; window pos -> bit (pos & 63) of word (pos >> 6);
       bit = 1 when the reverse complement is the canonical k-mer.  NULL = none wanted. */
    const uint32_t* seg_strand_off;
    uint64_t n_strand_words;
} pf_batch;

/* Result of pf_submit, valid until the next pf_submit / pf_destroy on the context. */
typedef struct {
    uint64_t n_instances;      /* trip count of panfeed.py:64 (x2 non-canonical), slow path excluded */
    uint64_t n_unique;         /* unique k-mers over all clusters (len(cluster_dict) summed) */
    uint64_t n_kept;           /* rows of kmers_to_hashes.tsv minus the per-cluster rows */
    uint64_t n_new_patterns;   /* rows this batch adds to hashes_to_patterns.tsv */
    uint32_t W;                /* uint32 words per presence row */
    uint32_t key_words;        /* KW = ceil(2k / 63): 1 (k <= 31), 2 (<= 63), 3 (<= 94), 4 (<= 126) */
    /* per cluster (batch order) */
    const uint64_t* cluster_kmer_off;  /* [n_clusters] first kept k-mer in kmer_* */
    const uint32_t* cluster_kmer_cnt;  /* [n_clusters] */
    const uint32_t* cluster_pattern;   /* [n_clusters] pattern id of the cluster row (panfeed.py:175-187) */
    const uint32_t* cluster_unique;    /* [n_clusters] len(cluster_dict) */
    /* per kept k-mer, in dict insertion order inside each cluster (panfeed.py:189) */
    const uint64_t* kmer_key;          /* [n_kept_total * key_words]: the k-mer's 2k-bit value (first base most
                                          significant), most significant word first, 63 bits per word; bit 63 of
                                          word 0 set: slow-path row, low 32 bits = index into the batch's extra_* arrays */
    const uint32_t* kmer_pattern;      /* [n_kept_total] pattern id */
    /* patterns first seen in this batch, sorted by first_seen = the order of hashes_to_patterns.tsv */
    const uint32_t* new_pattern_id;    /* [n_new_patterns] */
    /* run-global pattern pool, indexed by pattern id */
    uint64_t n_patterns;
    const uint8_t* pattern_md5;        /* [n_patterns * 16] md5 of the int64 / float64 image */
    const uint32_t* pattern_bits;      /* [n_patterns * W] */
    const uint32_t* pattern_nan;       /* [n_patterns * W] NaN positions (consider_missing), else NULL */
    const uint32_t* pattern_n;         /* [n_patterns] vector length; bit 31 set: int64 image (cluster row) */
    const uint64_t* pattern_first_seen;/* [n_patterns] cluster_ordinal << 32 | rank inside the cluster */
    const uint64_t* strand_bits;       /* [n_strand_words] or NULL */
} pf_result;

/* device-side timing of the last pf_submit, milliseconds (hipEvent on the context's stream) */
typedef struct {
    float total_ms;
    float scan_ms;     /* kmer_scan_kernel launches */
    float rows_ms;     /* rows_kernel */
    float emit_ms;     /* cluster_base_kernel + emit_kernel (+ extra_fill_kernel) */
    uint32_t scan_launches;
    uint32_t n_items;  /* (cluster, key-partition) work items scanned, retries included */
    uint32_t n_retried;/* clusters whose table overflowed and were re-run with more partitions */
    uint32_t n_dedup_clusters; /* clusters scanned through distinct-sequence representatives */
    uint64_t scan_packed_bytes; /* packed sequence bytes the scan kernels were asked to read */
    float dedup_ms;    /* cluster_dedup_kernel */
    float patrows_ms;  /* pattern_rows_kernel (emit_ms excludes it) */
    float md5_ms;      /* md5_kernel (emit_ms excludes it) */
    float finish_ms;   /* finish_kernel (fused rows+emit of single-item deduplicated clusters) */
    uint32_t n_wide_clusters;  /* clusters that went through the wide dedup class (more than 64 distinct sequences, ...) */
    uint32_t n_binned_clusters;/* clusters whose windows were sorted by key partition before the scan, retries included */
    uint32_t n_scratch_grown;  /* times the context re-made its scratch for a cluster of more work items than max_items (since pf_create) */
    uint32_t n_device_planned; /* clusters whose work items were laid out on the device (no host round trip before their scan) */
    uint32_t n_side_launches;  /* launches whose fused finish kernels ran on the context's second stream, beside the general path's */
    uint32_t reserved;
} pf_timing;

const char* pf_last_error(void);
const char* pf_version(void);

/* Number of visible HIP devices, or a negative pf_status. */
int pf_device_count(void);

int pf_create(pf_ctx** out, int device, const pf_opts* opts);
void pf_destroy(pf_ctx* ctx);

/* Forget the run-global patterns (a new run on the same context). */
int pf_reset_patterns(pf_ctx* ctx);

/* Run cluster_cutter + pattern_hasher over one batch.  Results stay on the device until
 * pf_fetch; `res` (may be NULL) receives the counters only. */
int pf_submit(pf_ctx* ctx, const pf_batch* batch, pf_result* counters);

/*
 * Genomes resident in HBM (SURVEY 8f N1 on the device).  pf_genomes_upload packs the contigs (A/C/G/T in either
 * case -> 2 bits, anything else -> an arbitrary code the caller must never ask for) into the context's genome
 * store and returns each contig's word offset.  pf_submit_gather is pf_submit for a batch whose packed input is
 * produced on the device: segment s is the range [src_start, src_start + seg_len) of the source at src_off --
 * the genome store, or (flag bit 0) b->packed, the words the host packed itself -- copied, or (flag bit 1)
 * reverse-complemented (what `-seq` of input.py:431,443 yields), to b->seg_word_off[s] of a device buffer of
 * g->n_words words.  Replaces the per-base host work of input.py:413-446 + the packing of the boundary.
 */
typedef struct {
    uint64_t n_words;
    const uint64_t* src_off; const uint32_t* src_start; const uint32_t* src_flags;   /* [b->n_segs] */
} pf_gather;
int pf_genomes_upload(pf_ctx* ctx, uint32_t n_contigs, const char* const* ascii, const uint64_t* len,
                      uint64_t* word_off_out);
int pf_genomes_clear(pf_ctx* ctx);
int pf_submit_gather(pf_ctx* ctx, const pf_batch* b, const pf_gather* g, pf_result* counters);

/* Copy the arrays of the last batch's result to host memory owned by the context. */
int pf_fetch(pf_ctx* ctx, pf_result* res);

int pf_get_timing(pf_ctx* ctx, pf_timing* t);

/* Multi-GPU (one process per GPU): first-seen bookkeeping for the run-global pattern dedup.
 * pf_export_patterns: (md5[16], first_seen) of every pattern this rank holds;
 * the driver all-gathers them (RCCL) and keeps, per digest, the rank with the lowest first_seen. */
int pf_export_patterns(pf_ctx* ctx, uint64_t* n, const uint8_t** md5, const uint64_t** first_seen);
/* Same, device to device: copies min(count, cap) entries into caller-owned DEVICE buffers
 * (d_md5: 16 B each, d_first_seen: 8 B each) so the all-gather can start from HBM. */
int pf_export_patterns_dev(pf_ctx* ctx, uint64_t cap, void* d_md5, void* d_first_seen, uint64_t* n);
/* After the all-gather: d_gathered (DEVICE) holds n_total rows {md5[0:8], md5[8:16], first_seen} (3 x uint64) of all
 * ranks; this rank's rows are [my_first, my_first+my_count).  Writes d_keep[j] = 1 (DEVICE, uint8) when this rank's
 * row j holds the lowest first_seen of its digest, and *n_global = number of distinct digests.  O(n_total) atomics
 * in a scratch hash table, no sort. */
int pf_merge_patterns(pf_ctx* ctx, const void* d_gathered, uint64_t n_total, uint64_t my_first, uint64_t my_count,
                      void* d_keep, uint64_t* n_global);
/* Same on the padded buffer all_gather_into_tensor leaves: `world` slots of slot_rows rows, of which the first
 * d_slot_counts[r] (DEVICE int64) are real; this rank's rows are the first my_count of slot `rank`. */
int pf_merge_patterns_padded(pf_ctx* ctx, const void* d_gathered, uint64_t world, uint64_t slot_rows,
                             const void* d_slot_counts, uint64_t rank, uint64_t my_count, void* d_keep,
                             uint64_t* n_global);
/* A checksum of checksums of the last pf_submit's results, computed on the device (no pf_fetch): out[0] = wrapping sum
 * over every kept k-mer of h(cluster index in the batch, its position in the cluster's output order, key words, MD5
 * digest of its pattern) -- i.e. of what its kmers_to_hashes.tsv row holds and where it stands (panfeed.py:208);
 * out[1] = the same per cluster over (kept, unique, digest of the cluster's own row, :177); out[2] = kept k-mers.
 * Independent of arena placement and pattern ids: two runs that would write the same files give the same three
 * numbers.  For parity checks at sizes no CPU checker covers (bench.py: identical-sequence shortcut on vs off at
 * BASELINE's full size; tests). */
int pf_result_checksum(pf_ctx* ctx, uint64_t out[3]);
/* Number of patterns in the run-global set after the last pf_submit. */
int pf_pattern_count(pf_ctx* ctx, uint64_t* n);
/* Test hook: pattern tables of more than max_slots slots are refused as if the device were out of memory (0: no
 * limit) -- the growth paths of the reference's unbounded `patterns` set (panfeed.py:146-150) have failure branches
 * that a 288 GB device never takes on its own. */
int pf_debug_limit_pattern_slots(pf_ctx* ctx, uint64_t max_slots);
/* Test hook, process-wide: a single device allocation of the library's growable buffers above max_bytes is refused as if
 * the device were out of memory (0: no limit).  stats (may be NULL) receives, and resets, what was seen since the last
 * call: stats[0] = the largest buffer size asked for, stats[1] = allocations whose size-plus-slack request failed and
 * whose exact-size retry succeeded.  A buffer's slack is a convenience: a run must not fail while the bytes it needs
 * exist (the reference's structures grow until the machine is full, panfeed.py:146-150). */
int pf_debug_limit_alloc(uint64_t max_bytes, uint64_t stats[2]);

/* Device buffers for callers that keep batches resident (bench.py, tests): plain hipMalloc /
 * hipMemcpy / hipFree on the context's device. */
int pf_dev_alloc(pf_ctx* ctx, uint64_t bytes, void** dptr);
int pf_dev_free(pf_ctx* ctx, void* dptr);
int pf_dev_upload(pf_ctx* ctx, void* dptr, const void* src, uint64_t bytes);
int pf_dev_download(pf_ctx* ctx, void* dst, const void* dptr, uint64_t bytes);

/* Synthetic-input helper (bench.py): expand allele pools into per-sample packed segments on the
 * device.  seg i becomes a copy of allele seg_allele[i]: words
 * [allele_word_off[a], allele_word_off[a] + ceil16(len)) -> packed[seg_word_off[i] ...].
 * All pointers are device pointers. */
int pf_synth_expand(pf_ctx* ctx, const uint64_t* allele_words, const uint64_t* allele_word_off,
                    const uint32_t* seg_allele, const uint64_t* seg_word_off, const uint32_t* seg_len,
                    uint32_t n_segs, uint64_t* packed);

/* Host helper: pack ASCII A/C/G/T (upper case) to the layout above.  dst must hold
 * 2*ceil(len/64) words; returns the number of words written. */
uint64_t pf_pack_acgt(const char* seq, uint32_t len, uint64_t* dst);

/*
 * Host-side batch packer (no GPU involved): the Seqinfo records of a batch of clusters
 * (/root/reference/panfeed/classes.py:11-18, iteration order of panfeed.py:54-55, cluster-major) -> the segment
 * arrays of pf_batch, the slow-path rows for windows with a non-ACGT base (grouped as panfeed.py:64-88 does),
 * strand-bit offsets for target strains, and what the kmers.tsv writer needs per target sequence.
 */
typedef struct {
    uint32_t n_clusters, n_seqs;
    const char* const* seq;           /* [n_seqs] Seqinfo.sequence (upper case), seq_len bytes */
    const char* const* comp;          /* [n_seqs] Seqinfo.compsequence; checked to be the complement on A/C/G/T */
    const uint32_t* seq_len;          /* [n_seqs] */
    const uint32_t* seq_col;          /* [n_seqs] column of the strain in sorted(cluster.keys()) (panfeed.py:47-49) */
    const uint8_t* seq_target;        /* [n_seqs] `strain in stroi` (panfeed.py:90), may be NULL */
    const uint32_t* cluster_seq_off;  /* [n_clusters+1] */
    uint32_t klength, canon, W, want_strand;
    /* optional (NULL = every sequence is given as text): sequences that are ranges of the genomes resident in HBM
     * (pf_genomes_upload).  seq_flags bit 0: by reference -- seq/comp of that sequence are not read, the range is
     * pure A/C/G/T and not a target strain's; bit 1: the sequence is the reverse complement of the range. */
    const uint64_t* seq_src_off;      /* [n_seqs] word offset of the contig in the genome store */
    const uint32_t* seq_src_start;    /* [n_seqs] lowest base coordinate of the range in the contig */
    const uint32_t* seq_flags;        /* [n_seqs] */
} pf_pack_in;

typedef struct pf_packed pf_packed;

typedef struct {
    uint32_t n_segs, n_extra, n_targets, reserved;
    uint64_t n_words, n_strand_words, n_instances;
    const uint64_t* packed; const uint64_t* seg_word_off;
    const uint32_t* seg_len; const uint32_t* seg_sample; const uint32_t* seg_ord_base; const uint32_t* seg_strand_off;
    const uint32_t* cluster_seg_off;  /* [n_clusters+1] */
    const uint64_t* cluster_ninst;    /* [n_clusters] trip count of panfeed.py:64 (x2 non-canonical) */
    const uint32_t* extra_cluster; const uint32_t* extra_ord; const uint32_t* extra_bits;
    const char* extra_keys;           /* n_extra * klength bytes */
    /* target sequences (input index), their pure segments and slow-path windows (CSR) */
    const uint32_t* target_seq; const uint32_t* target_seg_off; const uint32_t* target_seg_index;
    const uint32_t* target_seg_start; const uint32_t* target_seg_nwin;
    const uint32_t* target_ambig_off; const uint32_t* target_ambig_pos; const int8_t* target_ambig_used;
    const char* target_ambig_keys;    /* klength bytes per slow-path window */
    /* with by-reference sequences: `packed` / n_words hold only the segments packed on the host ("literal"),
     * seg_word_off are offsets in the DEVICE buffer of n_words_dev words that pf_submit_gather fills; NULL / 0
     * when every sequence was given as text */
    uint64_t n_words_dev;
    const uint64_t* gather_src_off; const uint32_t* gather_src_start; const uint32_t* gather_src_flags;
} pf_packed_view_t;

int pf_pack_records(const pf_pack_in* in, pf_packed** out);
/* Binding helper for a CPython host side (panfeed_amd/packing.py hands Seqinfo.sequence / .compsequence,
 * classes.py:11-18, to pf_pack_records by address): out[i] = as_utf8(objs[i]) for an array of n object pointers, where
 * as_utf8 is the interpreter's PyUnicode_AsUTF8 -- one C loop instead of one FFI call per string, no assumption about the
 * object layout.  Call it with the GIL held (ctypes.PyDLL).  out[i] == 0: the API refused that object. */
int pf_py_str_addresses(void* const* objs, uint64_t n, const char* (*as_utf8)(void*), uint64_t* out);
/* The same for the two str attributes of every object of a Python list in one pass (Seqinfo.sequence / .compsequence,
 * classes.py:11-18; what pattern_hasher's records carry, panfeed.py:54-67): addresses of their UTF-8 bytes, their length,
 * flags[i] bit 0 = both str, equal length, all ASCII.  The interpreter's API comes as function pointers (PyList_GetItem,
 * PyObject_GetAttr, PyUnicode_AsUTF8AndSize, PyUnicode_GetLength, Py_DecRef, PyErr_Clear); the attribute objects stay
 * referenced in held[2 n] until pf_py_release.  Call both with the GIL held. */
typedef struct pf_py_api {
    void* (*list_get_item)(void*, long long);                 /* PyList_GetItem (borrowed reference) */
    void* (*get_attr)(void*, void*);                          /* PyObject_GetAttr (new reference) */
    const char* (*as_utf8_and_size)(void*, long long*);       /* PyUnicode_AsUTF8AndSize */
    long long (*get_length)(void*);                           /* PyUnicode_GetLength */
    void (*dec_ref)(void*);                                   /* Py_DecRef */
    void (*err_clear)(void);                                  /* PyErr_Clear */
} pf_py_api;
int pf_py_seqinfo_columns(void* list, uint64_t n, void* attr_seq, void* attr_comp, const pf_py_api* api,
                          uint64_t* a_seq, uint64_t* a_comp, uint32_t* len, uint8_t* flags, void** held);
int pf_py_release(void** held, uint64_t n, const pf_py_api* api);
int pf_packed_view(const pf_packed* p, pf_packed_view_t* view);
void pf_packed_free(pf_packed* p);

/*
 * Native reader in front of the path: the panaroo presence/absence table, GFF3 features and
 * FASTA sequences -> the Seqinfo records iter_gene_clusters yields
 * (/root/reference/panfeed/input.py:274-332 parse_gff, :335-468 iter_gene_clusters), as flat arrays whose
 * sequence pointers feed pf_pack_records directly.  PARITY UNPINNED at the pyfaidx boundary.
 */
typedef struct {
    const char* presence_absence_csv;       /* -p */
    uint32_t n_genomes, reserved;
    const char* const* genome_names;        /* file name minus ".gff" (input.py:31) */
    const char* const* gff_paths;
    const char* const* fasta_paths;         /* NULL, or per genome NULL = sequences embedded after ##FASTA */
    uint32_t upstream, downstream, downstream_start_codon, raise_missing;
    const char* const* target_strains; uint32_t n_targets, n_genes;
    const char* const* gene_list;           /* --genes; NULL = every cluster */
} pf_pangenome_opts;

typedef struct pf_pangenome pf_pangenome;
typedef struct pf_records pf_records;

typedef struct { uint32_t n_clusters, n_strains, next_cluster, reserved; } pf_pangenome_info_t;

typedef struct {
    uint32_t n_clusters, n_seqs, W, reserved;
    /* per Seqinfo, in the iteration order of panfeed.py:54-55, cluster-major */
    const char* const* seq; const char* const* comp; const char* const* id; const char* const* chromosome;
    const uint32_t* seq_len; const uint32_t* seq_col; const uint32_t* seq_strain;   /* strain: index in the cluster's dict */
    const uint8_t* seq_target; const int32_t* seq_strand;
    const int64_t* seq_start; const int64_t* seq_end; const int64_t* seq_offset;
    /* per cluster */
    const uint32_t* cluster_seq_off;        /* [n_clusters+1] */
    const char* const* cluster_name;
    const uint32_t* cluster_nstrains;       /* len(cluster.keys()) */
    const uint32_t* cluster_npresab;        /* len(clusterpresab) = strains of the table */
    const uint32_t* cluster_presab;         /* [n_clusters * W] bits over sorted(strains) */
    const uint32_t* cluster_strain_off;     /* [n_clusters+1] into cluster_strain: the dict keys in insertion order */
    const char* const* cluster_strain;
    /* after pf_pangenome_set_store: sequences given as ranges of the resident genomes (seq/comp NULL for those);
     * same meaning as the fields of pf_pack_in; NULL before */
    const uint64_t* seq_src_off; const uint32_t* seq_src_start; const uint32_t* seq_flags;
} pf_records_view_t;

int pf_pangenome_open(const pf_pangenome_opts* opts, pf_pangenome** out);
/* The same reader with the genomes going to the device AS THE FILES ARE READ (one pass over the input instead of
 * pf_pangenome_open's read + upper-casing copy and pf_genomes_upload's second copy): every file is read straight into
 * pinned memory, its GFF lines parsed there (input.py:274-332), its contigs measured without copying a base, and a kernel
 * de-wraps, upper-cases and packs the letters to 2 bits per base in the context's genome store while other files are
 * still being read.  Contig text stays on the host only for contigs with a letter other than A/C/G/T and for target
 * strains (what iter_gene_clusters cuts out of text, input.py:427-452).  The reader comes back in by-reference mode
 * (as after pf_pangenome_contigs / pf_genomes_upload / pf_pangenome_set_store); pf_pangenome_contigs is refused on it.
 * PF_ERR_CAPACITY: the store's estimate (half a byte per byte of input) did not hold -- contigs of a few letters each;
 * use pf_pangenome_open + pf_genomes_upload. */
int pf_pangenome_open_device(const pf_pangenome_opts* opts, pf_ctx* ctx, pf_pangenome** out);
/* The same with the context asked for when the first genome needs it (get_ctx(user), called once, from one of the reader's
 * threads): the caller may still be creating it while the reader parses the presence/absence table. */
int pf_pangenome_open_device_cb(const pf_pangenome_opts* opts, pf_ctx* (*get_ctx)(void*), void* user, pf_pangenome** out);
/* Test hook: the reader side of pf_pangenome_open_device without a device -- blocks of ordinary memory, the store filled
 * on the host by the kernel's addressing (letter j of a contig = byte text_off + j + (j / width) * eol of its block).
 * *store (pf_free_text) holds *nwords words; the reader is in by-reference mode. */
int pf_debug_open_hostsink(const pf_pangenome_opts* opts, pf_pangenome** out, uint64_t** store, uint64_t* nwords);
void pf_pangenome_close(pf_pangenome* p);
/* The same on a thread of its own: returns at once, the reader's memory (tables, features, gigabytes of contigs) is
 * given back in the background.  For a caller that has finished its run. */
void pf_pangenome_close_async(pf_pangenome* p);
int pf_pangenome_info(pf_pangenome* p, pf_pangenome_info_t* info);
const char* pf_pangenome_strain(pf_pangenome* p, uint32_t i, int sorted);
const char* pf_pangenome_take_log(pf_pangenome* p);   /* warnings the reference sends to logger.warning */
/* Records of the next (at most) max_clusters rows of the table; n_clusters == 0 at the end. */
int pf_pangenome_next(pf_pangenome* p, uint32_t max_clusters, pf_records** out, pf_records_view_t* view);
/* Multi-GPU sharding of the processing order (contiguous ranges of table rows per rank).  The clusters a
 * run processes are the table rows that pass --genes, in table order (input.py:352-355); pf_pangenome_weights gives
 * each one's number of gene entries (paralogs counted: the sequences iter_gene_clusters will cut, a proxy for its
 * k-mer instances) so that ranks can balance their ranges; pf_pangenome_set_range restricts pf_pangenome_next to
 * processed clusters [first, first + count) and rewinds the reader to `first`. */
int pf_pangenome_weights(pf_pangenome* p, uint32_t cap, uint32_t* weights, uint32_t* n_processed);
int pf_pangenome_set_range(pf_pangenome* p, uint32_t first, uint32_t count);
/* Contigs of all genomes in one flat order (upper-cased text), for pf_genomes_upload; then the word offsets it
 * returned: from here on pf_pangenome_next hands out non-target, pure-ACGT sequences by reference. */
int pf_pangenome_contigs(pf_pangenome* p, uint32_t* n, const char* const** ascii, const uint64_t** len);
int pf_pangenome_set_store(pf_pangenome* p, const uint64_t* contig_word_off, uint32_t n);
void pf_records_free(pf_records* r);

/* Host helper: md5 + base64 of a digest -- panfeed.py:175-176 -- for writers. */
void pf_b64_digest(const uint8_t md5[16], char out[24]);

/* Writers (host threads, after pf_fetch): the bytes pattern_hasher appends to kmers_to_hashes.tsv for the last
 * batch -- "idx\t\thash\n" per cluster then "idx\tkmer\thash\n" per kept k-mer (panfeed.py:177, 208) -- and to
 * hashes_to_patterns.tsv -- "hash\tv0\tv1...\n" per pattern first seen in this batch, in first-seen order
 * (panfeed.py:181-187, 217-223).  cluster_names: n_clusters NUL-terminated names (batch order); extra_keys: the
 * k-mer strings of the batch's slow-path rows (klength bytes each), may be NULL without such rows.
 * cluster_end (may be NULL): [n_clusters] byte offset where each cluster's rows end.
 * The buffers are malloc'ed; release them with pf_free_text. */
int pf_render_kmers_to_hashes(pf_ctx* ctx, const char* const* cluster_names, const char* const* extra_keys,
                              char** out, uint64_t* nbytes, uint64_t* cluster_end);
int pf_render_hashes_to_patterns(pf_ctx* ctx, char** out, uint64_t* nbytes);

/* One Seqinfo of a target strain (panfeed/classes.py:11-18) for the positional rows of kmers.tsv
 * (panfeed.py:90-107).  Its pure-ACGT windows take used_strand from the strand bits of the last fetched batch
 * (segment `seg_index[j]` covers windows seg_start[j] .. seg_start[j]+seg_nwin[j]-1 of the sequence); windows
 * that contain another base come with their canonical k-mer and used_strand from the caller's slow path. */
typedef struct {
    const char* cluster; const char* strain; const char* id; const char* chromosome;   /* NUL-terminated */
    const char* sequence; const char* compsequence;                                    /* `len` bytes each */
    uint32_t len;
    int32_t strand;
    int64_t start, end, offset;
    uint32_t n_segs; uint32_t n_ambig;
    const uint32_t* seg_index; const uint32_t* seg_start; const uint32_t* seg_nwin;
    const uint32_t* ambig_pos; const int8_t* ambig_used; const char* const* ambig_key;   /* sorted by ambig_pos */
} pf_target_seq;

/* Rows of kmers.tsv for `n` target sequences, in the given order: one row per window ("cluster, strain,
 * feature_id, contig, feature_strand, contig_start, contig_end, gene_start, gene_end, strand, k-mer"), two per
 * window in non-canonical mode (panfeed.py:104-107).  seg_strand_off: the batch's array.  Needs the last pf_submit
 * only (the windows' used_strand bits are copied from the device by the call itself; no pf_fetch).  Sizes and rows are
 * both produced by all host threads; *out is malloc'd (pf_free_text). */
int pf_render_kmers_tsv(pf_ctx* ctx, const pf_target_seq* seqs, uint32_t n, const uint32_t* seg_strand_off,
                        char** out, uint64_t* nbytes);
void pf_free_text(char* p);
/* The same rows (panfeed.py:90-107) written ON THE DEVICE: the used_strand bits and the packed bases of the last
 * pf_submit are there already, so neither they nor the sequences' letters cross PCIe -- per sequence only its five
 * constant fields, as text, and four numbers go up.  Sequences that are pure A/C/G/T (one segment covering every
 * window) are written by the GPU; the others (an 'N' inside, an over-long name) by the host renderer above and copied
 * to their places, so that the text is that of pf_render_kmers_tsv byte for byte, in the order of `seqs`.  The text
 * stays in device memory; *nbytes is its size.  With every sample a target (BASELINE configs[4]'s second pass) it is
 * the largest output of a run. */
int pf_render_kmers_tsv_device(pf_ctx* ctx, const pf_target_seq* seqs, uint32_t n, uint64_t* nbytes);
/* Bytes [offset, offset + *nbytes) of that text, *nbytes = min(max_bytes, what is left), in pinned host memory owned by
 * the context: valid until the next call of this function (the block after it is already being copied when the call
 * returns, so that the caller's write of one block overlaps the copy of the next). */
int pf_device_text_chunk(pf_ctx* ctx, uint64_t offset, uint64_t max_bytes, const char** ptr, uint64_t* nbytes);

/* The bodies of kmers_to_hashes.tsv and hashes_to_patterns.tsv of the last pf_submit written ON THE DEVICE
 * (panfeed.py:177,208 and :181-187,217-223): rows assembled in LDS, coalesced stores, one copy to pinned host memory.
 * No pf_fetch needed; the k-mer keys and pattern rows never cross PCIe, only the text does.  names: cluster names in
 * batch order; extra_keys: n_extra * klength bytes, the k-mer text of the batch's slow-path rows (NULL if none).
 * *kh / *hp point into pinned memory owned by the context (two blocks used alternately): valid until the call after
 * the next one, so that a writer thread can still be on the previous batch.  Not under multiple_files. */
int pf_render_device(pf_ctx* ctx, const char* const* names, const char* extra_keys, uint64_t n_extra,
                     const char** kh, uint64_t* kh_bytes, const char** hp, uint64_t* hp_bytes);
/* Same with flags: PF_RENDER_NO_PATTERN_ROWS leaves hashes_to_patterns.tsv out (*hp_bytes = 0) -- a rank of a
 * multi-GPU run writes its pattern rows only after the run-global merge has said which ones are its own. */
#define PF_RENDER_NO_PATTERN_ROWS 1u
int pf_render_device_ex(pf_ctx* ctx, const char* const* names, const char* extra_keys, uint64_t n_extra, uint32_t flags,
                        const char** kh, uint64_t* kh_bytes, const char** hp, uint64_t* hp_bytes);
/* Rows of hashes_to_patterns.tsv ("hash\tv0\tv1...\n", panfeed.py:181-187, 217-223) for ANY patterns of the
 * run-global pool, in the order given, written on the device: pids[n] are pattern ids (< pf_pattern_count).  The
 * multi-GPU driver calls it after pf_merge_patterns with the ids whose first_seen is the global minimum of their
 * digest, sorted by first_seen (rank-ordered concatenation then equals the --cores 1 file; reference writer:
 * __main__.py:67-81).  *text points into the context's pinned memory, valid until the call after the next one of
 * this function / pf_render_device. */
int pf_render_pattern_rows(pf_ctx* ctx, const uint32_t* pids, uint64_t n, const char** text, uint64_t* nbytes);

/* Parallel gzip of a block of output text (the reference: gzip.open(..., "wt", compresslevel=9),
 * /root/reference/panfeed/input.py:239-241,255-258 -- one core).  `data` is cut at line ends into chunks of about
 * chunk_bytes, each deflated as its own gzip member on its own host thread; *out (pf_free_text) is the members
 * concatenated, to be appended to the .gz file.  Decompresses to exactly `data`. */
int pf_gzip_members(const char* data, uint64_t n, int level, uint64_t chunk_bytes, char** out, uint64_t* out_n);

/*
 * Row filter of the downstream tools on the device: the rows of kmers_to_hashes.tsv whose
 * hashed_pattern is one of a set of hashes (/root/reference/panfeed/get_clusters.py:89-94, get_kmers.py:103-106:
 * `x[x['hashed_pattern'].isin(passing_hashes)]` over 100 000-row pandas chunks) and the rows of kmers.tsv whose cluster
 * is one of a set of clusters (get_kmers.py:131-134).  first_field = 1: the key is the first tab-separated field of a
 * line (cluster); 0: the last one (hashed_pattern).  pf_rowfilter_scan takes a block of the file that starts at a line
 * start, tests every complete line of it on the GPU (64-bit hashes of the key field against a device hash set, every
 * candidate then checked against the exact keys on the host) and returns the matching lines, in file order, as
 * [begin, end) byte ranges of `text` (end includes the newline; valid until the next call); *consumed = bytes of complete
 * lines: the caller puts text[consumed:] in front of its next block.  The file's header line is the caller's business.
 */
typedef struct pf_rowfilter pf_rowfilter;
int pf_rowfilter_create(int device, int first_field, const char* const* keys, const uint32_t* key_len, uint64_t n_keys,
                        pf_rowfilter** out);
int pf_rowfilter_scan(pf_rowfilter* f, const char* text, uint64_t nbytes, const uint64_t** line_begin,
                      const uint64_t** line_end, uint64_t* n_lines, uint64_t* consumed);
int pf_rowfilter_stats(pf_rowfilter* f, uint64_t* bytes_scanned, float* device_ms);
void pf_rowfilter_destroy(pf_rowfilter* f);

#ifdef __cplusplus
}
#endif
#endif
