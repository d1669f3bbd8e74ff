// pf_kernels.h -- HIP kernels of the panfeed hot path for gfx950 (MI355X, wave64, 160 KiB LDS/CU).
//
// Pipeline per batch of gene clusters (DESIGN.md has the full picture):
//   gather_segments_kernel  (optional) segments cut out of / reverse-complemented from genomes resident in HBM.
//   cluster_dedup_kernel  per cluster: identical segments -> one representative per distinct sequence (exact,
//                      one pass over the packed bytes).
//   unit_class_small_kernel / unit_class_kernel   per cluster of several distinct sequences: 64-window units that
//                      several of them hold unchanged at the same place -> one piece with all their column bits (exact).
//   scan_desc_kernel + kmer_scan_kernel   persistent 1024-thread workgroups, one (cluster, key partition) at a
//                      time: slide the k window over the 2-bit packed segments, canonicalise (panfeed.py:65-75),
//                      group k-mers in an LDS hash table {key, first-occurrence ordinal, 32-column presence word};
//                      columns are swept in chunks of 32, each chunk's words flushed coalesced.
//   finish_kernel      deduplicated clusters: everything after the scan in one workgroup (sample sets, allele-mask
//                      table, MAF / same-as-cluster filter, ranks from ordinal bitmaps, run-global pattern table,
//                      outputs).
//   rows_kernel -> cluster_base_kernel (+ bitmap_merge_kernel) -> emit_kernel -> pattern_rows_kernel   the general
//                      path: popcount / MAF + same-as-cluster filter (panfeed.py:190-204), 128-bit row hash, ordinal
//                      order (dict insertion order, :189; one merged ordinal bitmap per cluster of several items),
//                      run-global pattern table (first-seen rule, :179-187, 210-223), pattern rows.
//   extra_csr_kernel   slow-path rows (k-mers around a non-ACGT base) of a batch already in device memory: per-cluster
//                      counts and the check of their list.
//   md5_kernel         MD5 of the int64 / float64 image of each new pattern (panfeed.py:175, 206).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace pf {

// phase profiling (build with -DPF_PROF): shader-clock cycles per phase as workgroup thread 0 sees them
#ifdef PF_PROF
__device__ unsigned long long pf_prof[64];
#define PF_PROF_BEGIN() uint64_t prof_t_ = __builtin_readcyclecounter()
#define PF_PROF_STAMP(k) do { if (threadIdx.x == 0) { const uint64_t n_ = __builtin_readcyclecounter(); \
    atomicAdd(&pf_prof[k], (unsigned long long)(n_ - prof_t_)); prof_t_ = n_; } } while (0)
#else
#define PF_PROF_BEGIN() do { } while (0)
#define PF_PROF_STAMP(k) do { } while (0)
#endif
// timing experiments (never shipped): -DPF_KO_FINISH=n, n = 0 .. 4, makes finish_kernel return behind its phase n (0 set-up +
// sample sets, 1 mask table, 2 row evaluation, 3 ordinal bitmaps, 4 prefix counts) -- nothing has been claimed or written by
// then, the later passes find empty clusters; n = 7 keeps every phase but gives every pattern the id 0 without touching the
// run-global table (no pattern is made: MD5 has nothing to do).  A return behind phase 5 or 6 would leave claimed slots
// unpublished / rows unwritten for the kernels behind it: not offered.  profiles/r05/experiment_finish_knockout.txt
#ifdef PF_KO_FINISH
#define PF_KO_FINISH_AT(n) do { if ((n) == PF_KO_FINISH && (n) <= 4) return; } while (0)
#else
#define PF_KO_FINISH_AT(n) do { } while (0)
#endif

constexpr uint32_t SCAN_THREADS = 1024;
constexpr uint32_t SCAN_WAVES = SCAN_THREADS / 64;
constexpr uint32_t LDS_BYTES = 163840;           // 160 KiB, whole CU
constexpr uint32_t MISC_WORDS = 2560;            // counters, chunk offsets, staged segment metadata (10 KiB)
constexpr uint32_t MAX_CHUNKS = 256;             // 32 samples each -> max_strains <= 8192
constexpr uint32_t INSERT_SLACK = 2176;          // > 2*SCAN_THREADS: inserts that can land after the limit trips
                                                 // (a wave stops after the 64-window unit in which one of its own
                                                 // inserts sees the count past the limit; a unit inserts at most
                                                 // 2 keys per lane in non-canonical mode)
constexpr uint64_t EMPTY64 = ~0ull;
constexpr uint32_t NO_ORD = 0xFFFFFFFFu;
constexpr uint64_t KEY_EXTRA_FLAG = 1ull << 63;  // tab_key of a slow-path row

__host__ __device__ constexpr uint32_t nslots_max(int kw) {
    // keys (8*kw) + ord (4) + bits (4) per slot, MISC_WORDS*4 bytes aside, multiple of 64
    return ((LDS_BYTES - MISC_WORDS * 4) / (8u * kw + 8u)) / 64u * 64u;
}
__host__ __device__ inline uint32_t insert_limit(uint32_t ns) {
    uint32_t a = ns - INSERT_SLACK, b = (uint32_t)(((uint64_t)ns * 4) / 5);
    return a < b ? a : b;
}

constexpr uint32_t SCAN_BUCKET = 4;   // slots per bucket of the one-word-key table (ns is a multiple of 64)
template <int KW>
struct Key {
    uint64_t w[KW];
};

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
template <int KW>
__device__ __forceinline__ uint32_t key_hash(const Key<KW>& k) {
    // two 32-bit multiplies (quarter-rate on CDNA): fold the halves with a rotate, then xor-shift / multiply.
    // The slot takes the high bits (umulhi), the key partition the low 16.
    uint32_t x = (uint32_t)k.w[0] ^ rotl32((uint32_t)(k.w[0] >> 32), 15);
#pragma unroll
    for (int j = 1; j < KW; j++) x = rotl32(x, 11) ^ rotl32((uint32_t)k.w[j], 7) ^ rotl32((uint32_t)(k.w[j] >> 32), 23);
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    x *= 0x85EBCA77u;
    x ^= x >> 13;
    return x;
}
// swap the two bits of every 2-bit group
__device__ __forceinline__ uint64_t pairswap(uint64_t x) {
    return ((x & 0x5555555555555555ull) << 1) | ((x >> 1) & 0x5555555555555555ull);
}
// reverse the order of the 32 two-bit groups of x
__device__ __forceinline__ uint64_t rev_groups(uint64_t x) { return pairswap(__brevll(x)); }

// murmur3_x86_128 block / finalisation (Appleby, public domain algorithm) on 32-bit words
struct H128 {
    uint32_t h1, h2, h3, h4;
};
__device__ __forceinline__ void mm3_block(H128& s, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t k4) {
    const uint32_t c1 = 0x239b961b, c2 = 0xab0e9789, c3 = 0x38b34ae5, c4 = 0xa1e38b93;
    k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2; s.h1 ^= k1;
    s.h1 = rotl32(s.h1, 19); s.h1 += s.h2; s.h1 = s.h1 * 5 + 0x561ccd1b;
    k2 *= c2; k2 = rotl32(k2, 16); k2 *= c3; s.h2 ^= k2;
    s.h2 = rotl32(s.h2, 17); s.h2 += s.h3; s.h2 = s.h2 * 5 + 0x0bcaa747;
    k3 *= c3; k3 = rotl32(k3, 17); k3 *= c4; s.h3 ^= k3;
    s.h3 = rotl32(s.h3, 15); s.h3 += s.h4; s.h3 = s.h3 * 5 + 0x96cd1c35;
    k4 *= c4; k4 = rotl32(k4, 18); k4 *= c1; s.h4 ^= k4;
    s.h4 = rotl32(s.h4, 13); s.h4 += s.h1; s.h4 = s.h4 * 5 + 0x32ac3b17;
}
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6b; h ^= h >> 13; h *= 0xc2b2ae35; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ void mm3_final(H128& s, uint32_t len) {
    s.h1 ^= len; s.h2 ^= len; s.h3 ^= len; s.h4 ^= len;
    s.h1 += s.h2; s.h1 += s.h3; s.h1 += s.h4; s.h2 += s.h1; s.h3 += s.h1; s.h4 += s.h1;
    s.h1 = fmix32(s.h1); s.h2 = fmix32(s.h2); s.h3 = fmix32(s.h3); s.h4 = fmix32(s.h4);
    s.h1 += s.h2; s.h1 += s.h3; s.h1 += s.h4; s.h2 += s.h1; s.h3 += s.h1; s.h4 += s.h1;
}

// ---------------------------------------------------------------------------------------------
// kmer_scan_kernel
// ---------------------------------------------------------------------------------------------
// The scan reads a "view" of each cluster's segments: either the caller's segments as they are
// (mode 0) or, after cluster_dedup_kernel found identical segments, only one representative per
// distinct sequence with the distinct index in place of the sample column (mode 1).
struct ScanParams {
    const uint64_t* packed;
    const uint64_t* seg_word_off;     // view
    const uint32_t* seg_len;          // view
    const uint32_t* seg_sample;       // view: sample column (mode 0) / distinct index (mode 1)
    const uint32_t* seg_ord_base;     // view: instance ordinal of the segment's first window
    const uint32_t* seg_bits;         // view: presence bits a window of the segment sets in its chunk's word
    // the unit view (unit_class_kernel) of the clusters that have one: same five arrays in a pool of their own;
    // cluster_seg_off[c] carries VIEW_IN_POOL then
    const uint64_t* u_word_off; const uint32_t* u_len; const uint32_t* u_sample; const uint32_t* u_ord_base; const uint32_t* u_bits;
    const uint32_t* cluster_seg_off;  // first view segment of the cluster (view_off)
    const uint32_t* cluster_vnseg;    // view segments of the cluster
    const uint32_t* cluster_vnstr;    // columns of the view (len(cluster) / distinct sequences)
    // per item (item = work[blockIdx.x])
    const uint32_t* item_cluster;     // batch-local cluster index
    const uint32_t* item_part;
    const uint32_t* item_nparts;
    const uint32_t* item_nslots;
    const uint32_t* item_scratch;     // scratch slice of the item
    const uint32_t* item_compact;     // 1: deduplicated cluster finished by finish_kernel -> compact table dump:
                                      //    only occupied slots, {key, ordinal, 64-bit allele mask}, any order
    uint32_t* cmask_lo; uint32_t* cmask_hi;   // [slice][NS] allele mask words of the compact dump
    // scratch, indexed by slice
    uint64_t* tab_key;                // [slice][KW][NS]
    uint32_t* tab_ord;                // [slice][NS]
    uint32_t* chunkbits;              // [slice][W][NS]
    uint32_t* chunkmask;              // [slice][8]  bit ch set: chunk ch was flushed
    uint32_t* item_count;             // [item] unique keys in the item's table
    uint32_t* cluster_overflow;       // [cluster] nonzero when a partition overflowed: 64 * (units of the item / units
                                      // scanned when the table was full), the largest over its partitions (>= 64)
    const uint32_t* work;             // [n_work] item ids of this launch
    const struct ScanDesc* desc;      // [n_work] what a workgroup needs to start on work[i] (scan_desc_kernel)
    // binned clusters (bin_kernel): the windows of the cluster's view as (key, ordinal, presence bit) entries, sorted by
    // key partition and chunk; an item of such a cluster reads ITS entries instead of walking the whole view
    const uint32_t* item_binned;      // [item] nonzero: the item's windows come from the queue
    uint64_t* q_key; uint32_t* q_ord; uint32_t* q_bit;    // entries; word j of entry e's key: q_key[j * q_stride + e]
    uint64_t q_stride;
    uint32_t* q_off;                  // [item][BIN_CHUNKS + 1] first entry of every chunk of the item; [nchunks]: the end
    const uint32_t* bin_cluster; const uint32_t* bin_item0; const uint32_t* bin_nparts; const uint32_t* bin_base;   // [grid of bin_kernel]
    uint32_t n_work;
    uint32_t k;
    uint32_t W;
    uint32_t NS;                      // slots per scratch slice (= nslots_max(KW))
};

// One 64-byte record per work entry, so that a workgroup starts an item with one load instead of a chain of
// four dependent ones (work -> item arrays -> cluster arrays); the scan kernel requests the next one while it
// works on the current item.
constexpr uint32_t VIEW_IN_POOL = 0x80000000u;   // view_off flag: the cluster's view is in the unit-view pool
constexpr uint32_t BIN_CHUNKS = 32;              // chunks (32 view columns each) of a binned cluster's view at most
struct ScanDesc { uint32_t item, c, part, nparts, ns, slice, seg0, nseg, nstr, compact, binned, pad[5]; };
static_assert(sizeof(ScanDesc) == 64, "ScanDesc is read as 16 words");

__global__ __launch_bounds__(256) void scan_desc_kernel(ScanParams p, ScanDesc* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_work) return;
    ScanDesc d{};
    d.item = p.work[i];
    d.c = p.item_cluster[d.item];
    d.part = p.item_part[d.item]; d.nparts = p.item_nparts[d.item];
    d.ns = p.item_nslots[d.item]; d.slice = p.item_scratch[d.item];
    d.compact = p.item_compact[d.item];
    d.binned = p.item_binned ? p.item_binned[d.item] : 0u;
    d.seg0 = p.cluster_seg_off[d.c]; d.nseg = p.cluster_vnseg[d.c]; d.nstr = p.cluster_vnstr[d.c];
    out[i] = d;
}

// lower bound of `v` in seg_sample[a..b)
__device__ __forceinline__ uint32_t seg_lower_bound(const uint32_t* seg_sample, uint32_t a, uint32_t b, uint32_t v) {
    while (a < b) {
        uint32_t m = (a + b) >> 1;
        if (seg_sample[m] < v) a = m + 1; else b = m;
    }
    return a;
}

// insert-or-find `key` in the LDS table, then fold (ordinal, sample bit) into the slot.  Returns true for the lanes
// whose insert found the table past `limit` (the caller stops its wave; the cluster is re-run with more partitions).
template <int KW>
__device__ __forceinline__ bool table_update(uint64_t* keys, uint32_t* ord, uint32_t* bits, uint32_t* misc,
                                             uint32_t NS, uint32_t ns, uint32_t limit, bool active,
                                             const Key<KW>& key, uint32_t home, uint32_t myord, uint32_t bit) {
    uint32_t slot = home;
    bool inserted = false;
    if (KW == 1) {
        // four-slot buckets: `home` is a bucket index (ns / 4 buckets), the four keys come with two 128-bit reads
        // issued together, a new key takes the first empty slot of the first bucket that has one.  The loop runs as long
        // as the slowest of the 64 lanes, so what counts is how often ANY lane has to go past its home bucket: with
        // two-slot buckets (round 1) a unit took ~3.5 trips at the usual fill and the trips after the first were 38 % of
        // the kernel's time (profiles/r02/scan_time_breakdown_experiments.txt).  ONE loop for the wave, its condition uniform (`__any`), the
        // lanes predicated inside it.
        bool pending = active;
        uint32_t bucket = home;
        const uint32_t nb = ns / SCAN_BUCKET;
        do {
            ulonglong2 kk[SCAN_BUCKET / 2];
#pragma unroll
            for (uint32_t j = 0; j < SCAN_BUCKET / 2; j++) kk[j] = *reinterpret_cast<const ulonglong2*>(&keys[SCAN_BUCKET * bucket + 2 * j]);
            bool hit = false, emp = false;
            uint32_t ih = 0, ie = 0;
#pragma unroll
            for (int j = SCAN_BUCKET / 2 - 1; j >= 0; j--) {             // downwards: the lowest index wins
                if (kk[j].y == key.w[0]) { hit = true; ih = 2 * j + 1; }
                if (kk[j].x == key.w[0]) { hit = true; ih = 2 * j; }
                if (kk[j].y == EMPTY64) { emp = true; ie = 2 * j + 1; }
                if (kk[j].x == EMPTY64) { emp = true; ie = 2 * j; }
            }
            const uint32_t cand = SCAN_BUCKET * bucket + (hit ? ih : ie);
            bool got = pending && hit;
            if (pending && !hit && emp) {
                const uint64_t cur = atomicCAS((unsigned long long*)&keys[cand], (unsigned long long)EMPTY64,
                                               (unsigned long long)key.w[0]);
                inserted = cur == EMPTY64;
                got = inserted || cur == key.w[0];          // else the bucket changed under us: look at it again
            }
            if (got) { slot = cand; pending = false; }
            if (pending && !emp) bucket = bucket + 1 == nb ? 0 : bucket + 1;    // full of other keys: next bucket
        } while (__any(pending));
    } else {
        // word 0 is claimed by CAS, word 1 published right after.  A lane that sees word 0 match while word 1 is
        // still EMPTY leaves the probe loop and looks again in the next round of the outer, WAVE-UNIFORM loop: the
        // claimer -- possibly a lane of this wave -- is never waited for inside divergent code.
        bool todo = active;
        for (;;) {
            while (todo) {
                uint64_t cur = __hip_atomic_load(&keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cur == EMPTY64) {
                    cur = atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)EMPTY64,
                                    (unsigned long long)key.w[0]);
                    if (cur == EMPTY64) {
                        // the middle words first, the last word publishes (a lane's LDS stores complete in order)
#pragma unroll
                        for (int j = 1; j < KW - 1; j++)
                            __hip_atomic_store(&keys[(size_t)j * NS + slot], key.w[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(&keys[(size_t)(KW - 1) * NS + slot], key.w[KW - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        inserted = true; todo = false;
                        break;
                    }
                }
                if (cur == key.w[0]) {
                    const uint64_t c1 = __hip_atomic_load(&keys[(size_t)(KW - 1) * NS + slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (c1 == EMPTY64) break;              // not published yet: same slot again next round
                    bool same = c1 == key.w[KW - 1];
#pragma unroll
                    for (int j = 1; j < KW - 1; j++)
                        same = same && __hip_atomic_load(&keys[(size_t)j * NS + slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == key.w[j];
                    if (same) { todo = false; break; }
                }
                slot = slot + 1 == ns ? 0 : slot + 1;
            }
            if (!__any(todo)) break;
        }
    }
    bool over = false;
    if (active) {
        atomicMin(&ord[slot], myord);
        atomicOr(&bits[slot], bit);
        if (inserted) {
            uint32_t c = atomicAdd(&misc[0], 1u);
            if (c + 1 > limit) { misc[1] = 1; over = true; }   // overflow: the cluster is re-run with more partitions
        }
    }
    return over;
}

// forward and reverse-complement key of this lane's window; w[0..KW] = packed words at index 2u + (lane>>5) ...
// A key is the 2k-bit value of the k-mer (first base most significant, so integer order = the reference's string
// order on A/C/G/T) cut into KW words of 63 bits, most significant first -- 63 so that no word of a real key equals
// the EMPTY64 sentinel: k <= 31 one word, <= 63 two, <= 94 three, <= 126 four.
template <int KW>
__device__ __forceinline__ bool window_keys(uint32_t k, uint32_t lane, const uint64_t (&w)[KW + 1], Key<KW>& fwd, Key<KW>& rc) {
    const uint32_t sh = (lane & 31) << 1;
    if (KW == 1) {
        const uint64_t x = (w[0] << sh) | ((w[1] >> 1) >> (63 - sh));
        fwd.w[0] = x >> (64 - 2 * k);
        rc.w[0] = rev_groups(~x) & ((1ull << (2 * k)) - 1);   // low 2k bits of the reversed complement
        return rc.w[0] < fwd.w[0];
    } else {
        // x = the 64 KW bits that start at this lane's base, most significant word first
        uint64_t x[KW], f[KW], c[KW];
#pragma unroll
        for (int j = 0; j < KW; j++) x[j] = (w[j] << sh) | ((w[j + 1] >> 1) >> (63 - sh));
        // forward: the top 2k bits of x, right-aligned (shift right by r = 64 KW - 2k, 2 <= r < 64 (KW - 1) + 64)
        const uint32_t r = 64 * KW - 2 * k, rw = r >> 6, rb = r & 63;
#pragma unroll
        for (int j = 0; j < KW; j++) {
            const int a = j - (int)rw;                          // source word of the low part
            uint64_t v = 0;
#pragma unroll
            for (int t = 0; t < KW; t++) {
                if (t == a) v |= x[t] >> rb;
                if (t == a - 1 && rb) v |= x[t] << (64 - rb);
            }
            f[j] = v;
        }
        // reversed complement of the whole window: its low 2k bits are the k-mer's reverse complement
#pragma unroll
        for (int j = 0; j < KW; j++) c[j] = rev_groups(~x[KW - 1 - j]);
#pragma unroll
        for (int j = 0; j < KW; j++) {
            // keep bits below 2k: word j covers bits [64 (KW-1-j), 64 (KW-j))
            const int lo = 64 * (KW - 1 - j);
            const int keep = (int)(2 * k) - lo;                 // bits of this word that belong to the value
            if (keep <= 0) c[j] = 0;
            else if (keep < 64) c[j] &= (1ull << keep) - 1;
        }
        // 64-bit words -> 63-bit words (word KW-1 = bits 0..62, word KW-2 = bits 63..125, ...)
        auto split = [](const uint64_t (&v)[KW], Key<KW>& out) {
#pragma unroll
            for (int j = 0; j < KW; j++) {
                const int bit = 63 * (KW - 1 - j);              // lowest bit of 63-bit word j
                const int wi = KW - 1 - (bit >> 6), sb = bit & 63;      // 64-bit word holding it, offset inside
                uint64_t t = v[wi] >> sb;
                if (sb > 0 && wi > 0) t |= v[wi - 1] << (64 - sb);
                out.w[j] = t & 0x7FFFFFFFFFFFFFFFull;
            }
        };
        split(f, fwd);
        split(c, rc);
        bool less = false, decided = false;
#pragma unroll
        for (int j = 0; j < KW; j++) {
            if (!decided && rc.w[j] != fwd.w[j]) { less = rc.w[j] < fwd.w[j]; decided = true; }
        }
        return less;
    }
}

// One 64-window unit of one segment: fold this lane's window into the table.  True (wave-uniform) when an insert of
// this wave found the table past its limit.
template <int KW, bool CANON>
__device__ __forceinline__ bool scan_unit(uint64_t* keys, uint32_t* ord, uint32_t* bits, uint32_t* misc,
                                          uint32_t NS, uint32_t ns, uint32_t limit, uint32_t k, uint32_t lane,
                                          uint32_t part, uint32_t nparts, const uint64_t (&cw)[KW + 1],
                                          uint32_t u, uint32_t ninst, uint32_t ordb, uint32_t bit) {
    const uint32_t pos = (u << 6) + lane;
    const bool valid = pos < ninst;
    Key<KW> fwd, rc;
    const bool rc_smaller = window_keys<KW>(k, lane, cw, fwd, rc);
    const uint32_t nhome = KW == 1 ? ns / SCAN_BUCKET : ns;
    bool over;
    if (CANON) {
        Key<KW> key = rc_smaller ? rc : fwd;          // specseq <= revspecseq -> forward (panfeed.py:70)
        const uint32_t h = key_hash<KW>(key);
        const bool mine = nparts == 1 || (((h & 0xFFFFu) * nparts) >> 16) == part;
        over = table_update<KW>(keys, ord, bits, misc, NS, ns, limit, valid && mine, key, __umulhi(h, nhome), ordb + pos, bit);
    } else {
        // forward then reverse complement, both inserted (panfeed.py:82-88)
        uint32_t h = key_hash<KW>(fwd);
        bool mine = nparts == 1 || (((h & 0xFFFFu) * nparts) >> 16) == part;
        over = table_update<KW>(keys, ord, bits, misc, NS, ns, limit, valid && mine, fwd, __umulhi(h, nhome), 2 * (ordb + pos), bit);
        h = key_hash<KW>(rc);
        mine = nparts == 1 || (((h & 0xFFFFu) * nparts) >> 16) == part;
        over |= table_update<KW>(keys, ord, bits, misc, NS, ns, limit, valid && mine, rc, __umulhi(h, nhome), 2 * (ordb + pos) + 1, bit);
    }
    return __any(over);
}

// ---- key partitions: a queue of "mine" windows per wave
// An item that is one of P key partitions of its cluster looks at every window of the view and keeps the 1 / P whose key
// hashes into its partition.  Going through the table's insert loop once per unit -- the loop is wave-uniform and runs
// as long as its slowest lane -- made a partition's scan cost as much as a whole scan, for 64 / P inserts per trip.  The
// windows a wave keeps are therefore gathered over several units in its lanes (lane i < n holds pending entry i; a
// unit's kept windows are appended behind them, pulled from their lanes through a 64-byte list of lane numbers in LDS)
// and go through the insert loop 64 at a time.
template <int KW>
struct ScanQueue { Key<KW> key; uint32_t ord, bit; };
template <>
struct ScanQueue<1> { Key<1> key; uint32_t ord, bit, home; };   // home: the bucket the entry's walk through the table starts at
                                                                 // (one-word keys only: the wider kernels have no register for it)

constexpr uint32_t M_WQ_WORDS = 16;            // 64 lane numbers per wave

template <int KW>
__device__ __forceinline__ bool queue_flush(uint64_t* keys, uint32_t* ord, uint32_t* bits, uint32_t* misc, uint32_t NS,
                                            uint32_t ns, uint32_t limit, uint32_t lane, ScanQueue<KW>& pq, uint32_t& pn) {
    if (!pn) return false;                                           // (uniform)
    uint32_t home;
    if constexpr (KW == 1) home = pq.home;
    else home = __umulhi(key_hash<KW>(pq.key), ns);
    const bool over = table_update<KW>(keys, ord, bits, misc, NS, ns, limit, lane < pn, pq.key, home, pq.ord, pq.bit);
    pn = 0;
    return __any(over);
}

template <int KW>
__device__ __forceinline__ bool queue_push(uint64_t* keys, uint32_t* ord, uint32_t* bits, uint32_t* misc, uint32_t NS,
                                           uint32_t ns, uint32_t limit, uint32_t lane, uint8_t* wq, ScanQueue<KW>& pq,
                                           uint32_t& pn, bool mine, const Key<KW>& key, uint32_t home, uint32_t myord, uint32_t bit) {
    const uint64_t mm = __ballot(mine);
    if (!mm) return false;                                           // (uniform)
    const uint32_t m = (uint32_t)__popcll(mm);
    bool over = false;
    if (pn + m > 64) over = queue_flush<KW>(keys, ord, bits, misc, NS, ns, limit, lane, pq, pn);
    const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
    if (mine) wq[r] = (uint8_t)lane;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");           // (the list is this wave's own: its LDS operations run in order)
    __builtin_amdgcn_wave_barrier();
    const bool take = lane >= pn && lane < pn + m;
    const uint32_t src = take ? (uint32_t)wq[lane - pn] : lane;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();                                 // the list is rewritten by the next push
#pragma unroll
    for (int j = 0; j < KW; j++) {
        const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)key.w[j], (int)src);
        const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(key.w[j] >> 32), (int)src);
        if (take) pq.key.w[j] = ((uint64_t)hi << 32) | lo;
    }
    const uint32_t o = (uint32_t)__shfl((int)myord, (int)src), b = (uint32_t)__shfl((int)bit, (int)src);
    if (take) { pq.ord = o; pq.bit = b; }
    if constexpr (KW == 1) {
        const uint32_t hm = (uint32_t)__shfl((int)home, (int)src);
        if (take) pq.home = hm;
    }
    pn += m;
    return over;
}

// One 64-window unit of an item that is one key partition of several: this wave's kept windows join its queue.
template <int KW, bool CANON>
__device__ __forceinline__ bool scan_unit_queued(uint64_t* keys, uint32_t* ord, uint32_t* bits, uint32_t* misc, uint32_t NS,
                                                 uint32_t ns, uint32_t limit, uint32_t k, uint32_t lane, uint32_t part,
                                                 uint32_t nparts, const uint64_t (&cw)[KW + 1], uint32_t u, uint32_t ninst,
                                                 uint32_t ordb, uint32_t bit, uint8_t* wq, ScanQueue<KW>& pq, uint32_t& pn) {
    const uint32_t pos = (u << 6) + lane;
    const bool valid = pos < ninst;
    Key<KW> fwd, rc;
    const bool rc_smaller = window_keys<KW>(k, lane, cw, fwd, rc);
    const uint32_t nhome = KW == 1 ? ns / SCAN_BUCKET : ns;
    if (CANON) {
        Key<KW> key = rc_smaller ? rc : fwd;          // specseq <= revspecseq -> forward (panfeed.py:70)
        const uint32_t h = key_hash<KW>(key);
        const bool mine = valid && (((h & 0xFFFFu) * nparts) >> 16) == part;
        return queue_push<KW>(keys, ord, bits, misc, NS, ns, limit, lane, wq, pq, pn, mine, key, __umulhi(h, nhome), ordb + pos, bit);
    } else {
        uint32_t h = key_hash<KW>(fwd);
        bool mine = valid && (((h & 0xFFFFu) * nparts) >> 16) == part;
        bool over = queue_push<KW>(keys, ord, bits, misc, NS, ns, limit, lane, wq, pq, pn, mine, fwd, __umulhi(h, nhome), 2 * (ordb + pos), bit);
        h = key_hash<KW>(rc);
        mine = valid && (((h & 0xFFFFu) * nparts) >> 16) == part;
        over |= queue_push<KW>(keys, ord, bits, misc, NS, ns, limit, lane, wq, pq, pn, mine, rc, __umulhi(h, nhome), 2 * (ordb + pos) + 1, bit);
        return over;
    }
}

// ONE look at a one-word key's bucket (`bucket`: its home, or where an earlier look left off): the key is found or takes the
// first empty slot -- ordinal, presence bits and the table's key count follow -- or the lane is left over (true), to look at
// `next` later: the same bucket when it lost a race for the slot, the following one when the bucket is full of other keys.
__device__ __forceinline__ bool table_first_trip(uint64_t* keys, uint32_t* ord, uint32_t* bits, uint32_t* misc, uint32_t ns,
                                                 uint32_t limit, bool active, const Key<1>& key, uint32_t bucket, uint32_t myord,
                                                 uint32_t bit, bool& over, uint32_t& next) {
    const uint32_t nb = ns / SCAN_BUCKET;
    ulonglong2 kk[SCAN_BUCKET / 2];
#pragma unroll
    for (uint32_t j = 0; j < SCAN_BUCKET / 2; j++) kk[j] = *reinterpret_cast<const ulonglong2*>(&keys[SCAN_BUCKET * bucket + 2 * j]);
    bool hit = false, emp = false;
    uint32_t ih = 0, ie = 0;
#pragma unroll
    for (int j = SCAN_BUCKET / 2 - 1; j >= 0; j--) {             // downwards: the lowest index wins
        if (kk[j].y == key.w[0]) { hit = true; ih = 2 * j + 1; }
        if (kk[j].x == key.w[0]) { hit = true; ih = 2 * j; }
        if (kk[j].y == EMPTY64) { emp = true; ie = 2 * j + 1; }
        if (kk[j].x == EMPTY64) { emp = true; ie = 2 * j; }
    }
    const uint32_t cand = SCAN_BUCKET * bucket + (hit ? ih : ie);
    bool got = active && hit, inserted = false;
    if (active && !hit && emp) {
        const uint64_t cur = atomicCAS((unsigned long long*)&keys[cand], (unsigned long long)EMPTY64, (unsigned long long)key.w[0]);
        inserted = cur == EMPTY64;
        got = inserted || cur == key.w[0];
    }
    if (got) {
        atomicMin(&ord[cand], myord);
        atomicOr(&bits[cand], bit);
        if (inserted) {
            const uint32_t c = atomicAdd(&misc[0], 1u);
            if (c + 1 > limit) { misc[1] = 1; over = true; }   // overflow: the cluster is re-run with more partitions
        }
    }
    next = emp ? bucket : (bucket + 1 == nb ? 0 : bucket + 1);
    return active && !got;
}

// One-word keys, an item that is its cluster's only key partition: the table's insert loop runs as long as the slowest of a
// unit's 64 lanes, and the trips after the first serve the few lanes that found their home bucket full or lost a race for a
// slot -- 0.9 of the scan's 4.1 ms on the headline workload (one trip per unit and the rest dropped, timing only: 3.22 ms).
// Here a unit makes exactly ONE trip; the lanes it leaves unplaced join the wave's queue (the one key partitions use) and go
// through the full loop 64 at a time.  Order does not matter to what the table ends up holding (smallest ordinal, OR of bits).
template <bool CANON>
__device__ __forceinline__ bool scan_unit_deferred(uint64_t* keys, uint32_t* ord, uint32_t* bits, uint32_t* misc, uint32_t NS,
                                                   uint32_t ns, uint32_t limit, uint32_t k, uint32_t lane,
                                                   const uint64_t (&cw)[2], uint32_t u, uint32_t ninst, uint32_t ordb, uint32_t bit,
                                                   uint8_t* wq, ScanQueue<1>& pq, uint32_t& pn) {
    const uint32_t pos = (u << 6) + lane;
    const bool valid = pos < ninst;
    Key<1> fwd, rc;
    const bool rc_smaller = window_keys<1>(k, lane, cw, fwd, rc);
    const uint32_t nb = ns / SCAN_BUCKET;
    auto first_trip = [&](bool active, const Key<1>& key, uint32_t myord, bool& over, uint32_t& next) -> bool {
        return table_first_trip(keys, ord, bits, misc, ns, limit, active, key, __umulhi(key_hash<1>(key), nb), myord, bit, over, next);
    };
    bool over = false;
    if (CANON) {
        const Key<1> key = rc_smaller ? rc : fwd;          // specseq <= revspecseq -> forward (panfeed.py:70)
        uint32_t next;
        const bool left = first_trip(valid, key, ordb + pos, over, next);
        over |= queue_push<1>(keys, ord, bits, misc, NS, ns, limit, lane, wq, pq, pn, left, key, next, ordb + pos, bit);
    } else {
        // forward then reverse complement, both inserted (panfeed.py:82-88)
        uint32_t next;
        bool left = first_trip(valid, fwd, 2 * (ordb + pos), over, next);
        over |= queue_push<1>(keys, ord, bits, misc, NS, ns, limit, lane, wq, pq, pn, left, fwd, next, 2 * (ordb + pos), bit);
        left = first_trip(valid, rc, 2 * (ordb + pos) + 1, over, next);
        over |= queue_push<1>(keys, ord, bits, misc, NS, ns, limit, lane, wq, pq, pn, left, rc, next, 2 * (ordb + pos) + 1, bit);
    }
    return __any(over);
}

// chunkmask word 0 is accumulated by thread 0
__device__ __forceinline__ uint32_t mask_word_any(uint32_t mask_word, uint32_t tid) { return tid == 0 ? mask_word : 0u; }

// misc[] layout (uint32 words)
constexpr uint32_t M_COUNT = 0, M_OVERFLOW = 1, M_CHUNK = 2;            // chunk offsets: MAX_CHUNKS + 2 words
constexpr uint32_t SEG_TILE = 256;                                       // segments staged per tile
constexpr uint32_t M_WOFF = 272;                                         // [2*SEG_TILE] word offset (lo, hi)
constexpr uint32_t M_NINST = M_WOFF + 2 * SEG_TILE;
constexpr uint32_t M_ORDB = M_NINST + SEG_TILE;
constexpr uint32_t M_SAMPLE = M_ORDB + SEG_TILE;
constexpr uint32_t M_UPREF = M_SAMPLE + SEG_TILE;                        // [SEG_TILE + 1] unit prefix
constexpr uint32_t M_TMP = M_UPREF + SEG_TILE + 2;                       // one scratch word
constexpr uint32_t M_PROG = M_TMP + 1;                                   // unit index at which the table was found full
constexpr uint32_t M_DESC = M_TMP + 2;                                   // [16] the current item's ScanDesc
constexpr uint32_t M_NDESC = M_DESC + 16;                                // [16] the next item's
constexpr uint32_t M_BITS = M_NDESC + 16;                                // [SEG_TILE] presence bits of the staged segments
constexpr uint32_t M_WQ = M_BITS + SEG_TILE;                             // [SCAN_WAVES][M_WQ_WORDS] lane lists of the waves' queues
static_assert(M_WQ + SCAN_WAVES * M_WQ_WORDS <= MISC_WORDS, "misc area too small");

// Persistent: gridDim.x workgroups (one per CU: the table takes the whole LDS) walk work entries
// blockIdx.x, blockIdx.x + gridDim.x, ...  With one workgroup per CU nothing else hides the dependent global
// loads an item starts with, so the next item's descriptor is requested at the start of the current one and its
// first tile of segment metadata before the current table is dumped.
template <int KW, bool CANON>
__global__ __launch_bounds__(SCAN_THREADS) void kmer_scan_kernel(ScanParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t NS = p.NS;
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);          // [KW][NS]
    uint32_t* ord = reinterpret_cast<uint32_t*>(keys + (size_t)KW * NS);
    uint32_t* bits = ord + NS;
    uint32_t* misc = bits + NS;

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t k = p.k;

    // first-tile metadata of the item about to start, one segment per thread (tid < SEG_TILE)
    uint32_t pm_len = 0, pm_ordb = 0, pm_sample = 0, pm_bits = 0;
    uint64_t pm_wo = 0;
    auto fetch_tile0 = [&](uint32_t seg0raw, uint32_t nseg) {
        if (tid < min(nseg, SEG_TILE)) {
            const uint32_t s = (seg0raw & ~VIEW_IN_POOL) + tid;
            if (seg0raw & VIEW_IN_POOL) {
                pm_len = p.u_len[s]; pm_wo = p.u_word_off[s]; pm_ordb = p.u_ord_base[s]; pm_sample = p.u_sample[s];
                pm_bits = p.u_bits[s];
            } else {
                pm_len = p.seg_len[s]; pm_wo = p.seg_word_off[s]; pm_ordb = p.seg_ord_base[s]; pm_sample = p.seg_sample[s];
                pm_bits = p.seg_bits[s];
            }
        }
    };
    if (blockIdx.x < p.n_work) {
        if (tid < 16) misc[M_DESC + tid] = reinterpret_cast<const uint32_t*>(p.desc + blockIdx.x)[tid];
        __syncthreads();
        fetch_tile0(misc[M_DESC + 6], misc[M_DESC + 7]);
    }

    // The table is emptied ONCE here; after that every item leaves it empty behind itself -- the dump looks at every slot
    // anyway and resets the occupied ones (a third of them) as it goes, where a clearing pass over all slots at the start
    // of every item was 8 % of the kernel.
    for (uint32_t i = tid; i < NS; i += SCAN_THREADS) {
#pragma unroll
        for (int j = 0; j < KW; j++) keys[(size_t)j * NS + i] = EMPTY64;
        ord[i] = NO_ORD;
        bits[i] = 0;
    }
    PF_PROF_BEGIN();
    uint32_t dcur = M_DESC, dnxt = M_NDESC;      // the two descriptor slots take turns: nothing is copied between items
    for (uint32_t wi = blockIdx.x; wi < p.n_work; wi += gridDim.x) {
    // ---- the current item (its descriptor was written before the barrier in front of the previous item's dump)
    const uint32_t item = misc[dcur + 0], c = misc[dcur + 1];
    const uint32_t part = misc[dcur + 2], nparts = misc[dcur + 3];
    const uint32_t ns = misc[dcur + 4], slice = misc[dcur + 5];
    const bool in_pool = (misc[dcur + 6] & VIEW_IN_POOL) != 0;
    const uint32_t seg0 = misc[dcur + 6] & ~VIEW_IN_POOL, seg1 = seg0 + misc[dcur + 7];
    const uint64_t* const a_woff = in_pool ? p.u_word_off : p.seg_word_off;
    const uint32_t* const a_len = in_pool ? p.u_len : p.seg_len;
    const uint32_t* const a_sample = in_pool ? p.u_sample : p.seg_sample;
    const uint32_t* const a_ordb = in_pool ? p.u_ord_base : p.seg_ord_base;
    const uint32_t* const a_bits = in_pool ? p.u_bits : p.seg_bits;
    const uint32_t nstr = misc[dcur + 8];
    const bool compact = misc[dcur + 9] != 0;  // view has <= 64 columns: at most chunks 0 and 1
    const bool binned = misc[dcur + 10] != 0;  // the cluster's windows were sorted by key partition (bin_kernel)
    const uint32_t nchunks = (nstr + 31) >> 5;
    const uint32_t limit = insert_limit(ns);
    // request the next item's descriptor now; it is looked at after the scan loop
    const uint32_t wn = wi + gridDim.x;
    uint32_t nd_word = 0;
    if (wn < p.n_work && tid < 16) nd_word = reinterpret_cast<const uint32_t*>(p.desc + wn)[tid];

    if (tid < 2) misc[tid] = 0;
    if (tid == 2) misc[M_PROG] = 0xFFFFFFFFu;
    uint32_t ubefore = 0;     // units of the tiles already walked
    const bool one_tile = seg1 - seg0 <= SEG_TILE;
    if (!one_tile) {
        for (uint32_t ch = tid; ch <= nchunks; ch += SCAN_THREADS)
            misc[M_CHUNK + ch] = seg_lower_bound(a_sample, seg0, seg1, ch << 5);
    }
    {   // tile 0 from the registers filled during the previous item
        const uint32_t nseg = min(SEG_TILE, seg1 - seg0);
        if (tid < nseg) {
            misc[M_WOFF + 2 * tid] = (uint32_t)pm_wo;
            misc[M_WOFF + 2 * tid + 1] = (uint32_t)(pm_wo >> 32);
            misc[M_NINST + tid] = pm_len >= k ? pm_len - k + 1 : 0;
            misc[M_ORDB + tid] = pm_ordb;
            misc[M_SAMPLE + tid] = pm_sample;
            misc[M_BITS + tid] = pm_bits;
        }
    }
    __syncthreads();
    if (one_tile) {
        // chunk boundaries from the staged sample columns (sorted): no global search
        const uint32_t nseg = seg1 - seg0;
        for (uint32_t ch = tid; ch <= nchunks; ch += SCAN_THREADS) {
            uint32_t a = 0, b = nseg;
            const uint32_t v = ch << 5;
            while (a < b) { const uint32_t m = (a + b) >> 1; if (misc[M_SAMPLE + m] < v) a = m + 1; else b = m; }
            misc[M_CHUNK + ch] = seg0 + a;
        }
    }

    PF_PROF_STAMP(16);
    uint32_t mask_word = 0;   // thread t < 8 accumulates chunkmask word t
    bool overflow = false;
    uint32_t ch = 0;          // current sample chunk (32 columns); its bits[] words are live in LDS
    bool chunk_dirty = false; // some segment of chunk `ch` has been scanned since the last flush

    // flush the presence words of chunk `ch` (coalesced) and clear them
    auto flush_chunk = [&]() {
        uint32_t* dst = p.chunkbits + ((size_t)slice * p.W + ch) * NS;
        for (uint32_t i = tid; i < ns; i += SCAN_THREADS) {
            dst[i] = bits[i];
            bits[i] = 0;
        }
        if (tid == (ch >> 5)) mask_word |= 1u << (ch & 31);
    };

    if (binned) {
        // ---- the item's own windows, chunk after chunk, 64 entries a wave and trip.  (Walking the view, an item that
        // is one of P key partitions works out the key of EVERY window to keep one in P: at ten partitions the scan of
        // a cluster cost ten scans' worth of key arithmetic, 80 % of its time.)
        __syncthreads();                           // misc[M_CHUNK ..] was written above
        for (uint32_t i = tid; i <= nchunks; i += SCAN_THREADS) misc[M_CHUNK + i] = p.q_off[(size_t)item * (BIN_CHUNKS + 1) + i];
        __syncthreads();
        const uint32_t qbeg = misc[M_CHUNK];
        const uint32_t nhome = KW == 1 ? ns / SCAN_BUCKET : ns;
        for (uint32_t qc = 0; qc < nchunks && !overflow; qc++) {
            const uint32_t e0 = misc[M_CHUNK + qc], e1 = misc[M_CHUNK + qc + 1];
            if (e0 == e1) continue;                // (uniform)
            if (qc != ch) {
                if (chunk_dirty) { flush_chunk(); chunk_dirty = false; __syncthreads(); }
                ch = qc;
            }
            auto load = [&](uint32_t at, Key<KW>& kk, uint32_t& oo, uint32_t& bb) {
                const uint32_t i = min(at + lane, e1 - 1);
#pragma unroll
                for (int j = 0; j < KW; j++) kk.w[j] = p.q_key[(size_t)j * p.q_stride + i];
                oo = p.q_ord[i]; bb = p.q_bit[i];
            };
            uint32_t e = e0 + wave * 64;
            Key<KW> key{};
            uint32_t eo = 0, eb = 0;
            if (e < e1) load(e, key, eo, eb);
            while (e < e1) {
                const uint32_t en = e + SCAN_WAVES * 64;
                Key<KW> nkey;
                uint32_t no, nb;
                load(en < e1 ? en : e, nkey, no, nb);      // unconditional: one request in flight across the table work
                const uint32_t h = key_hash<KW>(key);
                // (one look per entry and the leftovers through the wave's queue, as scan_unit_deferred has it, was measured here
                // too: nothing, 6.5 - 6.7 ms of binning + scan either way at ~150 SURVEY alleles)
                const bool over = table_update<KW>(keys, ord, bits, misc, NS, ns, limit, e + lane < e1, key, __umulhi(h, nhome), eo, eb);
                if (__any(over)) {
                    if (lane == 0) atomicMin(&misc[M_PROG], (e - qbeg) >> 6);
                    break;
                }
                key = nkey; eo = no; eb = nb;
                e = en;
            }
            chunk_dirty = true;
            __syncthreads();
            if (misc[M_OVERFLOW]) overflow = true;
        }
        PF_PROF_STAMP(18);
    }
    for (uint32_t t0 = seg0; t0 < seg1 && !overflow && !binned; t0 += SEG_TILE) {
        const uint32_t nseg = min(SEG_TILE, seg1 - t0);
        // ---- stage this tile's segment metadata (coalesced; tile 0 is there already), then unit prefix by wave 0
        if (t0 != seg0 && tid < nseg) {
            const uint32_t s = t0 + tid;
            const uint32_t len = a_len[s];
            const uint64_t wo = a_woff[s];
            misc[M_WOFF + 2 * tid] = (uint32_t)wo;
            misc[M_WOFF + 2 * tid + 1] = (uint32_t)(wo >> 32);
            misc[M_NINST + tid] = len >= k ? len - k + 1 : 0;
            misc[M_ORDB + tid] = a_ordb[s];
            misc[M_SAMPLE + tid] = a_sample[s];
            misc[M_BITS + tid] = a_bits[s];
        }
        __syncthreads();
        if (wave == 0) {
            uint32_t v[4], sum = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t i = lane * 4 + j;
                v[j] = i < nseg ? (misc[M_NINST + i] + 63) >> 6 : 0;
                sum += v[j];
            }
            uint32_t x = sum;
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(x, d);
                if ((int)lane >= d) x += y;
            }
            uint32_t run = x - sum;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t i = lane * 4 + j;
                if (i <= nseg) misc[M_UPREF + i] = run;
                run += v[j];
            }
            if (lane == 63) misc[M_UPREF + SEG_TILE] = x;   // total (index 256 is not covered above)
        }
        __syncthreads();
        PF_PROF_STAMP(17);

        // ---- walk the chunks that intersect this tile
        uint32_t lo = 0;   // local segment index
        while (lo < nseg) {
            const uint32_t want = misc[M_SAMPLE + lo] >> 5;
            if (want != ch) {
                // a new chunk starts: flush the previous one first
                if (chunk_dirty) { flush_chunk(); chunk_dirty = false; __syncthreads(); }
                ch = want;
            }
            // local end of this chunk inside the tile: chunk offsets are absolute segment indices
            const uint32_t cend = misc[M_CHUNK + ch + 1];
            const uint32_t hi = min(nseg, cend - t0);
            const uint32_t ubase = misc[M_UPREF + lo], utot = misc[M_UPREF + hi] - ubase;
            // ---- this wave's units: ubase + wave, + 16, ... -- dealt round robin, NOT in contiguous shares: the units
            // of the chunk's first sequence insert 64 new keys each (CAS + counter round trips) and cost about twice
            // a unit of a later, nearly identical sequence; with contiguous shares the one or two waves that got the
            // first sequence kept the other fourteen waiting at the barrier for half of the chunk's time.  The next
            // unit's words are requested before this one is processed.
            uint32_t g = ubase + wave;
            const uint32_t gend = ubase + utot;
            if (g < gend) {
                uint32_t s = in_pool ? lo + (g - ubase) : lo;      // (a piece of the unit view holds exactly one unit)
                while (g >= misc[M_UPREF + s + 1]) s++;
                uint32_t send = misc[M_UPREF + s + 1];      // first unit past segment s
                uint32_t u = g - misc[M_UPREF + s];
                uint32_t ninst = misc[M_NINST + s], ordb = misc[M_ORDB + s], bit = misc[M_BITS + s];
                const uint64_t* q = p.packed + (((uint64_t)misc[M_WOFF + 2 * s + 1] << 32) | misc[M_WOFF + 2 * s]) +
                                    2 * (size_t)u + (lane >> 5);
                uint64_t cw[KW + 1];
#pragma unroll
                for (int j = 0; j <= KW; j++) cw[j] = q[j];
                // one of several key partitions (one- and two-word keys: the wider kernels have no registers to spare):
                // this wave's kept windows are queued and inserted 64 at a time
                const bool queued = KW <= 2 && nparts > 1;
                const bool deferred = KW == 1 && nparts == 1;       // (a unit makes one trip to the table; stragglers queue up)
                ScanQueue<KW> pq{};
                uint32_t pn = 0;
                uint8_t* const wq = reinterpret_cast<uint8_t*>(misc + M_WQ + wave * M_WQ_WORDS);
                bool stopped = false;
                while (g < gend) {
                    // unconditional prefetch (re-reads the current address past the end) so that the compiler
                    // keeps exactly one load in flight across the table work: s_waitcnt vmcnt(1), not 0
                    const uint32_t gn = g + SCAN_WAVES;
                    const uint64_t* qn = q;
                    uint32_t sn = s, sendn = send, un = u, ninstn = ninst, ordbn = ordb, bitn = bit;
                    if (gn < gend) {
                        if (gn < send) { qn = q + 2 * SCAN_WAVES; un = u + SCAN_WAVES; }
                        else {
                            sn = in_pool ? lo + (gn - ubase) : s + 1;
                            while (gn >= misc[M_UPREF + sn + 1]) sn++;
                            sendn = misc[M_UPREF + sn + 1];
                            un = gn - misc[M_UPREF + sn];
                            ninstn = misc[M_NINST + sn]; ordbn = misc[M_ORDB + sn];
                            bitn = misc[M_BITS + sn];
                            qn = p.packed + (((uint64_t)misc[M_WOFF + 2 * sn + 1] << 32) | misc[M_WOFF + 2 * sn]) +
                                 2 * (size_t)un + (lane >> 5);
                        }
                    }
                    uint64_t nw[KW + 1];
#pragma unroll
                    for (int j = 0; j <= KW; j++) nw[j] = qn[j];
                    // table past its limit: this wave stops inserting (the cluster is re-run with more key
                    // partitions).  A wave learns it from its own inserts -- no look at the flag per unit, that is an
                    // LDS round trip in front of every unit (looking every fourth unit was measured too: slower, 4.36 ->
                    // 4.55 ms) -- so after the limit trips every wave finishes at most the unit it is in:
                    // 16 waves x 64 lanes x 2 keys < INSERT_SLACK.
                    bool over;
                    if constexpr (KW == 1) {
                        over = deferred ? scan_unit_deferred<CANON>(keys, ord, bits, misc, NS, ns, limit, k, lane, cw, u, ninst, ordb, bit, wq, pq, pn)
                                        : scan_unit_queued<KW, CANON>(keys, ord, bits, misc, NS, ns, limit, k, lane, part, nparts, cw, u, ninst,
                                                                      ordb, bit, wq, pq, pn);
                    } else {
                        over = queued
                            ? scan_unit_queued<KW, CANON>(keys, ord, bits, misc, NS, ns, limit, k, lane, part, nparts, cw, u, ninst,
                                                          ordb, bit, wq, pq, pn)
                            : scan_unit<KW, CANON>(keys, ord, bits, misc, NS, ns, limit, k, lane, part, nparts, cw, u, ninst, ordb, bit);
                    }
                    if (over) {
                        if (lane == 0) atomicMin(&misc[M_PROG], ubefore + g);
                        stopped = true;
                        break;
                    }
#pragma unroll
                    for (int j = 0; j <= KW; j++) cw[j] = nw[j];
                    q = qn; s = sn; send = sendn; u = un; ninst = ninstn; ordb = ordbn; bit = bitn;
                    g = gn;
                }
                // what is still queued belongs to this chunk's words: in before they are flushed
                if ((queued || deferred) && !stopped && queue_flush<KW>(keys, ord, bits, misc, NS, ns, limit, lane, pq, pn)) {
                    if (lane == 0) atomicMin(&misc[M_PROG], ubefore + gend - 1);
                }
            }
            chunk_dirty = true;
            lo = hi;
            __syncthreads();                       // the chunk part is complete in LDS
            PF_PROF_STAMP(18);
            if (misc[M_OVERFLOW]) { overflow = true; break; }
        }
        ubefore += misc[M_UPREF + SEG_TILE];       // (rewritten only after the next tile's first barrier)
        // the next tile overwrites the staged metadata: everyone is past the barrier above
    }
    if (!overflow && chunk_dirty && !compact) { flush_chunk(); }
    // ---- the next item: descriptor to LDS, first tile of its segment metadata into registers (in flight
    // while this item's table is written out)
    if (tid < 16) misc[dnxt + tid] = nd_word;
    __syncthreads();
    if (wn < p.n_work) fetch_tile0(misc[dnxt + 6], misc[dnxt + 7]);
    PF_PROF_STAMP(19);

    if (overflow) {
        // How far the item had come when its table was full tells the host how many key partitions the cluster needs:
        // the same keys arrive at about the same rate in every partition and all through the item (the first sequence
        // of a cluster brings more new keys than the later ones: the extrapolation errs on the safe side).
        uint32_t tot = 0;
        if (binned) {
            if (tid == 0) tot = (misc[M_CHUNK + nchunks] - misc[M_CHUNK] + 63) >> 6;     // (its progress counts 64 entries)
        } else {
            for (uint32_t s = seg0 + tid; s < seg1; s += SCAN_THREADS) {
                const uint32_t len = a_len[s];
                tot += len >= k ? (len - k + 64) >> 6 : 0;
            }
        }
        for (int d = 1; d < 64; d <<= 1) tot += __shfl_xor(tot, d);
        if (tid == 0) misc[M_TMP] = 0;
        __syncthreads();
        if (lane == 0 && tot) atomicAdd(&misc[M_TMP], tot);
        __syncthreads();
        if (tid == 0) {
            const uint64_t all = misc[M_TMP], done = min(misc[M_PROG], (uint32_t)all) + 1;
            const uint64_t ratio = min((all * 64 + done - 1) / done, (uint64_t)0x7FFFFFFFu);
            atomicMax(&p.cluster_overflow[c], (uint32_t)max(ratio, (uint64_t)64));
            p.item_count[item] = 0;
        }
        for (uint32_t i = tid; i < ns; i += SCAN_THREADS) {      // nothing is dumped: the table is emptied as a whole
#pragma unroll
            for (int j = 0; j < KW; j++) keys[(size_t)j * NS + i] = EMPTY64;
            ord[i] = NO_ORD;
            bits[i] = 0;
        }
    } else if (compact) {
        // chunk 0 words were flushed to global memory iff a chunk 1 followed; the last chunk is still in bits[]
        const bool last_live = chunk_dirty;
        const bool c0_flushed = (mask_word_any(mask_word, tid) & 1u) != 0;
        if (tid == 0) { misc[M_TMP] = c0_flushed ? 1u : 0u; misc[M_COUNT] = 0; }
        __syncthreads();
        const bool have_c0 = misc[M_TMP] != 0;
        const uint32_t* g0 = p.chunkbits + ((size_t)slice * p.W) * NS;
        // occupied slots -> consecutive entries.  All of a thread's slots are looked at first (their LDS reads in flight
        // together), positions inside the wave's block come from ballots, and ONE LDS atomic per wave reserves the block:
        // an atomic per trip was ten dependent LDS round trips per wave with nothing to hide them.
        constexpr uint32_t DUMP_U = (nslots_max(KW) + SCAN_THREADS - 1) / SCAN_THREADS;
        uint32_t o_[DUMP_U], at_[DUMP_U];
        uint32_t tot = 0;
#pragma unroll
        for (uint32_t u = 0; u < DUMP_U; u++) {
            const uint32_t i = tid + u * SCAN_THREADS;
            o_[u] = i < ns ? ord[i] : NO_ORD;
        }
#pragma unroll
        for (uint32_t u = 0; u < DUMP_U; u++) {
            const uint64_t m = __ballot(o_[u] != NO_ORD);
            at_[u] = tot + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            tot += (uint32_t)__popcll(m);
        }
        uint32_t blk = 0;
        if (lane == 0 && tot) blk = atomicAdd(&misc[M_COUNT], tot);
        blk = __builtin_amdgcn_readfirstlane(blk);
#pragma unroll
        for (uint32_t u = 0; u < DUMP_U; u++) {
            const uint32_t i = tid + u * SCAN_THREADS;
            const uint32_t o = o_[u];
            if (o == NO_ORD) continue;
            uint32_t lo = 0, hi = 0;
            if (last_live && ch == 0) lo = bits[i];
            else if (have_c0) lo = g0[i];
            if (last_live && ch == 1) hi = bits[i];
            const uint32_t e = blk + at_[u];
#pragma unroll
            for (int j = 0; j < KW; j++) p.tab_key[((size_t)slice * KW + j) * NS + e] = keys[(size_t)j * NS + i];
            p.tab_ord[(size_t)slice * NS + e] = o;
            p.cmask_lo[(size_t)slice * NS + e] = lo;
            if (nstr > 32) p.cmask_hi[(size_t)slice * NS + e] = hi;      // (at most 32 columns: nobody reads the upper word)
#pragma unroll
            for (int j = 0; j < KW; j++) keys[(size_t)j * NS + i] = EMPTY64;
            ord[i] = NO_ORD;
            bits[i] = 0;
        }
        __syncthreads();
        if (tid == 0) p.item_count[item] = misc[M_COUNT];
    } else {
        for (uint32_t i = tid; i < ns; i += SCAN_THREADS) {
#pragma unroll
            for (int j = 0; j < KW; j++) p.tab_key[((size_t)slice * KW + j) * NS + i] = keys[(size_t)j * NS + i];
            p.tab_ord[(size_t)slice * NS + i] = ord[i];
#pragma unroll
            for (int j = 0; j < KW; j++) keys[(size_t)j * NS + i] = EMPTY64;
            ord[i] = NO_ORD;                                    // (bits[] went out, and to zero, with the last chunk)
        }
        if (tid < 8) p.chunkmask[slice * 8 + tid] = mask_word;
        if (tid == 0) p.item_count[item] = misc[M_COUNT];
    }
    __syncthreads();                               // the table and misc[] are free again
    PF_PROF_STAMP(20);
    { const uint32_t t = dcur; dcur = dnxt; dnxt = t; }
    PF_PROF_STAMP(21);
#ifdef PF_PROF
    if (tid == 0) atomicAdd(&pf_prof[24], 1ull);
#endif
    }
}

// ---------------------------------------------------------------------------------------------
// bin_kernel: the windows of a cluster that takes several key partitions, sorted by partition and chunk
// ---------------------------------------------------------------------------------------------
// One workgroup per cluster walks the cluster's view twice: it counts the windows of every (key partition, chunk) cell,
// gives every cell its place in the cluster's stretch of the entry arrays (partition-major: an item's entries are
// contiguous, chunk after chunk), then writes (key, ordinal, presence bit) of every window to its cell.  The key
// arithmetic is done twice per window here instead of once per window and partition in the scan.  The order of the
// entries inside a cell depends on the run; what the scan makes of them (smallest ordinal, OR of the bits) does not.
constexpr uint32_t BIN_THREADS = 1024, BIN_CELLS = 8192;      // partitions x chunks of a binned cluster at most
template <int KW, bool CANON>
__global__ __launch_bounds__(BIN_THREADS) void bin_kernel(ScanParams p) {
    __shared__ uint32_t cell[BIN_CELLS];           // windows per cell, then the cell's write position
    __shared__ uint32_t wave_tot[BIN_THREADS / 64 + 1];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t b = blockIdx.x;
    const uint32_t c = p.bin_cluster[b], item0 = p.bin_item0[b], P = p.bin_nparts[b], base = p.bin_base[b];
    const uint32_t raw = p.cluster_seg_off[c];
    const bool in_pool = (raw & VIEW_IN_POOL) != 0;
    const uint32_t seg0 = raw & ~VIEW_IN_POOL, seg1 = seg0 + p.cluster_vnseg[c];
    const uint64_t* const a_woff = in_pool ? p.u_word_off : p.seg_word_off;
    const uint32_t* const a_len = in_pool ? p.u_len : p.seg_len;
    const uint32_t* const a_sample = in_pool ? p.u_sample : p.seg_sample;
    const uint32_t* const a_ordb = in_pool ? p.u_ord_base : p.seg_ord_base;
    const uint32_t* const a_bits = in_pool ? p.u_bits : p.seg_bits;
    const uint32_t nch = (p.cluster_vnstr[c] + 31) >> 5;          // <= BIN_CHUNKS (host)
    const uint32_t cells = P * nch;                                // <= BIN_CELLS (host)
    const uint32_t k = p.k;
    for (uint32_t i = tid; i < BIN_CELLS; i += BIN_THREADS) cell[i] = 0;
    __syncthreads();
    // every window of the view, one segment per wave at a time: sink(valid, partition, chunk, key, ordinal, bit)
    auto walk = [&](auto&& sink) {
        if (in_pool) {
            // pieces of the unit view (one unit each, KW + 2 words at most): a wave takes 64 consecutive pieces, every lane
            // asks for ONE piece's numbers and then its words, and the pieces are handed out lane to lane -- two trips to
            // memory per 64 pieces.  (Piece by piece they were two per piece, with little else in flight: 70 % of the kernel.)
            for (uint32_t sb = seg0 + wave * 64; sb < seg1; sb += BIN_THREADS) {
                const uint32_t s = sb + lane;
                const bool has = s < seg1;
                const uint32_t m_len = has ? a_len[s] : 0u, m_ordb = has ? a_ordb[s] : 0u, m_bit = has ? a_bits[s] : 0u;
                const uint32_t m_ch = has ? a_sample[s] >> 5 : 0u;
                const uint64_t m_woff = has ? a_woff[s] : 0ull;
                uint32_t pw_lo[KW + 2], pw_hi[KW + 2];
#pragma unroll
                for (int j = 0; j < KW + 2; j++) {
                    const uint64_t v = has ? p.packed[m_woff + j] : 0ull;
                    pw_lo[j] = (uint32_t)v; pw_hi[j] = (uint32_t)(v >> 32);
                }
                const uint32_t cntp = min(64u, seg1 - sb);
                for (uint32_t i = 0; i < cntp; i++) {
                    const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)m_len, (int)i);
                    if (len < k) continue;
                    const uint32_t ninst = len - k + 1;
                    const uint32_t ordb = (uint32_t)__builtin_amdgcn_readlane((int)m_ordb, (int)i);
                    const uint32_t bit = (uint32_t)__builtin_amdgcn_readlane((int)m_bit, (int)i);
                    const uint32_t ch = (uint32_t)__builtin_amdgcn_readlane((int)m_ch, (int)i);
                    uint64_t uw[KW + 2];
#pragma unroll
                    for (int j = 0; j < KW + 2; j++)
                        uw[j] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)pw_hi[j], (int)i) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)pw_lo[j], (int)i);
                    uint64_t cw[KW + 1];
#pragma unroll
                    for (int j = 0; j <= KW; j++) cw[j] = (lane >> 5) ? uw[j + 1] : uw[j];
                    const bool valid = lane < ninst;
                    Key<KW> fwd, rc;
                    const bool rc_smaller = window_keys<KW>(k, lane, cw, fwd, rc);
                    if (CANON) {
                        const Key<KW> key = rc_smaller ? rc : fwd;
                        const uint32_t h = key_hash<KW>(key);
                        sink(valid, ((h & 0xFFFFu) * P) >> 16, ch, key, ordb + lane, bit);
                    } else {
                        uint32_t h = key_hash<KW>(fwd);
                        sink(valid, ((h & 0xFFFFu) * P) >> 16, ch, fwd, 2 * (ordb + lane), bit);
                        h = key_hash<KW>(rc);
                        sink(valid, ((h & 0xFFFFu) * P) >> 16, ch, rc, 2 * (ordb + lane) + 1, bit);
                    }
                }
            }
            return;
        }
        for (uint32_t s = seg0 + wave; s < seg1; s += BIN_THREADS / 64) {
            const uint32_t len = a_len[s];
            if (len < k) continue;
            const uint32_t ninst = len - k + 1, ordb = a_ordb[s], bit = a_bits[s], ch = a_sample[s] >> 5;
            const uint64_t* q = p.packed + a_woff[s] + (lane >> 5);
            uint64_t cw[KW + 1];
#pragma unroll
            for (int j = 0; j <= KW; j++) cw[j] = q[j];
            for (uint32_t u = 0; (u << 6) < ninst; u++) {
                const bool more = ((u + 1) << 6) < ninst;
                const uint64_t* qn = more ? q + 2 : q;                 // (unconditional: the next unit's words in flight)
                uint64_t nw[KW + 1];
#pragma unroll
                for (int j = 0; j <= KW; j++) nw[j] = qn[j];
                const uint32_t pos = (u << 6) + lane;
                const bool valid = pos < ninst;
                Key<KW> fwd, rc;
                const bool rc_smaller = window_keys<KW>(k, lane, cw, fwd, rc);
                if (CANON) {
                    const Key<KW> key = rc_smaller ? rc : fwd;
                    const uint32_t h = key_hash<KW>(key);
                    sink(valid, ((h & 0xFFFFu) * P) >> 16, ch, key, ordb + pos, bit);
                } else {
                    uint32_t h = key_hash<KW>(fwd);
                    sink(valid, ((h & 0xFFFFu) * P) >> 16, ch, fwd, 2 * (ordb + pos), bit);
                    h = key_hash<KW>(rc);
                    sink(valid, ((h & 0xFFFFu) * P) >> 16, ch, rc, 2 * (ordb + pos) + 1, bit);
                }
#pragma unroll
                for (int j = 0; j <= KW; j++) cw[j] = nw[j];
                q = qn;
            }
        }
    };
    walk([&](bool valid, uint32_t part, uint32_t ch, const Key<KW>&, uint32_t, uint32_t) {
        if (valid) atomicAdd(&cell[part * nch + ch], 1u);
    });
    __syncthreads();
    {   // cells -> positions (exclusive prefix in partition-major order), the items' chunk tables
        constexpr uint32_t PER = BIN_CELLS / BIN_THREADS;
        uint32_t v[PER], sum = 0;
#pragma unroll
        for (uint32_t j = 0; j < PER; j++) {
            const uint32_t i = tid * PER + j;
            v[j] = i < cells ? cell[i] : 0;
            sum += v[j];
        }
        uint32_t x = sum;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d);
            if ((int)lane >= d) x += y;
        }
        if (lane == 63) wave_tot[wave] = x;
        __syncthreads();
        if (tid == 0) {
            uint32_t run = 0;
            for (uint32_t w = 0; w < BIN_THREADS / 64; w++) { const uint32_t t = wave_tot[w]; wave_tot[w] = run; run += t; }
            wave_tot[BIN_THREADS / 64] = run;
        }
        __syncthreads();
        uint32_t run = base + wave_tot[wave] + x - sum;
#pragma unroll
        for (uint32_t j = 0; j < PER; j++) {
            const uint32_t i = tid * PER + j;
            if (i < cells) {
                const uint32_t part = i / nch, ch = i - part * nch;
                cell[i] = run;
                p.q_off[(size_t)(item0 + part) * (BIN_CHUNKS + 1) + ch] = run;
                if (ch == 0 && part) p.q_off[(size_t)(item0 + part - 1) * (BIN_CHUNKS + 1) + nch] = run;
                run += v[j];
            }
        }
        if (tid == 0) p.q_off[(size_t)(item0 + P - 1) * (BIN_CHUNKS + 1) + nch] = base + wave_tot[BIN_THREADS / 64];
    }
    __syncthreads();
    walk([&](bool valid, uint32_t part, uint32_t ch, const Key<KW>& key, uint32_t o, uint32_t bit) {
        if (!valid) return;
        const uint32_t e = atomicAdd(&cell[part * nch + ch], 1u);
#pragma unroll
        for (int j = 0; j < KW; j++) p.q_key[(size_t)j * p.q_stride + e] = key.w[j];
        p.q_ord[e] = o;
        p.q_bit[e] = bit;
    });
}

// ---------------------------------------------------------------------------------------------
// strand_bits_kernel: used_strand of panfeed.py:69-75 for the windows of target-strain segments
// (one wave per flagged segment; bit = 1 when the reverse complement is the canonical k-mer)
// ---------------------------------------------------------------------------------------------
template <int KW>
__global__ __launch_bounds__(256) void strand_bits_kernel(const uint64_t* packed, const uint64_t* seg_word_off,
                                                          const uint32_t* seg_len, const uint32_t* seg_strand_off,
                                                          uint32_t n_segs, uint32_t k, uint64_t* strand_bits) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t s = wave; s < n_segs; s += nwaves) {
        const uint32_t soff = seg_strand_off[s];
        if (soff == 0xFFFFFFFFu) continue;
        const uint32_t len = seg_len[s];
        const uint32_t ninst = len >= k ? len - k + 1 : 0;
        const uint64_t* wp = packed + seg_word_off[s];
        for (uint32_t u = 0; u < (ninst + 63) >> 6; u++) {
            const uint64_t* q = wp + 2 * (size_t)u + (lane >> 5);
            Key<KW> fwd, rc;
            uint64_t cw[KW + 1];
#pragma unroll
            for (int j = 0; j <= KW; j++) cw[j] = q[j];
            const bool rc_smaller = window_keys<KW>(k, lane, cw, fwd, rc);
            const uint64_t bal = __ballot(((u << 6) + lane) < ninst && rc_smaller);
            if (lane == 0) strand_bits[(size_t)soff + u] = bal;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// cluster_dedup_kernel: find identical segments inside a cluster (exact: hash, then word compare, one pass)
// ---------------------------------------------------------------------------------------------
// Samples that carry the same allele contribute the same k-mers at the same relative positions, so
// only one representative per distinct sequence has to be scanned: the representative is the copy
// with the lowest instance ordinal (its windows are the first occurrences, panfeed.py:77-79), the
// scan sets "distinct-sequence" bits instead of sample bits, and rows_kernel expands them through
// M[d] = set of samples that carry distinct sequence d.  Output is identical to scanning every copy.
// A cluster stays in mode 0 (scan everything) when dedup does not pay or does not fit.
constexpr uint32_t DEDUP_MAX_SEGS = 16384;    // segments of a cluster (one table-slot index of LDS each)
constexpr uint32_t DEDUP_MAX_D = 64;          // distinct sequences of a mode-1 cluster (two 32-bit presence words)
constexpr uint32_t DEDUP_MAX_D_WIDE = 1024;   // ... of a mode-2 cluster (up to 32 presence words per k-mer)
constexpr uint32_t DEDUP_MROWS = 4096;        // words of the M matrix: D * ceil4(W) <= this
constexpr uint32_t DENSE_WORDS = 8192;        // ordinal bitmap words held in LDS two at a time (262144 dense ordinals)
constexpr uint32_t DENSE_WIN = 16384;         // ... words of the ONE LDS bitmap a larger cluster's ordinal space is walked with,
                                              // stretch by stretch
constexpr uint32_t DENSE_WORDS_BIG = 32768;   // the words an item's bitmaps take in the scratch arrays: wide clusters of up to
                                              // 1 048 576 windows over their distinct sequences rank by bitmap
constexpr uint32_t MODE_RETRY_WIDE = 0x80u;   // v_mode flag: mode 0 only because the small class was too small
// How a cluster's k-mers get their rank (position in dict insertion order): from bitmaps over the cluster's dense ordinal
// space -- prefix popcounts, summed over the cluster's key partitions -- whenever that space fits (every mode-1 cluster, and
// a mode-2 cluster of up to DENSE_WORDS_BIG * 32 windows over its distinct sequences); otherwise every partition sorts its
// (ordinal, slot) pairs and a k-mer's rank is a binary search in each sibling partition.
__host__ __device__ inline bool ranks_by_bitmap(uint32_t mode, uint32_t v_dense) {
    return mode == 1 || (mode == 2 && v_dense <= DENSE_WORDS_BIG * 32);
}

// v_mode of a cluster: 0 = scan every segment (sample columns); 1 = scan one representative per distinct sequence,
// D <= 64, rows through the sample-set matrix M (LDS), ranks from ordinal bitmaps; 2 = the same view with D <= 1024
// ("wide"): rows gathered through the (distinct index, sample) list of the segments, ranks by sorting.

struct DedupParams {
    const uint64_t* packed; const uint64_t* seg_word_off; const uint32_t* seg_len;
    const uint32_t* seg_sample; const uint32_t* seg_ord_base;
    const uint32_t* cluster_seg_off; const uint32_t* cluster_nstrains;
    const uint32_t* extra_off;        // [C+1] extras per cluster (CSR)
    const uint32_t* extra_ord;        // [n_extra]
    uint64_t* v_word_off; uint32_t* v_len; uint32_t* v_sample; uint32_t* v_ord;   // view, [n_segs]
    uint32_t* seg_distinct;           // [n_segs] distinct index of every original segment (modes 1, 2)
    uint32_t* v_bits;                 // view, [n_segs]: the presence bits a window of the segment sets in its chunk's word
    uint32_t* v_nseg; uint32_t* v_nstr; uint32_t* v_mode; uint32_t* v_dense;       // [C]
    uint32_t* view_off;               // [C] first entry of the cluster's view in the view arrays
    // per-cluster outputs of the passes that follow, given their start values here (small class only: it sees every
    // cluster of a batch first) instead of by four memsets in front of the batch
    uint32_t* cl_overflow; uint32_t* cl_kmer_cnt; uint32_t* cl_unique; uint32_t* cl_pattern;
    uint32_t* extra_dense;            // [n_extra] ordinal of the extra row in the cluster's (dense) numbering
    uint32_t k, W, canon, enable;
    uint32_t cluster_base;            // first cluster of this launch (the batch's clusters may be launched in two halves)
    const uint32_t* cluster_list;     // or: the clusters of this launch (the wide class runs on the flagged ones only)
};

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
    return x;
}

// Contribution of 64-bit word `x` at word index idx of a segment to the segment's content hash.  Summed over the
// words (any order: eight lanes hold different words), then finalised once per segment with mix64.  A multiply of the
// two salted halves ("mum" folding): one v_mad_u64_u32 instead of the two 64-bit multiplies of a mix64 -- the pass is
// HBM-bound only as long as the hashing stays well inside the vector-issue budget (a 64-bit multiply is four
// quarter-rate 32-bit ones).  The hash only has to separate distinct sequences well: equal hashes are compared word for word.
__device__ __forceinline__ uint64_t dedup_word_hash(uint64_t x, uint32_t salt_lo, uint32_t salt_hi) {
    return (uint64_t)((uint32_t)x ^ salt_lo) * (uint64_t)((uint32_t)(x >> 32) ^ salt_hi);
}
__device__ __forceinline__ uint32_t dedup_salt_lo(uint32_t idx) { return 0x9E3779B1u * (2u * idx + 1u); }
__device__ __forceinline__ uint32_t dedup_salt_hi(uint32_t idx) { return 0x85EBCA77u * (2u * idx + 1u) + 0x165667B1u; }
typedef uint32_t pf_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ulonglong2 dedup_load16(const ulonglong2* p) {
    return *p;
}

constexpr uint32_t DEDUP_THREADS = 512;
constexpr uint32_t DEDUP_CH = 4;              // 16-byte chunks per lane kept in registers (CH * GL * 64 bases per segment)
constexpr uint32_t DEDUP_U = 1;       // segments in flight per 8-lane group
constexpr uint32_t DEDUP_GL = 8;              // lanes that share one segment
constexpr uint32_t DEDUP_UNSET = 0xFFFFFFFFu;
constexpr uint32_t DEDUP_EX_LDS = 1024;       // slow-path rows of a cluster whose ordinals are staged in LDS
constexpr uint32_t DEDUP_INGLOBAL = 0x80000000u;

// Two size classes.  Small: every cluster goes through it first -- ~46 KiB of LDS and 80 VGPRs, three workgroups per
// CU, which is what the HBM-bound pass over the packed bytes needs.  Wide: only the clusters the small class flags
// (more than 64 distinct sequences, or a sample-set matrix / ordinal bitmap that does not fit): one workgroup per CU.
struct DedupSmall { static constexpr uint32_t GTAB = 256, MAXD = DEDUP_MAX_D, POOL = 2048, MODE = 1; typedef uint8_t slot_t; };
struct DedupWide { static constexpr uint32_t GTAB = 2048, MAXD = DEDUP_MAX_D_WIDE, POOL = 1024, MODE = 2; typedef uint16_t slot_t; };

template <class CFG>
__global__ __launch_bounds__(DEDUP_THREADS) __attribute__((amdgpu_waves_per_eu(CFG::MODE == 1 ? 6 : 2, CFG::MODE == 1 ? 6 : 4)))
void cluster_dedup_kernel(DedupParams p) {
    constexpr uint32_t GTAB = CFG::GTAB, MAXD = CFG::MAXD, POOL = CFG::POOL;
    typedef typename CFG::slot_t slot_t;
    __shared__ uint64_t t_key[GTAB];           // content hash of the group
    __shared__ uint64_t t_val[GTAB];           // min over the group's members of (ord_base << 32 | local index)
    __shared__ uint32_t t_pool[GTAB];          // where the group's sequence sits: pool word offset, or INGLOBAL | index
    __shared__ uint32_t t_len[GTAB];           // its length in bases
    __shared__ __align__(16) uint64_t s_pool[POOL];   // u64 words of LDS holding one copy of distinct sequences (while they fit)
    __shared__ uint32_t t_woff[GTAB];          // word offset (relative to the cluster's first segment) of its first member
    __shared__ uint32_t t_rank[GTAB];          // distinct index of the group (representatives in ordinal order)
    __shared__ slot_t s_slot[DEDUP_MAX_SEGS];  // hash group (table slot) of every segment
    __shared__ uint32_t r_list[MAXD];          // occupied table slots, unordered
    __shared__ uint32_t r_rep[MAXD];           // by distinct index: local index of the representative
    __shared__ uint32_t r_ord0[MAXD], r_ninst[MAXD], r_dense[MAXD];   // by distinct index
    __shared__ uint32_t sh_bad, sh_many, sh_nrep, sh_total, sh_ngroups, sh_pool_used;
    __shared__ uint32_t s_ex[DEDUP_EX_LDS];    // the cluster's slow-path ordinals (read E times per row further down)

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t c = p.cluster_list ? p.cluster_list[blockIdx.x] : blockIdx.x + p.cluster_base;
    const uint32_t seg0 = p.cluster_seg_off[c], seg1 = p.cluster_seg_off[c + 1];
    const uint32_t n = seg1 - seg0;
    const uint32_t ex0 = p.extra_off[c], ex1 = p.extra_off[c + 1];
    const uint32_t k = p.k;
    const uint32_t mult = p.canon ? 1u : 2u;
    const uint32_t Wp = (p.W + 3) & ~3u;
    if (CFG::MODE == 1 && threadIdx.x == 0) {
        p.cl_overflow[c] = 0; p.cl_kmer_cnt[c] = 0; p.cl_unique[c] = 0; p.cl_pattern[c] = 0xFFFFFFFFu;
    }

    bool mode1 = p.enable && n >= 4 && n <= DEDUP_MAX_SEGS;
    bool retry_wide = false;                 // small class only: the wide class may still deduplicate this cluster
    const bool ex_lds = ex1 - ex0 <= DEDUP_EX_LDS;
    if (mode1 && ex_lds)
        for (uint32_t e = tid; e < ex1 - ex0; e += DEDUP_THREADS) s_ex[e] = p.extra_ord[ex0 + e];
    auto ex_ord = [&](uint32_t e) { return ex_lds ? s_ex[e - ex0] : p.extra_ord[e]; };
    if (tid == 0) { sh_bad = 0; sh_many = 0; sh_nrep = 0; sh_total = 0; sh_ngroups = 0; sh_pool_used = 0; }
    if (mode1) {
        for (uint32_t i = tid; i < GTAB; i += DEDUP_THREADS) { t_key[i] = EMPTY64; t_val[i] = EMPTY64; t_pool[i] = DEDUP_UNSET; }
    }
    __syncthreads();
    if (mode1) {
        const uint64_t wspan = p.seg_word_off[seg1 - 1] - p.seg_word_off[seg0];
        if (wspan >= 0xFFFFFFFFull) mode1 = false;      // uniform; a cluster this large is not worth it anyway
    }
    if (mode1) {
        // ---- 1. one pass over the packed bytes: DEDUP_GL lanes per segment, DEDUP_U segments in flight per group.  A segment is
        // read once into registers, hashed (64 bits), looked up in the group table; the first segment of a group
        // leaves its words in the LDS pool, every later one is compared with the pool word for word (exact).
        // Per wave the steps run in lockstep -- claim, publish, then wait -- so a waiting group never sits in front
        // of the group it waits for.
        const uint64_t wbase = p.seg_word_off[seg0];
        const ulonglong2* cbase = reinterpret_cast<const ulonglong2*>(p.packed + wbase);
        // Lanes per segment: as many as the cluster's segments need at DEDUP_CH pieces per lane (the sequences of a cluster
        // are alleles of one gene: about the same length) -- 5 for 1 200 bases.  With a fixed 8 (rounds 1-2) such a segment
        // filled 19 of its group's 32 places: the pass had 38 bytes in flight per lane where it could have 61.
        const uint32_t GL = min(DEDUP_GL, max(2u, (((p.seg_len[seg0] + 63) >> 6) + DEDUP_CH - 1) / DEDUP_CH));
        const uint32_t gpw = 64 / GL;                              // groups per wave (lanes past gpw * GL stay idle)
        const bool lane_used = lane < gpw * GL;
        const uint32_t gl = lane % GL, gbase = lane - gl;
        const uint32_t grp = (tid >> 6) * gpw + min(lane / GL, gpw - 1), ngrp = (DEDUP_THREADS / 64) * gpw;
        const uint32_t wave_grp0 = (tid >> 6) * gpw;
        // segment metadata is read from global memory one trip ahead (no LDS copy: clusters of thousands of samples
        // have thousands of segments)
        uint32_t nx_len[DEDUP_U], nx_woff[DEDUP_U], nx_ord[DEDUP_U];
#pragma unroll
        for (int u = 0; u < (int)DEDUP_U; u++) {
            const uint32_t sn = grp + u * ngrp;
            nx_len[u] = sn < n ? p.seg_len[seg0 + sn] : 0;
            nx_woff[u] = sn < n ? (uint32_t)(p.seg_word_off[seg0 + sn] - wbase) : 0;
            nx_ord[u] = sn < n ? p.seg_ord_base[seg0 + sn] : 0;
        }
        for (uint32_t sw = wave_grp0; sw < n; sw += DEDUP_U * ngrp) {
            // more distinct sequences than this class holds: the rest of the pass would be wasted
            if (__hip_atomic_load(&sh_many, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
            const uint32_t s = sw + (grp - wave_grp0);
            uint32_t len[DEDUP_U], pc[DEDUP_U], si[DEDUP_U], slot[DEDUP_U];
            bool has[DEDUP_U], registrar[DEDUP_U];
            const ulonglong2* w[DEDUP_U];
            ulonglong2 v[DEDUP_U][DEDUP_CH];
            uint64_t acc[DEDUP_U] = {};
            uint32_t my_woff[DEDUP_U], my_ord[DEDUP_U];
#pragma unroll
            for (int u = 0; u < (int)DEDUP_U; u++) {
                si[u] = s + u * ngrp;
                has[u] = lane_used && si[u] < n;
                len[u] = nx_len[u];
                pc[u] = lane_used ? (len[u] + 63) >> 6 : 0;
                w[u] = cbase + (nx_woff[u] >> 1);
            }
#pragma unroll
            for (int u = 0; u < (int)DEDUP_U; u++) { my_woff[u] = nx_woff[u]; my_ord[u] = nx_ord[u]; }
#pragma unroll
            for (int u = 0; u < (int)DEDUP_U; u++)
#pragma unroll
                for (uint32_t q = 0; q < DEDUP_CH; q++) {
                    const uint32_t j = gl + GL * q;
                    v[u][q] = j < pc[u] ? dedup_load16(&w[u][j]) : make_ulonglong2(0, 0);
                }
#pragma unroll
            for (int u = 0; u < (int)DEDUP_U; u++) {          // the next trip's metadata, behind this trip's data
                const uint32_t sn = si[u] + DEDUP_U * ngrp;
                nx_len[u] = sn < n ? p.seg_len[seg0 + sn] : 0;
                nx_woff[u] = sn < n ? (uint32_t)(p.seg_word_off[seg0 + sn] - wbase) : 0;
                nx_ord[u] = sn < n ? p.seg_ord_base[seg0 + sn] : 0;     // (read at claim time it was a global round trip per trip)
            }
#pragma unroll
            for (int u = 0; u < (int)DEDUP_U; u++) {
#pragma unroll
                for (uint32_t q = 0; q < DEDUP_CH; q++) {
                    const uint32_t j = gl + GL * q;
                    if (j < pc[u])
                        acc[u] += dedup_word_hash(v[u][q].x, dedup_salt_lo(2 * j), dedup_salt_hi(2 * j)) +
                                  dedup_word_hash(v[u][q].y, dedup_salt_lo(2 * j + 1), dedup_salt_hi(2 * j + 1));
                }
                for (uint32_t j = gl + GL * DEDUP_CH; j < pc[u]; j += GL) {      // longer than the registers hold
                    const ulonglong2 x = w[u][j];
                    acc[u] += dedup_word_hash(x.x, dedup_salt_lo(2 * j), dedup_salt_hi(2 * j)) +
                              dedup_word_hash(x.y, dedup_salt_lo(2 * j + 1), dedup_salt_hi(2 * j + 1));
                }
            }
#if defined(PF_KO_DEDUP) && PF_KO_DEDUP >= 2
            if (acc[0] == 0x123456789ull) sh_bad = 1;            // (timing experiment: loads + hash only)
            continue;
#endif
            // claim: lane 0 of the group finds or opens the hash group
#pragma unroll
            for (int u = 0; u < (int)DEDUP_U; u++) {
                uint64_t a = 0;                                  // the group's sum (GL is not a power of two in general)
                for (uint32_t i = 0; i < GL; i++) a += __shfl(acc[u], (int)(gbase + i));
                uint64_t h = mix64(a ^ ((uint64_t)len[u] << 40));
                if (h == EMPTY64) h = EMPTY64 - 1;
                uint32_t sl = 0, reg = 0;
                if (gl == 0 && has[u]) {
                    sl = (uint32_t)h & (GTAB - 1);
                    for (uint32_t probes = 0;; probes++) {
                        uint64_t cur = __hip_atomic_load(&t_key[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (cur == EMPTY64) {
                            if (__hip_atomic_load(&sh_ngroups, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= MAXD) {
                                sh_many = 1; sh_bad = 1; sl = GTAB; break;       // too many distinct sequences
                            }
                            cur = atomicCAS((unsigned long long*)&t_key[sl], (unsigned long long)EMPTY64, (unsigned long long)h);
                            if (cur == EMPTY64) {
                                reg = 1;
                                if (atomicAdd(&sh_ngroups, 1u) >= MAXD) { sh_many = 1; sh_bad = 1; }
                                break;
                            }
                        }
                        if (cur == h) break;
                        sl = (sl + 1) & (GTAB - 1);
                        if (probes >= GTAB) { sh_many = 1; sh_bad = 1; sl = GTAB; break; }
                    }
                    if (sl < GTAB) {
                        atomicMin((unsigned long long*)&t_val[sl],
                                  (unsigned long long)(((uint64_t)my_ord[u] << 32) | si[u]));
                        s_slot[si[u]] = (slot_t)sl;
                    }
                }
                slot[u] = __shfl(sl, (int)gbase);
                registrar[u] = __shfl(reg, (int)gbase) != 0;
                if (slot[u] >= GTAB) has[u] = false;      // cluster given up by this class
            }
            // publish: the first segment of a group leaves its words in the pool
#pragma unroll
            for (int u = 0; u < (int)DEDUP_U; u++) {
                if (has[u] && registrar[u]) {
                    uint32_t off = 0;
                    if (gl == 0) off = atomicAdd(&sh_pool_used, 2 * pc[u]);
                    off = __shfl(off, (int)gbase);
                    const bool fits = off + 2 * pc[u] <= POOL;
                    if (fits) {
#pragma unroll
                        for (uint32_t q = 0; q < DEDUP_CH; q++) {
                            const uint32_t j = gl + GL * q;
                            if (j < pc[u]) *reinterpret_cast<ulonglong2*>(&s_pool[off + 2 * j]) = v[u][q];     // off is even: 16-byte stores
                        }
                        for (uint32_t j = gl + GL * DEDUP_CH; j < pc[u]; j += GL) {
                            const ulonglong2 x = w[u][j];
                            s_pool[off + 2 * j] = x.x; s_pool[off + 2 * j + 1] = x.y;
                        }
                    }
                    if (gl == 0) { t_len[slot[u]] = len[u]; t_woff[slot[u]] = my_woff[u]; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    // all the group's stores are ahead of the offset: a wave's LDS stores complete in program order
                    if (gl == 0)
                        __hip_atomic_store(&t_pool[slot[u]], fits ? off : DEDUP_INGLOBAL, __ATOMIC_RELEASE,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
#if defined(PF_KO_DEDUP) && PF_KO_DEDUP >= 1
            continue;                                            // (timing experiment: no exact compare)
#endif
            // compare: every other segment of the group against the pool (or against the first one's global words)
#pragma unroll
            for (int u = 0; u < (int)DEDUP_U; u++) {
                if (has[u] && !registrar[u]) {
                    uint32_t pp;
                    while ((pp = __hip_atomic_load(&t_pool[slot[u]], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == DEDUP_UNSET)
                        __builtin_amdgcn_s_sleep(1);
                    bool diff = t_len[slot[u]] != len[u];
                    if (diff) {
                    } else if (pp & DEDUP_INGLOBAL) {
                        const ulonglong2* b = cbase + (t_woff[slot[u]] >> 1);
                        {
#pragma unroll
                            for (uint32_t q = 0; q < DEDUP_CH; q++) {
                                const uint32_t j = gl + GL * q;
                                if (j < pc[u]) { const ulonglong2 y = b[j]; diff = diff || y.x != v[u][q].x || y.y != v[u][q].y; }
                            }
                            for (uint32_t j = gl + GL * DEDUP_CH; j < pc[u]; j += GL) {
                                const ulonglong2 x = w[u][j], y = b[j];
                                diff = diff || x.x != y.x || x.y != y.y;
                            }
                        }
                    } else {
#pragma unroll
                        for (uint32_t q = 0; q < DEDUP_CH; q++) {
                            const uint32_t j = gl + GL * q;
                            if (j < pc[u]) {
                                const ulonglong2 y = *reinterpret_cast<const ulonglong2*>(&s_pool[pp + 2 * j]);   // one 16-byte read
                                diff = diff || y.x != v[u][q].x || y.y != v[u][q].y;
                            }
                        }
                        for (uint32_t j = gl + GL * DEDUP_CH; j < pc[u]; j += GL) {
                            const ulonglong2 x = w[u][j];
                            diff = diff || s_pool[pp + 2 * j] != x.x || s_pool[pp + 2 * j + 1] != x.y;
                        }
                    }
                    if (diff) sh_bad = 1;     // a 64-bit hash collision: give up on this cluster (mode 0), stay exact
                }
            }
        }
        __syncthreads();
        // ---- 4. hash groups -> distinct indices in the ordinal order of their representatives
        for (uint32_t t = tid; t < GTAB; t += DEDUP_THREADS)
            if (t_key[t] != EMPTY64) {
                const uint32_t at = atomicAdd(&sh_nrep, 1u);
                if (at < MAXD) r_list[at] = t;
            }
        __syncthreads();
        const uint32_t D = sh_nrep;
        // The view of distinct sequences is taken whenever it fits.  (Rounds 1-2 asked for at least half of the segments
        // to be copies: with fewer, mode 0's direct sample columns were as fast.  With the unit view the scan of the
        // distinct sequences is several times cheaper than the scan of every copy, and the rule only kept a third of the
        // accessory clusters of a many-allele pangenome on the every-copy path: profiles/r03/worth_rule_experiment.txt.)
        const bool worth = true;
        if (CFG::MODE == 1) {
            // what the wide class can still do for this cluster: more distinct sequences, no sample-set matrix, no
            // ordinal bitmap; (a hash collision or too few copies stay mode 0)
            retry_wide = sh_many != 0 || (!sh_bad && worth && D * Wp > DEDUP_MROWS);
            // (the view's columns are swept in chunks of 32 whose words live in chunkbits[slice][W][NS]: more chunks than
            // the W words of a sample row -- more distinct sequences than samples, through paralogs and sequences cut at
            // an 'N' -- would run into the next slice; found by the randomised test once such clusters reached mode 1)
            mode1 = !sh_bad && D <= MAXD && D * Wp <= DEDUP_MROWS && worth && ((D + 31) >> 5) <= p.W;
        } else {
            mode1 = !sh_bad && D <= MAXD && worth && ((D + 31) >> 5) <= p.W;
        }
        if (mode1) {
            for (uint32_t d = tid; d < D; d += DEDUP_THREADS) {
                const uint32_t slot = r_list[d];
                const uint64_t val = t_val[slot];            // (ord_base << 32 | local index) of the representative
                uint32_t rank = 0;
                for (uint32_t j = 0; j < D; j++) rank += t_val[r_list[j]] < val ? 1u : 0u;
                t_rank[slot] = rank;
                r_rep[rank] = (uint32_t)val;
                const uint32_t len = t_len[slot];
                r_ord0[rank] = (uint32_t)(val >> 32);
                r_ninst[rank] = len >= k ? len - k + 1 : 0;
            }
            __syncthreads();
            for (uint32_t d = tid; d < D; d += DEDUP_THREADS) {
                // dense ordinal of this representative's first window: scanned instances before it plus the
                // slow-path rows before it (those never fall inside a segment's ordinal range)
                uint32_t base = 0;
                for (uint32_t j = 0; j < d; j++) base += r_ninst[j];
                uint32_t xb = 0;
                const uint32_t o0 = r_ord0[d] * mult;
                for (uint32_t e = ex0; e < ex1; e++) xb += ex_ord(e) < o0 ? 1u : 0u;
                r_dense[d] = base + xb;
                if (d == D - 1) sh_total = base + r_ninst[d];
            }
            __syncthreads();
            const uint64_t dense_bits = ((uint64_t)sh_total + (ex1 - ex0)) * mult;
            if (CFG::MODE == 1) {
                mode1 = dense_bits <= (uint64_t)DENSE_WORDS * 32;
                if (!mode1) retry_wide = true;             // ranks by sorting need no bitmap
            } else {
                mode1 = dense_bits < 0xFFFFFFF0ull;
            }
            if (mode1) {
                for (uint32_t s = tid; s < n; s += DEDUP_THREADS) p.seg_distinct[seg0 + s] = t_rank[s_slot[s]];
                for (uint32_t d = tid; d < D; d += DEDUP_THREADS) {
                    const uint32_t s = r_rep[d];
                    p.v_word_off[seg0 + d] = p.seg_word_off[seg0 + s];
                    p.v_len[seg0 + d] = p.seg_len[seg0 + s];
                    p.v_sample[seg0 + d] = d;
                    p.v_bits[seg0 + d] = 1u << (d & 31);
                    p.v_ord[seg0 + d] = r_dense[d];
                }
                for (uint32_t e = ex0 + tid; e < ex1; e += DEDUP_THREADS) {
                    // F(o) = scanned instances below o + slow-path rows below o  (monotone in the reference order)
                    const uint32_t o = ex_ord(e);
                    const uint32_t oi = o / mult;
                    uint32_t below = 0;
                    for (uint32_t d = 0; d < D; d++) {
                        const uint32_t a = r_ord0[d];
                        below += oi <= a ? 0u : min(oi - a, r_ninst[d]);
                    }
                    uint32_t er = 0;
                    for (uint32_t f = ex0; f < ex1; f++) er += ex_ord(f) < o ? 1u : 0u;
                    p.extra_dense[e] = (below + er) * mult;
                }
                if (tid == 0) {
                    p.view_off[c] = seg0;
                    p.v_nseg[c] = D; p.v_nstr[c] = D; p.v_mode[c] = CFG::MODE;
                    p.v_dense[c] = (uint32_t)min(dense_bits, (uint64_t)0xFFFFFFFFu);
                }
            }
        }
    }
    if (!mode1) {
        // mode 0: the view is the caller's segment list
        for (uint32_t s = tid; s < n; s += DEDUP_THREADS) {
            p.v_word_off[seg0 + s] = p.seg_word_off[seg0 + s];
            p.v_len[seg0 + s] = p.seg_len[seg0 + s];
            const uint32_t smp = p.seg_sample[seg0 + s];
            p.v_sample[seg0 + s] = smp;
            p.v_bits[seg0 + s] = 1u << (smp & 31);
            p.v_ord[seg0 + s] = p.seg_ord_base[seg0 + s];
        }
        for (uint32_t e = ex0 + tid; e < ex1; e += DEDUP_THREADS) p.extra_dense[e] = p.extra_ord[e];
        if (tid == 0) {
            p.view_off[c] = seg0;
            p.v_nseg[c] = n; p.v_nstr[c] = p.cluster_nstrains[c]; p.v_dense[c] = 0;
            p.v_mode[c] = retry_wide ? MODE_RETRY_WIDE : 0;
        }
    }
    (void)lane;
}

// ---------------------------------------------------------------------------------------------
// unit_class_kernel: identical 64-window units among the distinct sequences of a cluster
// ---------------------------------------------------------------------------------------------
// The distinct sequences of a cluster are alleles of one gene: they differ in a few positions, so most of their
// 64-window units -- the 64 + k - 1 bases that the windows starting at bases 64u .. 64u + 63 cover -- are the same
// bases at the same place in many of them.  Units of different distinct sequences with the same u and the same
// content form a class: its windows are the same k-mers at the same offsets, so ONE scan of the class's first member
// (lowest distinct index = lowest ordinals: the view is in ordinal order, so its windows are the first occurrences,
// panfeed.py:77-79) with ALL the members' column bits gives what scanning every member would.  Exact: members are
// compared with the class's first member word for word; two different contents with one hash send the cluster back to
// the plain view.  The scan view of such a cluster becomes a list of one-unit pieces {words, bases, ordinal of the
// first window, chunk of 32 columns, the members' bits in that chunk}, sorted by chunk -- a class whose members spread
// over several chunks is listed once per chunk.
struct UnitParams {
    const uint64_t* packed;
    const uint32_t* cluster_seg_off;       // the plain view of cluster c sits at [seg0, seg0 + D) of the view arrays
    const uint32_t* v_nstr;                // [C] D
    const uint32_t* list_cluster;          // [n] clusters of this launch
    const uint32_t* list_base;             // [n] where the cluster's pieces go in the view arrays (room: its units)
    const uint64_t* v_word_off; const uint32_t* v_len; const uint32_t* v_ord;     // the plain view
    uint64_t* u_word_off; uint32_t* u_len; uint32_t* u_sample; uint32_t* u_ord; uint32_t* u_bits;   // the pool
    uint32_t* v_nseg; uint32_t* view_off;  // [C] rewritten when the cluster takes the unit view
    uint32_t k;
    uint32_t tmp_off;                      // unit_class_kernel: a second stretch of the pool arrays, this far behind the first,
                                           // takes a cluster's pieces in the order they are found
};
constexpr uint32_t UNIT_THREADS = 256;
constexpr uint32_t UNIT_TAB = 4096;            // class-table slots per batch of unit positions
constexpr uint32_t UNIT_PAIRS = 2048;          // (distinct sequence, unit) pairs per batch: half the table
constexpr uint32_t UNIT_PER_THREAD = UNIT_PAIRS / UNIT_THREADS;
constexpr uint32_t UNIT_MAX_WORDS = 6;         // ceil((63 + PF_MAX_K) / 32)

__global__ __launch_bounds__(UNIT_THREADS) void unit_class_kernel(UnitParams p) {
    __shared__ uint64_t t_hash[UNIT_TAB];
    __shared__ uint32_t t_min[UNIT_TAB];                 // lowest pair index of the class = its first member
    __shared__ uint32_t d_len[DEDUP_MAX_D_WIDE], d_ord[DEDUP_MAX_D_WIDE];
    __shared__ uint32_t d_wlo[DEDUP_MAX_D_WIDE], d_whi[DEDUP_MAX_D_WIDE];
    __shared__ uint32_t ch_cnt[DEDUP_MAX_D_WIDE / 32], ch_base[DEDUP_MAX_D_WIDE / 32 + 1];
    __shared__ uint32_t sh_bad;

    const uint32_t tid = threadIdx.x, lane = tid & 63, half = (lane >> 5) << 5;
    const uint32_t c = p.list_cluster[blockIdx.x], base = p.list_base[blockIdx.x];
    const uint32_t seg0 = p.cluster_seg_off[c], D = p.v_nstr[c], k = p.k;
    if (D < 2 || D > DEDUP_MAX_D_WIDE) return;
    const uint32_t Dp = (D + 31) & ~31u, nchunks = Dp >> 5;
    uint32_t maxlen = 0;
    for (uint32_t d = tid; d < D; d += UNIT_THREADS) {
        const uint64_t wo = p.v_word_off[seg0 + d];
        const uint32_t len = p.v_len[seg0 + d];
        d_len[d] = len; d_ord[d] = p.v_ord[seg0 + d]; d_wlo[d] = (uint32_t)wo; d_whi[d] = (uint32_t)(wo >> 32);
        maxlen = max(maxlen, len);
    }
    for (uint32_t i = tid; i < nchunks; i += UNIT_THREADS) ch_cnt[i] = 0;
    if (tid == 0) sh_bad = 0;
    for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o));
    __shared__ uint32_t sh_max[UNIT_THREADS / 64];
    if (lane == 0) sh_max[tid >> 6] = maxlen;
    __syncthreads();
    maxlen = max(max(sh_max[0], sh_max[1]), max(sh_max[2], sh_max[3]));
    if (maxlen < k) return;                                          // no window anywhere: the plain view is empty work
    const uint32_t nunits = (maxlen - k + 64) >> 6;                  // unit positions of the longest sequence
    const uint32_t UB = max(1u, UNIT_PAIRS / Dp);                    // unit positions per batch (Dp <= 1024 <= UNIT_PAIRS)
    const uint32_t span = 63 + k;                                    // bases a full unit covers

    // ONE walk: the pieces go to the cluster's second stretch of the pool as they are found, counted per chunk; they are put
    // in chunk order afterwards (a few thousand 24-byte records: the walk -- hashing every unit of every distinct sequence,
    // the class table, the word-for-word checks -- is what the kernel's time is, and it used to be done twice).
    __shared__ uint32_t sh_npieces;
    if (tid == 0) sh_npieces = 0;
    const uint32_t tb = base + p.tmp_off;
    {
        for (uint32_t u0 = 0; u0 < nunits; u0 += UB) {
            const uint32_t ub_n = min(UB, nunits - u0), npairs = ub_n * Dp;
            for (uint32_t i = tid; i < UNIT_TAB; i += UNIT_THREADS) { t_hash[i] = EMPTY64; t_min[i] = 0xFFFFFFFFu; }
            __syncthreads();
            // ---- every pair finds or opens its class (hash of the unit's bases, its length and its position)
            uint32_t my_slot[UNIT_PER_THREAD];
#pragma unroll
            for (uint32_t r = 0; r < UNIT_PER_THREAD; r++) {
                my_slot[r] = 0xFFFFFFFFu;
                const uint32_t pi = tid + r * UNIT_THREADS;
                if (pi >= npairs) continue;
                const uint32_t ub = pi / Dp, d = pi - ub * Dp, u = u0 + ub;
                if (d >= D) continue;
                const uint32_t len = d_len[d];
                if (len < k || 64 * u + k > len) continue;            // the sequence has no window in this unit
                const uint32_t nb = min(len - 64 * u, span), nw = (nb + 31) >> 5;
                const uint64_t* w = p.packed + (((uint64_t)d_whi[d] << 32) | d_wlo[d]) + 2 * (size_t)u;
                uint64_t h = 0x9E3779B97F4A7C15ull * (nb + 1) + u;
                for (uint32_t j = 0; j < nw; j++) {
                    uint64_t x = w[j];
                    if (j + 1 == nw && (nb & 31)) x &= ~0ull << (64 - 2 * (nb & 31));      // bases past the unit's span
                    h = mix64(h ^ x) + 0xC2B2AE3D27D4EB4Full * (j + 1);
                }
                if (h == EMPTY64) h = EMPTY64 - 1;
                uint32_t sl = (uint32_t)(h >> 20) & (UNIT_TAB - 1);
                for (;;) {
                    uint64_t cur = __hip_atomic_load(&t_hash[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (cur == EMPTY64)
                        cur = atomicCAS((unsigned long long*)&t_hash[sl], (unsigned long long)EMPTY64, (unsigned long long)h);
                    if (cur == EMPTY64 || cur == h) break;
                    sl = (sl + 1) & (UNIT_TAB - 1);                   // at most UNIT_PAIRS entries: an empty slot exists
                }
                atomicMin(&t_min[sl], pi);
                my_slot[r] = sl;
            }
            __syncthreads();
            // ---- exactness: every member against the class's first member, word for word
#pragma unroll
            for (uint32_t r = 0; r < UNIT_PER_THREAD; r++) {
                if (my_slot[r] == 0xFFFFFFFFu) continue;
                const uint32_t pi = tid + r * UNIT_THREADS, fi = t_min[my_slot[r]];
                if (fi == pi) continue;
                const uint32_t ub = pi / Dp, d = pi - ub * Dp, u = u0 + ub, fd = fi - (fi / Dp) * Dp;
                bool diff = fi / Dp != ub;
                const uint32_t nb = min(d_len[d] - 64 * u, span), nw = (nb + 31) >> 5;
                if (!diff) diff = min(d_len[fd] - 64 * u, span) != nb;
                if (!diff) {
                    const uint64_t* a = p.packed + (((uint64_t)d_whi[d] << 32) | d_wlo[d]) + 2 * (size_t)u;
                    const uint64_t* b = p.packed + (((uint64_t)d_whi[fd] << 32) | d_wlo[fd]) + 2 * (size_t)u;
                    for (uint32_t j = 0; j < nw; j++) {
                        uint64_t x = a[j] ^ b[j];
                        if (j + 1 == nw && (nb & 31)) x &= ~0ull << (64 - 2 * (nb & 31));
                        diff = diff || x != 0;
                    }
                }
                if (diff) sh_bad = 1;
            }
            // ---- the members of a class inside one chunk of 32 columns -> one piece.  Pair index = ub * Dp + d with Dp a
            // multiple of 32: the 32 lanes of a half-wave hold the 32 columns of one (unit, chunk).
#pragma unroll
            for (uint32_t r = 0; r < UNIT_PER_THREAD; r++) {
                const uint32_t pi0 = r * UNIT_THREADS + (tid & ~63u);            // first pair of this wave in round r
                if (pi0 >= npairs) continue;                                      // wave-uniform
                const uint32_t pi = tid + r * UNIT_THREADS;
                const uint32_t ub = pi / Dp, d = pi - ub * Dp, u = u0 + ub, chunk = d >> 5;
                bool todo = my_slot[r] != 0xFFFFFFFFu;
                while (__any(todo)) {
                    const uint64_t bal = __ballot(todo);
                    const uint32_t mine = (uint32_t)(bal >> half);               // this half-wave's lanes
                    uint32_t leader = mine ? (uint32_t)__builtin_ctz(mine) + half : lane;
                    const uint32_t lslot = __shfl(my_slot[r], leader);
                    const bool same = todo && my_slot[r] == lslot;
                    const uint32_t members = (uint32_t)(__ballot(same) >> half);
                    if (todo && lane == leader) {
                        const uint32_t fi = t_min[my_slot[r]], fd = fi - (fi / Dp) * Dp;
                        atomicAdd(&ch_cnt[chunk], 1u);
                        const uint32_t at = tb + atomicAdd(&sh_npieces, 1u);
                        p.u_word_off[at] = (((uint64_t)d_whi[fd] << 32) | d_wlo[fd]) + 2 * (uint64_t)u;
                        p.u_len[at] = min(d_len[fd] - 64 * u, span);
                        p.u_ord[at] = d_ord[fd] + 64 * u;
                        p.u_sample[at] = chunk << 5;
                        p.u_bits[at] = members;
                    }
                    if (same) todo = false;
                }
            }
            __syncthreads();
        }
        if (sh_bad) return;                                          // (uniform) the cluster keeps its plain view
        if (tid == 0) {
            uint32_t run = 0;
            for (uint32_t i = 0; i < nchunks; i++) { ch_base[i] = run; run += ch_cnt[i]; ch_cnt[i] = 0; }
            ch_base[nchunks] = run;
        }
        __syncthreads();                                             // (the pieces written above: visible to the workgroup)
        for (uint32_t i = tid, n = sh_npieces; i < n; i += UNIT_THREADS) {
            const uint32_t smp = p.u_sample[tb + i], chunk = smp >> 5;
            const uint32_t at = base + ch_base[chunk] + atomicAdd(&ch_cnt[chunk], 1u);
            p.u_word_off[at] = p.u_word_off[tb + i]; p.u_len[at] = p.u_len[tb + i]; p.u_ord[at] = p.u_ord[tb + i];
            p.u_sample[at] = smp; p.u_bits[at] = p.u_bits[tb + i];
        }
        __syncthreads();
    }
    if (tid == 0) { p.view_off[c] = base | VIEW_IN_POOL; p.v_nseg[c] = ch_base[nchunks]; }
}

// The same for a cluster of at most 64 distinct sequences (every mode-1 cluster: the usual case), without a table: one
// WAVE per cluster, lane d holds distinct sequence d, and for every unit position the lanes are grouped by comparing
// their unit's words with the lowest remaining lane's (v_readlane: the leader is wave-uniform) -- exact by
// construction, no hash, no LDS, no barrier.  A class gives one piece per chunk of 32 columns it has members in; chunk-1
// pieces go behind all chunk-0 pieces, so a cluster of more than 32 distinct sequences is walked twice (the first walk
// only counts).
constexpr uint32_t UNIT_SMALL_MAX_D = 64;
template <int NW>     // 64-bit words a unit's 63 + k bases take: ceil((63 + k) / 32)
__global__ __launch_bounds__(256) void unit_class_small_kernel(UnitParams p, uint32_t n) {
    // One wave per cluster.  The wave's lanes are G = 4 .. 64 columns (the next power of two >= D) x 64 / G unit positions:
    // with lane = distinct sequence alone, a cluster of 8 alleles left 56 lanes idle and the kernel was bound by
    // instruction issue (9 000 wave-instructions per cluster, 0.7 ms per 50 000 clusters).  Classes inside a group of G
    // lanes: the group's lowest remaining lane leads, everyone compares its unit with the leader's (ds_bpermute), the
    // members leave; a class gives one piece per chunk of 32 columns it has members in.  Chunk-1 pieces go behind all
    // chunk-0 pieces: they are parked in the cluster's second stretch of the pool and moved there at the end.
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wi = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wi >= n) return;                                             // (whole waves)
    const uint32_t c = p.list_cluster[wi], base = p.list_base[wi];
    const uint32_t seg0 = p.cluster_seg_off[c], D = p.v_nstr[c], k = p.k;
    if (D < 2 || D > UNIT_SMALL_MAX_D) return;
    const uint32_t G = D <= 4 ? 4u : D <= 8 ? 8u : D <= 16 ? 16u : D <= 32 ? 32u : 64u, UPR = 64 / G;
    const uint32_t d = lane & (G - 1), gbase = lane & ~(G - 1), ug = lane / G;
    const uint64_t gmask = G == 64 ? ~0ull : (1ull << G) - 1;
    uint32_t len = 0, ordb = 0;
    uint64_t wo = 0;
    if (d < D) { len = p.v_len[seg0 + d]; ordb = p.v_ord[seg0 + d]; wo = p.v_word_off[seg0 + d]; }
    uint32_t maxlen = len;
    for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o));
    if (maxlen < k) return;
    const uint32_t nunits = (maxlen - k + 64) >> 6, span = 63 + k;
    const uint64_t* w = p.packed + wo;
    const uint64_t below = lane ? ~0ull >> (64 - lane) : 0ull;       // the lanes in front of this one
    const uint32_t tb = base + p.tmp_off;
    {
        uint32_t i0 = 0, i1 = 0;
        for (uint32_t u0 = 0; u0 < nunits; u0 += UPR) {
            const uint32_t u = u0 + ug;
            const bool valid = u < nunits && len >= k && 64 * u + k <= len;
            const uint32_t nb = valid ? min(len - 64 * u, span) : 0u, nw = (nb + 31) >> 5;
            uint64_t x[NW];
#pragma unroll
            for (int j = 0; j < NW; j++) x[j] = (uint32_t)j < nw ? w[2 * (size_t)u + j] : 0ull;
#pragma unroll
            for (int j = 0; j < NW; j++)
                if ((uint32_t)j + 1 == nw && (nb & 31)) x[j] &= ~0ull << (64 - 2 * (nb & 31));      // bases past the unit's span
            bool todo = valid;
            while (__any(todo)) {
                const uint64_t grp = (__ballot(todo) >> gbase) & gmask;            // this group's remaining lanes
                const uint32_t leader = gbase + (grp ? (uint32_t)__builtin_ctzll(grp) : 0u);
                bool same = todo && (uint32_t)__shfl((int)nb, (int)leader) == nb;
#pragma unroll
                for (int j = 0; j < NW; j++) {
                    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)x[j], (int)leader);
                    const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(x[j] >> 32), (int)leader);
                    same = same && x[j] == (((uint64_t)hi << 32) | lo);
                }
                const uint64_t members = (__ballot(same) >> gbase) & gmask;        // (the leader is one of them)
                const uint32_t m0 = (uint32_t)members, m1 = (uint32_t)(members >> 32);
                const bool lead = todo && lane == leader;
                const uint64_t b0 = __ballot(lead && m0 != 0), b1 = __ballot(lead && m1 != 0);
                if (lead) {
                    if (m0) {
                        const uint32_t at = base + i0 + (uint32_t)__popcll(b0 & below);
                        p.u_word_off[at] = wo + 2 * (uint64_t)u; p.u_len[at] = nb; p.u_ord[at] = ordb + 64 * u;
                        p.u_sample[at] = 0; p.u_bits[at] = m0;
                    }
                    if (m1) {
                        const uint32_t at = tb + i1 + (uint32_t)__popcll(b1 & below);
                        p.u_word_off[at] = wo + 2 * (uint64_t)u; p.u_len[at] = nb; p.u_ord[at] = ordb + 64 * u;
                        p.u_sample[at] = 32; p.u_bits[at] = m1;
                    }
                }
                i0 += (uint32_t)__popcll(b0); i1 += (uint32_t)__popcll(b1);
                if (same) todo = false;
            }
        }
        if (i1) {                                                    // (uniform) chunk 1's pieces behind chunk 0's
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's stores above, before its loads below
            for (uint32_t j = lane; j < i1; j += 64) {
                const uint32_t at = base + i0 + j;
                p.u_word_off[at] = p.u_word_off[tb + j]; p.u_len[at] = p.u_len[tb + j]; p.u_ord[at] = p.u_ord[tb + j];
                p.u_sample[at] = 32; p.u_bits[at] = p.u_bits[tb + j];
            }
        }
        if (lane == 0) { p.view_off[c] = base | VIEW_IN_POOL; p.v_nseg[c] = i0 + i1; }
    }
}

// ---------------------------------------------------------------------------------------------
// extra_csr_kernel: first slow-path row of every cluster, for a batch whose arrays are already in device memory
// ---------------------------------------------------------------------------------------------
// extra_cluster is non-decreasing: ex_first[c] = rows of clusters below c = lower bound of c (c = 0 .. C).  bad != 0 when
// the list is out of order or names a cluster >= C (the host checks the list itself when it has it).
__global__ __launch_bounds__(256) void extra_csr_kernel(const uint32_t* extra_cluster, uint32_t n_extra, uint32_t C,
                                                        uint32_t* ex_first, uint32_t* bad) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (uint32_t c = t; c <= C; c += stride) {
        uint32_t a = 0, b = n_extra;
        while (a < b) { const uint32_t m = (a + b) >> 1; if (extra_cluster[m] < c) a = m + 1; else b = m; }
        ex_first[c] = a;
    }
    for (uint32_t e = t; e < n_extra; e += stride)
        if (extra_cluster[e] >= C || (e && extra_cluster[e] < extra_cluster[e - 1])) *bad = 1u;
}

// ---------------------------------------------------------------------------------------------
// slow-path rows -> a prebuilt table in an item's scratch slice (same layout the scan kernel leaves)
// ---------------------------------------------------------------------------------------------
struct ExtraParams {
    const uint32_t* extra_ord;     // ordinals in the cluster's numbering (cluster_dedup_kernel's extra_dense)
    const uint32_t* extra_bits;    // [n_extra][W]
    const uint32_t* item_first;    // [n] first extra row of the item
    const uint32_t* item_nslots;   // [n] rows of the item
    const uint32_t* item_scratch;
    uint64_t* tab_key; uint32_t* tab_ord; uint32_t* chunkbits; uint32_t* chunkmask; uint32_t* item_count;
    const uint32_t* work;          // [gridDim.x] item ids of this launch
    uint32_t W, NS, KW;
};
__global__ void extra_fill_kernel(ExtraParams p) {
    const uint32_t item = p.work[blockIdx.x];
    const uint32_t first = p.item_first[item], n = p.item_nslots[item], slice = p.item_scratch[item];
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
        p.tab_key[((size_t)slice * p.KW) * p.NS + e] = KEY_EXTRA_FLAG | (uint64_t)(first + e);
        for (uint32_t j = 1; j < p.KW; j++) p.tab_key[((size_t)slice * p.KW + j) * p.NS + e] = 0;
        p.tab_ord[(size_t)slice * p.NS + e] = p.extra_ord[first + e];
        for (uint32_t w = 0; w < p.W; w++)
            p.chunkbits[((size_t)slice * p.W + w) * p.NS + e] = p.extra_bits[(size_t)(first + e) * p.W + w];
    }
    if (threadIdx.x < 8) p.chunkmask[slice * 8 + threadIdx.x] = 0xFFFFFFFFu;
    if (threadIdx.x == 0) p.item_count[item] = n;
}

// ---------------------------------------------------------------------------------------------
// rows_kernel
// ---------------------------------------------------------------------------------------------
struct RowsParams {
    const uint32_t* item_cluster; const uint32_t* item_nslots; const uint32_t* item_scratch;
    const uint32_t* item_count; const uint32_t* item_is_extra;
    const uint32_t* cluster_overflow;
    const uint32_t* cluster_seg_off;                                   // caller's segments (for M)
    const uint32_t* seg_sample; const uint32_t* seg_distinct;          // caller's sample column / distinct index
    const uint32_t* v_mode; const uint32_t* v_nstr; const uint32_t* v_dense;
    const uint32_t* cluster_nstrains; const uint32_t* cluster_npresab; const uint32_t* cluster_presab;
    const uint64_t* cluster_ordinal;
    const uint32_t* maf_lo; const uint32_t* maf_hi;
    const uint32_t* tab_ord; const uint32_t* chunkbits; const uint32_t* chunkmask;
    uint4* slot_hash;        // [slice][NS]
    // mode 0 (sorted)
    uint64_t* sorted_pair;   // [slice][NS]  ord << 32 | slot, ascending
    uint32_t* kept_prefix;   // [slice][NS+1] exclusive prefix of keep flags in sorted order
    // mode 1 (ordinal bitmaps)
    uint4* bm4;              // [slice][DENSE_WORDS_BIG] per ordinal word {occupied, kept, ordinals before it, kept ones before it}:
                             // one 16-byte record, because emit_kernel asks for all four at a random word per kept k-mer
    uint2* bm2;              // [slice][DENSE_WORDS_BIG] {occupied, kept} of an item that is one of several of its cluster: what
                             // bitmap_merge_kernel reads (it writes the cluster's records into the first item's bm4)
    const uint32_t* item_nsib;   // [item] items of the item's cluster
    uint32_t* mrows;         // [slice][DEDUP_MROWS]  M[d][Wp]: samples that carry distinct sequence d
    uint32_t* item_unique;   // [item]
    uint32_t* item_kept;     // [item]
    const uint32_t* work;    // [gridDim.x] item ids of this launch
    uint32_t W, NS;
    uint32_t consider_missing, patfilt, multiple_files;
};

constexpr uint32_t ROWS_THREADS = 1024;
constexpr uint32_t AT_SLOTS = 1024, AT_LIMIT = 768;   // distinct allele masks per item held in LDS (mode 1)
constexpr uint32_t SORT_MAX = 8192;   // >= insert_limit(nslots_max(1)), power of two

// exclusive scan of one value per thread over the block; returns the exclusive prefix, *total = sum
__device__ __forceinline__ uint32_t block_exscan(uint32_t v, uint32_t* wave_tot /*[17]*/, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t x = v;
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d);
        if ((int)lane >= d) x += y;
    }
    if (lane == 63) wave_tot[wave] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t w = 0; w < nw; w++) { uint32_t t = wave_tot[w]; wave_tot[w] = run; run += t; }
        wave_tot[nw] = run;
    }
    __syncthreads();
    uint32_t ex = wave_tot[wave] + x - v;
    *total = wave_tot[nw];
    __syncthreads();
    return ex;
}

__global__ __launch_bounds__(ROWS_THREADS) void rows_kernel(RowsParams p) {
    // mode 0: pairs u64[8192] | keepf u8[16384]          mode 1: M u32[4096] | occ u32[8192] | keep u32[8192]
    __shared__ __align__(16) uint32_t rsh[20480];
    __shared__ uint32_t cmask[8];
    __shared__ uint32_t wave_tot[ROWS_THREADS / 64 + 1];
    __shared__ uint32_t sh_cnt, sh_npres, at_count, sh_more;
    __shared__ uint64_t at_key[AT_SLOTS];      // mode 1: distinct allele masks of the item; mode 2: hash50 << 14 | slot
    __shared__ uint4 at_hash[AT_SLOTS];
    __shared__ uint32_t at_keep[AT_SLOTS];
    // mode 2 ("wide": up to 1024 distinct sequences, a k-mer's allele mask is up to 32 words)
    __shared__ uint16_t slot_tag[9600];        // per slot: table position of its mask in this round, or a WIDE_ state
    __shared__ uint32_t mstage[ROWS_THREADS / 32][4][33];   // the four masks being expanded, per half-wave; word 32 stays zero
    __shared__ uint32_t wstart[MAX_CHUNKS + 1];          // first segment of every 32-sample word
    // mode 2, masks of ONE distinct sequence (a k-mer private to an allele: 98 % of the slots of a cluster of 150 alleles
    // that each carry their own substitutions, 40 % where the alleles descend from one another): row, hash and keep flag
    // per distinct sequence, once, from the sample sets M -- such slots never see the mask table
    constexpr uint32_t SINGLE_MAX = 512;
    constexpr uint16_t SINGLE_TAG = 0x8000u;             // slot_tag: SINGLE_TAG | distinct index
    __shared__ uint4 single_hash[SINGLE_MAX];
    __shared__ uint32_t single_keep[SINGLE_MAX / 32];
    // ... and the slots of several bits of such an item, listed by step A's first pass for its dense second one (the list lies
    // where the round's row hashes will: at_hash is written behind step A only)
    __shared__ uint32_t ml_count;
    constexpr uint32_t ML_CAP = AT_SLOTS * sizeof(uint4) / sizeof(uint16_t);
    uint16_t* const mlist = reinterpret_cast<uint16_t*>(at_hash);

    PF_PROF_BEGIN();
    const uint32_t tid = threadIdx.x;
    const uint32_t item = p.work[blockIdx.x];
    const uint32_t c = p.item_cluster[item];
    const uint32_t slice = p.item_scratch[item];
    const uint32_t NS = p.NS, W = p.W;
    if (p.cluster_overflow[c]) {
        if (tid == 0) { p.item_unique[item] = 0; p.item_kept[item] = 0; }
        return;
    }
    const uint32_t ns = p.item_nslots[item];
    const uint32_t mode = p.v_mode[c] & 3u;
    const bool expand = mode == 1 && !p.item_is_extra[item];
    const bool wide = mode == 2 && !p.item_is_extra[item];
    const bool bitmaps = ranks_by_bitmap(mode, p.v_dense[c]);     // ranks from ordinal bitmaps; else sorted pairs
    const uint32_t nstr = p.cluster_nstrains[c], npres = p.cluster_npresab[c];
    const uint32_t nchunks = (nstr + 31) >> 5;
    const uint32_t Wp = (W + 3) & ~3u;
    const uint32_t* presab = p.cluster_presab + (size_t)c * W;

    uint64_t* pairs = reinterpret_cast<uint64_t*>(rsh);
    uint8_t* keepf = reinterpret_cast<uint8_t*>(rsh + 2 * SORT_MAX);
    uint32_t* M = rsh;
    uint32_t* occ = rsh + DEDUP_MROWS;
    uint32_t* keepbm = occ + DENSE_WORDS;
    const uint32_t dense_words = bitmaps ? (p.v_dense[c] + 31) >> 5 : 0;
    // more dense ordinals than two LDS bitmaps hold (a wide cluster of up to 1 048 576 windows over its distinct sequences):
    // ONE bitmap of DENSE_WIN words, per stretch of the ordinal space used twice -- occupied ordinals, then kept ones -- with
    // the slots' keep flags parked in slot_tag meanwhile.  (Such clusters sorted their (ordinal, slot) pairs per item and searched every
    // sibling item per k-mer before: 130 dependent global reads per kept k-mer at ten key partitions.)
    const bool big = bitmaps && dense_words > DENSE_WORDS;
    uint32_t* bigbm = rsh;
    uint16_t* segd = reinterpret_cast<uint16_t*>(rsh);   // mode 2: (distinct index << 5 | sample & 31) per segment;
                                                         // lives where `pairs` will be, until the rows are evaluated

    if (tid < 8) cmask[tid] = p.chunkmask[slice * 8 + tid];
    if (tid == 0) {
        sh_cnt = 0;
        uint32_t np = 0;
        for (uint32_t w = 0; w < W; w++) np += __popc(presab[w]);
        sh_npres = np;
    }
    if (wide) {
        for (uint32_t w = tid; w <= MAX_CHUNKS; w += ROWS_THREADS) wstart[w] = 0;
    } else if (bitmaps) {
        for (uint32_t i = tid; i < DEDUP_MROWS + 2 * DENSE_WORDS; i += ROWS_THREADS) rsh[i] = 0;
    } else {
        for (uint32_t i = tid; i < SORT_MAX; i += ROWS_THREADS) pairs[i] = EMPTY64;
    }
    __syncthreads();
    if (expand) {
        // M[d] = samples whose sequence is distinct sequence d, from the caller's segment list
        const uint32_t s0 = p.cluster_seg_off[c], s1 = p.cluster_seg_off[c + 1];
        for (uint32_t s = s0 + tid; s < s1; s += ROWS_THREADS) {
            const uint32_t d = p.seg_distinct[s], smp = p.seg_sample[s];
            atomicOr(&M[d * Wp + (smp >> 5)], 1u << (smp & 31));
        }
        __syncthreads();
        const uint32_t mw = p.v_nstr[c] * Wp;
        for (uint32_t i = tid; i < mw; i += ROWS_THREADS) p.mrows[(size_t)slice * DEDUP_MROWS + i] = M[i];
        PF_PROF_STAMP(57);
    }
    const uint32_t npresent = sh_npres;
    // denominators of panfeed.py:191 / :196
    const uint32_t n_eff = p.consider_missing ? npresent : nstr;
    const uint32_t lo = p.maf_lo[n_eff], hi = p.maf_hi[n_eff];
    // tuple(vec) == tuple(clusterpresab) can only hold for equal lengths and a NaN-free vector (panfeed.py:203)
    const bool same_possible = !p.patfilt && nstr == npres && (!p.consider_missing || npresent == nstr);
    const uint64_t ordinal = p.cluster_ordinal[c];

    const uint32_t* ordp = p.tab_ord + (size_t)slice * NS;
    const uint32_t* cb = p.chunkbits + (size_t)slice * W * NS;
    const bool f0 = (cmask[0] & 1) != 0, f1 = (cmask[0] & 2) != 0;

    // presence row of one k-mer -> (128-bit row hash, keep flag).  The row comes either from the chunk words of
    // slot i, or (mode 1) from the union of the sample sets of the distinct sequences in `amask`.
    auto row_eval = [&](bool from_mask, uint64_t amask, uint32_t i, uint4& hout) -> bool {
        H128 s;
        s.h1 = 0x9747b28cu ^ nstr; s.h2 = 0x1b873593u; s.h3 = 0xe6546b64u; s.h4 = 0x85ebca6bu;
        if (p.multiple_files) { s.h2 ^= (uint32_t)ordinal; s.h3 ^= (uint32_t)(ordinal >> 32); }
        uint32_t cnt = 0;
        bool eq = true;
        auto fold = [&](uint32_t ch, const uint32_t (&wv)[4]) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (ch + j < nchunks) { cnt += __popc(wv[j]); eq = eq && (wv[j] == presab[ch + j]); }
            mm3_block(s, wv[0], wv[1], wv[2], wv[3]);
        };
        if (from_mask) {
            for (uint32_t ch = 0; ch < nchunks; ch += 4) {
                uint32_t wv[4] = {0, 0, 0, 0};
                uint64_t t = amask;
                while (t) {
                    const uint32_t d = __ffsll((unsigned long long)t) - 1;
                    t &= t - 1;
                    const uint4 m = *reinterpret_cast<const uint4*>(&M[d * Wp + ch]);
                    wv[0] |= m.x; wv[1] |= m.y; wv[2] |= m.z; wv[3] |= m.w;
                }
                fold(ch, wv);
            }
        } else {
            // (sixteen chunk words of the slot asked for at a time, NS words apart in the scan's dump: the hash is a chain, the
            // loads need not be -- four at a time a row of 1 000 samples was eight trips to memory one after the other)
            for (uint32_t ch0 = 0; ch0 < nchunks; ch0 += 16) {
                uint32_t w16[16];
#pragma unroll
                for (uint32_t j = 0; j < 16; j++) w16[j] = cb[(size_t)min(ch0 + j, nchunks - 1) * NS + i];
#pragma unroll
                for (uint32_t j = 0; j < 16; j++) {
                    asm volatile("" : "+v"(w16[j]));
                    const uint32_t cc = ch0 + j;
                    if (!(cc < nchunks && ((cmask[cc >> 5] >> (cc & 31)) & 1))) w16[j] = 0;
                }
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    if (ch0 + 4 * q >= nchunks) break;
                    const uint32_t wv[4] = {w16[4 * q], w16[4 * q + 1], w16[4 * q + 2], w16[4 * q + 3]};
                    fold(ch0 + 4 * q, wv);
                }
            }
        }
        if (p.consider_missing) {
            // NaN where clusterpresab == 0 (panfeed.py:19): the image depends on presab too
            for (uint32_t ch = 0; ch < nchunks; ch += 4) {
                uint32_t wv[4];
#pragma unroll
                for (int j = 0; j < 4; j++) wv[j] = (ch + j < nchunks) ? ~presab[ch + j] : 0;
                mm3_block(s, wv[0], wv[1], wv[2], wv[3]);
            }
        }
        mm3_final(s, nchunks * 4);
        bool keep = cnt >= lo && cnt <= hi;                 // panfeed.py:197-200 (host-tabulated float64)
        if (same_possible && eq) keep = false;              // panfeed.py:202-204
        hout = make_uint4(s.h1, s.h2, s.h3, s.h4);
        return keep;
    };

    // slot_tag of a wide item's slot: position of its mask in the table of the round (< AT_SLOTS), or one of these
    constexpr uint16_t WIDE_PENDING = 0xFFFFu, WIDE_KEEP = 0xFFFEu, WIDE_DROP = 0xFFFDu, WIDE_EMPTY = 0xFFFCu;
    if (wide) {
        // ---- mode 2.  A k-mer's allele mask (which distinct sequences contain it) is nmw = ceil(D / 32) chunk
        // words; its presence row is gathered through the segment list: sample s carries the k-mer iff one of its
        // segments is a copy of a distinct sequence of the mask.  Distinct masks are evaluated once (table keyed by a
        // 50-bit hash of the words, verified word for word against the slot that opened the entry), in rounds of up
        // to AT_LIMIT distinct masks: a mask that finds the table full waits for the next round.
        const uint32_t s0 = p.cluster_seg_off[c], s1 = p.cluster_seg_off[c + 1], nsegs = s1 - s0;
        const uint32_t D = p.v_nstr[c], nmw = (D + 31) >> 5;
        // (the segments are sorted by sample: the first segment of a 32-sample word is the count of those before it)
        for (uint32_t s = tid; s < nsegs; s += ROWS_THREADS) {
            const uint32_t smp = p.seg_sample[s0 + s];
            segd[s] = (uint16_t)((p.seg_distinct[s0 + s] << 5) | (smp & 31u));
            atomicAdd(&wstart[(smp >> 5) + 1], 1u);
        }
        // (every slot starts as waiting; step A, which asks for a slot's mask words anyway, asks for its ordinal with them and
        // finds the empty ones -- a pass of its own over the table's ordinals was 9 600 more loads per item)
        for (uint32_t i = tid; i < ns; i += ROWS_THREADS) slot_tag[i] = WIDE_PENDING;
        __syncthreads();
        if (tid < 64) {
            constexpr uint32_t PL = (MAX_CHUNKS + 1 + 63) / 64;          // entries per lane
            uint32_t v[PL], sum = 0;
#pragma unroll
            for (uint32_t j = 0; j < PL; j++) {
                const uint32_t w = tid * PL + j;
                v[j] = w <= MAX_CHUNKS ? wstart[w] : 0;
                sum += v[j];
            }
            uint32_t x = sum;
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(x, d);
                if ((int)tid >= d) x += y;
            }
            uint32_t run = x - sum;
#pragma unroll
            for (uint32_t j = 0; j < PL; j++) {
                const uint32_t w = tid * PL + j;
                run += v[j];
                if (w <= MAX_CHUNKS) wstart[w] = run;
            }
        }
        __syncthreads();
        PF_PROF_STAMP(40);
        const uint32_t cm0 = cmask[0];                    // nmw <= 32: the flushed-chunk flags of the mask words
        // The words of a mask lie NS words apart in the scan's table dump; a slot asks for them eight at a time
        // (unconditionally, with a clamped index), so that it costs one trip to memory per eight words and not one per
        // word -- which, word after word through a hash chain, was 35-40 % of this kernel at 150 distinct sequences.
        constexpr uint32_t MWB = 8, MWMAX = DEDUP_MAX_D_WIDE / 32;
        // the round's table: at_key = hash50 << 14 | state (entry number once its words are written), the masks' words
        // in LDS behind the segment list (entry e: tabw[e * nmw ..]), their rows in this slice's bitmap arrays (which
        // are written at the very end of this kernel only), four entries to a stripe of RW words
        constexpr uint32_t ENT_BUSY = 0x3FFEu, ENT_DEAD = 0x3FFFu;
        uint32_t* tabw = rsh + ((nsegs + 1) >> 1);
        const uint32_t RW = (nchunks + 3) & ~3u;
        const uint32_t cap = min(min(AT_LIMIT, (20480u - ((nsegs + 1) >> 1)) / nmw), 4u * (DENSE_WORDS_BIG / RW));
        auto row_of = [&](uint32_t e) -> uint32_t* {
            return reinterpret_cast<uint32_t*>(p.bm4 + (size_t)slice * DENSE_WORDS_BIG) + (size_t)e * RW;
        };
        // one half-wave per mask: lane j gathers row word 32 r + j in round r, the 32 words of a round then go through
        // the row hash in order (eight blocks), every lane running it on the words read back from LDS (one broadcast
        // 128-bit read per block).  The segments a lane looks at in round 0 are the same for every mask: they are
        // held in registers (two to a register, the unused places point at a distinct index whose mask word is the
        // zero word 32), so that a mask costs WIDE_SEGREG independent LDS reads per lane and no dependent pair.
        // (What one lane writes to LDS for the other lanes of ITS wave needs no wait: a wave's LDS operations run in
        // the order they were issued.  The fences are of wavefront scope, for the compiler -- a workgroup-scope release
        // waits for every memory operation of the wave, the next mask's words on their way from global memory among
        // them, which is exactly what should stay in flight.)
        const uint32_t hw = tid >> 5, hl = tid & 31u;
        constexpr uint32_t WIDE_SEGREG = 32;
        uint32_t sg[WIDE_SEGREG / 2];
        uint32_t my_q0 = 0, my_q1 = 0, sg_used = 0;
        if (hl < 4) mstage[hw][hl][32] = 0;
        // FOUR masks a call (entries e, e + 32, e + 64, e + 96 of the round's table): a segment's place and bit numbers are
        // worked out once and looked up in the four masks -- three instructions a mask and segment where one mask a
        // call took seven, and the step is bound by exactly those.
        constexpr uint32_t GM = 4;
        auto gather_rows = [&](uint32_t e0, uint32_t n_ent) {
#pragma unroll
            for (uint32_t m = 0; m < GM; m++) {
                const uint32_t e = e0 + 32 * m;
                if (hl < nmw) mstage[hw][m][hl] = e < n_ent ? tabw[e * nmw + hl] : 0u;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t* ms = mstage[hw][0];
            constexpr uint32_t MS = 33;                                // words from one mask's stage to the next
            for (uint32_t w0 = 0; w0 < RW; w0 += 32) {
                const uint32_t w = w0 + hl;
                uint32_t word[GM];
#pragma unroll
                for (uint32_t m = 0; m < GM; m++) word[m] = 0;
                auto take = [&](uint32_t e) {                              // e: distinct index << 5 | sample & 31
                    const uint32_t at = e >> 10, sh = (e >> 5) & 31u;
#pragma unroll
                    for (uint32_t m = 0; m < GM; m++) word[m] |= ((ms[m * MS + at] >> sh) & 1u) << (e & 31u);
                };
                if (w0 == 0) {
#pragma unroll
                    for (uint32_t jb = 0; jb < WIDE_SEGREG / 2; jb += 4) {
                        if (2 * jb >= sg_used) break;                         // wave-uniform; eight segments a step
#pragma unroll
                        for (uint32_t j = jb; j < jb + 4; j++) {
                            // (opaque copy: decoded here, every time -- the compiler would otherwise keep the decoded
                            // values of every segment alive across the loop over the masks and spill)
                            uint32_t pr = sg[j];
                            asm volatile("" : "+v"(pr));
                            take(pr & 0xFFFFu);
                            take(pr >> 16);
                        }
                    }
                    for (uint32_t q = my_q0 + WIDE_SEGREG; q < my_q1; q++) take(segd[q]);   // a word with more segments than that
                } else if (w < nchunks) {
                    for (uint32_t q = wstart[w], qe = wstart[w + 1]; q < qe; q++) take(segd[q]);
                }
                if (w < RW) {                                        // (words past nchunks: zero, as the row hash wants them)
#pragma unroll
                    for (uint32_t m = 0; m < GM; m++)
                        if (e0 + 32 * m < n_ent) row_of(e0 + 32 * m)[w] = word[m];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();                    // mstage[hw] is rewritten by the next call
        };
        // the row of one mask, read back by ONE lane: hash, count, comparison with the cluster's own row
        auto row_hash = [&](const uint32_t* row, uint4& hout) -> bool {
            H128 st;
            st.h1 = 0x9747b28cu ^ nstr; st.h2 = 0x1b873593u; st.h3 = 0xe6546b64u; st.h4 = 0x85ebca6bu;
            if (p.multiple_files) { st.h2 ^= (uint32_t)ordinal; st.h3 ^= (uint32_t)(ordinal >> 32); }
            uint32_t cnt = 0;
            bool eq = true;
            // (four 16-byte pieces of the row asked for at a time: the hash is a chain, the loads need not be -- piece by piece
            // a row of 1 000 samples was eight trips to L2 one after the other per mask, 40 at 5 000)
            for (uint32_t ch0 = 0; ch0 < nchunks; ch0 += 16) {
                uint4 vq[4];
#pragma unroll
                for (uint32_t q = 0; q < 4; q++)
                    vq[q] = *reinterpret_cast<const uint4*>(row + min(ch0 + 4 * q, RW - 4));
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    const uint32_t ch = ch0 + 4 * q;
                    if (ch >= nchunks) break;
                    const uint32_t wv[4] = {vq[q].x, vq[q].y, vq[q].z, vq[q].w};
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (ch + j < nchunks) { cnt += __popc(wv[j]); eq = eq && (wv[j] == presab[ch + j]); }
                    mm3_block(st, wv[0], wv[1], wv[2], wv[3]);
                }
            }
            if (p.consider_missing) {
                for (uint32_t ch = 0; ch < nchunks; ch += 4) {
                    uint32_t wv[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) wv[j] = (ch + j < nchunks) ? ~presab[ch + j] : 0;
                    mm3_block(st, wv[0], wv[1], wv[2], wv[3]);
                }
            }
            mm3_final(st, nchunks * 4);
            bool keep = cnt >= lo && cnt <= hi;                 // panfeed.py:197-200
            if (same_possible && eq) keep = false;              // panfeed.py:202-204
            hout = make_uint4(st.h1, st.h2, st.h3, st.h4);
            return keep;
        };
        // ---- singleton masks: M[d] = the samples that carry distinct sequence d, in the place the round's table will
        // take afterwards; one lane per d hashes its row
        const uint32_t m_base = (((nsegs + 1) >> 1) + 3u) & ~3u;
        // (for clusters of three or more work items: there an allele's private k-mers are nearly all there is and its mask
        // comes back in every item -- 2 000 clusters of ~150 SURVEY alleles, ten items each: rows_kernel 6.2 -> 5.7 ms; with
        // related alleles, two items a cluster, the set-up cost what it saved: 1.40 -> 1.54 ms)
        const bool singles = D <= SINGLE_MAX && m_base + D * RW <= 20480u && p.item_nsib[item] >= 3;
        if (singles) {
            uint32_t* Ms = rsh + m_base;
            for (uint32_t i = tid; i < D * RW; i += ROWS_THREADS) Ms[i] = 0;
            if (tid < SINGLE_MAX / 32) single_keep[tid] = 0;
            __syncthreads();
            for (uint32_t s = tid; s < nsegs; s += ROWS_THREADS) {
                const uint32_t e = segd[s];                                   // distinct index << 5 | sample & 31
                // (the sample's word: the segments are sorted by sample, wstart[w] = segments of the words before w)
                uint32_t lo_w = 0, hi_w = nchunks;
                while (lo_w + 1 < hi_w) { const uint32_t mid = (lo_w + hi_w) >> 1; if (wstart[mid] <= s) lo_w = mid; else hi_w = mid; }
                atomicOr(&Ms[(e >> 5) * RW + lo_w], 1u << (e & 31u));
            }
            __syncthreads();
            if (tid < D) {
                uint4 h;
                const bool keep = row_hash(Ms + tid * RW, h);
                single_hash[tid] = h;
                if (keep) atomicOr(&single_keep[tid >> 5], 1u << (tid & 31u));
            }
            __syncthreads();
        }
        for (;;) {
            for (uint32_t t = tid; t < AT_SLOTS; t += ROWS_THREADS) at_key[t] = 0;
            if (tid == 0) { at_count = 0; sh_more = 0; ml_count = 0; }
            __syncthreads();
            PF_PROF_STAMP(55);
            // A: every waiting slot finds its mask in the table (hash, then word for word against the entry's copy) or
            // opens an entry for it.  Nobody waits in place for an entry whose words are still being written: the lane
            // goes round the loop again, the writer -- which may be a lane of the same wave -- finishes within ITS turn.
            // (The trips to memory are what this step waits for -- one workgroup per CU, sixteen waves: the words of four
            // slots are asked for at a time where a mask is eight words at most, of two up to sixteen words.)
            auto step_a = [&](auto nbc) {
                constexpr uint32_t NWA = decltype(nbc)::value * MWB, U = NWA == MWB ? 4 : NWA == 2 * MWB ? 2 : 1;
                // slot i's mask (words w) into the round's table: its entry's number becomes the slot's tag
                auto resolve = [&](uint32_t i, const uint32_t (&w)[NWA]) {
                    uint64_t a1 = 0x9E3779B97F4A7C15ull, a2 = 0xC2B2AE3D27D4EB4Full;
#pragma unroll
                    for (uint32_t j = 0; j < NWA; j++) {              // sums of word x odd constant of its place
                        const uint32_t kj = (2 * j + 1) * 0x9E3779B1u;
                        a1 += (uint64_t)w[j] * (kj | 1u);
                        a2 += (uint64_t)w[j] * (((kj >> 9) | (kj << 23)) | 1u);
                    }
                    const uint64_t hsh = mix64(a1 ^ ((a2 << 32) | (a2 >> 32)));
                    uint64_t h50 = hsh >> 14;
                    if (!h50) h50 = 1;
                    uint32_t a = (uint32_t)(hsh & (AT_SLOTS - 1));
                    uint32_t tag = WIDE_PENDING, probes = 0, spins = 0;
                    bool done = false;
                    while (!done) {
                        uint64_t cur = __hip_atomic_load(&at_key[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (cur == 0) {
                            if (__hip_atomic_load(&at_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= cap) {
                                done = true;                                     // full: next round
                            } else {
                                cur = atomicCAS((unsigned long long*)&at_key[a], 0ull, (unsigned long long)((h50 << 14) | ENT_BUSY));
                                if (cur == 0) {
                                    const uint32_t e = atomicAdd(&at_count, 1u);
                                    if (e >= cap) {
                                        __hip_atomic_store(&at_key[a], (h50 << 14) | ENT_DEAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    } else {
#pragma unroll
                                        for (uint32_t j = 0; j < NWA; j++)
                                            if (j < nmw) tabw[e * nmw + j] = w[j];
                                        __hip_atomic_store(&at_key[a], (h50 << 14) | e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        tag = e;
                                    }
                                    done = true;
                                }
                            }
                        }
                        if (!done) {                                             // cur: somebody's entry
                            const uint32_t st = (uint32_t)cur & 0x3FFFu;
                            uint32_t adv = 1;
                            if ((cur >> 14) == h50) {
                                if (st == ENT_BUSY) { adv = 0; spins++; }        // look again on the next trip
                                else if (st == ENT_DEAD) done = true;
                                else {
                                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                                    uint32_t diff = 0;
#pragma unroll
                                    for (uint32_t j = 0; j < NWA; j++)
                                        if (j < nmw) diff |= tabw[st * nmw + j] ^ w[j];
                                    if (!diff) { tag = st; done = true; }
                                }
                            }
                            a = (a + adv) & (AT_SLOTS - 1);
                            probes += adv;
                            if (probes >= AT_SLOTS || spins >= 4096u) done = true;   // (gives up for this round; never met)
                        }
                    }
                    if (tag == WIDE_PENDING) sh_more = 1;
                    else slot_tag[i] = (uint16_t)tag;
                };
                for (uint32_t i0 = tid; i0 < ns; i0 += U * ROWS_THREADS) {
                    uint32_t wu[U][NWA], ou[U];
                    bool pend[U], any = false;
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) {
                        const uint32_t i = i0 + u * ROWS_THREADS;
                        pend[u] = i < ns && slot_tag[min(i, ns - 1)] == WIDE_PENDING;
                        any = any || pend[u];
                    }
                    if (!any) continue;
                    // (every load of the trip first, then an opaque use of every value: left to itself the compiler
                    // turns "clamped index, then select" back into a branch around each load and waits for each one)
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) {
                        const uint32_t i = min(i0 + u * ROWS_THREADS, ns - 1);
#pragma unroll
                        for (uint32_t j = 0; j < NWA; j++) wu[u][j] = cb[min(j, nmw - 1) * NS + i];   // (32-bit index: a slice is W * NS words)
                        ou[u] = ordp[i];
                    }
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) {
#pragma unroll
                        for (uint32_t j = 0; j < NWA; j++) asm volatile("" : "+v"(wu[u][j]));
                        asm volatile("" : "+v"(ou[u]));
                    }
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) {
#pragma unroll
                        for (uint32_t j = 0; j < NWA; j++)
                            if (!(j < nmw && ((cm0 >> j) & 1))) wu[u][j] = 0;
                    }
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) {
                        if (!pend[u]) continue;
                        const uint32_t i = i0 + u * ROWS_THREADS;
                        if (ou[u] == NO_ORD) { slot_tag[i] = WIDE_EMPTY; continue; }
                        const uint32_t (&w)[NWA] = wu[u];
                        if (singles) {
                            uint32_t pc = 0, at = 0;
#pragma unroll
                            for (uint32_t j = 0; j < NWA; j++) { pc += __popc(w[j]); if (w[j]) at = 32u * j + (uint32_t)__ffs((int)w[j]) - 1u; }
                            if (pc == 1) { slot_tag[i] = (uint16_t)(SINGLE_TAG | at); continue; }
                        }
                        if (singles) {
                            // (a slot of several bits: on the list, looked at by a dense pass below -- here the hash and
                            // the walk through the table ran with the one or two lanes of a wave that had such a slot,
                            // ten times per wave: 40 000 of this kernel's 155 000 cycles per item at ~150 SURVEY alleles)
                            const uint32_t q = atomicAdd(&ml_count, 1u);
                            if (q < ML_CAP) { mlist[q] = (uint16_t)i; continue; }
                        }
                        resolve(i, w);
                    }
                }
                if (singles) {
                    // the listed slots, 64 to a wave and trip: words in, mask into the table
                    __syncthreads();
                    const uint32_t nl = min(ml_count, ML_CAP);
                    for (uint32_t q = tid; q < nl; q += ROWS_THREADS) {
                        const uint32_t i = mlist[q];
                        uint32_t w[NWA];
#pragma unroll
                        for (uint32_t j = 0; j < NWA; j++) w[j] = cb[min(j, nmw - 1) * NS + i];
#pragma unroll
                        for (uint32_t j = 0; j < NWA; j++) {
                            asm volatile("" : "+v"(w[j]));
                            if (!(j < nmw && ((cm0 >> j) & 1))) w[j] = 0;
                        }
                        resolve(i, w);
                    }
                }
            };
            if (nmw <= MWB) step_a(std::integral_constant<uint32_t, 1>{});
            else if (nmw <= 2 * MWB) step_a(std::integral_constant<uint32_t, 2>{});
            else step_a(std::integral_constant<uint32_t, MWMAX / MWB>{});
            PF_PROF_STAMP(56);
            __syncthreads();
            PF_PROF_STAMP(41);
            const uint32_t n_ent = min(at_count, cap);
            // (the lane's segments, decoded into registers for the round's gathers: not held across step A, which
            // wants the registers for the words of its slots)
            if (hl < nchunks) { my_q0 = wstart[hl]; my_q1 = wstart[hl + 1]; }
#pragma unroll
            for (uint32_t j = 0; j < WIDE_SEGREG / 2; j++) {
                const uint32_t q = my_q0 + 2 * j;
                const uint32_t e0 = q < my_q1 ? segd[q] : 0x8000u, e1 = q + 1 < my_q1 ? segd[q + 1] : 0x8000u;
                sg[j] = e0 | (e1 << 16);
            }
            sg_used = min(my_q1 - my_q0, WIDE_SEGREG);          // the most any lane of the wave holds
            for (int dd = 1; dd < 64; dd <<= 1) sg_used = max(sg_used, (uint32_t)__shfl_xor(sg_used, dd));
            // C: the rows of the round's masks, one half-wave per mask, into the row stripes ...
            for (uint32_t e = hw; e < n_ent; e += GM * (ROWS_THREADS / 32)) gather_rows(e, n_ent);
            __syncthreads();
            PF_PROF_STAMP(42);
            // ... and their hashes and keep flags, one LANE per mask (half-waves that ran the hash in step, as rounds 2-3
            // had it, spent two thirds of their issue slots on 32 copies of the same quarter-rate multiplications)
            for (uint32_t e = tid; e < n_ent; e += ROWS_THREADS) {
                uint4 h;
                const bool keep = row_hash(row_of(e), h);
                at_hash[e] = h; at_keep[e] = keep ? 1u : 0u;
            }
            __syncthreads();
            // D: every slot of the round takes its mask's result
            for (uint32_t i = tid; i < ns; i += ROWS_THREADS) {
                const uint32_t tag = slot_tag[i];
                if (tag >= SINGLE_TAG && tag < SINGLE_TAG + SINGLE_MAX) {
                    const uint32_t dd = tag - SINGLE_TAG;
                    const bool kp = (single_keep[dd >> 5] >> (dd & 31u)) & 1u;
                    if (kp) p.slot_hash[(size_t)slice * NS + i] = single_hash[dd];
                    slot_tag[i] = kp ? WIDE_KEEP : WIDE_DROP;
                    continue;
                }
                if (tag >= AT_SLOTS) continue;
                if (at_keep[tag]) p.slot_hash[(size_t)slice * NS + i] = at_hash[tag];    // (emit_kernel reads kept k-mers' only)
                slot_tag[i] = at_keep[tag] ? WIDE_KEEP : WIDE_DROP;
            }
            __syncthreads();
            const bool more = sh_more != 0;
            PF_PROF_STAMP(43);
#ifdef PF_PROF
            if (tid == 0) { atomicAdd(&pf_prof[48], (unsigned long long)n_ent); atomicAdd(&pf_prof[50], 1ull); }
#endif
            __syncthreads();
            if (!more) break;
        }
#ifdef PF_PROF
        if (tid == 0) { atomicAdd(&pf_prof[47], 1ull); atomicAdd(&pf_prof[49], (unsigned long long)ns); }
#endif
        // segd is no longer needed: its place is the bitmaps' or the pairs'
        if (big) { }
        else if (bitmaps) { for (uint32_t i = tid; i < 2 * DENSE_WORDS; i += ROWS_THREADS) occ[i] = 0; }
        else { for (uint32_t i = tid; i < SORT_MAX; i += ROWS_THREADS) pairs[i] = EMPTY64; }
        __syncthreads();
    }
    // mode 1: ordinal and the two mask words of XS slots of a thread, loaded together
    constexpr uint32_t XS = 5;
    uint32_t xo_[XS], xl_[XS], xh_[XS];
    auto load_slots = [&](uint32_t i0) {
#pragma unroll
        for (uint32_t u = 0; u < XS; u++) {
            const uint32_t i = min(i0 + u * ROWS_THREADS, ns - 1);
            xo_[u] = ordp[i]; xl_[u] = cb[i]; xh_[u] = cb[(W > 1 ? NS : 0u) + i];     // (one plane only when W == 1; f1 is off then)
        }
#pragma unroll
        for (uint32_t u = 0; u < XS; u++) { asm volatile("" : "+v"(xo_[u])); asm volatile("" : "+v"(xl_[u])); asm volatile("" : "+v"(xh_[u])); }
    };
    if (expand) {
        // Few distinct sequences -> few distinct masks: evaluate each distinct mask once (LDS table), then
        // hand the result to every slot that carries it.
        for (uint32_t t = tid; t < AT_SLOTS; t += ROWS_THREADS) at_key[t] = 0;
        if (tid == 0) at_count = 0;
        __syncthreads();
        // (five slots a trip: their ordinals and mask words are asked for together -- slot by slot, ordinal then words, this
        // loop and the one further down were 70 % of the kernel on clusters of 30-64 distinct sequences)
        for (uint32_t i0 = tid; i0 < ns; i0 += XS * ROWS_THREADS)
        for (uint32_t u = 0, first = 1; u < XS; u++, first = 0) {
            if (first) load_slots(i0);
            const uint32_t i = i0 + u * ROWS_THREADS;
            if (i >= ns || xo_[u] == NO_ORD) continue;
            const uint64_t amask = (f0 ? (uint64_t)xl_[u] : 0) | (f1 ? (uint64_t)xh_[u] << 32 : 0);
            if (!amask) continue;
            uint32_t a = (uint32_t)mix64(amask) & (AT_SLOTS - 1);
            for (uint32_t probes = 0; probes < AT_SLOTS; probes++) {
                uint64_t cur = __hip_atomic_load(&at_key[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cur == 0) {
                    if (__hip_atomic_load(&at_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= AT_LIMIT) break;
                    cur = atomicCAS((unsigned long long*)&at_key[a], 0ull, (unsigned long long)amask);
                    if (cur == 0) { atomicAdd(&at_count, 1u); break; }
                }
                if (cur == amask) break;
                a = (a + 1) & (AT_SLOTS - 1);
            }
        }
        __syncthreads();
        PF_PROF_STAMP(58);
        for (uint32_t t = tid; t < AT_SLOTS; t += ROWS_THREADS) {
            const uint64_t key = at_key[t];
            if (!key) continue;
            uint4 h;
            const bool keep = row_eval(true, key, 0, h);         // (eight lanes per mask, as finish_kernel has it, measured
            at_hash[t] = h;                                      // slower here: 1.94 -> 2.13 ms at ~60 distinct sequences)
            at_keep[t] = keep ? 1u : 0u;
        }
        __syncthreads();
        PF_PROF_STAMP(59);
#ifdef PF_PROF
        if (tid == 0) { atomicAdd(&pf_prof[60], 1ull); atomicAdd(&pf_prof[61], (unsigned long long)at_count); }
#endif
    }
    // a k-mer's ordinal and keep flag into the item's bitmaps (or its list of pairs to sort)
    auto place = [&](uint32_t i, uint32_t o, bool keep) {
        if (!bitmaps) {
            keepf[i] = keep ? 1 : 0;
            const uint32_t at = atomicAdd(&sh_cnt, 1u);
            if (at < SORT_MAX) pairs[at] = ((uint64_t)o << 32) | i;
        } else if (big) {
            slot_tag[i] = keep ? 1 : 0;                    // (the bits: stretch by stretch, below)
        } else if ((o >> 5) < dense_words) {
            atomicOr(&occ[o >> 5], 1u << (o & 31));
            if (keep) atomicOr(&keepbm[o >> 5], 1u << (o & 31));
        }
        // the kept k-mers of the item, (ordinal, slot) in any order, for emit_kernel -- which otherwise looks at every slot
        // of the table and finds out slot by slot, load after load, that most are empty or dropped
        if (bitmaps && keep && (o >> 5) < dense_words)
            p.sorted_pair[(size_t)slice * NS + atomicAdd(&sh_cnt, 1u)] = ((uint64_t)o << 32) | i;
    };
    if (wide) {
        // (hash and flag of every slot: round D above; here only the ordinals are read, five slots' at a time)
        constexpr uint32_t SU = 5;
        for (uint32_t i0 = tid; i0 < ns; i0 += SU * ROWS_THREADS) {
            uint32_t o_[SU];
#pragma unroll
            for (uint32_t u = 0; u < SU; u++) o_[u] = ordp[min(i0 + u * ROWS_THREADS, ns - 1)];
#pragma unroll
            for (uint32_t u = 0; u < SU; u++) asm volatile("" : "+v"(o_[u]));
#pragma unroll
            for (uint32_t u = 0; u < SU; u++) {
                const uint32_t i = i0 + u * ROWS_THREADS;
                if (i < ns && o_[u] != NO_ORD) place(i, o_[u], slot_tag[i] == WIDE_KEEP);
            }
        }
    }
    if (expand) {
        for (uint32_t i0 = tid; i0 < ns; i0 += XS * ROWS_THREADS)
        for (uint32_t u = 0, first = 1; u < XS; u++, first = 0) {
            if (first) load_slots(i0);
            const uint32_t i = i0 + u * ROWS_THREADS;
            const uint32_t o = xo_[u];
            if (i >= ns || o == NO_ORD) continue;
            uint4 h;
            bool keep;
            const uint64_t amask = (f0 ? (uint64_t)xl_[u] : 0) | (f1 ? (uint64_t)xh_[u] << 32 : 0);
            bool found = false;
            uint32_t a = (uint32_t)mix64(amask) & (AT_SLOTS - 1);
            for (uint32_t probes = 0; amask && probes < AT_SLOTS; probes++) {
                const uint64_t cur = at_key[a];
                if (cur == 0) break;
                if (cur == amask) { found = true; break; }
                a = (a + 1) & (AT_SLOTS - 1);
            }
            if (found) { h = at_hash[a]; keep = at_keep[a] != 0; }
            else keep = row_eval(true, amask, i, h);          // table was full: evaluate this slot on its own
            if (keep) p.slot_hash[(size_t)slice * NS + i] = h;
            place(i, o, keep);
        }
    }
    for (uint32_t i = tid; i < ns && !wide && !expand; i += ROWS_THREADS) {
        const uint32_t o = ordp[i];
        if (o == NO_ORD) continue;
        uint4 h;
        bool keep;
        if (expand) {
            const uint64_t amask = (f0 ? (uint64_t)cb[i] : 0) | (f1 ? (uint64_t)cb[(size_t)NS + i] << 32 : 0);
            bool found = false;
            uint32_t a = (uint32_t)mix64(amask) & (AT_SLOTS - 1);
            for (uint32_t probes = 0; amask && probes < AT_SLOTS; probes++) {
                const uint64_t cur = at_key[a];
                if (cur == 0) break;
                if (cur == amask) { found = true; break; }
                a = (a + 1) & (AT_SLOTS - 1);
            }
            if (found) { h = at_hash[a]; keep = at_keep[a] != 0; }
            else keep = row_eval(true, amask, i, h);          // table was full: evaluate this slot on its own
        } else {
            keep = row_eval(false, 0, i, h);
        }
        if (keep) p.slot_hash[(size_t)slice * NS + i] = h;
        place(i, o, keep);
    }
    __syncthreads();
    if (bitmaps && tid == 0) p.kept_prefix[(size_t)slice * (NS + 1)] = sh_cnt;
    PF_PROF_STAMP(44);

    if (big) {
        // stretches of DENSE_WIN words of the ordinal space, one after the other; in each the LDS bitmap is used twice --
        // the occupied ordinals of the stretch, then the kept ones -- with the counts of the stretches before it carried on
        constexpr uint32_t PWB = DENSE_WIN / ROWS_THREADS;           // 16 words per thread
        const size_t gb = (size_t)slice * DENSE_WORDS_BIG;
        const bool alone = p.item_nsib[item] == 1;
        uint32_t base_o = 0, base_k = 0;
        for (uint32_t w0 = 0; w0 < dense_words; w0 += DENSE_WIN) {
            for (int round = 0; round < 2; round++) {
                __syncthreads();
                for (uint32_t i = tid; i < DENSE_WIN; i += ROWS_THREADS) bigbm[i] = 0;
                __syncthreads();
                for (uint32_t i = tid; i < ns; i += ROWS_THREADS) {
                    const uint32_t o = ordp[i];
                    if (o == NO_ORD || (o >> 5) >= dense_words || (o >> 5) < w0 || (o >> 5) >= w0 + DENSE_WIN) continue;
                    if (!round || slot_tag[i] == 1) atomicOr(&bigbm[(o >> 5) - w0], 1u << (o & 31));
                }
                __syncthreads();
                uint32_t sm = 0;
#pragma unroll
                for (uint32_t j = 0; j < PWB; j++) sm += __popc(bigbm[tid * PWB + j]);
                uint32_t tot;
                uint32_t run = (round ? base_k : base_o) + block_exscan(sm, wave_tot, &tot);
                uint32_t* rec = reinterpret_cast<uint32_t*>(p.bm4 + gb) + (round ? 1 : 0);      // .x / .y, and .z / .w two further on
                uint32_t* rec2 = reinterpret_cast<uint32_t*>(p.bm2 + gb) + (round ? 1 : 0);
#pragma unroll
                for (uint32_t j = 0; j < PWB; j++) {
                    const uint32_t w = w0 + tid * PWB + j;
                    if (w < dense_words) {
                        const uint32_t a = bigbm[tid * PWB + j];
                        if (alone) { rec[4 * (size_t)w] = a; rec[4 * (size_t)w + 2] = run; }
                        else rec2[2 * (size_t)w] = a;
                        run += __popc(a);
                    }
                }
                if (round) base_k += tot; else base_o += tot;
            }
        }
        if (tid == 0) { p.item_unique[item] = base_o; p.item_kept[item] = base_k; }
        PF_PROF_STAMP(45);
        return;
    }

    if (bitmaps) {
        // ranks come from the ordinal bitmaps: prefix popcounts per word, stored for emit_kernel
        constexpr uint32_t PW = DENSE_WORDS / ROWS_THREADS;   // 8 words per thread
        uint32_t so = 0, sk = 0;
#pragma unroll
        for (uint32_t j = 0; j < PW; j++) {
            const uint32_t w = tid * PW + j;
            so += __popc(occ[w]);
            sk += __popc(keepbm[w]);
        }
        uint32_t tot_o, tot_k;
        uint32_t bo = block_exscan(so, wave_tot, &tot_o);
        uint32_t bk = block_exscan(sk, wave_tot, &tot_k);
        const size_t gb = (size_t)slice * DENSE_WORDS_BIG;
        const bool alone = p.item_nsib[item] == 1;
#pragma unroll
        for (uint32_t j = 0; j < PW; j++) {
            const uint32_t w = tid * PW + j;
            if (w < dense_words) {
                const uint32_t a = occ[w], b = keepbm[w];
                if (alone) p.bm4[gb + w] = make_uint4(a, b, bo, bk);
                else p.bm2[gb + w] = make_uint2(a, b);
                bo += __popc(a); bk += __popc(b);
            }
        }
        if (tid == 0) { p.item_unique[item] = tot_o; p.item_kept[item] = tot_k; }
        PF_PROF_STAMP(45);
        return;
    }

    const uint32_t found = sh_cnt;
    uint32_t Msz = 64;
    while (Msz < found && Msz < SORT_MAX) Msz <<= 1;
    // bitonic sort of pairs[0..Msz) ascending (EMPTY64 pads sort to the end)
    for (uint32_t k2 = 2; k2 <= Msz; k2 <<= 1) {
        for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < (Msz >> 1); t += ROWS_THREADS) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t l = i | j;
                const bool up = (i & k2) == 0;
                const uint64_t a = pairs[i], b = pairs[l];
                if ((a > b) == up) { pairs[i] = b; pairs[l] = a; }
            }
            __syncthreads();
        }
    }
    const uint32_t n = found < SORT_MAX ? found : SORT_MAX;
    // keep-prefix in sorted order: thread t owns positions [t*PER, t*PER+PER)
    constexpr uint32_t PER = SORT_MAX / ROWS_THREADS;
    uint32_t loc[PER];
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) {
        const uint32_t r = tid * PER + j;
        uint32_t kf = 0;
        if (r < n) kf = keepf[(uint32_t)pairs[r]];
        loc[j] = sum;
        sum += kf;
    }
    uint32_t total;
    const uint32_t base = block_exscan(sum, wave_tot, &total);
    uint64_t* sp = p.sorted_pair + (size_t)slice * NS;
    uint32_t* kp = p.kept_prefix + (size_t)slice * (NS + 1);
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) {
        const uint32_t r = tid * PER + j;
        if (r < n) { sp[r] = pairs[r]; kp[r] = base + loc[j]; }
    }
    if (tid == 0) {
        kp[n] = total;
        p.item_unique[item] = n;
        p.item_kept[item] = total;
    }
    PF_PROF_STAMP(46);
}

// ---------------------------------------------------------------------------------------------
// bitmap_merge_kernel: one ordinal bitmap per cluster out of its items'
// ---------------------------------------------------------------------------------------------
// A cluster of several items (key partitions, slow-path rows) whose ranks come from ordinal bitmaps: every item marked
// the ordinals of ITS k-mers; a k-mer's rank in the cluster counts the ordinals below its own in all of them.  The items'
// bitmaps are disjoint, so their OR with one prefix count per word answers that with one look instead of one per item
// (emit_kernel asked every sibling, four loads each, for every kept k-mer).  The result replaces the first item's arrays.
struct BitmapMergeParams {
    const uint32_t* sub_cluster; const uint32_t* cluster_item0; const uint32_t* cluster_nitems;   // as BaseParams
    const uint32_t* item_scratch; const uint32_t* cluster_overflow; const uint32_t* v_mode; const uint32_t* v_dense;
    const uint32_t* item_fused;        // [item] nonzero: finished by finish_kernel, which keeps its bitmaps in LDS
    uint4* bm4;                        // [slice][DENSE_WORDS_BIG] {occupied, kept, ordinals before, kept before} per ordinal word
    const uint2* bm2;                  // [slice][DENSE_WORDS_BIG] the items' own {occupied, kept}
};
constexpr uint32_t BM_THREADS = 256, BM_WPT = 4, BM_SIB = 64;   // words per thread and round; sibling slices staged at a time
__global__ __launch_bounds__(BM_THREADS) void bitmap_merge_kernel(BitmapMergeParams p) {
    // Round 3's form took one word per thread and round and walked the siblings in a loop whose every trip was two
    // dependent loads (the sibling's slice number, then its word): 2 ni latencies in a row per 256 words, 92 % of the wave
    // cycles waiting -- 2.6 ms per 2 000 clusters of ten key partitions for 0.75 GB of traffic.  Now the slice numbers are
    // staged in LDS once, a thread takes BM_WPT words a round and asks for eight siblings' words of all of them before it
    // looks at any.
    __shared__ uint32_t wt_o[BM_THREADS / 64 + 1], wt_k[BM_THREADS / 64 + 1];
    __shared__ uint32_t sib_slice[BM_SIB];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t c = p.sub_cluster[blockIdx.x];
    const uint32_t i0 = p.cluster_item0[blockIdx.x], ni = p.cluster_nitems[blockIdx.x];
    if (ni < 2 || p.item_fused[i0] || p.cluster_overflow[c] || !ranks_by_bitmap(p.v_mode[c] & 3u, p.v_dense[c])) return;
    const uint32_t dense_words = (p.v_dense[c] + 31) >> 5;
    const size_t g0 = (size_t)p.item_scratch[i0] * DENSE_WORDS_BIG;
    uint32_t run_o = 0, run_k = 0;                       // ordinals / kept ordinals in the words before this round's
    constexpr uint32_t RW = BM_THREADS * BM_WPT;
    for (uint32_t w0 = 0; w0 < dense_words; w0 += RW) {
        // thread t owns words w0 + t * BM_WPT .. + BM_WPT - 1 (consecutive: one 32-byte piece of every sibling's array)
        const uint32_t wb = w0 + tid * BM_WPT;
        uint32_t o[BM_WPT], k[BM_WPT];
#pragma unroll
        for (uint32_t j = 0; j < BM_WPT; j++) { o[j] = 0; k[j] = 0; }
        for (uint32_t q0 = 0; q0 < ni; q0 += BM_SIB) {
            const uint32_t nq = min(ni - q0, BM_SIB);
            __syncthreads();
            if (tid < nq) sib_slice[tid] = p.item_scratch[i0 + q0 + tid];
            __syncthreads();
            for (uint32_t q = 0; q < nq; q += 8) {
                uint2 v[8][BM_WPT];
#pragma unroll
                for (uint32_t u = 0; u < 8; u++) {
                    const size_t g = (size_t)sib_slice[min(q + u, nq - 1)] * DENSE_WORDS_BIG;
#pragma unroll
                    for (uint32_t j = 0; j < BM_WPT; j++) v[u][j] = p.bm2[g + min(wb + j, dense_words - 1)];
                }
#pragma unroll
                for (uint32_t u = 0; u < 8; u++) {
#pragma unroll
                    for (uint32_t j = 0; j < BM_WPT; j++)
                        if (q + u < nq && wb + j < dense_words) { o[j] |= v[u][j].x; k[j] |= v[u][j].y; }
                }
            }
        }
        uint32_t so = 0, sk = 0;
#pragma unroll
        for (uint32_t j = 0; j < BM_WPT; j++) { so += __popc(o[j]); sk += __popc(k[j]); }
        uint32_t xo = so, xk = sk;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t yo = __shfl_up(xo, d), yk = __shfl_up(xk, d);
            if ((int)lane >= d) { xo += yo; xk += yk; }
        }
        __syncthreads();
        if (lane == 63) { wt_o[wave] = xo; wt_k[wave] = xk; }
        __syncthreads();
        uint32_t bo = run_o + xo - so, bk = run_k + xk - sk, to = 0, tk = 0;
        for (uint32_t v = 0; v < BM_THREADS / 64; v++) {
            if (v < wave) { bo += wt_o[v]; bk += wt_k[v]; }
            to += wt_o[v]; tk += wt_k[v];
        }
#pragma unroll
        for (uint32_t j = 0; j < BM_WPT; j++) {
            if (wb + j < dense_words) p.bm4[g0 + wb + j] = make_uint4(o[j], k[j], bo, bk);
            bo += __popc(o[j]); bk += __popc(k[j]);
        }
        run_o += to; run_k += tk;
    }
}

// ---------------------------------------------------------------------------------------------
// cluster_base_kernel: cluster_kmer_off / cluster_kmer_cnt for the clusters of one sub-batch
// ---------------------------------------------------------------------------------------------
struct BaseParams {
    const uint32_t* sub_cluster;       // [n] batch-local cluster ids of this sub-batch
    const uint32_t* cluster_item0;     // [n] first item (absolute) of the cluster in this pass
    const uint32_t* cluster_nitems;    // [n]
    const uint32_t* item_kept; const uint32_t* item_unique;
    const uint32_t* cluster_overflow;
    uint64_t* cluster_kmer_off; uint32_t* cluster_kmer_cnt; uint32_t* cluster_unique;
    uint64_t* cursor;                  // [0] next free output index  [1] total unique  [2] total kept
    uint32_t n;
};
__global__ __launch_bounds__(1024) void cluster_base_kernel(BaseParams p) {
    // (output room is CLAIMED -- one atomic add per 1 024 clusters -- not read and written back: the fused finish kernels of
    // the same sub-batch may be running beside this kernel on the context's second stream, and they claim theirs the same way)
    __shared__ uint32_t wave_tot[17];
    __shared__ uint64_t sh_base;
    const uint32_t tid = threadIdx.x;
    uint64_t uniq_sum = 0;
    for (uint32_t start = 0; start < p.n; start += 1024) {
        const uint32_t i = start + tid;
        uint32_t kept = 0, uniq = 0, c = 0;
        bool live = false;
        if (i < p.n) {
            c = p.sub_cluster[i];
            live = !p.cluster_overflow[c];
            if (live) {
                const uint32_t i0 = p.cluster_item0[i], ni = p.cluster_nitems[i];
                for (uint32_t q = 0; q < ni; q++) { kept += p.item_kept[i0 + q]; uniq += p.item_unique[i0 + q]; }
            }
        }
        uint32_t total;
        const uint32_t ex = block_exscan(kept, wave_tot, &total);
        if (tid == 0) {
            sh_base = atomicAdd((unsigned long long*)&p.cursor[0], (unsigned long long)total);
            if (total) atomicAdd((unsigned long long*)&p.cursor[2], (unsigned long long)total);
        }
        __syncthreads();
        if (i < p.n && live) {
            p.cluster_kmer_off[c] = sh_base + ex;
            p.cluster_kmer_cnt[c] = kept;
            p.cluster_unique[c] = uniq;
            uniq_sum += uniq;
        }
        __syncthreads();
    }
    // totals
    for (int d = 32; d > 0; d >>= 1) uniq_sum += __shfl_down(uniq_sum, d);
    if ((tid & 63) == 0 && uniq_sum) atomicAdd((unsigned long long*)&p.cursor[1], (unsigned long long)uniq_sum);
}

// ---------------------------------------------------------------------------------------------
// run-global pattern table
// ---------------------------------------------------------------------------------------------
struct PatternTable {
    uint64_t* lo;          // [cap] 64 bits of the row hash, EMPTY64 = free; claimed by CAS
    uint64_t* val;         // [cap] (32 more hash bits) << 32 | pattern id, EMPTY64 until published
    uint64_t* first_seen;  // [pool] by pattern id, atomicMin
    uint32_t* counters;    // [0] pattern ids asked for  [1] out of ids / table full  [2] output arena overflow
    uint64_t cap;          // power of two
    uint32_t pool;         // pattern ids available
};

// identity = (lo, hi31) = 95 bits of the 128-bit row hash (the top bit of hi32 is dropped so that a published
// `val` can never equal the EMPTY64 sentinel, whatever the pattern id)
//
// First half of an insert: find the pattern or claim a free slot WITHOUT allocating an id and without ever
// waiting.  Returns 0 = found (*pid), 1 = claimed (*slot is ours, val still unpublished), 2 = some other
// lane / workgroup has claimed a slot with our `lo` and not published it yet (look again later), 3 = the table is
// full or the run has already failed (counters[1]): *pid = PID_NONE.
constexpr uint32_t PID_NONE = 0xFFFFFFFFu;
__device__ __forceinline__ int pattern_find_or_claim(const PatternTable& t, uint64_t lo, uint32_t hi32,
                                                     uint64_t* slot_out, uint32_t* pid) {
    if (lo == EMPTY64) lo = EMPTY64 - 1;
    hi32 &= 0x7FFFFFFFu;
    uint64_t slot = (lo ^ ((uint64_t)hi32 * 0x9E3779B97F4A7C15ull)) & (t.cap - 1);
    for (uint64_t probes = 0; probes < t.cap; probes++) {
        uint64_t cur = __hip_atomic_load(&t.lo[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == EMPTY64) {
            cur = atomicCAS((unsigned long long*)&t.lo[slot], (unsigned long long)EMPTY64, (unsigned long long)lo);
            if (cur == EMPTY64) { *slot_out = slot; return 1; }
        }
        if (cur == lo) {
            const uint64_t v = __hip_atomic_load(&t.val[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v == EMPTY64) return 2;
            if ((uint32_t)(v >> 32) == hi32) { *pid = (uint32_t)v; return 0; }
        }
        slot = (slot + 1) & (t.cap - 1);
        // a full pool has been noticed: the batch is re-run with a larger table, nothing inserted now matters
        if ((probes & 63) == 63 && __hip_atomic_load(&t.counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    t.counters[1] = 1;
    *pid = PID_NONE;
    return 3;
}
__device__ __forceinline__ void pattern_publish(const PatternTable& t, uint64_t slot, uint32_t hi32, uint32_t pid) {
    __hip_atomic_store(&t.val[slot], ((uint64_t)(hi32 & 0x7FFFFFFFu) << 32) | pid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Blocking insert; atomicMin on first_seen; *lowered = this call lowered it (the caller then writes the pattern's
// row).  Claim -> publish is straight-line code inside a loop whose condition is WAVE-UNIFORM: a lane that met an
// unpublished claim (state 2) goes round again together with the whole wave, so the claimer -- even a lane of the
// same wave -- has published before the next look, whatever the compiler does with the divergent parts.
__device__ __forceinline__ uint32_t pattern_insert_lower(const PatternTable& t, uint64_t lo, uint32_t hi32,
                                                         uint64_t first_seen, bool* lowered) {
    uint32_t pid = PID_NONE;
    int st = 2;
    for (;;) {
        uint64_t slot = 0;
        if (st == 2) st = pattern_find_or_claim(t, lo, hi32, &slot, &pid);
        // ids for the lanes that claimed a slot: ONE add on the run-global counter per wave and trip (every new pattern of a
        // batch used to make its own atomic on that one address -- 5 M of them per 2 000 clusters of 150 alleles, which L2
        // serves one after the other)
        const uint64_t claimed = __ballot(st == 1);
        if (claimed) {
            const int leader = __ffsll((unsigned long long)claimed) - 1;
            const uint32_t lane = threadIdx.x & 63;
            uint32_t base = 0;
            if ((int)lane == leader) base = atomicAdd(&t.counters[0], (uint32_t)__popcll(claimed));
            base = __shfl(base, leader);
            if (st == 1) {
                const uint32_t id = base + (uint32_t)__popcll(claimed & ((1ull << lane) - 1ull));
                if (id >= t.pool) { t.counters[1] = 1; pid = PID_NONE; } else pid = id;
                pattern_publish(t, slot, hi32, pid);        // published even on failure: nobody waits for ever
                st = 0;
            }
        }
        if (!__any(st == 2)) break;
        __builtin_amdgcn_s_sleep(1);
    }
    *lowered = false;
    if (pid == PID_NONE) { t.counters[1] = 1; return pid; }
    const uint64_t old = atomicMin((unsigned long long*)&t.first_seen[pid], (unsigned long long)first_seen);
    *lowered = old > first_seen;
    return pid;
}
__device__ __forceinline__ uint32_t pattern_insert(const PatternTable& t, uint64_t lo, uint32_t hi32,
                                                   uint64_t first_seen) {
    bool lw;
    return pattern_insert_lower(t, lo, hi32, first_seen, &lw);
}
__device__ __forceinline__ uint32_t pattern_insert(const PatternTable& t, uint4 h, uint64_t first_seen) {
    return pattern_insert(t, ((uint64_t)h.x << 32) | h.y, h.z, first_seen);
}

// One insert per thread of a WORKGROUP (every thread calls it; `active` says who has a pattern), the ids of the patterns the
// workgroup claims taken with ONE add on the run-global counter: that word serves ~96 returning adds per microsecond whoever
// asks (MI355X_MICROARCH.md, "dequeue"; measured here: 5.3 M inserts by 316 000 waves, one add each, took 4.2 ms -- 0.9 ms with
// the add compiled out, profiles/r05/experiment_pattern_id_counter.txt).  lds: two words, zero, not touched by anyone else
// until the call returns; finish_kernel has the same steps inline.
__device__ __forceinline__ uint32_t pattern_insert_block(const PatternTable& t, bool active, uint64_t lo, uint32_t hi32,
                                                         uint64_t first_seen, uint32_t* lds) {
    uint64_t slot = 0;
    uint32_t pid = PID_NONE, newidx = 0;
    int st = -1;
    if (active) {
        st = pattern_find_or_claim(t, lo, hi32, &slot, &pid);
        if (st == 1) newidx = atomicAdd(&lds[0], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t nnew = lds[0];
        uint32_t base = 0;
        if (nnew) {
            base = atomicAdd(&t.counters[0], nnew);
            if ((uint64_t)base + nnew > t.pool) t.counters[1] = 1;
        }
        lds[1] = base;
    }
    __syncthreads();
    if (st == 1) {
        const uint32_t id = lds[1] + newidx;
        pid = id < t.pool ? id : PID_NONE;
        pattern_publish(t, slot, hi32, pid);            // published even on failure: nobody waits for ever
    }
    // EVERY claim of the workgroup is published before any of its lanes waits for somebody else's: left to itself the compiler
    // put the waiting lanes' loop in FRONT of the claiming lanes' store (the two branches are disjoint sets of lanes, their
    // order is its choice), and two waves of sibling items, each with a claim the other's lanes were waiting for, never came
    // back (tests/fuzz_parity.py 80 7, case 8).  A barrier is a point no lane's code moves across.
    __syncthreads();
    if (st == 2) return pattern_insert(t, lo, hi32, first_seen);     // somebody's claim, not yet published: the waiting form
    if (st == 0 || st == 1) {
        if (pid == PID_NONE) t.counters[1] = 1;
        else atomicMin((unsigned long long*)&t.first_seen[pid], (unsigned long long)first_seen);
    }
    return pid;
}

// pattern table of a larger capacity: re-insert the entries of the old one whose pattern id is < keep_below (the
// patterns of earlier batches; whatever a failed batch added is dropped)
struct RehashParams {
    const uint64_t* old_lo; const uint64_t* old_val; uint64_t old_cap;
    uint64_t* new_lo; uint64_t* new_val; uint64_t new_cap;
    uint32_t keep_below;
};
__global__ __launch_bounds__(256) void pattern_rehash_kernel(RehashParams p) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < p.old_cap; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t lo = p.old_lo[i], v = p.old_val[i];
        if (lo == EMPTY64 || v == EMPTY64 || (uint32_t)v >= p.keep_below) continue;
        const uint32_t hi32 = (uint32_t)(v >> 32);
        uint64_t slot = (lo ^ ((uint64_t)hi32 * 0x9E3779B97F4A7C15ull)) & (p.new_cap - 1);
        for (uint64_t probes = 0; probes < p.new_cap; probes++) {
            // entries are distinct: claiming a free slot is the whole insert
            if (atomicCAS((unsigned long long*)&p.new_lo[slot], (unsigned long long)EMPTY64, (unsigned long long)lo) == EMPTY64) {
                p.new_val[slot] = v;
                break;
            }
            slot = (slot + 1) & (p.new_cap - 1);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// emit_kernel
// ---------------------------------------------------------------------------------------------
struct EmitParams {
    const uint32_t* item_cluster; const uint32_t* item_scratch; const uint32_t* item_unique;
    const uint32_t* item_nslots;
    const uint32_t* item_sib0;       // [item] first item of the same cluster (absolute)
    const uint32_t* item_nsib;       // [item] items of that cluster
    const uint32_t* cluster_overflow;
    const uint32_t* v_mode; const uint32_t* v_dense;
    const uint32_t* cluster_nstrains; const uint32_t* cluster_npresab; const uint32_t* cluster_presab;
    const uint64_t* cluster_ordinal;
    const uint64_t* cluster_kmer_off;
    const uint64_t* tab_key; const uint32_t* tab_ord; const uint4* slot_hash;
 const uint64_t* sorted_pair; const uint32_t* kept_prefix;                                       // mode 0
    uint32_t* kept_prefix_rw;        // (the same array: with bitmaps its words 1.. take the kept k-mers' output indices)
    const uint4* bm4;                // ranks by bitmap: {occupied, kept, ordinals before, kept before} per ordinal word
    uint32_t* slot_out;              // [slice][NS] mode 1: index of the slot's k-mer inside the cluster's output
    uint64_t* out_key; uint32_t* out_pid; uint64_t* out_first;     // out_first: first_seen each k-mer offered
    uint32_t* cluster_pattern; uint64_t* cluster_first;
    PatternTable pt;
    uint64_t out_base;      // global index of out_key[0] (arena base)
    uint64_t out_cap;       // entries in the arena
    const uint32_t* work;   // [gridDim.x] item ids of this launch
    uint32_t W, NS, KW;
    uint32_t consider_missing, multiple_files;
};

__device__ __forceinline__ uint32_t pair_lower_bound(const uint64_t* sp, uint32_t n, uint32_t ord) {
    uint32_t a = 0, b = n;
    while (a < b) {
        uint32_t m = (a + b) >> 1;
        if ((uint32_t)(sp[m] >> 32) < ord) a = m + 1; else b = m;
    }
    return a;
}

constexpr uint32_t EMIT_THREADS = 1024;
// Per-item pattern table in LDS.  1024 slots = 28 KiB: two 1024-thread workgroups per CU instead of the one that 4096
// slots (114 KiB) allowed -- the kernel is a chain of dependent global loads per k-mer, and twice the waves hide twice
// the latency (2 000 clusters of ~140 related alleles: emit 3.33 -> 2.12 ms at 2048 slots, 1.68 at 1024, where the three
// passes over the table shrink too; 370 alleles: 11.4 -> 7.2 ms; tools/lt_exp.sh).  An item with more than LT_LIMIT
// distinct patterns sends the rest straight to the run-global table.
constexpr uint32_t LT_SLOTS = 1024;
constexpr uint32_t LT_LIMIT = LT_SLOTS / 4 * 3;

__global__ __launch_bounds__(EMIT_THREADS) void emit_kernel(EmitParams p) {
    // The item's patterns are first deduplicated in LDS (identity: the 128-bit row hash) so that the run-global
    // table sees one insert per distinct pattern of the item instead of one per k-mer.
    __shared__ uint64_t lt_lo[LT_SLOTS];
    __shared__ uint64_t lt_hi[LT_SLOTS];
    __shared__ uint64_t lt_first[LT_SLOTS];
    __shared__ uint32_t lt_pid[LT_SLOTS];
    __shared__ uint32_t lt_count;
    __shared__ uint32_t ins_cnt[2];

    PF_PROF_BEGIN();
    const uint32_t tid = threadIdx.x;
    const uint32_t item = p.work[blockIdx.x];
    const uint32_t c = p.item_cluster[item];
    if (p.cluster_overflow[c]) return;
    const uint32_t slice = p.item_scratch[item];
    const uint32_t NS = p.NS, W = p.W, KW = p.KW;
    const uint32_t U = p.item_unique[item];
    const uint32_t sib0 = p.item_sib0[item], nsib = p.item_nsib[item];
    const uint64_t ordinal = p.cluster_ordinal[c];
    const uint64_t obase = p.cluster_kmer_off[c] - p.out_base;
    const uint32_t mode = p.v_mode[c] & 3u;
    const bool sorted = !ranks_by_bitmap(mode, p.v_dense[c]);    // the item's k-mers sorted by ordinal (rows_kernel), or
                                                                 // ordinal bitmaps

    if (item == sib0 && tid == 0) {
        // the cluster's own row: md5 of the int64 image of clusterpresab (panfeed.py:175-187)
        const uint32_t npres = p.cluster_npresab[c];
        const uint32_t nw = (npres + 31) >> 5;
        const uint32_t* presab = p.cluster_presab + (size_t)c * W;
        H128 s;
        s.h1 = 0x9747b28cu ^ npres; s.h2 = 0x1b873593u ^ 0x5bd1e995u; s.h3 = 0xe6546b64u; s.h4 = 0x85ebca6bu;
        if (p.multiple_files) { s.h2 ^= (uint32_t)ordinal; s.h3 ^= (uint32_t)(ordinal >> 32); }
        for (uint32_t w = 0; w < nw; w += 4) {
            uint32_t wv[4];
            for (int j = 0; j < 4; j++) wv[j] = (w + j < nw) ? presab[w + j] : 0;
            mm3_block(s, wv[0], wv[1], wv[2], wv[3]);
        }
        mm3_final(s, nw * 4 + 1);
        const uint64_t fs = ordinal << 32;
        p.cluster_pattern[c] = pattern_insert(p.pt, make_uint4(s.h1, s.h2, s.h3, s.h4), fs);
        p.cluster_first[c] = fs;
    }

    const uint32_t ns = p.item_nslots[item];
    const size_t gb0 = (size_t)p.item_scratch[sib0] * DENSE_WORDS_BIG;
    const uint64_t* sp = p.sorted_pair + (size_t)slice * NS;
    const uint32_t* kp = p.kept_prefix + (size_t)slice * (NS + 1);
    uint32_t* sout = p.slot_out + (size_t)slice * NS;    // per entry (slot / sorted position): index of its k-mer in
                                                         // the cluster's output, NONE when it is not kept
    // entries: the item's k-mers in ordinal order (sorted), or its KEPT k-mers in any order (bitmaps: rows_kernel left
    // their (ordinal, slot) pairs in sorted_pair and their number in kept_prefix[0]; kept_prefix[1 + i] takes entry i's
    // output index here)
    const uint32_t n_entries = sorted ? U : min(kp[0], ns);
    uint32_t* kres = p.kept_prefix_rw + (size_t)slice * (NS + 1) + 1;
    for (uint32_t i = tid; i < LT_SLOTS; i += EMIT_THREADS) { lt_lo[i] = EMPTY64; lt_hi[i] = EMPTY64; lt_first[i] = EMPTY64; }
    if (tid == 0) { lt_count = 0; ins_cnt[0] = 0; }
    // (ranks by bitmap: nobody reads a slot's output index -- pattern_rows_kernel and pass 3 go by entry, kres[] -- so the
    // array is not initialised slot by slot (38 KB of stores per item) nor written per kept k-mer; its first n_entries words
    // take, per entry, where pass 1 found the k-mer's pattern in the local table, so that pass 3 need not read the row hash
    // and probe again)
    constexpr uint32_t LS_NONE = 0xFFFFFFFFu;
    __syncthreads();
    PF_PROF_STAMP(32);
    auto row_id = [&](uint32_t slot, uint64_t& lo, uint64_t& hi) {
        const uint4 h = p.slot_hash[(size_t)slice * NS + slot];
        lo = ((uint64_t)h.x << 32) | h.y; hi = ((uint64_t)h.z << 32) | h.w;
        if (lo == EMPTY64) lo--;
        if (hi == EMPTY64) hi--;
    };
    // pass 1: rank, output index, item-local pattern table
    for (uint32_t i0 = 0; i0 < n_entries; i0 += EMIT_THREADS) {
        const uint32_t i = i0 + tid;
        uint32_t res = 0xFFFFFFFFu, slot = 0;
        uint64_t fs = 0, lo = 0, hi = 0;
#ifdef PF_PROF
        const uint64_t tp0 = __builtin_readcyclecounter();
#endif
        if (i < n_entries) {
            if (sorted) {
                const uint32_t kb = kp[i];
                if (kp[i + 1] != kb) {
                    const uint64_t pr = sp[i];
                    const uint32_t ord = (uint32_t)(pr >> 32);
                    slot = (uint32_t)pr;
                    uint32_t rank = i, kept_before = kb;
                    for (uint32_t q = 0; q < nsib; q++) {
                        const uint32_t it = sib0 + q;
                        if (it == item) continue;
                        const uint32_t sl = p.item_scratch[it];
                        const uint32_t lb = pair_lower_bound(p.sorted_pair + (size_t)sl * NS, p.item_unique[it], ord);
                        rank += lb;
                        kept_before += p.kept_prefix[(size_t)sl * (NS + 1) + lb];
                    }
                    res = kept_before;
                    fs = (ordinal << 32) | (uint64_t)(rank + 1);
                }
                sout[i] = res;
            } else {
                const uint64_t pr = sp[i];
                const uint32_t o = (uint32_t)(pr >> 32);
                slot = (uint32_t)pr;
                const uint32_t below = (1u << (o & 31)) - 1;
                // the cluster's bitmaps: this item's own, or (several items) their union, which
                // bitmap_merge_kernel left in the first item's place -- four loads that go out together
                const size_t g2 = gb0 + (o >> 5);
                const uint4 rec = p.bm4[g2];
                const uint32_t bo = rec.x, kw = rec.y, po = rec.z, pk = rec.w;
                row_id(slot, lo, hi);                          // (and the row hash with them)
                res = pk + __popc(kw & below);
                fs = (ordinal << 32) | (uint64_t)(po + __popc(bo & below) + 1);
                kres[i] = res;
            }
        }
        // local table: find or claim; an entry whose second word is not published yet is looked at again in the next
        // round of a WAVE-UNIFORM loop (no lane ever spins inside divergent code)
#ifdef PF_PROF
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t tp1 = __builtin_readcyclecounter();
        if (tid == 0) atomicAdd(&pf_prof[62], (unsigned long long)(tp1 - tp0));
#endif
        int st = res != 0xFFFFFFFFu ? 2 : 0;
        uint32_t ls = 0;
        if (st == 2) {
            const uint64_t o_idx = obase + res;
            if (o_idx < p.out_cap) p.out_first[o_idx] = fs;
            if (sorted) row_id(slot, lo, hi);
            ls = (uint32_t)(lo ^ (hi >> 7)) & (LT_SLOTS - 1);
        }
        for (;;) {
            if (st == 2) {
                st = 3;                                       // table full: straight to the run-global table in pass 3
                for (uint32_t probes = 0; probes < LT_SLOTS; probes++) {
                    uint64_t cur = __hip_atomic_load(&lt_lo[ls], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (cur == EMPTY64) {
                        if (__hip_atomic_load(&lt_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= LT_LIMIT) break;
                        cur = atomicCAS((unsigned long long*)&lt_lo[ls], (unsigned long long)EMPTY64, (unsigned long long)lo);
                        if (cur == EMPTY64) {
                            __hip_atomic_store(&lt_hi[ls], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            atomicAdd(&lt_count, 1u);
                            st = 1;
                            break;
                        }
                    }
                    if (cur == lo) {
                        const uint64_t h2 = __hip_atomic_load(&lt_hi[ls], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (h2 == EMPTY64) { st = 2; break; }      // claimed, not yet published
                        if (h2 == hi) { st = 1; break; }
                    }
                    ls = (ls + 1) & (LT_SLOTS - 1);
                }
            }
            if (!__any(st == 2)) break;
        }
        if (st == 1) atomicMin((unsigned long long*)&lt_first[ls], (unsigned long long)fs);
        if (!sorted && i < n_entries) sout[i] = st == 1 ? ls : LS_NONE;
#ifdef PF_PROF
        if (tid == 0) { atomicAdd(&pf_prof[63], (unsigned long long)(__builtin_readcyclecounter() - tp1)); atomicAdd(&pf_prof[38], 1ull); }
#endif
    }
    __syncthreads();
    PF_PROF_STAMP(33);
    // pass 2: one insert into the run-global table per distinct pattern of this item, the ids of the new ones from ONE add
    // per item (pattern_insert_block: with an add per wave this pass was 2 of the kernel's 4.4 ms at ~150 SURVEY alleles)
    static_assert(LT_SLOTS == EMIT_THREADS, "one place of the local table per thread");
    {
        const uint64_t lo = lt_lo[tid];
        const uint32_t pid = pattern_insert_block(p.pt, lo != EMPTY64, lo, (uint32_t)(lt_hi[tid] >> 32), lt_first[tid], ins_cnt);
        if (lo != EMPTY64) lt_pid[tid] = pid;
    }
    __syncthreads();
    PF_PROF_STAMP(34);
    // pass 3: pattern id per kept k-mer, outputs
    for (uint32_t i = tid; i < n_entries; i += EMIT_THREADS) {
        const uint32_t kb = sorted ? sout[i] : kres[i];
        if (kb == 0xFFFFFFFFu) continue;
        const uint64_t o_idx = obase + kb;
        if (o_idx >= p.out_cap) { p.pt.counters[2] = 1; continue; }   // cannot happen: the arena holds every item's limit
        const uint32_t slot = (uint32_t)sp[i];
        uint32_t pid = 0xFFFFFFFFu;
        bool found = false;
        uint64_t lo = 0, hi = 0;
        if (!sorted) {
            const uint32_t lsv = sout[i];                      // pass 1's place in the local table
            if (lsv != LS_NONE) { pid = lt_pid[lsv]; found = true; }
            else row_id(slot, lo, hi);
        } else {
            row_id(slot, lo, hi);
            uint32_t ls = (uint32_t)(lo ^ (hi >> 7)) & (LT_SLOTS - 1);
            for (uint32_t probes = 0; probes < LT_SLOTS; probes++) {
                const uint64_t cur = lt_lo[ls];
                if (cur == EMPTY64) break;
                if (cur == lo && lt_hi[ls] == hi) { pid = lt_pid[ls]; found = true; break; }
                ls = (ls + 1) & (LT_SLOTS - 1);
            }
        }
        // not in the local table (it was full): straight to the run-global table
        if (!found) pid = pattern_insert(p.pt, lo, (uint32_t)(hi >> 32), p.out_first[o_idx]);
        p.out_key[o_idx * KW] = p.tab_key[((size_t)slice * KW) * NS + slot];
        for (uint32_t j = 1; j < KW; j++) p.out_key[o_idx * KW + j] = p.tab_key[((size_t)slice * KW + j) * NS + slot];
        p.out_pid[o_idx] = pid;
    }
#ifdef PF_PROF
    __syncthreads();
    PF_PROF_STAMP(35);
    if (threadIdx.x == 0) { atomicAdd(&pf_prof[36], 1ull); atomicAdd(&pf_prof[37], (unsigned long long)lt_count); }
#endif
}

// ---------------------------------------------------------------------------------------------
// finish_kernel: rows + emit in one workgroup for a deduplicated cluster that is a single work item
// ---------------------------------------------------------------------------------------------
// (mode 1, one key partition, no slow-path rows, dense ordinal space <= FUSED_DENSE_WORDS*32.)  Everything
// between the scan's table dump and the outputs happens in LDS: sample sets M, the table of distinct allele
// masks with their row hash / keep flag, ordinal bitmaps for ranks, one insert into the run-global pattern
// table per distinct mask, and the pattern row is written by whoever lowers the pattern's first_seen (all
// writers of one pattern write identical bytes).
// Two size classes: the small one (most clusters) keeps four 512-thread workgroups resident per CU, which is
// what hides the latency of the global atomics and dependent loads this kernel is made of.
struct FinSmall { static constexpr uint32_t THREADS = 512, DW = 1024, MR = 1024, AT = 256, ATL = 192; typedef uint8_t tag_t; };
struct FinLarge { static constexpr uint32_t THREADS = 1024, DW = 2048, MR = 2048, AT = 512, ATL = 384; typedef uint16_t tag_t; };
// The large class in its MULTI form (several key partitions) has no per-slot tags: the LDS they would take holds a mask table
// of twice the size -- 70 KiB, still two workgroups per CU.  (Alleles that each carry their own substitutions give a cluster
// about as many distinct masks as the gene has windows; every mask past the table is evaluated slot by slot.)
struct FinLargeM { static constexpr uint32_t THREADS = 1024, DW = 2048, MR = 2048, AT = 1024, ATL = 768; typedef uint16_t tag_t; };
// A third class for clusters whose distinct sequences carry up to 131 071 windows (60 sequences of 1 100 bases + flanks, or
// 30 of 2 500): always in the MULTI form (no per-slot tags), 75 KiB, two workgroups per CU.  The 16-bit prefix counts are
// kept: those of the upper half of the bitmap are relative to its first word (sh_half_*).
struct FinHuge { static constexpr uint32_t THREADS = 1024, DW = 4096, MR = 2048, AT = 512, ATL = 384; typedef uint16_t tag_t; };
constexpr uint32_t FUSED_DENSE_WORDS = FinHuge::DW;    // dense ordinals of a fused cluster / 32
constexpr uint32_t FUSED_MROWS = FinLarge::MR;         // D * ceil4(W) words of M
constexpr uint32_t FUSED_MAX_EXTRA = 1024;             // slow-path rows of a cluster the fused kernel takes along
static_assert(nslots_max(1) <= 9600, "slot_at too small");

struct FinishParams {
    const uint32_t* work;            // [gridDim.x] item ids
    const uint32_t* item_cluster; const uint32_t* item_nslots; const uint32_t* item_scratch;
    const uint32_t* item_nparts;     // key partitions of the item's cluster (MULTI: items item .. item+nparts-1)
    const uint32_t* cluster_overflow;
    const uint32_t* cluster_seg_off; const uint32_t* seg_sample; const uint32_t* seg_distinct;
    const uint32_t* v_nstr; const uint32_t* v_dense;
    const uint32_t* cluster_nstrains; const uint32_t* cluster_npresab; const uint32_t* cluster_presab;
    const uint64_t* cluster_ordinal;
    const uint32_t* maf_lo; const uint32_t* maf_hi;
    const uint32_t* item_count;      // entries of the item's compact table
    const uint64_t* tab_key; const uint32_t* tab_ord; const uint32_t* cmask_lo; const uint32_t* cmask_hi;
    // slow-path rows of the cluster (k-mers with a non-ACGT base, grouped by the caller): CSR per cluster, their
    // ordinals in the cluster's dense numbering (cluster_dedup_kernel), their presence rows
    const uint32_t* extra_off; const uint32_t* extra_dense; const uint32_t* extra_bits;
    uint64_t* out_key; uint32_t* out_pid;
    uint64_t* cluster_kmer_off; uint32_t* cluster_kmer_cnt; uint32_t* cluster_unique; uint32_t* cluster_pattern;
    uint64_t* cursor;                // [0] next free output index [1] unique total [2] kept total
    PatternTable pt;
    uint32_t* pat_bits; uint32_t* pat_nan; uint32_t* pat_n;
    uint64_t out_base, out_cap;
    uint32_t W, NS, KW;
    uint32_t consider_missing, patfilt, multiple_files;
};

// MULTI: the work item is the first of several key partitions of the cluster; the slot loops run over all of
// them (the mask table, ordinal bitmaps and M are per cluster anyway) and slot tags are looked up again instead
// of being kept per slot.
// (finish_kernel's stamps are off unless -DPF_PROF_FINISH is given too: all together they make the kernel fault on batches of
// thousands of clusters -- round 5; the stamps index nothing, what changes is the register allocation of a kernel that spills
// 110 - 175 SGPRs -- and took the whole profiling build with them.  In two halves they work: -DPF_PROF_FINISH_MASK=0x0F, then
// =0xF0 (profiles/r05/phase_profile_finish_headline.txt); the phases also have knock-out builds, PF_KO_FINISH)
#if defined(PF_PROF) && !defined(PF_PROF_FINISH)
#undef PF_PROF_BEGIN
#undef PF_PROF_STAMP
#define PF_PROF_BEGIN() do { } while (0)
#define PF_PROF_STAMP(k) do { } while (0)
#elif defined(PF_PROF) && defined(PF_PROF_FINISH_MASK)
#undef PF_PROF_STAMP
#define PF_PROF_STAMP(k) do { if (((PF_PROF_FINISH_MASK >> (k)) & 1) && threadIdx.x == 0) { const uint64_t n_ = __builtin_readcyclecounter(); \
    atomicAdd(&pf_prof[k], (unsigned long long)(n_ - prof_t_)); prof_t_ = n_; } } while (0)
#endif
template <class CFG, bool MULTI>
__global__ __launch_bounds__(CFG::THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void finish_kernel(FinishParams p) {
    constexpr uint32_t T = CFG::THREADS, DW = CFG::DW, AT = CFG::AT;
    __shared__ __align__(16) uint32_t M[CFG::MR];
    __shared__ uint32_t occ[DW], keepbm[DW];
    __shared__ uint16_t pocc[DW], pkeep[DW];                     // exclusive prefix popcounts (< 65536)
    __shared__ uint64_t at_key[AT];
    __shared__ uint4 at_hash[AT];
    __shared__ uint32_t at_keep[AT], at_minord[AT], at_pid[AT];
    typedef typename CFG::tag_t tag_t;
    constexpr uint32_t PROBE = AT - 1;                           // table positions 0..AT-2; AT-1 = "not in the table"
    constexpr tag_t UNTABLED = (tag_t)(AT - 1);
    __shared__ tag_t slot_at[MULTI ? 1 : 9600];                  // per occupied slot: its mask's table position
    __shared__ uint32_t wave_tot[T / 64 + 1];
    __shared__ uint32_t sh_npres, at_count, wl_count;
    __shared__ uint32_t sh_half_o, sh_half_k;                    // DW > 2048: the prefix counts at word 2048
    __shared__ uint64_t sh_base;
    static_assert(DW <= 4096 && (DW <= 2048 || (DW / T) * (T / 2) == 2048), "prefix counts are 16-bit per half");

    const uint32_t tid = threadIdx.x;
    PF_PROF_BEGIN();
    const uint32_t item = p.work[blockIdx.x];
    const uint32_t c = p.item_cluster[item];
    if (p.cluster_overflow[c]) return;
    const uint32_t NS = p.NS, W = p.W, KW = p.KW;
    const uint32_t nparts = MULTI ? p.item_nparts[item] : 1;
    const uint32_t nstr = p.cluster_nstrains[c], npres = p.cluster_npresab[c];
    const uint32_t nchunks = (nstr + 31) >> 5;
    const uint32_t Wp = (W + 3) & ~3u;
    const uint32_t* presab = p.cluster_presab + (size_t)c * W;
    const uint32_t dense_words = (p.v_dense[c] + 31) >> 5;      // <= DW (host-checked)
    const uint64_t ordinal = p.cluster_ordinal[c];
    // per key partition q: scratch slice and its compact table (occupied entries only, any order):
    // ordinals, 64-bit allele masks (which distinct sequences contain the k-mer), keys
    uint32_t slice = 0, ns = 0;
    const uint32_t* ordp = nullptr;
    const uint32_t* mlo = nullptr;
    const uint32_t* mhi = nullptr;
    const bool has_hi = p.v_nstr[c] > 32;                       // more than 32 distinct sequences: the masks' upper word exists
    auto set_part = [&](uint32_t q) {
        slice = p.item_scratch[item + q];
        ns = p.item_count[item + q];
        ordp = p.tab_ord + (size_t)slice * NS;
        mlo = p.cmask_lo + (size_t)slice * NS;
        mhi = p.cmask_hi + (size_t)slice * NS;
    };
    auto find_tag = [&](uint64_t amask) -> tag_t {
        uint32_t a = (uint32_t)mix64(amask) % PROBE;
        for (uint32_t probes = 0; amask && probes < PROBE; probes++) {
            const uint64_t cur = at_key[a];
            if (cur == 0) break;
            if (cur == amask) return (tag_t)a;
            a = a + 1 == PROBE ? 0 : a + 1;
        }
        return UNTABLED;
    };

    for (uint32_t i = tid; i < CFG::MR; i += T) M[i] = 0;
    for (uint32_t i = tid; i < DW; i += T) { occ[i] = 0; keepbm[i] = 0; }
    for (uint32_t i = tid; i < AT; i += T) { at_key[i] = 0; at_minord[i] = NO_ORD; }
    if (tid == 0) {
        at_count = 0; wl_count = 0;
        uint32_t np = 0;
        for (uint32_t w = 0; w < W; w++) np += __popc(presab[w]);
        sh_npres = np;
    }
    __syncthreads();
    {   // M[d] = samples that carry distinct sequence d
        const uint32_t s0 = p.cluster_seg_off[c], s1 = p.cluster_seg_off[c + 1];
        for (uint32_t s = s0 + tid; s < s1; s += T) {
            const uint32_t d = p.seg_distinct[s], smp = p.seg_sample[s];
            atomicOr(&M[d * Wp + (smp >> 5)], 1u << (smp & 31));
        }
    }
    PF_PROF_STAMP(0);
    PF_KO_FINISH_AT(0);
    // phase A: distinct allele masks
    for (uint32_t q = 0; q < nparts; q++) {
    set_part(q);
    // four slots per thread and trip: their loads are issued together (the loop is latency-, not bandwidth-bound)
    for (uint32_t i0 = tid; i0 < ns; i0 += 4 * T) {
      uint32_t ml_[4], mh_[4];
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) {
          const uint32_t i = i0 + u * T;
          ml_[u] = i < ns ? mlo[i] : 0; mh_[u] = (has_hi && i < ns) ? mhi[i] : 0;
      }
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) {
        const uint32_t i = i0 + u * T;
        if (i >= ns) break;
        const uint64_t amask = (uint64_t)ml_[u] | ((uint64_t)mh_[u] << 32);
        tag_t tag = UNTABLED;
        {
            uint32_t a = (uint32_t)mix64(amask) % PROBE;
            for (uint32_t probes = 0; amask && probes < PROBE; probes++) {
                uint64_t cur = __hip_atomic_load(&at_key[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cur == 0) {
                    if (__hip_atomic_load(&at_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= CFG::ATL) break;
                    cur = atomicCAS((unsigned long long*)&at_key[a], 0ull, (unsigned long long)amask);
                    if (cur == 0) { atomicAdd(&at_count, 1u); cur = amask; }
                }
                if (cur == amask) { tag = (tag_t)a; break; }
                a = a + 1 == PROBE ? 0 : a + 1;
            }
        }
        if (!MULTI) slot_at[i] = tag;
      }
    }
    }
    __syncthreads();
    PF_PROF_STAMP(1);
    PF_KO_FINISH_AT(1);
    const uint32_t npresent = sh_npres;
    const uint32_t n_eff = p.consider_missing ? npresent : nstr;               // panfeed.py:191 / :196
    const uint32_t lo = p.maf_lo[n_eff], hi = p.maf_hi[n_eff];
    const bool same_possible = !p.patfilt && nstr == npres && (!p.consider_missing || npresent == nstr);

    auto row_word4 = [&](uint64_t amask, uint32_t ch, uint32_t wv[4]) {
        wv[0] = wv[1] = wv[2] = wv[3] = 0;
        uint64_t t = amask;
        while (t) {
            const uint32_t d = __ffsll((unsigned long long)t) - 1;
            t &= t - 1;
            const uint4 m = *reinterpret_cast<const uint4*>(&M[d * Wp + ch]);
            wv[0] |= m.x; wv[1] |= m.y; wv[2] |= m.z; wv[3] |= m.w;
        }
    };
    auto row_eval = [&](uint64_t amask, uint4& hout) -> bool {
        H128 s;
        s.h1 = 0x9747b28cu ^ nstr; s.h2 = 0x1b873593u; s.h3 = 0xe6546b64u; s.h4 = 0x85ebca6bu;
        if (p.multiple_files) { s.h2 ^= (uint32_t)ordinal; s.h3 ^= (uint32_t)(ordinal >> 32); }
        uint32_t cnt = 0;
        bool eq = true;
        for (uint32_t ch = 0; ch < nchunks; ch += 4) {
            uint32_t wv[4];
            row_word4(amask, ch, wv);
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (ch + j < nchunks) { cnt += __popc(wv[j]); if (same_possible) eq = eq && (wv[j] == presab[ch + j]); }
            mm3_block(s, wv[0], wv[1], wv[2], wv[3]);
        }
        if (p.consider_missing) {
            for (uint32_t ch = 0; ch < nchunks; ch += 4) {
                uint32_t wv[4];
#pragma unroll
                for (int j = 0; j < 4; j++) wv[j] = (ch + j < nchunks) ? ~presab[ch + j] : 0;
                mm3_block(s, wv[0], wv[1], wv[2], wv[3]);
            }
        }
        mm3_final(s, nchunks * 4);
        bool keep = cnt >= lo && cnt <= hi;                 // panfeed.py:197-200
        if (same_possible && eq) keep = false;              // panfeed.py:202-204
        hout = make_uint4(s.h1, s.h2, s.h3, s.h4);
        return keep;
    };
    // the same for a row given word by word (the cluster's slow-path rows)
    auto row_eval_words = [&](const uint32_t* words, uint4& hout) -> bool {
        H128 s;
        s.h1 = 0x9747b28cu ^ nstr; s.h2 = 0x1b873593u; s.h3 = 0xe6546b64u; s.h4 = 0x85ebca6bu;
        if (p.multiple_files) { s.h2 ^= (uint32_t)ordinal; s.h3 ^= (uint32_t)(ordinal >> 32); }
        uint32_t cnt = 0;
        bool eq = true;
        for (uint32_t ch = 0; ch < nchunks; ch += 4) {
            uint32_t wv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                wv[j] = ch + j < nchunks ? words[ch + j] : 0;
                if (ch + j < nchunks) { cnt += __popc(wv[j]); if (same_possible) eq = eq && (wv[j] == presab[ch + j]); }
            }
            mm3_block(s, wv[0], wv[1], wv[2], wv[3]);
        }
        if (p.consider_missing) {
            for (uint32_t ch = 0; ch < nchunks; ch += 4) {
                uint32_t wv[4];
#pragma unroll
                for (int j = 0; j < 4; j++) wv[j] = (ch + j < nchunks) ? ~presab[ch + j] : 0;
                mm3_block(s, wv[0], wv[1], wv[2], wv[3]);
            }
        }
        mm3_final(s, nchunks * 4);
        bool keep = cnt >= lo && cnt <= hi;                 // panfeed.py:197-200
        if (same_possible && eq) keep = false;              // panfeed.py:202-204
        hout = make_uint4(s.h1, s.h2, s.h3, s.h4);
        return keep;
    };
    const uint32_t ex0 = p.extra_off ? p.extra_off[c] : 0, ex1 = p.extra_off ? p.extra_off[c + 1] : 0;
    auto write_row = [&](uint32_t pid, uint64_t amask) {
        for (uint32_t w = 0; w < W; w += 4) {
            uint32_t wv[4] = {0, 0, 0, 0};
            if (w < nchunks) row_word4(amask, w, wv);
            for (uint32_t j = 0; j < 4 && w + j < W; j++) {
                p.pat_bits[(size_t)pid * W + w + j] = (w + j < nchunks) ? wv[j] : 0;
                if (p.pat_nan) {
                    uint32_t nn = 0;
                    if (p.consider_missing && w + j < nchunks) {
                        nn = ~presab[w + j];
                        const uint32_t rem = nstr - ((w + j) << 5);
                        if (rem < 32) nn &= (1u << rem) - 1;
                    }
                    p.pat_nan[(size_t)pid * W + w + j] = nn;
                }
            }
        }
        p.pat_n[pid] = nstr;
    };

    // the same two, eight lanes per mask: lane `sub` gathers words [32 r + 4 sub, +4) of the row; the row hash is
    // sequential, so all eight run it on the shuffled words; popcount / equality are reduced over the eight lanes
    auto row_eval8 = [&](uint64_t amask, uint32_t sub, uint4& hout) -> bool {
        H128 s;
        s.h1 = 0x9747b28cu ^ nstr; s.h2 = 0x1b873593u; s.h3 = 0xe6546b64u; s.h4 = 0x85ebca6bu;
        if (p.multiple_files) { s.h2 ^= (uint32_t)ordinal; s.h3 ^= (uint32_t)(ordinal >> 32); }
        uint32_t cnt = 0;
        bool eq = true;
        const uint32_t gbase = (tid & 63u) & ~7u;
        for (uint32_t ch0 = 0; ch0 < nchunks; ch0 += 32) {
            const uint32_t ch = ch0 + 4 * sub;
            uint32_t wv[4] = {0, 0, 0, 0};
            if (ch < nchunks) {
                row_word4(amask, ch, wv);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (ch + j < nchunks) { cnt += __popc(wv[j]); if (same_possible) eq = eq && (wv[j] == presab[ch + j]); }
            }
            const uint32_t nb = min(8u, (nchunks - ch0 + 3) >> 2);
            for (uint32_t j = 0; j < nb; j++) {
                const uint32_t b0 = __shfl(wv[0], gbase + j), b1 = __shfl(wv[1], gbase + j);
                const uint32_t b2 = __shfl(wv[2], gbase + j), b3 = __shfl(wv[3], gbase + j);
                mm3_block(s, b0, b1, b2, b3);
            }
        }
        if (p.consider_missing) {
            for (uint32_t ch = 0; ch < nchunks; ch += 4) {
                uint32_t wv[4];
#pragma unroll
                for (int j = 0; j < 4; j++) wv[j] = (ch + j < nchunks) ? ~presab[ch + j] : 0;
                mm3_block(s, wv[0], wv[1], wv[2], wv[3]);
            }
        }
        mm3_final(s, nchunks * 4);
        int ne = eq ? 0 : 1;
        for (int d = 1; d < 8; d <<= 1) { cnt += __shfl_xor(cnt, d); ne |= __shfl_xor(ne, d); }
        bool keep = cnt >= lo && cnt <= hi;                 // panfeed.py:197-200
        if (same_possible && !ne) keep = false;             // panfeed.py:202-204
        hout = make_uint4(s.h1, s.h2, s.h3, s.h4);
        return keep;
    };
    auto write_row8 = [&](uint32_t pid, uint64_t amask, uint32_t sub) {
        for (uint32_t w0 = 0; w0 < W; w0 += 32) {
            const uint32_t w = w0 + 4 * sub;
            if (w >= W) break;
            uint32_t wv[4] = {0, 0, 0, 0};
            if (w < nchunks) row_word4(amask, w, wv);
            for (uint32_t j = 0; j < 4 && w + j < W; j++) {
                p.pat_bits[(size_t)pid * W + w + j] = (w + j < nchunks) ? wv[j] : 0;
                if (p.pat_nan) {
                    uint32_t nn = 0;
                    if (p.consider_missing && w + j < nchunks) {
                        nn = ~presab[w + j];
                        const uint32_t rem = nstr - ((w + j) << 5);
                        if (rem < 32) nn &= (1u << rem) - 1;
                    }
                    p.pat_nan[(size_t)pid * W + w + j] = nn;
                }
            }
        }
        if (sub == 0) p.pat_n[pid] = nstr;
    };

    // phase B: one row evaluation per distinct mask (eight lanes each)
    for (uint32_t t = tid >> 3; t < AT; t += T >> 3) {
        const uint64_t key = at_key[t];
        if (!key) continue;
        uint4 h;
        const bool kp = row_eval8(key, tid & 7u, h);
        if ((tid & 7u) == 0) { at_keep[t] = kp ? 1u : 0u; at_hash[t] = h; }
    }
    __syncthreads();
    PF_PROF_STAMP(2);
    PF_KO_FINISH_AT(2);
    // phase C: ordinal bitmaps, lowest ordinal per mask.  Slots whose mask did not fit the table (rare) are left to
    // a second, plain loop so that the batched one stays small.
    bool saw_untabled = false;
    for (uint32_t q = 0; q < nparts; q++) {
    set_part(q);
    for (uint32_t i0 = tid; i0 < ns; i0 += 4 * T) {
      uint32_t o_[4], ml_[4], mh_[4];
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) {
          const uint32_t i = i0 + u * T;
          o_[u] = i < ns ? ordp[i] : 0xFFFFFFFFu;
          if (MULTI) { ml_[u] = i < ns ? mlo[i] : 0; mh_[u] = (has_hi && i < ns) ? mhi[i] : 0; }
      }
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) {
        const uint32_t i = i0 + u * T;
        const uint32_t o = o_[u];
        if (i >= ns || (o >> 5) >= dense_words) continue;
        const tag_t tag = MULTI ? find_tag((uint64_t)ml_[u] | ((uint64_t)mh_[u] << 32)) : slot_at[i];
        atomicOr(&occ[o >> 5], 1u << (o & 31));
        if (tag == UNTABLED) { saw_untabled = true; continue; }
        if (at_keep[tag] != 0) {
            atomicMin(&at_minord[tag], o);
            atomicOr(&keepbm[o >> 5], 1u << (o & 31));
        }
      }
    }
    }
    if (saw_untabled) {
        for (uint32_t q = 0; q < nparts; q++) {
            set_part(q);
#pragma unroll 1
            for (uint32_t i = tid; i < ns; i += T) {
                const uint32_t o = ordp[i];
                if ((o >> 5) >= dense_words) continue;
                const uint64_t amask = (uint64_t)mlo[i] | ((uint64_t)(has_hi ? mhi[i] : 0u) << 32);
                if ((MULTI ? find_tag(amask) : slot_at[i]) != UNTABLED) continue;
                uint4 h;
                if (row_eval(amask, h)) atomicOr(&keepbm[o >> 5], 1u << (o & 31));
            }
        }
    }
    // the cluster's slow-path rows: one ordinal each, kept or not by their own row
#pragma unroll 1
    for (uint32_t e = ex0 + tid; e < ex1; e += T) {
        const uint32_t o = p.extra_dense[e];
        if ((o >> 5) >= dense_words) continue;
        atomicOr(&occ[o >> 5], 1u << (o & 31));
        uint4 h;
        if (row_eval_words(p.extra_bits + (size_t)e * W, h)) atomicOr(&keepbm[o >> 5], 1u << (o & 31));
    }
    __syncthreads();
    PF_PROF_STAMP(3);
    PF_KO_FINISH_AT(3);
    // prefix popcounts over the bitmap words
    uint32_t tot_o, tot_k;
    {
        constexpr uint32_t PW = DW / T;   // 2
        uint32_t so = 0, sk = 0;
#pragma unroll
        for (uint32_t j = 0; j < PW; j++) { so += __popc(occ[tid * PW + j]); sk += __popc(keepbm[tid * PW + j]); }
        uint32_t bo = block_exscan(so, wave_tot, &tot_o);
        uint32_t bk = block_exscan(sk, wave_tot, &tot_k);
        if (DW > 2048) {
            if (tid * PW == 2048) { sh_half_o = bo; sh_half_k = bk; }
            __syncthreads();
            if (tid * PW >= 2048) { bo -= sh_half_o; bk -= sh_half_k; }
        }
#pragma unroll
        for (uint32_t j = 0; j < PW; j++) {
            const uint32_t w = tid * PW + j;
            pocc[w] = (uint16_t)bo; pkeep[w] = (uint16_t)bk;
            bo += __popc(occ[w]); bk += __popc(keepbm[w]);
        }
    }
    if (tid == 0) at_count = 0;      // from here on: number of pattern-table slots this workgroup claims
    __syncthreads();
    PF_PROF_STAMP(4);
    PF_KO_FINISH_AT(4);
    const uint32_t half_o = DW > 2048 ? sh_half_o : 0, half_k = DW > 2048 ? sh_half_k : 0;
    auto rank_of = [&](uint32_t o) -> uint32_t {
        return pocc[o >> 5] + __popc(occ[o >> 5] & ((1u << (o & 31)) - 1)) + (DW > 2048 && (o >> 5) >= 2048 ? half_o : 0u);
    };
    auto kept_before = [&](uint32_t o) -> uint32_t {
        return pkeep[o >> 5] + __popc(keepbm[o >> 5] & ((1u << (o & 31)) - 1)) + (DW > 2048 && (o >> 5) >= 2048 ? half_k : 0u);
    };
    // ---- run-global pattern table, bulk protocol.  Entry AT-1 of the mask table (never a hash position) stands
    // for the cluster's own row.  Step 1: every entry finds its pattern or claims a free slot (no waiting, no id);
    // the output range is reserved meanwhile.  Step 2: ONE atomicAdd on the global id counter for all claims of
    // this workgroup, ids published.  Step 3: entries that ran into another workgroup's unpublished claim go
    // through the ordinary blocking insert (safe now: nobody waits on us).  Step 4: first_seen minima; whoever
    // lowers one writes the pattern's row.
    const bool is_row_entry = tid == AT - 1;      // T >= AT
    uint64_t e_lo = 0, e_fs = 0, e_gslot = 0;
    uint32_t e_hi = 0, e_pid = 0xFFFFFFFFu, e_newidx = 0;
    int e_state = -1;                             // -1 no entry, 0 found, 1 claimed, 2 deferred
    uint64_t e_key = 0;
    static_assert(T >= AT, "one thread per mask-table entry");
    if (tid == (AT == T ? T - 2 : T - 1)) {       // (not the row entry's thread, which has its hash to work out meanwhile)
        sh_base = atomicAdd((unsigned long long*)&p.cursor[0], (unsigned long long)tot_k);
        atomicAdd((unsigned long long*)&p.cursor[1], (unsigned long long)tot_o);
        atomicAdd((unsigned long long*)&p.cursor[2], (unsigned long long)tot_k);
        p.cluster_kmer_off[c] = sh_base;
        p.cluster_kmer_cnt[c] = tot_k;
        p.cluster_unique[c] = tot_o;
    }
    if (is_row_entry) {
        // the cluster's own row: md5 of the int64 image of clusterpresab (panfeed.py:175-187)
        const uint32_t nw = (npres + 31) >> 5;
        H128 hs;
        hs.h1 = 0x9747b28cu ^ npres; hs.h2 = 0x1b873593u ^ 0x5bd1e995u; hs.h3 = 0xe6546b64u; hs.h4 = 0x85ebca6bu;
        if (p.multiple_files) { hs.h2 ^= (uint32_t)ordinal; hs.h3 ^= (uint32_t)(ordinal >> 32); }
        for (uint32_t w = 0; w < nw; w += 4) {
            uint32_t wv[4];
            for (int j = 0; j < 4; j++) wv[j] = (w + j < nw) ? presab[w + j] : 0;
            mm3_block(hs, wv[0], wv[1], wv[2], wv[3]);
        }
        mm3_final(hs, nw * 4 + 1);
        e_lo = ((uint64_t)hs.h1 << 32) | hs.h2; e_hi = hs.h3; e_fs = ordinal << 32;
        e_state = 0;
    } else if (tid < AT - 1) {
        e_key = at_key[tid];
        if (e_key && at_keep[tid]) {
            const uint32_t mo = at_minord[tid];
            if (mo != NO_ORD) {
                const uint4 h = at_hash[tid];
                e_lo = ((uint64_t)h.x << 32) | h.y; e_hi = h.z;
                e_fs = (ordinal << 32) | (uint64_t)(rank_of(mo) + 1);
                e_state = 0;
            }
        }
        if (e_state < 0) at_pid[tid] = 0xFFFFFFFFu;
    }
#if defined(PF_KO_FINISH) && PF_KO_FINISH == 7
    if (e_state == 0) e_pid = 0;                                   // (timing experiment: no run-global table)
#else
    if (e_state == 0) {
        e_state = pattern_find_or_claim(p.pt, e_lo, e_hi, &e_gslot, &e_pid);
        if (e_state == 3) e_state = 0;                             // table full: e_pid = PID_NONE, the batch is re-run
        if (e_state == 1) e_newidx = atomicAdd(&at_count, 1u);     // at_count: claims of this workgroup (reused)
    }
#endif
    __syncthreads();
    PF_PROF_STAMP(5);
    PF_KO_FINISH_AT(5);
    if (tid == 0) {
        const uint32_t nnew = at_count;
        uint32_t base = 0;
        if (nnew) {
            base = atomicAdd(&p.pt.counters[0], nnew);
            if ((uint64_t)base + nnew > p.pt.pool) p.pt.counters[1] = 1;
        }
        sh_npres = base;                                           // reused: first id of this workgroup's claims
    }
    __syncthreads();
    if (e_state == 1) {
        const uint32_t id = sh_npres + e_newidx;
        e_pid = id < p.pt.pool ? id : PID_NONE;
        pattern_publish(p.pt, e_gslot, e_hi, e_pid);
    }
    // The publish above must come before the waiting form below in the CODE the compiler emits, not only here: the two
    // branches are disjoint sets of lanes and their order is its choice -- pattern_insert_block says what happened when it
    // chose the other one.  The barrier pins it (measured: no cost, finish_kernel 3.91 ms with and without).
    __syncthreads();
    bool lowered = false;
    if (e_state == 2) {
        e_pid = pattern_insert_lower(p.pt, e_lo, e_hi, e_fs, &lowered);
    } else if (e_state >= 0 && e_pid < p.pt.pool) {
#if !(defined(PF_KO_FINISH) && PF_KO_FINISH == 7)
        const uint64_t old = atomicMin((unsigned long long*)&p.pt.first_seen[e_pid], (unsigned long long)e_fs);
        lowered = old > e_fs;
#endif
    }
    if (e_state >= 0) {
        if (is_row_entry) {
            p.cluster_pattern[c] = e_pid;
            if (lowered && e_pid < p.pt.pool) {
                for (uint32_t w = 0; w < W; w++) {
                    p.pat_bits[(size_t)e_pid * W + w] = presab[w];
                    if (p.pat_nan) p.pat_nan[(size_t)e_pid * W + w] = 0;
                }
                p.pat_n[e_pid] = npres | 0x80000000u;
            }
        } else {
            at_pid[tid] = e_pid;
            if (lowered && e_pid < p.pt.pool) at_minord[atomicAdd(&wl_count, 1u)] = tid;   // row written below, 8 lanes each
        }
    }
    __syncthreads();
    PF_PROF_STAMP(6);
    PF_KO_FINISH_AT(6);
    const uint64_t obase = sh_base - p.out_base;
    // rows of the patterns whose first_seen this workgroup lowered (at_minord is free by now: the list of their
    // mask-table positions)
    for (uint32_t g = tid >> 3, nw = wl_count; g < nw; g += T >> 3) {
        const uint32_t t = at_minord[g];
        write_row8(at_pid[t], at_key[t], tid & 7u);
    }
    // outputs: key + pattern id per kept k-mer, in first-occurrence order
    for (uint32_t q = 0; q < nparts; q++) {
    set_part(q);
    for (uint32_t i0 = tid; i0 < ns; i0 += 4 * T) {
      uint32_t o_[4], ml_[4], mh_[4];
      uint64_t k0_[4];
      bool kp_[4];
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) {
          const uint32_t i = i0 + u * T;
          o_[u] = i < ns ? ordp[i] : 0xFFFFFFFFu;
      }
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) {
          const uint32_t i = i0 + u * T, o = o_[u];
          kp_[u] = (o >> 5) < dense_words && ((keepbm[o >> 5] >> (o & 31)) & 1);
          k0_[u] = kp_[u] ? p.tab_key[((size_t)slice * KW) * NS + i] : 0;
          if (MULTI) { ml_[u] = kp_[u] ? mlo[i] : 0; mh_[u] = (has_hi && kp_[u]) ? mhi[i] : 0; }
      }
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) {
        const uint32_t i = i0 + u * T;
        if (!kp_[u]) continue;
        const uint32_t o = o_[u];
        const tag_t tag = MULTI ? find_tag((uint64_t)ml_[u] | ((uint64_t)mh_[u] << 32)) : slot_at[i];
        if (tag == UNTABLED) continue;           // second loop below
        const uint64_t oi = obase + kept_before(o);
        if (oi >= p.out_cap) { p.pt.counters[2] = 1; continue; }
        p.out_key[oi * KW] = k0_[u];
        for (uint32_t j = 1; j < KW; j++) p.out_key[oi * KW + j] = p.tab_key[((size_t)slice * KW + j) * NS + i];
        p.out_pid[oi] = at_pid[tag];
      }
    }
    }
    if (saw_untabled) {
        // mask table was full: these slots go to the run-global table on their own
        for (uint32_t q = 0; q < nparts; q++) {
            set_part(q);
#pragma unroll 1
            for (uint32_t i = tid; i < ns; i += T) {
                const uint32_t o = ordp[i];
                if ((o >> 5) >= dense_words || !((keepbm[o >> 5] >> (o & 31)) & 1)) continue;
                const uint64_t amask = (uint64_t)mlo[i] | ((uint64_t)(has_hi ? mhi[i] : 0u) << 32);
                if ((MULTI ? find_tag(amask) : slot_at[i]) != UNTABLED) continue;
                uint4 h;
                row_eval(amask, h);
                bool lw;
                const uint32_t pid = pattern_insert_lower(p.pt, ((uint64_t)h.x << 32) | h.y, h.z,
                                                          (ordinal << 32) | (uint64_t)(rank_of(o) + 1), &lw);
                if (lw && pid < p.pt.pool) write_row(pid, amask);
                const uint64_t oi = obase + kept_before(o);
                if (oi >= p.out_cap) { p.pt.counters[2] = 1; continue; }
                p.out_key[oi * KW] = p.tab_key[((size_t)slice * KW) * NS + i];
                for (uint32_t j = 1; j < KW; j++) p.out_key[oi * KW + j] = p.tab_key[((size_t)slice * KW + j) * NS + i];
                p.out_pid[oi] = pid;
            }
        }
    }
    // slow-path rows that are kept: their own insert into the run-global table, their row if they are its first
#pragma unroll 1
    for (uint32_t e = ex0 + tid; e < ex1; e += T) {
        const uint32_t o = p.extra_dense[e];
        if ((o >> 5) >= dense_words || !((keepbm[o >> 5] >> (o & 31)) & 1)) continue;
        const uint32_t* words = p.extra_bits + (size_t)e * W;
        uint4 h;
        row_eval_words(words, h);
        bool lw;
        const uint32_t pid = pattern_insert_lower(p.pt, ((uint64_t)h.x << 32) | h.y, h.z,
                                                  (ordinal << 32) | (uint64_t)(rank_of(o) + 1), &lw);
        if (lw && pid < p.pt.pool) {
            for (uint32_t w = 0; w < W; w++) {
                p.pat_bits[(size_t)pid * W + w] = w < nchunks ? words[w] : 0;
                if (p.pat_nan) {
                    uint32_t nn = 0;
                    if (p.consider_missing && w < nchunks) {
                        nn = ~presab[w];
                        const uint32_t rem = nstr - (w << 5);
                        if (rem < 32) nn &= (1u << rem) - 1;
                    }
                    p.pat_nan[(size_t)pid * W + w] = nn;
                }
            }
            p.pat_n[pid] = nstr;
        }
        const uint64_t oi = obase + kept_before(o);
        if (oi >= p.out_cap) { p.pt.counters[2] = 1; continue; }
        p.out_key[oi * KW] = KEY_EXTRA_FLAG | (uint64_t)e;
        for (uint32_t j = 1; j < KW; j++) p.out_key[oi * KW + j] = 0;
        p.out_pid[oi] = pid;
    }
    __syncthreads();
    PF_PROF_STAMP(7);
#ifdef PF_PROF
    if (tid == 0) atomicAdd(&pf_prof[8], 1ull);
#endif
}

// ---------------------------------------------------------------------------------------------
// pattern_rows_kernel: the k-mer (or cluster row) that holds a pattern's first_seen writes its row
// ---------------------------------------------------------------------------------------------
#if defined(PF_PROF) && !defined(PF_PROF_FINISH)
#undef PF_PROF_BEGIN
#undef PF_PROF_STAMP
#define PF_PROF_BEGIN() uint64_t prof_t_ = __builtin_readcyclecounter()
#define PF_PROF_STAMP(k) do { if (threadIdx.x == 0) { const uint64_t n_ = __builtin_readcyclecounter(); \
    atomicAdd(&pf_prof[k], (unsigned long long)(n_ - prof_t_)); prof_t_ = n_; } } while (0)
#endif
struct PatRowsParams {
    const uint32_t* item_cluster; const uint32_t* item_scratch; const uint32_t* item_unique;
    const uint32_t* item_nslots; const uint32_t* item_is_extra;
    const uint32_t* item_sib0; const uint32_t* item_nsib;
    const uint32_t* cluster_overflow;
    const uint32_t* v_mode; const uint32_t* v_nstr; const uint32_t* v_dense;
    const uint32_t* cluster_seg_off; const uint32_t* seg_sample; const uint32_t* seg_distinct;   // caller's segments (mode 2)
    const uint32_t* cluster_nstrains; const uint32_t* cluster_npresab; const uint32_t* cluster_presab;
    const uint64_t* cluster_kmer_off;
    const uint64_t* sorted_pair; const uint32_t* kept_prefix; const uint32_t* chunkbits; const uint32_t* chunkmask;
    const uint32_t* slot_out; const uint32_t* mrows;
    const uint32_t* out_pid; const uint64_t* out_first;
    const uint32_t* cluster_pattern; const uint64_t* cluster_first;
    const uint64_t* pat_first_seen;
    uint32_t* pat_bits; uint32_t* pat_nan; uint32_t* pat_n;
    uint64_t out_base, out_cap;
    uint32_t pool;
    const uint32_t* work;   // [gridDim.x] item ids of this launch
    uint32_t W, NS;
    uint32_t consider_missing;
};
constexpr uint32_t PR_LIST = 2048;   // winners collected per round
// 512 threads: three workgroups = 24 waves per CU (the 50 KiB of LDS allow three; with 256 threads that was 12 waves, with
// 1024 two workgroups = 32 waves but longer barriers).  Phase 1 is three dependent global loads per slot and nothing else:
// the waves in flight are all that hides them.  2 000 clusters of ~140 related alleles: 2.05 ms at 256, 1.45 at 512,
// 1.89 at 1024 (tools/pr_exp.sh).
constexpr uint32_t PR_THREADS = 512;

__global__ __launch_bounds__(PR_THREADS) void pattern_rows_kernel(PatRowsParams p) {
    // phase 1: every thread looks for k-mers whose first_seen won their pattern and appends (slot, pid) to an
    // LDS list; phase 2: one wave per list entry builds the row with its lanes spread over the words.
    __shared__ uint32_t l_slot[PR_LIST];
    __shared__ uint32_t l_pid[PR_LIST];
    __shared__ uint32_t l_count;
    // mode 2: the cluster's (distinct index << 5 | sample & 31) list, the first segment of every 32-sample word, and
    // the mask a wave is expanding
    __shared__ __align__(16) uint16_t segd[DEDUP_MAX_SEGS];
    __shared__ uint32_t wstart[MAX_CHUNKS + 1];
    __shared__ uint32_t mstage[PR_THREADS / 64][33];          // word 32 stays zero (the place unused segment registers point at)
    // mode 1: the item's sample-set matrix M (rows_kernel left it in global memory), which a pattern's row is the OR of up
    // to 64 rows of -- read from LDS, not through the cache, once per pattern word and member (it shares the place of the
    // mode-2 segment list: an item is one or the other)
    uint32_t* const Ml = reinterpret_cast<uint32_t*>(segd);
    static_assert(sizeof(uint16_t) * DEDUP_MAX_SEGS >= sizeof(uint32_t) * DEDUP_MROWS, "M does not fit the segment list's place");

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const uint32_t item = p.work[blockIdx.x];
    const uint32_t c = p.item_cluster[item];
    if (p.cluster_overflow[c]) return;
    const uint32_t slice = p.item_scratch[item];
    const uint32_t NS = p.NS, W = p.W;
    const uint32_t U = p.item_unique[item];
    const uint32_t sib0 = p.item_sib0[item];
    const uint32_t nstr = p.cluster_nstrains[c];
    const uint32_t nchunks = (nstr + 31) >> 5;
    const uint32_t* presab = p.cluster_presab + (size_t)c * W;
    const uint64_t obase = p.cluster_kmer_off[c] - p.out_base;
    const uint32_t* cb = p.chunkbits + (size_t)slice * W * NS;
    const uint32_t* cm = p.chunkmask + slice * 8;
    const uint32_t mode = p.v_mode[c] & 3u;
    const uint32_t ns = p.item_nslots[item];
    const bool expand = mode == 1 && !p.item_is_extra[item];
    const bool wide = mode == 2 && !p.item_is_extra[item];
    const uint32_t nmw = (p.v_nstr[c] + 31) >> 5;
    const uint32_t Wp = (W + 3) & ~3u;
    const uint32_t* M = p.mrows + (size_t)slice * DEDUP_MROWS;
    const uint32_t* sout = p.slot_out + (size_t)slice * NS;
    const uint64_t* sp = p.sorted_pair + (size_t)slice * NS;

    if (item == sib0 && tid == 0) {
        const uint32_t pid = p.cluster_pattern[c];
        if (pid < p.pool && p.pat_first_seen[pid] == p.cluster_first[c]) {
            const uint32_t npres = p.cluster_npresab[c];
            for (uint32_t w = 0; w < W; w++) {
                p.pat_bits[(size_t)pid * W + w] = presab[w];
                if (p.pat_nan) p.pat_nan[(size_t)pid * W + w] = 0;
            }
            p.pat_n[pid] = npres | 0x80000000u;
        }
    }
    auto nan_word = [&](uint32_t w) -> uint32_t {
        uint32_t nn = 0;
        if (p.consider_missing && w < nchunks) {
            nn = ~presab[w];
            const uint32_t rem = nstr - (w << 5);
            if (rem < 32) nn &= (1u << rem) - 1;
        }
        return nn;
    };
    if (expand) {
        const uint32_t mw = min(p.v_nstr[c] * Wp, DEDUP_MROWS);
        for (uint32_t i = tid; i < mw; i += blockDim.x) Ml[i] = M[i];
        __syncthreads();
    }
    if (wide) {
        const uint32_t s0 = p.cluster_seg_off[c], s1 = p.cluster_seg_off[c + 1];
        for (uint32_t s = tid; s < s1 - s0; s += blockDim.x)
            segd[s] = (uint16_t)((p.seg_distinct[s0 + s] << 5) | (p.seg_sample[s0 + s] & 31u));
        for (uint32_t w = tid; w <= nchunks; w += blockDim.x)
            wstart[w] = seg_lower_bound(p.seg_sample, s0, s1, w << 5) - s0;
        if (lane == 0) mstage[wave][32] = 0;
        __syncthreads();
    }
    // mode 2: the segments lane `lane` looks at for row word `lane` are the same for every pattern: held in registers as in
    // rows_kernel (two to a register, unused places = distinct index 1024, whose mask word is the zero word 32)
    constexpr uint32_t PR_SEGREG = 32;
    uint32_t sg[PR_SEGREG / 2];
    uint32_t my_q0 = 0, my_q1 = 0;
    if (wide && lane < nchunks) { my_q0 = wstart[lane]; my_q1 = wstart[lane + 1]; }
#pragma unroll
    for (uint32_t j = 0; j < PR_SEGREG / 2; j++) {
        const uint32_t q = my_q0 + 2 * j;
        const uint32_t e0 = q < my_q1 ? segd[q] : 0x8000u, e1 = q + 1 < my_q1 ? segd[q + 1] : 0x8000u;
        sg[j] = e0 | (e1 << 16);
    }
    uint32_t sg_used = min(my_q1 - my_q0, PR_SEGREG);
    for (int dd = 1; dd < 64; dd <<= 1) sg_used = max(sg_used, (uint32_t)__shfl_xor(sg_used, dd));
    // entries = sorted positions, or (ranks by bitmap) the item's kept k-mers: (ordinal, slot) pairs from rows_kernel,
    // their output indices from emit_kernel
    const bool sorted = !ranks_by_bitmap(mode, p.v_dense[c]);
    const uint32_t* kres = p.kept_prefix + (size_t)slice * (NS + 1) + 1;
    const uint32_t total = sorted ? U : min(kres[-1], ns);
    const uint32_t stride = blockDim.x;
    const uint32_t rounds_total = (total + stride - 1) / stride;
    uint32_t round = 0;
    while (round < rounds_total) {          // uniform: every thread walks the same number of rounds
        if (tid == 0) l_count = 0;
        __syncthreads();
        // PR_LIST / stride rounds fit the list even if every thread appends in every round
        // (the rounds of one list fill, four at most, side by side: each entry is a chain of three trips to memory -- output
        // index, pattern id + the k-mer's first_seen, the pattern's first_seen -- and round after round that was twelve)
        constexpr uint32_t PRR = PR_LIST / PR_THREADS;
        static_assert(PRR * PR_THREADS == PR_LIST, "whole rounds per list fill");
        const uint32_t nr = min(PRR, rounds_total - round);
        uint32_t kb_[PRR], sl_[PRR], pid_[PRR];
        uint64_t of_[PRR], pf_[PRR];
        const uint32_t tclamp = total - 1;                       // (total > 0 here: rounds_total > 0)
#pragma unroll
        for (uint32_t r = 0; r < PRR; r++) {
            const uint32_t i = min((round + r) * stride + tid, tclamp);
            // emit_kernel left the output index of every kept entry (sorted position / slot)
            kb_[r] = sorted ? sout[i] : kres[i];
            sl_[r] = (uint32_t)sp[i];
        }
#pragma unroll
        for (uint32_t r = 0; r < PRR; r++) {
            asm volatile("" : "+v"(kb_[r])); asm volatile("" : "+v"(sl_[r]));
            const uint32_t i = (round + r) * stride + tid;
            if (!(r < nr && i < total)) kb_[r] = 0xFFFFFFFFu;
        }
        bool ok_[PRR];
#pragma unroll
        for (uint32_t r = 0; r < PRR; r++) {
            const uint64_t o = obase + kb_[r];
            ok_[r] = kb_[r] != 0xFFFFFFFFu && o < p.out_cap;
            const uint64_t oc = ok_[r] ? o : 0;                  // (an arena holds at least one entry)
            pid_[r] = p.out_pid[oc];                             // (unconditional: the loads of the four rounds go out together)
            of_[r] = p.out_first[oc];
        }
#pragma unroll
        for (uint32_t r = 0; r < PRR; r++) {
            asm volatile("" : "+v"(pid_[r]));
            if (!ok_[r]) pid_[r] = 0xFFFFFFFFu;
        }
#pragma unroll
        for (uint32_t r = 0; r < PRR; r++) pf_[r] = p.pat_first_seen[min(pid_[r], p.pool - 1)];
#pragma unroll
        for (uint32_t r = 0; r < PRR; r++) {
            if (pid_[r] < p.pool && pf_[r] == of_[r]) {
                const uint32_t at = atomicAdd(&l_count, 1u);
                if (at < PR_LIST) { l_slot[at] = sl_[r]; l_pid[at] = pid_[r]; }
            }
        }
        round += nr;
        __syncthreads();
        const uint32_t cnt = min(l_count, PR_LIST);
        for (uint32_t e = wave; e < cnt; e += nwaves) {
            const uint32_t slot = l_slot[e], pid = l_pid[e];
            uint64_t amask = 0;
            if (expand) {
                const bool f0 = (cm[0] & 1) != 0, f1 = (cm[0] & 2) != 0;
                amask = (f0 ? (uint64_t)cb[slot] : 0) | (f1 ? (uint64_t)cb[(size_t)NS + slot] << 32 : 0);
            }
            if (wide) {
                __builtin_amdgcn_wave_barrier();
                if (lane < nmw) mstage[wave][lane] = ((cm[lane >> 5] >> (lane & 31)) & 1) ? cb[(size_t)lane * NS + slot] : 0u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
            for (uint32_t w = lane; w < W; w += 64) {
                uint32_t v = 0;
                if (w < nchunks) {
                    if (wide) {
                        const uint32_t* ms = mstage[wave];
                        uint32_t q = wstart[w];
                        const uint32_t qe = wstart[w + 1];
                        if (w == lane) {                            // the first 64 words: from the registers
#pragma unroll
                            for (uint32_t j = 0; j < PR_SEGREG / 2; j++) {
                                if (2 * j < sg_used) {              // wave-uniform
                                    uint32_t pr2 = sg[j];
                                    asm volatile("" : "+v"(pr2));   // decoded here, every time (see rows_kernel)
                                    const uint32_t e0 = pr2 & 0xFFFFu, e1 = pr2 >> 16;
                                    v |= ((ms[e0 >> 10] >> ((e0 >> 5) & 31u)) & 1u) << (e0 & 31u);
                                    v |= ((ms[e1 >> 10] >> ((e1 >> 5) & 31u)) & 1u) << (e1 & 31u);
                                }
                            }
                            q += PR_SEGREG;
                        }
                        for (; q < qe; q++) {
                            const uint32_t e = segd[q], d = e >> 5;
                            v |= ((ms[d >> 5] >> (d & 31u)) & 1u) << (e & 31u);
                        }
                    } else if (expand) {
                        uint64_t t = amask;
                        while (t) {
                            const uint32_t d = __ffsll((unsigned long long)t) - 1;
                            t &= t - 1;
                            v |= Ml[d * Wp + w];
                        }
                    } else if ((cm[w >> 5] >> (w & 31)) & 1) {
                        v = cb[(size_t)w * NS + slot];
                    }
                }
                p.pat_bits[(size_t)pid * W + w] = v;
                if (p.pat_nan) p.pat_nan[(size_t)pid * W + w] = nan_word(w);
            }
            if (lane == 0) p.pat_n[pid] = nstr;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// md5_kernel: one lane per pattern, MD5 (RFC 1321) of the 8*n-byte image generated from the bits
// ---------------------------------------------------------------------------------------------
// MD5 (RFC 1321), fully unrolled: the four round functions are ONE v_bitop3_b32 each (truth tables over
// src0 = 0xF0, src1 = 0xCC, src2 = 0xAA) -- left to itself the compiler builds H from two v_xor and uses 1.5 vector
// instructions per logic function; K and the rotation amounts are immediates.  Issue costs on gfx950
// (profiles/r02/valu_issue_costs_gfx950.txt): v_bitop3 / v_add_u32 2.3 cycles per wave and SIMD, v_add3_u32 /
// v_alignbit_b32 4.15: a step is 13 cycles of issue, 15.4 with a message word.
#define MD5_F(x, y, z) __builtin_amdgcn_bitop3_b32((x), (y), (z), 0xCA)     // (x & y) | (~x & z)
#define MD5_G(x, y, z) __builtin_amdgcn_bitop3_b32((x), (y), (z), 0xE4)     // (x & z) | (y & ~z)
#define MD5_H(x, y, z) __builtin_amdgcn_bitop3_b32((x), (y), (z), 0x96)     // x ^ y ^ z
#define MD5_I(x, y, z) __builtin_amdgcn_bitop3_b32((x), (y), (z), 0x39)     // y ^ (x | ~z)
#define MD5_STEP(f, a, b, c, d, k, s, mw) a += f(b, c, d) + (k) + (mw); a = b + __builtin_rotateleft32(a, s);
__device__ __forceinline__ void md5_block(uint32_t st[4], const uint32_t m[16]) {
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3];
    MD5_STEP(MD5_F, a, b, c, d, 0xd76aa478u, 7, m[0])
    MD5_STEP(MD5_F, d, a, b, c, 0xe8c7b756u, 12, m[1])
    MD5_STEP(MD5_F, c, d, a, b, 0x242070dbu, 17, m[2])
    MD5_STEP(MD5_F, b, c, d, a, 0xc1bdceeeu, 22, m[3])
    MD5_STEP(MD5_F, a, b, c, d, 0xf57c0fafu, 7, m[4])
    MD5_STEP(MD5_F, d, a, b, c, 0x4787c62au, 12, m[5])
    MD5_STEP(MD5_F, c, d, a, b, 0xa8304613u, 17, m[6])
    MD5_STEP(MD5_F, b, c, d, a, 0xfd469501u, 22, m[7])
    MD5_STEP(MD5_F, a, b, c, d, 0x698098d8u, 7, m[8])
    MD5_STEP(MD5_F, d, a, b, c, 0x8b44f7afu, 12, m[9])
    MD5_STEP(MD5_F, c, d, a, b, 0xffff5bb1u, 17, m[10])
    MD5_STEP(MD5_F, b, c, d, a, 0x895cd7beu, 22, m[11])
    MD5_STEP(MD5_F, a, b, c, d, 0x6b901122u, 7, m[12])
    MD5_STEP(MD5_F, d, a, b, c, 0xfd987193u, 12, m[13])
    MD5_STEP(MD5_F, c, d, a, b, 0xa679438eu, 17, m[14])
    MD5_STEP(MD5_F, b, c, d, a, 0x49b40821u, 22, m[15])
    MD5_STEP(MD5_G, a, b, c, d, 0xf61e2562u, 5, m[1])
    MD5_STEP(MD5_G, d, a, b, c, 0xc040b340u, 9, m[6])
    MD5_STEP(MD5_G, c, d, a, b, 0x265e5a51u, 14, m[11])
    MD5_STEP(MD5_G, b, c, d, a, 0xe9b6c7aau, 20, m[0])
    MD5_STEP(MD5_G, a, b, c, d, 0xd62f105du, 5, m[5])
    MD5_STEP(MD5_G, d, a, b, c, 0x02441453u, 9, m[10])
    MD5_STEP(MD5_G, c, d, a, b, 0xd8a1e681u, 14, m[15])
    MD5_STEP(MD5_G, b, c, d, a, 0xe7d3fbc8u, 20, m[4])
    MD5_STEP(MD5_G, a, b, c, d, 0x21e1cde6u, 5, m[9])
    MD5_STEP(MD5_G, d, a, b, c, 0xc33707d6u, 9, m[14])
    MD5_STEP(MD5_G, c, d, a, b, 0xf4d50d87u, 14, m[3])
    MD5_STEP(MD5_G, b, c, d, a, 0x455a14edu, 20, m[8])
    MD5_STEP(MD5_G, a, b, c, d, 0xa9e3e905u, 5, m[13])
    MD5_STEP(MD5_G, d, a, b, c, 0xfcefa3f8u, 9, m[2])
    MD5_STEP(MD5_G, c, d, a, b, 0x676f02d9u, 14, m[7])
    MD5_STEP(MD5_G, b, c, d, a, 0x8d2a4c8au, 20, m[12])
    MD5_STEP(MD5_H, a, b, c, d, 0xfffa3942u, 4, m[5])
    MD5_STEP(MD5_H, d, a, b, c, 0x8771f681u, 11, m[8])
    MD5_STEP(MD5_H, c, d, a, b, 0x6d9d6122u, 16, m[11])
    MD5_STEP(MD5_H, b, c, d, a, 0xfde5380cu, 23, m[14])
    MD5_STEP(MD5_H, a, b, c, d, 0xa4beea44u, 4, m[1])
    MD5_STEP(MD5_H, d, a, b, c, 0x4bdecfa9u, 11, m[4])
    MD5_STEP(MD5_H, c, d, a, b, 0xf6bb4b60u, 16, m[7])
    MD5_STEP(MD5_H, b, c, d, a, 0xbebfbc70u, 23, m[10])
    MD5_STEP(MD5_H, a, b, c, d, 0x289b7ec6u, 4, m[13])
    MD5_STEP(MD5_H, d, a, b, c, 0xeaa127fau, 11, m[0])
    MD5_STEP(MD5_H, c, d, a, b, 0xd4ef3085u, 16, m[3])
    MD5_STEP(MD5_H, b, c, d, a, 0x04881d05u, 23, m[6])
    MD5_STEP(MD5_H, a, b, c, d, 0xd9d4d039u, 4, m[9])
    MD5_STEP(MD5_H, d, a, b, c, 0xe6db99e5u, 11, m[12])
    MD5_STEP(MD5_H, c, d, a, b, 0x1fa27cf8u, 16, m[15])
    MD5_STEP(MD5_H, b, c, d, a, 0xc4ac5665u, 23, m[2])
    MD5_STEP(MD5_I, a, b, c, d, 0xf4292244u, 6, m[0])
    MD5_STEP(MD5_I, d, a, b, c, 0x432aff97u, 10, m[7])
    MD5_STEP(MD5_I, c, d, a, b, 0xab9423a7u, 15, m[14])
    MD5_STEP(MD5_I, b, c, d, a, 0xfc93a039u, 21, m[5])
    MD5_STEP(MD5_I, a, b, c, d, 0x655b59c3u, 6, m[12])
    MD5_STEP(MD5_I, d, a, b, c, 0x8f0ccc92u, 10, m[3])
    MD5_STEP(MD5_I, c, d, a, b, 0xffeff47du, 15, m[10])
    MD5_STEP(MD5_I, b, c, d, a, 0x85845dd1u, 21, m[1])
    MD5_STEP(MD5_I, a, b, c, d, 0x6fa87e4fu, 6, m[8])
    MD5_STEP(MD5_I, d, a, b, c, 0xfe2ce6e0u, 10, m[15])
    MD5_STEP(MD5_I, c, d, a, b, 0xa3014314u, 15, m[6])
    MD5_STEP(MD5_I, b, c, d, a, 0x4e0811a1u, 21, m[13])
    MD5_STEP(MD5_I, a, b, c, d, 0xf7537e82u, 6, m[4])
    MD5_STEP(MD5_I, d, a, b, c, 0xbd3af235u, 10, m[11])
    MD5_STEP(MD5_I, c, d, a, b, 0x2ad7d2bbu, 15, m[2])
    MD5_STEP(MD5_I, b, c, d, a, 0xeb86d391u, 21, m[9])
    st[0] += a; st[1] += b; st[2] += c; st[3] += d;
}

struct Md5Params {
    const uint32_t* pat_bits; const uint32_t* pat_nan; const uint32_t* pat_n;
    uint8_t* pat_md5;
    const uint32_t* cluster_pattern;   // [n_clusters] the int pass: the clusters' own rows are the int64 rows
    uint32_t n_clusters;
    const uint32_t* range;   // device: {first id, one past the last id} of this launch, or null:
    uint32_t pid0, pid1, W;  // ... the range given here
};
constexpr uint32_t MD5_THREADS = 256;

// One lane per pattern, no LDS and no barriers: every lane reads its own row, four words (16 MD5 blocks) ahead of the
// words it is hashing, so a wave never waits for another and only registers limit the waves per SIMD (52: eight).
// FLOAT_ROWS: the pass over all new patterns, specialised for the float64 image of k-mer rows (98 % of the patterns):
// the low word of every element is zero, which the inlined md5_block folds away (32 of its 64 message additions); the
// few int64 rows (the clusters' own, one per cluster at most) are only listed.  !FLOAT_ROWS: the listed int64 rows.
// HAS_NAN: a NaN mask exists (--consider-missing-cluster); compiled apart, the common case pays nothing for it.
// Vector-issue bound: 310 instructions per block at the ~4 cycles per wave-instruction this mix of 2- and 4-cycle
// operations sustains on a SIMD (tools/micro/valu_chain.hip: 4.05 with one to eight waves, one to four chains per wave).
// The int pass goes by cluster (its rows are the clusters' own: cluster_pattern[c], hashed by whichever cluster names a new
// one -- clusters with the same row write the same digest), so it does not wait for the float pass: the two run side by
// side on two streams (round 5: listed by the float pass, the int rows were a second launch behind it -- one wave's chain of
// 126 blocks, 66 us, with the chip empty).
template <bool FLOAT_ROWS, bool HAS_NAN>
__global__ __launch_bounds__(MD5_THREADS) void md5_kernel(Md5Params p) {
    // rows blockIdx.x * 256 + tid, + gridDim.x * 256, ... of the id range (the grid is sized before the range is known)
    const uint32_t r0 = FLOAT_ROWS ? (p.range ? p.range[0] : p.pid0) : 0;
    const uint32_t r1 = FLOAT_ROWS ? (p.range ? p.range[1] : p.pid1) : p.n_clusters;
    const uint32_t W = p.W;
    for (uint64_t base = (uint64_t)r0 + (uint64_t)blockIdx.x * MD5_THREADS; base < r1; base += (uint64_t)gridDim.x * MD5_THREADS) {
        const uint32_t idx = (uint32_t)base + threadIdx.x;
        if (idx >= r1) continue;
        const uint32_t pid = FLOAT_ROWS ? idx : p.cluster_pattern[idx];
        if (!FLOAT_ROWS && (pid < p.pid0 || pid >= p.pid1)) continue;      // a row of an earlier batch (hashed then), or none
        const uint32_t nk = p.pat_n[pid];
        if (FLOAT_ROWS == ((nk >> 31) != 0)) continue;        // an int64 row is the int pass's, a float64 row the float pass's
        const uint32_t n = nk & 0x7FFFFFFFu;
        const uint32_t* rb = p.pat_bits + (size_t)pid * W;
        const uint32_t* rn = HAS_NAN ? p.pat_nan + (size_t)pid * W : nullptr;
        uint32_t st[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
        const uint64_t nbytes = (uint64_t)n * 8;
        const uint32_t full = n >> 3, rem = n & 7;            // whole 64-byte blocks = 8 elements each, and the rest
        uint32_t m[16];
        uint32_t qb[4], qn[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            qb[j] = rb[min((uint32_t)j, W - 1)];
            if (HAS_NAN) qn[j] = rn[min((uint32_t)j, W - 1)];
        }
        uint32_t tail_bw = 0, tail_nw = 0;
        for (uint32_t w0 = 0; w0 < W; w0 += 4) {
            uint32_t nb[4], nn[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; j++) {                     // the next four words, in flight while these are hashed
                nb[j] = rb[min(w0 + 4 + j, W - 1)];
                if (HAS_NAN) nn[j] = rn[min(w0 + 4 + j, W - 1)];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t blk0 = (w0 + j) * 4;           // a row word = 32 elements = 4 blocks
                const uint32_t nblk = full > blk0 ? min(4u, full - blk0) : 0;
                for (uint32_t q = 0; q < nblk; q++) {
                    const uint32_t bw = (qb[j] >> (q * 8)) & 0xFF;
                    const uint32_t nw = HAS_NAN ? (qn[j] >> (q * 8)) & 0xFF : 0;
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        // int64 LE: 01 00.. ; float64 LE: 1.0 = 0x3FF00000:00000000, NaN = 0x7FF80000:00000000 (np.nan);
                        // branch-free: masks of the element's bit / NaN flag ANDed with constants
                        const uint32_t bm = 0u - ((bw >> e) & 1), nm = 0u - ((nw >> e) & 1);
                        m[2 * e] = FLOAT_ROWS ? 0 : bm & 1u;
                        m[2 * e + 1] = FLOAT_ROWS ? (bm & 0x3FF00000u) | (nm & 0x7FF80000u) : 0;
                    }
                    md5_block(st, m);
                }
                if (rem && (full >> 2) == w0 + j) {           // the (at most 7) elements after the last whole block
                    tail_bw = (qb[j] >> ((full & 3) * 8)) & 0xFF;
                    tail_nw = HAS_NAN ? (qn[j] >> ((full & 3) * 8)) & 0xFF : 0;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) { qb[j] = nb[j]; qn[j] = nn[j]; }
        }
        // tail: remaining elements, 0x80, zero pad, 64-bit length
#pragma unroll
        for (int j = 0; j < 16; j++) m[j] = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t bit = (tail_bw >> j) & 1, isn = (tail_nw >> j) & 1;
            if ((uint32_t)j < rem) {
                m[2 * j] = FLOAT_ROWS ? 0 : bit;
                m[2 * j + 1] = FLOAT_ROWS ? (bit ? 0x3FF00000u : (isn ? 0x7FF80000u : 0)) : 0;
            } else if ((uint32_t)j == rem) {
                m[2 * j] = 0x80;                              // first pad byte right after the data
            }
        }
        if (rem == 7) {                                       // 56 data bytes + 0x80 leaves no room for the length
            md5_block(st, m);
#pragma unroll
            for (int j = 0; j < 16; j++) m[j] = 0;
        }
        m[14] = (uint32_t)(nbytes << 3);
        m[15] = (uint32_t)((nbytes << 3) >> 32);
        md5_block(st, m);
        // little-endian words = MD5 byte order
        *reinterpret_cast<uint4*>(p.pat_md5 + (size_t)pid * 16) = make_uint4(st[0], st[1], st[2], st[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// result_checksum_kernel: a checksum of checksums of a batch's results, for full-size parity checks
// ---------------------------------------------------------------------------------------------
// One workgroup per cluster.  acc[0] += sum over the cluster's kept k-mers of h(cluster index, position in the cluster's
// output order, key words, MD5 digest of the k-mer's pattern); acc[1] += h(cluster index, kept, unique, digest of the
// cluster's own pattern); acc[2] += kept.  Sums wrap; every term depends on WHERE in the file a row would stand and
// on the bytes it would hold, nothing on arena placement or pattern ids.
__global__ __launch_bounds__(256) void result_checksum_kernel(const uint64_t* const* key_ptr, const uint32_t* const* pid_ptr,
                                                              const uint32_t* cnt, const uint32_t* uniq, const uint32_t* cpat,
                                                              const uint8_t* pat_md5, uint32_t KW, unsigned long long* acc) {
    const uint32_t c = blockIdx.x;
    const uint32_t n = cnt[c];
    const uint64_t* keys = key_ptr[c];
    const uint32_t* pids = pid_ptr[c];
    uint64_t sum = 0;
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
        uint64_t h = mix64(((uint64_t)c << 32) ^ j ^ 0x9E3779B97F4A7C15ull);
        for (uint32_t w = 0; w < KW; w++) h = mix64(h ^ keys[(size_t)j * KW + w]);
        const uint64_t* d = reinterpret_cast<const uint64_t*>(pat_md5 + (size_t)pids[j] * 16);
        sum += mix64(mix64(h ^ d[0]) + d[1]);
    }
    for (int d = 1; d < 64; d <<= 1) sum += __shfl_xor(sum, d);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd(&acc[0], (unsigned long long)sum);
    if (threadIdx.x == 0) {
        uint64_t h = mix64(((uint64_t)c << 32) ^ n) ^ mix64(((uint64_t)uniq[c] << 32) | 0x5bd1e995u);
        if (cpat[c] != 0xFFFFFFFFu) {
            const uint64_t* d = reinterpret_cast<const uint64_t*>(pat_md5 + (size_t)cpat[c] * 16);
            h = mix64(mix64(h ^ d[0]) + d[1]);
        }
        atomicAdd(&acc[1], (unsigned long long)mix64(h));
        atomicAdd(&acc[2], (unsigned long long)n);
    }
}

// ---------------------------------------------------------------------------------------------
// merge_digests_kernel: multi-GPU pattern dedup after the all-gather
// ---------------------------------------------------------------------------------------------
// gathered[i] = {md5 lo, md5 hi, first_seen} from every rank.  Pass 1 inserts every row into an open-addressing
// table keyed by the full 128-bit digest and keeps the minimum first_seen per digest; pass 2 looks this rank's
// own rows up again: a row is kept iff it holds that minimum (first_seen values are unique across the run).
struct MergeParams {
    const uint64_t* gathered;   // [n][3]
    uint64_t n;
    uint64_t* tab;              // [cap][4] = {digest lo, digest hi, min first_seen, -}, EMPTY64-filled: an entry is one
                                // 32-byte line segment (three parallel arrays were three cache lines per row)
    uint64_t cap;               // power of two
    uint64_t my_first, my_count;
    const int64_t* slot_counts; // padded layout: rows come in equal slots of `slot_rows` per rank, only the first
    uint64_t slot_rows;         // slot_counts[rank] rows of a slot are real (0 / null: every row is real)
    uint8_t* keep;              // [my_count]
    unsigned long long* n_global;
};
__device__ __forceinline__ uint64_t merge_slot(const MergeParams& p, uint64_t lo, uint64_t hi, bool insert, uint32_t* claimed) {
    // md5 words are already uniform; keep EMPTY64 out of the key space
    if (lo == EMPTY64) lo = EMPTY64 - 1;
    if (hi == EMPTY64) hi = EMPTY64 - 1;
    uint64_t slot = (lo ^ (hi >> 17)) & (p.cap - 1), res = EMPTY64;
    // an entry whose second word is not published yet is looked at again in the next round of the outer, wave-uniform
    // loop: no lane spins inside divergent code on a value another lane of its wave may have to publish
    int st = 2;
    for (;;) {
        if (st == 2) {
            st = 0;
            for (uint64_t probes = 0; probes < p.cap; probes++) {
                uint64_t cur = __hip_atomic_load(&p.tab[4 * slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == EMPTY64) {
                    if (!insert) break;
                    cur = atomicCAS((unsigned long long*)&p.tab[4 * slot], (unsigned long long)EMPTY64, (unsigned long long)lo);
                    if (cur == EMPTY64) {
                        __hip_atomic_store(&p.tab[4 * slot + 1], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ++*claimed;
                        res = slot;
                        break;
                    }
                }
                if (cur == lo) {
                    const uint64_t h = __hip_atomic_load(&p.tab[4 * slot + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (h == EMPTY64) { st = 2; break; }            // claimed, not yet published
                    if (h == hi) { res = slot; break; }
                }
                slot = (slot + 1) & (p.cap - 1);
            }
        }
        if (!__any(st == 2)) break;
        __builtin_amdgcn_s_sleep(1);
    }
    return res;
}
__global__ __launch_bounds__(256) void merge_insert_kernel(MergeParams p) {
    uint32_t claimed = 0;      // distinct digests this thread was first to insert
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < p.n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (p.slot_rows && (int64_t)(i % p.slot_rows) >= p.slot_counts[i / p.slot_rows]) continue;   // padding
        const uint64_t lo = p.gathered[3 * i], hi = p.gathered[3 * i + 1], fs = p.gathered[3 * i + 2];
        const uint64_t slot = merge_slot(p, lo, hi, true, &claimed);
        if (slot != EMPTY64) atomicMin((unsigned long long*)&p.tab[4 * slot + 2], (unsigned long long)fs);
    }
    for (int d = 32; d > 0; d >>= 1) claimed += __shfl_down(claimed, d);
    if ((threadIdx.x & 63) == 0 && claimed) atomicAdd(p.n_global, (unsigned long long)claimed);   // one per wave
}
__global__ __launch_bounds__(256) void merge_lookup_kernel(MergeParams p) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < p.my_count; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = p.my_first + j;
        const uint64_t lo = p.gathered[3 * i], hi = p.gathered[3 * i + 1], fs = p.gathered[3 * i + 2];
        uint32_t unused = 0;
        const uint64_t slot = merge_slot(p, lo, hi, false, &unused);
        p.keep[j] = (slot != EMPTY64 && p.tab[4 * slot + 2] == fs) ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// misc kernels
// ---------------------------------------------------------------------------------------------
// per-cluster instance counts: the trip count of panfeed.py:64 over the caller's segments, and the
// instances / packed words the scan will actually visit (the view); one wave per cluster.  With the cluster's mode,
// dense ordinal space and number of view columns they make the 40-byte record the host plans the batch's work from.
struct ClusterRec { uint64_t ninst, vinst, words; uint32_t mode, dense, vnstr, pad; };
static_assert(sizeof(ClusterRec) == 40, "ClusterRec is copied to the host as 40-byte records");
// What plan_kernel needs to know of a cluster, worked out here where every cluster has a wave of its own (the first cut
// did it inside plan_kernel's ONE workgroup: a float64 division and a dozen branches per cluster made a part of 37 500
// clusters 0.18 ms of a single CU): ClusterRec.pad = PLAN_CLS bits | table size | weight class, plan_room[c] = room of the
// cluster's unit view (0: none), plan_arena[c] = output entries to reserve.
struct PlanClassify {
    const uint32_t* extra_off;       // [C + 1] slow-path rows per cluster (CSR); nullptr: no classification (pad = 0)
    uint32_t* plan_room; uint32_t* plan_arena;
    uint32_t mult, NS, W, unit_view, reg_ready, pad;
    double share, reg_a, reg_b, reg_half_sd;     // the key-partition estimate (pf_api.hip, prep_half)
};
constexpr uint32_t PLAN_PLANNED = 1u;            // ClusterRec.pad bit 0: laid out by plan_kernel
constexpr uint32_t PLAN_CLS_SHIFT = 8, PLAN_NS_SHIFT = 12, PLAN_W_SHIFT = 16;   // class (1, 2, 5), table (0: NS, 1: 4096, 2: 6144), 64 - log2(windows)

__global__ __launch_bounds__(256) void cluster_ninst_kernel(const uint32_t* cluster_seg_off, const uint32_t* seg_len,
                                                            const uint32_t* v_len, const uint32_t* v_nseg, uint32_t k,
                                                            uint32_t c_first, uint32_t c_end, const uint32_t* list,
                                                            const uint32_t* v_mode, const uint32_t* v_dense,
                                                            const uint32_t* v_nstr, ClusterRec* rec, PlanClassify pc) {
    // clusters c_first .. c_end - 1, or (list) list[c_first .. c_end - 1]
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t idx = c_first + ((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (idx >= c_end) return;
    const uint32_t c = list ? list[idx] : idx;
    const uint32_t s0 = cluster_seg_off[c], s1 = cluster_seg_off[c + 1], sv = s0 + v_nseg[c];
    uint64_t n = 0, nv = 0, wv = 0;
    for (uint32_t s = s0 + lane; s < s1; s += 64) {
        const uint32_t len = seg_len[s];
        if (len >= k) n += len - k + 1;
    }
    for (uint32_t s = s0 + lane; s < sv; s += 64) {
        const uint32_t len = v_len[s];
        if (len >= k) nv += len - k + 1;
        wv += 2 * (uint64_t)((len + 63) >> 6);
    }
    for (int d = 32; d > 0; d >>= 1) {
        n += __shfl_down(n, d); nv += __shfl_down(nv, d); wv += __shfl_down(wv, d);
    }
    if (lane == 0) {
        ClusterRec r;
        r.ninst = n; r.vinst = nv; r.words = wv; r.mode = v_mode[c]; r.dense = v_dense[c]; r.vnstr = v_nstr[c]; r.pad = 0;
        if (pc.extra_off) {
            const uint32_t nex = pc.extra_off[c + 1] - pc.extra_off[c];
            const uint64_t inst = r.vinst * pc.mult;
            uint32_t cls = 0, nsi = 0, room_u = 0, arena = 0;
            if (r.mode == 1 && nex <= FUSED_MAX_EXTRA && r.ninst * pc.mult < 0xFFFFFFF0ull) {
                const double room = 0.9 * (double)insert_limit(pc.NS);
                bool one = true;                                  // one key partition by the estimate
                if (r.vnstr) {
                    const double D = (double)r.vnstr, L = (double)inst / D;
                    if (D >= 2.0 && pc.reg_ready) {
                        const double g = fmax(0.0, pc.reg_a + pc.reg_b * L) + pc.reg_half_sd;
                        one = !(L + g * (D - 1.0) > room);
                    } else {
                        one = !(L * (1.0 + pc.share * (D - 1.0)) > room) || D > 24.0;
                    }
                }
                const uint32_t mwords = r.vnstr * ((pc.W + 3) & ~3u);
                if (one) {
                    if (r.dense < FinSmall::DW * 32 - 1 && mwords <= FinSmall::MR) cls = 1;
                    else if (r.dense < FinLarge::DW * 32 - 1 && mwords <= FinLarge::MR) cls = 2;
                    else if (r.dense < FinHuge::DW * 32 - 1 && mwords <= FinHuge::MR) cls = 5;
                }
                uint32_t ns = pc.NS;
                if (pc.NS > 4096 + INSERT_SLACK && inst <= insert_limit(4096)) { ns = 4096; nsi = 1; }
                else if (pc.NS > 6144 + INSERT_SLACK && inst <= insert_limit(6144)) { ns = 6144; nsi = 2; }
                if (cls) {
                    const uint32_t wcls = r.vinst ? 63 - __clzll((long long)r.vinst) : 0;
                    r.pad = (cls << PLAN_CLS_SHIFT) | (nsi << PLAN_NS_SHIFT) | ((64 - wcls) << PLAN_W_SHIFT);
                    arena = (uint32_t)(inst < (uint64_t)insert_limit(ns) ? inst : (uint64_t)insert_limit(ns)) + nex;
                    if (pc.unit_view && r.vnstr >= 2 && r.words) room_u = (uint32_t)min(r.words / 2, (uint64_t)0x7FFFFFFFu);
                }
            }
            pc.plan_room[c] = room_u; pc.plan_arena[c] = arena;
        }
        rec[c] = r;
    }
}

// ---------------------------------------------------------------------------------------------
// plan_kernel: the work items of a part's SIMPLE clusters, laid out on the device
// ---------------------------------------------------------------------------------------------
// The reference hands a cluster from `cluster_cutter` to `pattern_hasher` through a queue (__main__.py:39-52); here the
// hand-off between the dedup pass and the scan was the host: it waited for the 40-byte records, walked them (key-partition
// estimate, fused class, table size, unit-view room), built the item arrays and work lists and sent them up -- 0.15-0.25 ms
// during which a small batch's GPU idled.  For the clusters that need nothing special -- a view of at most 64 distinct
// sequences (mode 1), ONE key partition by the estimate, a fused finish class, few slow-path rows: 90-100 % of SURVEY 8d's
// clusters -- this kernel does the same arithmetic on the device: item i of the pass = the i-th such cluster (its scratch
// slice is i), work lists heaviest first (counting sort by log2 of the view's windows, as the host's), the unit-view list
// with its pool offsets, and a 40-byte summary the host needs for the launches' grid sizes and the arena.  The host reads
// the summary (one small copy), launches, and builds the REST of the part (several partitions, wide or every-copy views,
// re-runs) the old way while the GPU is busy.
struct PlanOut { uint32_t n_items, n_fin, n_fin2, n_fin5, n_unit, n_unit_small; uint64_t unit_room, arena_cap; };
static_assert(sizeof(PlanOut) == 40, "PlanOut is copied to the host as one 40-byte record");
struct PlanParams {
    ClusterRec* rec;                 // [C]; pad: class bits from cluster_ninst_kernel, PLAN_PLANNED set here
    const uint32_t* plan_room; const uint32_t* plan_arena;            // [C] from cluster_ninst_kernel
    uint32_t c0, c1;                 // the part's clusters
    uint32_t NS, max_items;
    uint32_t* it_cluster; uint32_t* it_nslots;                        // [n] by item
    uint32_t* w_scan; uint32_t* w_fin; uint32_t* w_fin2; uint32_t* w_fin5;    // [n] work lists (item numbers)
    uint32_t* unit_cluster; uint32_t* unit_base;                      // [n] clusters that get a unit view, their pool offsets
    uint32_t* blk;                   // [PLAN_BLK_WORDS per workgroup] counts of a workgroup, then (plan_scan_kernel) its bases
    PlanOut* out;
};
// Three small launches, every workgroup 1 024 clusters: count -> scan over the workgroups (one workgroup) -> scatter.
// (The first cuts did all of it in ONE workgroup with the item numbers as a running count: 0.12-0.28 ms for a part of
// 37 500 clusters on a single CU, the rest of the GPU idle -- more than the host round trip it replaced.)
constexpr uint32_t PLAN_THREADS = 1024;
constexpr uint32_t PLAN_BINS = 66;               // weight classes 64 - log2(windows): 0 .. 64, and one spare
// per workgroup: [0] items [1] unit views [2,3] unit room (u64) [4,5] arena (u64) [6] fin5 items [7] accepted; then the weight
// histograms of the scan / fin / fin2 lists
constexpr uint32_t PLAN_BLK_HIST = 8, PLAN_BLK_WORDS = PLAN_BLK_HIST + 3 * PLAN_BINS;
// `active` lanes count themselves into LDS counter h[bin]; returns the lane's place among the counter's takers (the
// counter's old value + the lane's rank among the lanes of its wave that asked for the same bin).  Wave-aggregated: the
// clusters of a batch have two or three weight classes between them, and 64 lanes' atomics on ONE LDS address run one
// after the other.
__device__ __forceinline__ uint32_t wave_bin_take(uint32_t* h, bool active, uint32_t bin) {
    const uint32_t lane = threadIdx.x & 63;
    uint64_t todo = __ballot(active);
    uint32_t place = 0;
    while (todo) {
        const int leader = __ffsll((unsigned long long)todo) - 1;
        const uint32_t b = __shfl(bin, leader);
        const uint64_t same = __ballot(active && bin == b);
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(&h[b], (uint32_t)__popcll(same));
        base = __shfl(base, leader);
        if (active && bin == b) place = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return place;
}
__device__ __forceinline__ void plan_info(const PlanParams& p, uint32_t i, uint32_t n, uint32_t& info, uint32_t& uroom) {
    info = i < n ? p.rec[p.c0 + i].pad >> PLAN_CLS_SHIFT : 0u;             // nonzero: a simple cluster
    uroom = info ? p.plan_room[p.c0 + i] : 0u;
}
__global__ __launch_bounds__(PLAN_THREADS) void plan_count_kernel(PlanParams p) {
    __shared__ uint32_t cnt[PLAN_BLK_WORDS];
    __shared__ unsigned long long s64[2];
    const uint32_t tid = threadIdx.x, n = p.c1 - p.c0, i = blockIdx.x * PLAN_THREADS + tid;
    for (uint32_t k = tid; k < PLAN_BLK_WORDS; k += PLAN_THREADS) cnt[k] = 0;
    if (tid < 2) s64[tid] = 0;
    __syncthreads();
    uint32_t info, uroom;
    plan_info(p, i, n, info, uroom);
    const uint32_t cls = info & 15u, b = (info >> (PLAN_W_SHIFT - PLAN_CLS_SHIFT)) & 127u;
    (void)wave_bin_take(cnt + PLAN_BLK_HIST, info != 0, b);
    (void)wave_bin_take(cnt + PLAN_BLK_HIST + PLAN_BINS, cls == 1, b);
    (void)wave_bin_take(cnt + PLAN_BLK_HIST + 2 * PLAN_BINS, cls == 2, b);
    (void)wave_bin_take(cnt, info != 0, 0);
    (void)wave_bin_take(cnt, uroom != 0, 1);
    (void)wave_bin_take(cnt, cls == 5, 6);
    uint64_t room = uroom, arena = info ? p.plan_arena[p.c0 + i] : 0u;
    for (int d = 32; d > 0; d >>= 1) { room += __shfl_down(room, d); arena += __shfl_down(arena, d); }
    if ((tid & 63) == 0) {
        if (room) atomicAdd(&s64[0], (unsigned long long)room);
        if (arena) atomicAdd(&s64[1], (unsigned long long)arena);
    }
    __syncthreads();
    uint32_t* o = p.blk + (size_t)blockIdx.x * PLAN_BLK_WORDS;
    for (uint32_t k = tid; k < PLAN_BLK_WORDS; k += PLAN_THREADS) {
        uint32_t v = cnt[k];
        if (k == 2) v = (uint32_t)s64[0]; else if (k == 3) v = (uint32_t)(s64[0] >> 32);
        else if (k == 4) v = (uint32_t)s64[1]; else if (k == 5) v = (uint32_t)(s64[1] >> 32);
        o[k] = v;
    }
}
// counts -> bases, in place.  A sub-batch holds max_items items: workgroups are taken while their items fit, the clusters
// of the others are the host's (whole workgroups, so that the histograms stay those of the accepted clusters).
__global__ __launch_bounds__(256) void plan_scan_kernel(PlanParams p, uint32_t nblk) {
    __shared__ uint32_t sh_acc;                  // accepted workgroups
    __shared__ uint32_t list_tot[3], bin_base[3][PLAN_BINS];
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        uint32_t items = 0, units = 0, fin5 = 0, acc = 0;
        uint64_t room = 0, arena = 0;
        for (uint32_t b = 0; b < nblk; b++) {
            uint32_t* o = p.blk + (size_t)b * PLAN_BLK_WORDS;
            const uint32_t ni = o[0], nu = o[1], n5 = o[6];
            const uint64_t r = o[2] | ((uint64_t)o[3] << 32), a = o[4] | ((uint64_t)o[5] << 32);
            const bool ok = acc == b && items + ni <= p.max_items;
            o[7] = ok ? 1u : 0u;
            if (!ok) continue;
            o[0] = items; o[1] = units; o[2] = (uint32_t)room; o[3] = (uint32_t)(room >> 32); o[6] = fin5;
            items += ni; units += nu; room += r; arena += a; fin5 += n5; acc = b + 1;
        }
        sh_acc = acc;
        PlanOut out;
        out.n_items = items; out.n_fin = 0; out.n_fin2 = 0; out.n_fin5 = fin5; out.n_unit = units; out.n_unit_small = units;
        out.unit_room = room; out.arena_cap = arena;
        *p.out = out;
    }
    __syncthreads();
    const uint32_t acc = sh_acc;
    // every (list, weight class): its count over the accepted workgroups, then the workgroups' bases inside it
    for (uint32_t e = tid; e < 3 * PLAN_BINS; e += 256) {
        uint32_t run = 0;
        for (uint32_t b = 0; b < acc; b++) {
            uint32_t* o = p.blk + (size_t)b * PLAN_BLK_WORDS + PLAN_BLK_HIST + e;
            const uint32_t v = *o; *o = run; run += v;
        }
        bin_base[e / PLAN_BINS][e % PLAN_BINS] = run;           // (for now: the class's total)
    }
    __syncthreads();
    if (tid < 3) {                                               // heaviest class first
        uint32_t run = 0;
        for (uint32_t b = 0; b < PLAN_BINS; b++) { const uint32_t v = bin_base[tid][b]; bin_base[tid][b] = run; run += v; }
        list_tot[tid] = run;
    }
    __syncthreads();
    for (uint32_t e = tid; e < 3 * PLAN_BINS; e += 256) {
        const uint32_t add = bin_base[e / PLAN_BINS][e % PLAN_BINS];
        for (uint32_t b = 0; b < acc; b++) p.blk[(size_t)b * PLAN_BLK_WORDS + PLAN_BLK_HIST + e] += add;
    }
    if (tid == 0) { p.out->n_fin = list_tot[1]; p.out->n_fin2 = list_tot[2]; }
}
__global__ __launch_bounds__(PLAN_THREADS) void plan_scatter_kernel(PlanParams p) {
    __shared__ uint32_t base[PLAN_BLK_WORDS];
    __shared__ uint32_t wave_tot[PLAN_THREADS / 64 + 1];
    const uint32_t tid = threadIdx.x, n = p.c1 - p.c0, i = blockIdx.x * PLAN_THREADS + tid;
    const uint32_t* o = p.blk + (size_t)blockIdx.x * PLAN_BLK_WORDS;
    if (!o[7]) return;                                           // past max_items: the host's clusters (pad keeps no PLAN_PLANNED)
    for (uint32_t k = tid; k < PLAN_BLK_WORDS; k += PLAN_THREADS) base[k] = o[k];
    __syncthreads();
    uint32_t info, uroom;
    plan_info(p, i, n, info, uroom);
    uint32_t tot;
    const uint32_t idx = base[0] + block_exscan(info ? 1u : 0u, wave_tot, &tot);
    const uint32_t uidx = base[1] + block_exscan(uroom ? 1u : 0u, wave_tot, &tot);
    const uint64_t ubase = (base[2] | ((uint64_t)base[3] << 32)) + block_exscan(uroom, wave_tot, &tot);
    const uint32_t cls = info & 15u, nsi = (info >> (PLAN_NS_SHIFT - PLAN_CLS_SHIFT)) & 3u;
    const uint32_t b = (info >> (PLAN_W_SHIFT - PLAN_CLS_SHIFT)) & 127u;
    const uint32_t at0 = wave_bin_take(base + PLAN_BLK_HIST, info != 0, b);
    const uint32_t at1 = wave_bin_take(base + PLAN_BLK_HIST + PLAN_BINS, cls == 1, b);
    const uint32_t at2 = wave_bin_take(base + PLAN_BLK_HIST + 2 * PLAN_BINS, cls == 2, b);
    const uint32_t at5 = wave_bin_take(base, cls == 5, 6);
    if (!info) return;
    const uint32_t c = p.c0 + i;
    p.it_cluster[idx] = c;
    p.it_nslots[idx] = nsi == 1 ? 4096u : nsi == 2 ? 6144u : p.NS;
    p.rec[c].pad = (info << PLAN_CLS_SHIFT) | PLAN_PLANNED;
    p.w_scan[at0] = idx;
    if (cls == 1) p.w_fin[at1] = idx;
    else if (cls == 2) p.w_fin2[at2] = idx;
    else p.w_fin5[at5] = idx;
    if (uroom) { p.unit_cluster[uidx] = c; p.unit_base[uidx] = (uint32_t)ubase; }   // (a part whose pieces pass 2^31 is refused by
}                                                                                   // the host from the summary's unit_room)

// ---------------------------------------------------------------------------------------------
// Text of the output files, written on the device (row N2): the rows are assembled in LDS, one row per thread,
// and leave the workgroup as coalesced 16-byte stores.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ char b64_char(uint32_t v) {
    v &= 63;
    return (char)(v < 26 ? 'A' + v : v < 52 ? 'a' + (v - 26) : v < 62 ? '0' + (v - 52) : v == 62 ? '+' : '/');
}
// binascii.b2a_base64(digest)[:24] for patterns [pid0, pid1)  (panfeed.py:176, 207)
__global__ __launch_bounds__(256) void b64_kernel(const uint8_t* md5, uint32_t pid0, uint32_t pid1, char* out) {
    const uint32_t pid = pid0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= pid1) return;
    const uint8_t* d = md5 + (size_t)pid * 16;
    char* o = out + (size_t)pid * 24;
    for (int i = 0, q = 0; i < 15; i += 3, q += 4) {
        const uint32_t v = ((uint32_t)d[i] << 16) | ((uint32_t)d[i + 1] << 8) | d[i + 2];
        o[q] = b64_char(v >> 18); o[q + 1] = b64_char(v >> 12); o[q + 2] = b64_char(v >> 6); o[q + 3] = b64_char(v);
    }
    const uint32_t v = (uint32_t)d[15] << 16;
    o[20] = b64_char(v >> 18); o[21] = b64_char(v >> 12); o[22] = '='; o[23] = '=';
}

constexpr uint32_t TEXT_TILE = 24576;        // bytes of text a workgroup assembles at a time
constexpr uint32_t TEXT_MAX_ARENAS = 16;
struct KhTextParams {
    // per cluster
    const uint64_t* text_off;        // [C+1] byte offset of the cluster's rows in the text
    const uint32_t* name_off;        // [C+1] into names
    const char* names;
    const uint64_t* kmer_off;        // [C] index of the cluster's first kept k-mer inside its arena
    const uint32_t* kmer_cnt; const uint32_t* cluster_pattern; const uint32_t* cluster_arena;
    const uint32_t* block_cluster;   // [gridDim.x] cluster of the workgroup
    const uint32_t* block_row0;      // [gridDim.x] its first row (0 = the cluster's own row)
    const uint64_t* arena_key[TEXT_MAX_ARENAS]; const uint32_t* arena_pid[TEXT_MAX_ARENAS];
    const char* b64; const char* extra_keys;   // extra_keys: k bytes per slow-path row of the batch
    char* text;
    uint32_t k, KW, rows_per_block;
};

// kmers_to_hashes.tsv: "<idx>\t\t<hash>\n" then "<idx>\t<k-mer>\t<hash>\n" per kept k-mer  (panfeed.py:177, 208)
__global__ __launch_bounds__(256) void kh_text_kernel(KhTextParams p) {
    __shared__ __align__(16) char tile[TEXT_TILE + 32];
    const uint32_t c = p.block_cluster[blockIdx.x], row0 = p.block_row0[blockIdx.x];
    const uint32_t n0 = p.name_off[c], L = p.name_off[c + 1] - n0;
    const uint32_t k = p.k, cnt = p.kmer_cnt[c];
    const uint32_t head = L + 2 + 24 + 1, rowlen = L + 1 + k + 1 + 24 + 1;
    const uint32_t nrows_total = cnt + 1;                                   // row 0 is the cluster's own row
    const uint32_t nrows = min(p.rows_per_block, nrows_total - row0);
    // byte range of these rows inside the cluster's text
    const uint64_t b0 = row0 == 0 ? 0 : head + (uint64_t)(row0 - 1) * rowlen;
    const uint64_t b1 = head + (uint64_t)(row0 + nrows - 1) * rowlen;
    const uint64_t gbase = p.text_off[c] + b0;
    const uint32_t mis = (uint32_t)(gbase & 15);                            // tile[mis + i] <-> text[gbase + i]
    const uint32_t a = p.cluster_arena[c];
    const uint64_t* keys = p.arena_key[a];
    const uint32_t* pids = p.arena_pid[a];
    const uint64_t koff = p.kmer_off[c];
    for (uint32_t r = threadIdx.x; r < nrows; r += blockDim.x) {
        const uint32_t row = row0 + r;
        char* w = tile + mis + (row == 0 ? 0 : head + (uint64_t)(row - 1) * rowlen - b0);
        for (uint32_t i = 0; i < L; i++) w[i] = p.names[n0 + i];
        w += L;
        *w++ = '\t';
        uint32_t pid;
        if (row == 0) {
            pid = p.cluster_pattern[c];
        } else {
            const uint64_t j = koff + (row - 1);
            const uint64_t k0 = keys[j * p.KW];
            if (k0 >> 63) {                                                  // a slow-path row: its text is at hand
                const char* e = p.extra_keys + (size_t)(uint32_t)k0 * k;
                for (uint32_t q = 0; q < k; q++) w[q] = e[q];
            } else {
                // bit b of the 2k-bit value lives in 63-bit word KW - 1 - b / 63, at bit b % 63
                for (uint32_t q = 0; q < k; q++) {
                    const uint32_t b0 = 2 * (k - 1 - q), b1 = b0 + 1;
                    const uint64_t w0 = keys[j * p.KW + (p.KW - 1 - b0 / 63)], w1 = keys[j * p.KW + (p.KW - 1 - b1 / 63)];
                    const uint32_t code = (uint32_t)((w0 >> (b0 % 63)) & 1) | ((uint32_t)((w1 >> (b1 % 63)) & 1) << 1);
                    w[q] = (char)(0x54474341u >> (8 * code));               // "ACGT"
                }
            }
            w += k;
            pid = pids[j];
        }
        *w++ = '\t';
        const char* h = p.b64 + (size_t)pid * 24;
        for (uint32_t i = 0; i < 24; i++) w[i] = h[i];
        w[24] = '\n';
    }
    __syncthreads();
    // out: the unaligned head and tail byte by byte, the middle as 16-byte pieces
    const uint32_t nbytes = (uint32_t)(b1 - b0);
    char* g = p.text + gbase;
    const uint32_t lead = min(nbytes, (16 - mis) & 15);
    for (uint32_t i = threadIdx.x; i < lead; i += blockDim.x) g[i] = tile[mis + i];
    const uint32_t mid = (nbytes - lead) >> 4;
    const uint4* src = reinterpret_cast<const uint4*>(tile + mis + lead);
    uint4* dst = reinterpret_cast<uint4*>(g + lead);
    for (uint32_t i = threadIdx.x; i < mid; i += blockDim.x) dst[i] = src[i];
    for (uint32_t i = lead + (mid << 4) + threadIdx.x; i < nbytes; i += blockDim.x) g[i] = tile[mis + i];
}

// ---------------------------------------------------------------------------------------------
// kmers.tsv on the device: the positional rows of target strains (panfeed.py:90-107)
// ---------------------------------------------------------------------------------------------
// One row per window of a target strain's sequence (two in non-canonical mode):
//   "<idx>\t<strain>\t<feature_id>\t<contig>\t<feature_strand>\t<contig_start>\t<contig_end>\t<gene_start>\t<gene_end>\t<strand>\t<k-mer>\n"
// The first five fields are the same for every row of a sequence (the host sends them once per sequence, as text); the
// four coordinates follow from the window's position (panfeed.py:91-102), the strand used from the strand bits
// strand_bits_kernel left for the batch, the k-mer's letters from the packed bases.  A workgroup takes KT_ROWS
// consecutive rows of one sequence: kt_len_kernel adds up their bytes (the rows' lengths vary with the digits of their
// numbers), the host turns the tiles' sizes into offsets, kt_text_kernel writes the rows into an LDS tile -- every thread
// its own row, at the place a block scan of the lengths gives it -- and the tile leaves as coalesced 16-byte stores.
constexpr uint32_t KT_ROWS = 256;            // rows per workgroup (one per thread)
constexpr uint32_t KT_TILE = 61440;          // bytes of LDS tile: a sequence whose rows may be longer than KT_TILE / KT_ROWS stays on the host
struct KtSeq {
    int64_t base;                            // seq.start (feature strand > 0) or seq.end: panfeed.py:91-99
    int64_t offset;                          // panfeed.py:101-102
    int64_t strand;                          // feature strand as the record holds it
    uint32_t seg;                            // its segment in the batch (pure A/C/G/T: one segment, every window)
    uint32_t nk;                             // windows
    uint32_t prefix_off, prefix_len;         // "<idx>\t<strain>\t<feature_id>\t<contig>\t<feature_strand>\t"
};
struct KtParams {
    const KtSeq* seqs; const uint2* tiles;   // tile = (sequence, first row)
    const char* prefix;
    const uint64_t* packed; const uint64_t* seg_word_off; const uint32_t* seg_strand_off; const uint64_t* strand_bits;
    uint32_t* tile_bytes;                    // kt_len_kernel out
    const uint64_t* tile_off; char* text;    // kt_text_kernel in / out
    uint32_t k, canon;
};
__device__ __forceinline__ uint32_t kt_dec_len(int64_t v) {
    uint64_t a = v < 0 ? 0ull - (uint64_t)v : (uint64_t)v;
    uint32_t n = v < 0 ? 2u : 1u;
    while (a > 0xFFFFFFFFull) { a /= 10; n++; }
    uint32_t b = (uint32_t)a;
    while (b >= 10) { b /= 10; n++; }
    return n;
}
__device__ __forceinline__ char* kt_put_dec(char* w, int64_t v) {
    uint64_t a = v < 0 ? 0ull - (uint64_t)v : (uint64_t)v;
    if (v < 0) *w++ = '-';
    char t[20];
    int n = 0;
    while (a > 0xFFFFFFFFull) { t[n++] = (char)('0' + a % 10); a /= 10; }
    uint32_t b = (uint32_t)a;
    do { t[n++] = (char)('0' + b % 10); b /= 10; } while (b);
    while (n) *w++ = t[--n];
    return w;
}
struct KtRow { int64_t ts, te, gs, ge, us; uint32_t pos; bool rc; };
__device__ __forceinline__ bool kt_row(const KtParams& p, const KtSeq& s, uint32_t row, KtRow& r) {
    const uint32_t reps = p.canon ? 1u : 2u;
    if (row >= s.nk * reps) return false;
    const uint32_t pos = row / reps;
    const int64_t k = p.k;
    r.pos = pos;
    if (s.strand > 0) { r.ts = s.base + pos; r.te = s.base + pos + k; }          // panfeed.py:91-94
    else { r.te = s.base - pos; r.ts = s.base - pos - k; }                        // panfeed.py:96-99
    r.gs = (int64_t)pos - s.offset; r.ge = (int64_t)pos + k - s.offset;          // panfeed.py:101-102
    if (p.canon) {
        const uint64_t w = p.strand_bits[(size_t)p.seg_strand_off[s.seg] + (pos >> 6)];
        r.rc = (w >> (pos & 63)) & 1;
        r.us = r.rc ? -1 : 1;                                                     // panfeed.py:69-75
    } else {
        r.rc = (row & 1u) != 0;                                                   // panfeed.py:106-107
        r.us = r.rc ? -s.strand : s.strand;
    }
    return true;
}
__device__ __forceinline__ uint32_t kt_row_len(const KtParams& p, const KtSeq& s, const KtRow& r) {
    return s.prefix_len + kt_dec_len(r.ts) + kt_dec_len(r.te) + kt_dec_len(r.gs) + kt_dec_len(r.ge) + kt_dec_len(r.us) + 5 + p.k + 1;
}
__global__ __launch_bounds__(KT_ROWS) void kt_len_kernel(KtParams p) {
    __shared__ uint32_t wsum[KT_ROWS / 64];
    const uint2 t = p.tiles[blockIdx.x];
    const KtSeq s = p.seqs[t.x];
    KtRow r;
    uint32_t len = kt_row(p, s, t.y + threadIdx.x, r) ? kt_row_len(p, s, r) : 0u;
    for (int d = 32; d > 0; d >>= 1) len += __shfl_down(len, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = len;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t a = 0; for (uint32_t i = 0; i < KT_ROWS / 64; i++) a += wsum[i]; p.tile_bytes[blockIdx.x] = a; }
}
__global__ __launch_bounds__(KT_ROWS) void kt_text_kernel(KtParams p) {
    __shared__ __align__(16) char tile[KT_TILE + 32];
    __shared__ uint32_t wsum[KT_ROWS / 64 + 1];
    const uint2 t = p.tiles[blockIdx.x];
    const KtSeq s = p.seqs[t.x];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    KtRow r;
    const bool have = kt_row(p, s, t.y + threadIdx.x, r);
    const uint32_t len = have ? kt_row_len(p, s, r) : 0u;
    uint32_t x = len;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t before = x - len, total = 0;
    for (uint32_t i = 0; i < KT_ROWS / 64; i++) { if (i < wave) before += wsum[i]; total += wsum[i]; }
    const uint64_t gbase = p.tile_off[blockIdx.x];
    const uint32_t mis = (uint32_t)(gbase & 15);                             // tile[mis + i] <-> text[gbase + i]
    if (have && before + len <= KT_TILE) {
        char* w = tile + mis + before;
        const char* pre = p.prefix + s.prefix_off;
        for (uint32_t i = 0; i < s.prefix_len; i++) w[i] = pre[i];
        w += s.prefix_len;
        w = kt_put_dec(w, r.ts); *w++ = '\t';
        w = kt_put_dec(w, r.te); *w++ = '\t';
        w = kt_put_dec(w, r.gs); *w++ = '\t';
        w = kt_put_dec(w, r.ge); *w++ = '\t';
        w = kt_put_dec(w, r.us); *w++ = '\t';
        // the window's letters (forward), or its reverse complement's: base i of the segment is bits 63-2(i%32)..62-2(i%32)
        const uint64_t* words = p.packed + p.seg_word_off[s.seg];
        const uint32_t k = p.k;
        for (uint32_t q = 0; q < k; q++) {
            const uint32_t i = r.rc ? r.pos + k - 1 - q : r.pos + q;
            uint32_t code = (uint32_t)(words[i >> 5] >> (62 - 2 * (i & 31))) & 3u;
            if (r.rc) code = 3u - code;
            w[q] = (char)(0x54474341u >> (8 * code));                        // "ACGT"
        }
        w[k] = '\n';
    }
    __syncthreads();
    const uint32_t nbytes = min(total, KT_TILE);
    char* g = p.text + gbase;
    const uint32_t lead = min(nbytes, (16 - mis) & 15);
    for (uint32_t i = threadIdx.x; i < lead; i += blockDim.x) g[i] = tile[mis + i];
    const uint32_t mid = (nbytes - lead) >> 4;
    const uint4* src = reinterpret_cast<const uint4*>(tile + mis + lead);
    uint4* dst = reinterpret_cast<uint4*>(g + lead);
    for (uint32_t i = threadIdx.x; i < mid; i += blockDim.x) dst[i] = src[i];
    for (uint32_t i = lead + (mid << 4) + threadIdx.x; i < nbytes; i += blockDim.x) g[i] = tile[mis + i];
}

struct HpTextParams {
    const uint32_t* order;           // [n] new pattern ids in first-seen order
    const uint64_t* row_off;         // [n+1] byte offsets of the rows
    const uint32_t* pat_bits; const uint32_t* pat_nan; const uint32_t* pat_n;
    const char* b64;
    char* text;
    uint32_t n, W;
};
// bytes of a hashes_to_patterns row: 24 + one tab per cell + one digit per non-NaN cell + newline
// pattern i of the list: order[i], or pid0 + i without a list
__global__ __launch_bounds__(256) void hp_rowlen_kernel(const uint32_t* pat_n, const uint32_t* pat_nan, uint32_t W,
                                                        uint32_t pid0, const uint32_t* order, uint32_t n, uint32_t* len) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t pid = order ? order[i] : pid0 + i, nk = pat_n[pid], cells = nk & 0x7FFFFFFFu;
    uint32_t nn = 0;
    if (pat_nan && !(nk >> 31))
        for (uint32_t w = 0; w < W; w++) nn += __popc(pat_nan[(size_t)pid * W + w]);
    len[i] = 24 + cells + (cells - nn) + 1;
}
// hashes_to_patterns.tsv: "<hash>\t0\t1...\n", '' for a NaN cell  (panfeed.py:181-187, 217-223); one workgroup per row
__global__ __launch_bounds__(256) void hp_text_kernel(HpTextParams p) {
    __shared__ uint32_t pre[257];
    const uint32_t i = blockIdx.x;
    const uint32_t pid = p.order[i], nk = p.pat_n[pid], cells = nk & 0x7FFFFFFFu;
    const bool use_nan = p.pat_nan && !(nk >> 31);
    const uint32_t* bits = p.pat_bits + (size_t)pid * p.W;
    const uint32_t* nan = use_nan ? p.pat_nan + (size_t)pid * p.W : nullptr;
    char* g = p.text + p.row_off[i];
    if (threadIdx.x < 24) g[threadIdx.x] = p.b64[(size_t)pid * 24 + threadIdx.x];
    // NaN cells before each 32-cell word (only with a NaN mask)
    const uint32_t nwords = (cells + 31) >> 5;
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t w = 0; w < nwords && w < 256; w++) { pre[w] = run; run += nan ? __popc(nan[w]) : 0; }
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < cells; e += blockDim.x) {
        const uint32_t w = e >> 5, b = e & 31;
        uint32_t before = 0;
        bool isn = false;
        if (nan) {
            const uint32_t nw = nan[w];
            before = (w < 256 ? pre[w] : 0) + __popc(nw & ((1u << b) - 1));
            isn = (nw >> b) & 1;
        }
        char* q = g + 24 + 2 * (size_t)e - before;                          // every earlier NaN cell is one byte shorter
        q[0] = '\t';
        if (!isn) q[1] = ((bits[w] >> b) & 1) ? '1' : '0';
    }
    if (threadIdx.x == 0) g[p.row_off[i + 1] - p.row_off[i] - 1] = '\n';
}

// ---------------------------------------------------------------------------------------------
// Genomes resident in HBM (row N1 on the device): contigs as 2 bits per base, segments gathered from them
// ---------------------------------------------------------------------------------------------
// A contig sits at an even word offset of the store, 32 bases per word (first base in bits 63:62), followed by at
// least two zero words.  Non-ACGT bytes get an arbitrary code: the host knows where they are and never asks for a
// range that contains one.
struct PackPiece {
    uint64_t ascii_off;      // byte offset of the piece in the staged ASCII block (multiple of 32)
    uint64_t dst_word;       // first word of the piece in the store
    uint32_t nbases;         // bases of the piece (a multiple of 32 unless it ends its contig)
    uint32_t nwords;         // words to write: ceil(nbases / 32) plus the contig's zero padding behind its last piece
    uint32_t block0;         // first 256-thread block of the piece
    uint32_t pad;
};

__device__ __forceinline__ uint32_t codes16(uint64_t b) {
    // eight ASCII bytes -> eight 2-bit codes, first byte most significant: A 0, C 1, G 2, T 3 = ((c >> 1) ^ (c >> 2)) & 3
    const uint64_t x = ((b >> 1) ^ (b >> 2)) & 0x0303030303030303ull;
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) r |= (uint32_t)((x >> (8 * j)) & 3u) << (14 - 2 * j);
    return r;
}

__global__ __launch_bounds__(256) void genome_pack_kernel(const uint8_t* ascii, const PackPiece* pieces, uint32_t npieces,
                                                          uint64_t* store) {
    uint32_t a = 0, b = npieces;                         // last piece with block0 <= blockIdx.x
    while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (pieces[m].block0 <= blockIdx.x) a = m; else b = m; }
    const PackPiece pc = pieces[a];
    const uint32_t w = (blockIdx.x - pc.block0) * 256 + threadIdx.x;
    if (w >= pc.nwords) return;
    uint64_t out = 0;
    const uint64_t base = (uint64_t)w * 32;
    if (base < pc.nbases) {
        const ulonglong2* src = reinterpret_cast<const ulonglong2*>(ascii + pc.ascii_off + base);
        const ulonglong2 lo = src[0], hi = src[1];
        out = ((uint64_t)codes16(lo.x) << 48) | ((uint64_t)codes16(lo.y) << 32) | ((uint64_t)codes16(hi.x) << 16) |
              (uint64_t)codes16(hi.y);
        const uint32_t nb = (uint32_t)min((uint64_t)32, pc.nbases - base);
        if (nb < 32) out &= ~0ull << (2 * (32 - nb));
    }
    store[pc.dst_word + w] = out;
}

// One-pass ingest (pf_pangenome_open_device): a contig's letters as they lie in the FASTA text -- wrapped at `width` letters
// with `eol` bytes between lines, upper or lower case -- to 2 bits per base.  Letter j is byte text_off + j + (j / width) * eol.
struct TextPiece {
    uint64_t text_off;       // byte offset of the contig's first letter in the block
    uint64_t dst_word;       // first word of the contig in the store
    uint32_t nbases;
    uint32_t nwords;         // words to write: ceil(nbases / 32) plus the contig's zero padding
    uint32_t width, eol;     // width 0: not wrapped
    uint32_t block0;         // first 256-thread block of the piece
    uint32_t pad;
};
__global__ __launch_bounds__(256) void genome_pack_text_kernel(const uint8_t* text, const TextPiece* pieces, uint32_t npieces,
                                                               uint64_t* store) {
    uint32_t a = 0, b = npieces;                         // last piece with block0 <= blockIdx.x
    while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (pieces[m].block0 <= blockIdx.x) a = m; else b = m; }
    const TextPiece pc = pieces[a];
    const uint32_t w = (blockIdx.x - pc.block0) * 256 + threadIdx.x;
    if (w >= pc.nwords) return;
    uint64_t out = 0;
    const uint64_t j0 = (uint64_t)w * 32;
    if (j0 < pc.nbases) {
        const uint32_t nb = (uint32_t)min((uint64_t)32, pc.nbases - j0);
        uint64_t at = pc.text_off + j0;
        uint32_t col = 0, width = pc.width ? pc.width : 0xFFFFFFFFu;
        if (pc.width) { const uint64_t line = j0 / pc.width; at += line * pc.eol; col = (uint32_t)(j0 - line * pc.width); }
        for (uint32_t q = 0; q < nb; q++) {
            const uint32_t ch = text[at];
            out |= (uint64_t)(((ch >> 1) ^ (ch >> 2)) & 3u) << (62 - 2 * q);      // A/a 0, C/c 1, G/g 2, T/t 3
            at++;
            if (++col == width) { col = 0; at += pc.eol; }
        }
    }
    store[pc.dst_word + w] = out;
}

struct GatherParams {
    const uint64_t* store;        // genome store
    const uint64_t* literal;      // packed words the host prepared itself (sequences with non-ACGT bases, target strains)
    const uint64_t* src_off;      // [n_segs] word offset of the source inside its buffer
    const uint32_t* src_start;    // [n_segs] lowest base coordinate of the range inside the source
    const uint32_t* src_flags;    // [n_segs] bit 0: source is `literal`; bit 1: reverse complement
    const uint64_t* seg_word_off; const uint32_t* seg_len;
    uint64_t* packed;
    uint32_t n_segs;
};

// 16 lanes per segment, one destination word per lane and trip
__global__ __launch_bounds__(256) void gather_segments_kernel(GatherParams p) {
    const uint32_t seg = blockIdx.x * 16 + (threadIdx.x >> 4), gl = threadIdx.x & 15;
    if (seg >= p.n_segs) return;
    const uint32_t len = p.seg_len[seg], flags = p.src_flags[seg], start = p.src_start[seg];
    const uint64_t* src = ((flags & 1u) ? p.literal : p.store) + p.src_off[seg];
    uint64_t* dst = p.packed + p.seg_word_off[seg];
    const uint32_t nw = 2 * ((len + 63) >> 6);
    for (uint32_t w = gl; w < nw; w += 16) {
        uint64_t out = 0;
        if (w * 32 < len) {
            const uint32_t nb = min(32u, len - w * 32);
            if (!(flags & 2u)) {
                const uint32_t pos = start + w * 32, i = pos >> 5, sh = (pos & 31) << 1;
                out = src[i] << sh;
                if (sh) out |= src[i + 1] >> (64 - sh);
            } else {
                // destination base j = complement of source base start + len - 1 - j
                const int64_t qhi = (int64_t)start + len - 1 - (int64_t)w * 32, ws = qhi - 31;
                uint64_t y;
                if (ws >= 0) {
                    const uint32_t i = (uint32_t)(ws >> 5), sh = ((uint32_t)ws & 31) << 1;
                    y = src[i] << sh;
                    if (sh) y |= src[i + 1] >> (64 - sh);
                } else {
                    y = src[0] >> (2 * (uint32_t)(-ws));
                }
                out = rev_groups(~y);
            }
            if (nb < 32) out &= ~0ull << (2 * (32 - nb));
        }
        dst[w] = out;
    }
}

__global__ void fill_u64_kernel(uint64_t* p, uint64_t v, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = v;
}

// bench helper: copy allele words into per-sample segments, 16 bytes per lane
__global__ void synth_expand_kernel(const uint64_t* allele_words, const uint64_t* allele_word_off,
                                    const uint32_t* seg_allele, const uint64_t* seg_word_off,
                                    const uint32_t* seg_len, uint32_t n_segs, uint64_t* packed) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t s = wave; s < n_segs; s += nwaves) {
        const uint32_t pieces = (seg_len[s] + 63) >> 6;     // 16-byte pieces
        const ulonglong2* src = reinterpret_cast<const ulonglong2*>(allele_words + allele_word_off[seg_allele[s]]);
        ulonglong2* dst = reinterpret_cast<ulonglong2*>(packed + seg_word_off[s]);
        for (uint32_t i = lane; i < pieces; i += 64) dst[i] = src[i];
    }
}

}  // namespace pf
