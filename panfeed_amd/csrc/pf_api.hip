// pf_api.hip -- C ABI (include/panfeed_hip.h) over the kernels in pf_kernels.h.
//
// Host orchestration of one batch (pf_submit):
//   dedup (identical sequences) -> per-cluster counts back to the host -> unit classes (identical units) ->
//   work items (cluster x key partition, plus prebuilt items for slow-path rows) -> sub-batches of <= max_items
//   items, each: scan -> fused finish, or rows -> base -> emit -> pattern rows; clusters whose LDS table overflowed are
//   re-run with the partitions the failed scan asked for; MD5 of new patterns last.  A batch of >= 8 192 clusters goes
//   through in two parts so that the host builds a part's items while the GPU is on the other.
// Everything is stream-ordered on one HIP stream; the host syncs are the dedup results per part, the overflow / cursor /
// pattern-counter read-back of the last pass and the end of the batch.
#include "pf_host.h"
#include "pf_kernels.h"
#include "../../include/panfeed_hip.h"
#include "pf_ingest.h"

#include <sys/stat.h>
#include <condition_variable>
#include <mutex>
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <thread>
#include <string>
#include <atomic>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(e_ == hipErrorOutOfMemory ? PF_ERR_OOM : PF_ERR_HIP, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                             \
    } while (0)
#define PFCHK(expr)             \
    do {                        \
        int r_ = (expr);        \
        if (r_ != PF_OK) return r_; \
    } while (0)

// test hook (pf_debug_limit_alloc): single device allocations above the limit are refused as if the device were out of
// memory; the largest request and the exact-size retries that succeeded are counted
std::atomic<uint64_t> g_alloc_limit{0}, g_alloc_max_request{0}, g_alloc_exact_retries{0};

hipError_t dev_malloc(void** p, size_t bytes) {
    const uint64_t lim = g_alloc_limit.load(std::memory_order_relaxed);
    if (lim && bytes > lim) { *p = nullptr; return hipErrorOutOfMemory; }
    return hipMalloc(p, bytes);
}

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool view = false;     // points into another DevBuf (staged uploads): never freed, never grown
    int ensure(size_t bytes) {
        if (view) { p = nullptr; cap = 0; view = false; }
        if (bytes <= cap) return PF_OK;
        const bool regrow = p != nullptr;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        uint64_t seen = g_alloc_max_request.load(std::memory_order_relaxed);
        while (bytes > seen && !g_alloc_max_request.compare_exchange_weak(seen, bytes)) {}
        // (a buffer that has to be re-made gets a quarter of slack: hipFree + hipMalloc of a multi-gigabyte buffer was seen to
        // take 0.25 s in the middle of a submit when a batch's key-partition queues came out a little larger than the batch
        // before's; a first allocation -- the scratch slices are 123 GB in bench.py -- gets a sixteenth).  The slack is a
        // convenience, never a requirement: when it does not fit, the exact size is asked for before giving up.
        size_t want = bytes + (regrow ? bytes / 4 : bytes / 16) + 256;
        hipError_t e = dev_malloc(&p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            want = bytes;
            e = dev_malloc(&p, want);
            if (e == hipSuccess) g_alloc_exact_retries.fetch_add(1, std::memory_order_relaxed);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            return fail(PF_ERR_OOM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        }
        cap = want;
        return PF_OK;
    }
    void release() { if (p && !view) (void)hipFree(p); p = nullptr; cap = 0; view = false; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct Item {
    uint32_t cluster, part, nparts, nslots, slice, sib0, nsib, extra_first, is_extra;
};

struct Arena {
    DevBuf key, pid, first;
    uint64_t cap = 0;        // entries
    uint64_t base = 0;       // global index of entry 0
    uint64_t used = 0;
};

struct EvPair { hipEvent_t a, b; int cat; };

}  // namespace

struct pf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;            // the fused finish kernels of a SMALL launch run here, beside the general path's kernels
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    pf_opts o{};
    int KW = 1;
    uint32_t NS = 0, W = 0, max_items = 0;
    std::vector<uint32_t> maf_lo, maf_hi;
    DevBuf d_maf_lo, d_maf_hi;
    // pattern table + pool
    pf::PatternTable pt{};
    DevBuf pt_lo, pt_val, pt_first, pt_counters, pat_bits, pat_nan, pat_n, pat_md5;
    // what this context has seen of its clusters of many distinct sequences: g = new k-mers per further sequence against
    // L = windows of one sequence, as running sums for the line g = a + b L (related alleles: a few tens whatever L is;
    // SURVEY 8d's: flanks + a share of L) -- see the key-partition estimate
    double reg_n = 0, reg_x = 0, reg_y = 0, reg_xx = 0, reg_xy = 0, reg_yy = 0;
    uint64_t n_submits = 0;
    std::vector<uint32_t> hs_count, hs_plan;
    uint32_t n_patterns = 0;       // patterns allocated after the last submit
    uint32_t pid0 = 0;             // first pattern id of the last submit
    // scratch slices
    DevBuf tab_key, tab_ord, chunkbits, chunkmask, slot_hash, sorted_pair, kept_prefix;
    // uploaded batch (when the caller passes host pointers)
    DevBuf b_packed, b_seg_word_off, b_seg_len, b_seg_sample, b_seg_ord, b_cl_seg_off, b_cl_nstr, b_cl_npres,
        b_cl_presab, b_cl_ordinal, b_extra_ord, b_extra_bits, b_seg_strand_off;
    // per batch device arrays
    DevBuf cl_rec, cl_overflow, cl_kmer_off, cl_kmer_cnt, cl_unique, cl_pattern, cl_first, cursor;
    // scan view built by cluster_dedup_kernel
    DevBuf v_word_off, v_len, v_sample, v_ord, seg_distinct, v_nseg, v_nstr, v_mode, v_dense, extra_off, extra_dense;
    DevBuf bm4, bm2, mrows, slot_out, it_is_extra, cmask_lo, cmask_hi, it_compact;
    DevBuf strand_bits, scan_desc, md5_list, wide_list;
    DevBuf v_bits, view_off;
    // unit view (unit_class_kernel): one pool of view entries per part of a batch's first pass, and the list
    // {clusters, their places in the pool} that goes up with it
    struct UPool { DevBuf word_off, len, sample, ord, bits, list; uint32_t* pin = nullptr; size_t pin_cap = 0; };
    DevBuf pat_b64, txt_dev, txt_meta;   // device-side rendering: base64 of every digest, the text, its per-row tables
    uint32_t b64_done = 0;               // patterns whose base64 is in pat_b64
    char* txt_pins[2] = {nullptr, nullptr};   // pinned host copies of the rendered text, used alternately so that a
    size_t txt_pin_caps[2] = {0, 0};          // writer thread may still be on the previous batch's
    int txt_slot = 0;
    // kmers.tsv written on the device (pf_render_kmers_tsv_device): descriptors, tiles, the text; pinned blocks for its way out
    DevBuf kt_seqs, kt_tiles, kt_prefix, kt_tbytes, kt_toff, kt_text;
    uint64_t kt_bytes = 0;
    uint32_t kt_host_seqs = 0;
    char* kt_pins[2] = {nullptr, nullptr};
    size_t kt_pin_caps[2] = {0, 0};
    bool kt_pref_valid = false;
    uint64_t kt_pref_off = 0, kt_pref_n = 0;
    int kt_pref_slot = 0;
    pf_batch last{};                      // the last pf_submit's batch arrays as device pointers (valid until the next submit)
    uint32_t last_nseg = 0;
    DevBuf g_store, b_literal, g_src_off, g_src_start, g_src_flags;   // genomes resident in HBM + per-batch gather lists
    uint64_t g_words = 0;
    const pf_gather* pending_gather = nullptr;
    // host scratch of pf_submit, kept between calls (capacity persists: no allocation / page faults in steady state)
    std::vector<Item> hs_items;
    std::vector<uint8_t> hs_fused;
    std::vector<uint32_t> hs_v[10], hs_w[7], hs_sub[3];
    int n_cu = 256;
    DevBuf it_binned, bin_lists, q_key, q_ord, q_bit, q_off;      // key-partition queues of binned clusters (bin_kernel)
    std::vector<uint32_t> hs_binned, hs_bin[4];
    DevBuf it_cluster, it_part, it_nparts, it_nslots, it_slice, it_sib0, it_nsib, it_extra_first, it_count,
        it_unique, it_kept, work_scan, work_extra, work_fin, work_fin2, work_fin3, work_fin5, work_rows, sub_cluster, sub_item0, sub_nitems;
    std::vector<Arena*> arenas;
    uint32_t n_passes = 0;                 // arenas the last pf_submit used (arenas[] itself only ever grows)
    DevBuf rp_order, rp_rlen, rp_rowoff;   // pf_render_pattern_rows: the id list, row lengths, row offsets
    uint32_t n_grown = 0;                  // times the pattern table / pool were enlarged
    uint32_t n_scratch_grown = 0;          // times the scratch slices were re-made for a cluster of more items than max_items
    uint64_t pt_slot_limit = 0;            // test hook (pf_debug_limit_pattern_slots): allocations above it fail as if out of memory
    uint32_t pregrow_failed_pool = 0;      // pool size at which growing ahead of need failed: not tried again at this size
    bool pt_stale = false;                 // a batch failed and its patterns could not be dropped (growth failed): reset first
    DevBuf mg_lo, mg_cnt;   // pf_merge_patterns scratch table ([cap][4] words) and its counter
    // the small per-pass arrays: one device block + its pinned host mirror, two of each because the two halves of a
    // batch's first pass are in flight together (stage_slot picks the pair)
    DevBuf stage_devs[2];
    void* stage_pins[2] = {nullptr, nullptr};
    size_t stage_pin_caps[2] = {0, 0};
    int stage_slot = 0;
    void* pin_dedup = nullptr;     // pinned host copies of the per-cluster arrays the dedup kernel leaves
    size_t pin_dedup_cap = 0;
    uint64_t* pin_small = nullptr; // pinned scratch: cursor values going up [0..15], cursor read-backs of deferred passes [16..47],
                                   // the last pass's cursor triple [48..50] and pattern counters [52..53]
    uint32_t* pin_ovf = nullptr;   // pinned: the per-cluster overflow words of the last pass (a pageable destination makes
    size_t pin_ovf_cap = 0;        // hipMemcpyAsync a staged, blocking copy)
    static constexpr int MAX_PARTS = 8;
    UPool upool[2 * MAX_PARTS];            // [2 h]: the device-planned clusters of part h, [2 h + 1]: the host-planned rest
    hipEvent_t ev_part[MAX_PARTS] = {};    // a part's dedup results have arrived in pinned memory
    // plan_kernel's output per part: item arrays, work lists, unit-view list (one device block), its 40-byte summary in
    // pinned memory; and the constant item arrays (zeros | ones | 0, 1, 2, ...) every device-planned pass shares
    struct DPlan { DevBuf block, it_count, out; pf::PlanOut* pin_out = nullptr; uint32_t n = 0; };
    DPlan dplan[MAX_PARTS];
    DevBuf dp_const, plan_room, plan_arena;
    uint32_t dp_const_n = 0;
    hipEvent_t ev_stage[2] = {nullptr, nullptr};   // a staging slot's upload has left the pinned block

    // last batch bookkeeping
    bool have_batch = false;
    bool h_strand_fresh = false;        // h_strand holds the strand bits of the last submit
    uint32_t n_clusters = 0;
    uint64_t n_strand_words = 0;
    std::vector<uint32_t> cluster_arena;   // arena index per cluster
    pf_result counters{};
    pf_timing timing{};
    std::vector<EvPair> events;
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    // host result storage
    std::vector<uint64_t> h_kmer_off, h_kmer_key, h_first_seen, h_strand;
    std::vector<uint32_t> h_kmer_cnt, h_cl_pattern, h_cl_unique, h_kmer_pid, h_new_pid, h_pat_bits, h_pat_nan, h_pat_n;
    std::vector<uint8_t> h_pat_md5;
    std::vector<char> h_b64;       // 24 chars per pattern
};

namespace {

int get_event(pf_ctx* c, hipEvent_t* ev) {
    if (!c->ev_pool.empty()) { *ev = c->ev_pool.back(); c->ev_pool.pop_back(); return PF_OK; }
    HIPCHK(hipEventCreate(ev));
    return PF_OK;
}
// a timed stretch of one category on `s` (the context's stream unless the launches go to the side stream); the pairs are
// read after the batch's last synchronisation, when both streams have drained
int mark_begin(pf_ctx* c, int cat, hipStream_t s = nullptr) {
    EvPair e; e.cat = cat;
    PFCHK(get_event(c, &e.a));
    PFCHK(get_event(c, &e.b));
    HIPCHK(hipEventRecord(e.a, s ? s : c->stream));
    c->events.push_back(e);
    return PF_OK;
}
int mark_end(pf_ctx* c, hipStream_t s = nullptr) {
    HIPCHK(hipEventRecord(c->events.back().b, s ? s : c->stream));
    return PF_OK;
}

template <class T>
int upload(pf_ctx* c, DevBuf& b, const T* src, size_t n, const T** out) {
    PFCHK(b.ensure(std::max<size_t>(n, 1) * sizeof(T)));
    if (n) HIPCHK(hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    *out = b.as<T>();
    return PF_OK;
}
template <class T>
int upload_vec(pf_ctx* c, DevBuf& b, const std::vector<T>& v) {
    const T* dummy;
    return upload(c, b, v.data(), v.size(), &dummy);
}

int fill_u64(pf_ctx* c, void* p, uint64_t v, uint64_t n) {
    if (!n) return PF_OK;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(pf::fill_u64_kernel, dim3(blocks), dim3(256), 0, c->stream, (uint64_t*)p, v, n);
    HIPCHK(hipGetLastError());
    return PF_OK;
}

int reset_patterns(pf_ctx* c) {
    PFCHK(fill_u64(c, c->pt_lo.p, pf::EMPTY64, c->pt.cap));
    PFCHK(fill_u64(c, c->pt_val.p, pf::EMPTY64, c->pt.cap));
    PFCHK(fill_u64(c, c->pt_first.p, pf::EMPTY64, c->pt.pool));
    HIPCHK(hipMemsetAsync(c->pt_counters.p, 0, 16, c->stream));
    c->n_patterns = 0;
    c->pid0 = 0;
    c->b64_done = 0;
    c->pt_stale = false;
    c->h_pat_bits.clear(); c->h_pat_nan.clear(); c->h_pat_n.clear(); c->h_pat_md5.clear(); c->h_first_seen.clear();
    c->h_b64.clear();
    return PF_OK;
}

// Upload many small uint32 arrays with ONE pinned-host -> device copy; each DevBuf becomes a view into stage_dev.
int staged_upload(pf_ctx* c, std::vector<std::pair<DevBuf*, const std::vector<uint32_t>*>>& arrs) {
    size_t total = 0;
    std::vector<size_t> off(arrs.size());
    for (size_t i = 0; i < arrs.size(); i++) {
        off[i] = total;
        total += (std::max<size_t>(arrs[i].second->size(), 1) * 4 + 255) & ~(size_t)255;
    }
    for (auto& a : arrs) if (!a.first->view) a.first->release();
    DevBuf& stage_dev = c->stage_devs[c->stage_slot];
    void*& stage_pin = c->stage_pins[c->stage_slot];
    size_t& stage_pin_cap = c->stage_pin_caps[c->stage_slot];
    // several passes are queued without a host sync in between: the copy that last used this slot (two passes ago) has
    // to have left the pinned block before it is written again
    HIPCHK(hipEventSynchronize(c->ev_stage[c->stage_slot]));
    PFCHK(stage_dev.ensure(total));
    if (total > stage_pin_cap) {
        if (stage_pin) (void)hipHostFree(stage_pin);
        stage_pin = nullptr; stage_pin_cap = 0;
        const size_t want = total + total / 4;
        hipError_t e = hipHostMalloc(&stage_pin, want, hipHostMallocDefault);
        if (e != hipSuccess) return fail(PF_ERR_OOM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        stage_pin_cap = want;
    }
    for (size_t i = 0; i < arrs.size(); i++) {
        const auto& v = *arrs[i].second;
        if (!v.empty()) memcpy((char*)stage_pin + off[i], v.data(), v.size() * 4);
        arrs[i].first->p = (char*)stage_dev.p + off[i];
        arrs[i].first->cap = 0;
        arrs[i].first->view = true;
    }
    HIPCHK(hipMemcpyAsync(stage_dev.p, stage_pin, total, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipEventRecord(c->ev_stage[c->stage_slot], c->stream));
    return PF_OK;
}

// the run-global pattern table (`slots` = power of two) and the pool arrays indexed by pattern id (slots / 2 ids)
struct PatternBufs {
    DevBuf lo, val, first, bits, nan, n, md5, b64;
    uint64_t cap = 0;
    uint32_t pool = 0;
    void release() { lo.release(); val.release(); first.release(); bits.release(); nan.release(); n.release(); md5.release(); b64.release(); }
};
int alloc_pattern_bufs(pf_ctx* c, uint64_t slots, bool with_b64, PatternBufs& pb) {
    if (c->pt_slot_limit && slots > c->pt_slot_limit)
        return fail(PF_ERR_OOM, "pattern table of %llu slots refused (limit %llu set by pf_debug_limit_pattern_slots)",
                    (unsigned long long)slots, (unsigned long long)c->pt_slot_limit);
    pb.cap = slots;
    pb.pool = (uint32_t)std::min<uint64_t>(slots / 2, 0x7FFFFFF0ull);
    const size_t W = c->W, pool = pb.pool;
    PFCHK(pb.lo.ensure(slots * 8));
    PFCHK(pb.val.ensure(slots * 8));
    PFCHK(pb.first.ensure(pool * 8));
    PFCHK(pb.bits.ensure(pool * W * 4));
    PFCHK(pb.n.ensure(pool * 4));
    PFCHK(pb.md5.ensure(pool * 16));
    if (c->o.consider_missing) PFCHK(pb.nan.ensure(pool * W * 4));
    if (with_b64) PFCHK(pb.b64.ensure(pool * 24));
    return PF_OK;
}
// the context takes the buffers over (its old ones, if any, are the caller's to release: they are moved into `pb`)
void adopt_pattern_bufs(pf_ctx* c, PatternBufs& pb) {
    std::swap(c->pt_lo, pb.lo); std::swap(c->pt_val, pb.val); std::swap(c->pt_first, pb.first);
    std::swap(c->pat_bits, pb.bits); std::swap(c->pat_nan, pb.nan); std::swap(c->pat_n, pb.n);
    std::swap(c->pat_md5, pb.md5); std::swap(c->pat_b64, pb.b64);
    std::swap(c->pt.cap, pb.cap); std::swap(c->pt.pool, pb.pool);
    c->pt.lo = c->pt_lo.as<uint64_t>();
    c->pt.val = c->pt_val.as<uint64_t>();
    c->pt.first_seen = c->pt_first.as<uint64_t>();
}
int alloc_patterns(pf_ctx* c, uint64_t slots) {
    PatternBufs pb;
    int rc = alloc_pattern_bufs(c, slots, false, pb);
    if (rc == PF_OK) rc = c->pt_counters.ensure(16);
    if (rc != PF_OK) { pb.release(); return rc; }
    adopt_pattern_bufs(c, pb);
    c->pt.counters = c->pt_counters.as<uint32_t>();
    pb.release();
    return PF_OK;
}

// The reference's `patterns` is an unbounded set (panfeed.py:146-150): when a batch runs out of pattern ids (or
// comes close), the table and the pool are re-made larger, the patterns of earlier batches re-inserted
// (pattern_rehash_kernel; whatever the failed batch added is dropped) and the batch is run again.  The new table is
// built beside the old one and takes its place only when every step has succeeded: on a failure (out of memory with
// both resident, a failed copy) the context keeps the table it had and the error is returned.
int grow_patterns(pf_ctx* c, uint64_t min_pool) {
    uint64_t slots = c->pt.cap * 2;
    while (slots / 2 < min_pool + min_pool / 4) slots <<= 1;
    if (slots / 2 > 0x7FFFFFF0ull) return fail(PF_ERR_CAPACITY, "more than 2^31 distinct patterns");
    const uint32_t keep = c->n_patterns;            // ids of the batches that completed
    const size_t W = c->W;
    PatternBufs nb;
    int rc = alloc_pattern_bufs(c, slots, c->pat_b64.p != nullptr, nb);
    auto copy = [&](DevBuf& dst, DevBuf& src, size_t bytes) -> int {
        if (bytes && src.p) HIPCHK(hipMemcpyAsync(dst.p, src.p, bytes, hipMemcpyDeviceToDevice, c->stream));
        return PF_OK;
    };
    if (rc == PF_OK) rc = fill_u64(c, nb.lo.p, pf::EMPTY64, slots);
    if (rc == PF_OK) rc = fill_u64(c, nb.val.p, pf::EMPTY64, slots);
    if (rc == PF_OK) rc = fill_u64(c, nb.first.p, pf::EMPTY64, nb.pool);
    if (rc == PF_OK) rc = copy(nb.first, c->pt_first, (size_t)keep * 8);
    if (rc == PF_OK) rc = copy(nb.bits, c->pat_bits, (size_t)keep * W * 4);
    if (rc == PF_OK && c->o.consider_missing) rc = copy(nb.nan, c->pat_nan, (size_t)keep * W * 4);
    if (rc == PF_OK) rc = copy(nb.n, c->pat_n, (size_t)keep * 4);
    if (rc == PF_OK) rc = copy(nb.md5, c->pat_md5, (size_t)keep * 16);
    if (rc == PF_OK && nb.b64.p) rc = copy(nb.b64, c->pat_b64, (size_t)std::min(c->b64_done, keep) * 24);
    if (rc == PF_OK) {
        pf::RehashParams rp{};
        rp.old_lo = c->pt_lo.as<uint64_t>(); rp.old_val = c->pt_val.as<uint64_t>(); rp.old_cap = c->pt.cap;
        rp.new_lo = nb.lo.as<uint64_t>(); rp.new_val = nb.val.as<uint64_t>(); rp.new_cap = slots; rp.keep_below = keep;
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((c->pt.cap + 255) / 256, 8192);
        hipLaunchKernelGGL(pf::pattern_rehash_kernel, dim3(blocks), dim3(256), 0, c->stream, rp);
        if (hipGetLastError() != hipSuccess) rc = fail(PF_ERR_HIP, "pattern_rehash_kernel launch failed");
    }
    if (rc == PF_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(PF_ERR_HIP, "growing the pattern table failed");
    if (rc != PF_OK) {
        (void)hipStreamSynchronize(c->stream);       // nothing queued above may still touch the buffers released here
        nb.release();
        // the failed batch's additions are still in the old table: forget them, as the successful path does
        const uint32_t cnt[4] = {keep, 0, 0, 0};
        (void)hipMemcpy(c->pt_counters.p, cnt, 16, hipMemcpyHostToDevice);
        return rc;
    }
    adopt_pattern_bufs(c, nb);                       // nb now holds the old buffers
    nb.release();
    const uint32_t cnt[4] = {keep, 0, 0, 0};
    HIPCHK(hipMemcpy(c->pt_counters.p, cnt, 16, hipMemcpyHostToDevice));
    c->b64_done = std::min(c->b64_done, keep);
    c->n_grown++;
    return PF_OK;
}

// bytes of scratch one work item (cluster x key partition) keeps while its sub-batch is in flight
uint64_t slice_bytes(const pf_ctx* c) {
    return (uint64_t)c->NS * (8ull * c->KW + 4 + 4ull * c->W + 16 + 8 + 4 + 4 + 4 + 4) + (uint64_t)pf::DENSE_WORDS_BIG * 24 +
           (uint64_t)pf::DEDUP_MROWS * 4 + 64;
}
// the scratch slices of `items` work items (DevBuf::ensure: buffers that are large enough stay)
int alloc_scratch(pf_ctx* c, uint32_t items) {
    const size_t NS = c->NS, S = items, W = c->W;
    PFCHK(c->tab_key.ensure(S * NS * 8 * c->KW)); PFCHK(c->tab_ord.ensure(S * NS * 4));
    PFCHK(c->chunkbits.ensure(S * NS * W * 4)); PFCHK(c->chunkmask.ensure(S * 8 * 4));
    PFCHK(c->slot_hash.ensure(S * NS * 16)); PFCHK(c->sorted_pair.ensure(S * NS * 8));
    PFCHK(c->kept_prefix.ensure(S * (NS + 1) * 4));
    PFCHK(c->bm4.ensure(S * pf::DENSE_WORDS_BIG * 16)); PFCHK(c->bm2.ensure(S * pf::DENSE_WORDS_BIG * 8));
    PFCHK(c->mrows.ensure(S * pf::DEDUP_MROWS * 4)); PFCHK(c->slot_out.ensure(S * NS * 4));
    PFCHK(c->cmask_lo.ensure(S * NS * 4)); PFCHK(c->cmask_hi.ensure(S * NS * 4));
    return PF_OK;
}
// A cluster asks for more work items than a sub-batch holds (a very divergent or very wide cluster; an overflow retry
// multiplies its key partitions): the scratch is re-made for `need` items -- when that fits half of the device memory
// that is free once the old scratch is gone -- instead of failing the run.  Nothing may be in flight: the caller's
// earlier passes keep their results in the arenas, not in the scratch.
int grow_scratch(pf_ctx* c, uint32_t need) {
    HIPCHK(hipStreamSynchronize(c->stream));
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const uint64_t sb = slice_bytes(c);
    const uint64_t have = (uint64_t)c->max_items * sb;
    uint64_t want = std::max<uint64_t>(need, std::min<uint64_t>(2ull * c->max_items, 65536));
    if (want * sb > (free_b + have) / 2) want = need;
    if (want * sb > (free_b + have) / 2)
        return fail(PF_ERR_CAPACITY, "a cluster needs %u work items (%.1f GB of scratch); %.1f GB of device memory are free",
                    need, (double)need * sb / 1e9, (double)(free_b + have) / 1e9);
    DevBuf* bufs[] = {&c->tab_key, &c->tab_ord, &c->chunkbits, &c->chunkmask, &c->slot_hash, &c->sorted_pair, &c->kept_prefix,
                      &c->bm4, &c->bm2, &c->mrows, &c->slot_out, &c->cmask_lo, &c->cmask_hi};
    for (DevBuf* b : bufs) b->release();          // (freed first: old and new need not fit side by side)
    const int rc = alloc_scratch(c, (uint32_t)want);
    if (rc != PF_OK) {                            // back to what it was; if even that fails the context is unusable
        for (DevBuf* b : bufs) b->release();
        if (alloc_scratch(c, c->max_items) != PF_OK) c->max_items = 0;
        return rc;
    }
    c->max_items = (uint32_t)want;
    c->n_scratch_grown++;
    return PF_OK;
}

template <int KW, bool CANON>
int scan_attr_t(pf_ctx* c) {
    // ~100-160 KB of dynamic LDS: the limit is raised once per context (= per device), at pf_create
    const uint32_t lds = c->NS * (8u * KW + 8u) + pf::MISC_WORDS * 4;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(pf::kmer_scan_kernel<KW, CANON>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return PF_OK;
}
int scan_attr(pf_ctx* c) {
    switch (c->KW) {
        case 1: return c->o.canon ? scan_attr_t<1, true>(c) : scan_attr_t<1, false>(c);
        case 2: return c->o.canon ? scan_attr_t<2, true>(c) : scan_attr_t<2, false>(c);
        case 3: return c->o.canon ? scan_attr_t<3, true>(c) : scan_attr_t<3, false>(c);
        default: return c->o.canon ? scan_attr_t<4, true>(c) : scan_attr_t<4, false>(c);
    }
}

template <int KW, bool CANON>
int launch_scan_t(pf_ctx* c, const pf::ScanParams& sp, uint32_t n) {
    auto kern = pf::kmer_scan_kernel<KW, CANON>;
    const uint32_t lds = c->NS * (8u * KW + 8u) + pf::MISC_WORDS * 4;
    // descriptors first (one thread per work entry), then one persistent workgroup per CU
    PFCHK(c->scan_desc.ensure((size_t)n * sizeof(pf::ScanDesc)));
    pf::ScanParams q = sp;
    q.n_work = n;
    q.desc = c->scan_desc.as<pf::ScanDesc>();
    hipLaunchKernelGGL(pf::scan_desc_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, q, c->scan_desc.as<pf::ScanDesc>());
    HIPCHK(hipGetLastError());
    const uint32_t grid = n < (uint32_t)c->n_cu ? n : (uint32_t)c->n_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(pf::SCAN_THREADS), lds, c->stream, q);
    HIPCHK(hipGetLastError());
    return PF_OK;
}
int launch_bin(pf_ctx* c, const pf::ScanParams& sp, uint32_t n) {
    const dim3 g(n), b(pf::BIN_THREADS);
    switch (c->KW) {
        case 1:
            if (c->o.canon) hipLaunchKernelGGL((pf::bin_kernel<1, true>), g, b, 0, c->stream, sp);
            else hipLaunchKernelGGL((pf::bin_kernel<1, false>), g, b, 0, c->stream, sp);
            break;
        case 2:
            if (c->o.canon) hipLaunchKernelGGL((pf::bin_kernel<2, true>), g, b, 0, c->stream, sp);
            else hipLaunchKernelGGL((pf::bin_kernel<2, false>), g, b, 0, c->stream, sp);
            break;
        default: return fail(PF_ERR_STATE, "bin_kernel: keys of more than two words are not binned");
    }
    HIPCHK(hipGetLastError());
    return PF_OK;
}
int launch_scan(pf_ctx* c, const pf::ScanParams& sp, uint32_t n) {
    switch (c->KW) {
        case 1: return c->o.canon ? launch_scan_t<1, true>(c, sp, n) : launch_scan_t<1, false>(c, sp, n);
        case 2: return c->o.canon ? launch_scan_t<2, true>(c, sp, n) : launch_scan_t<2, false>(c, sp, n);
        case 3: return c->o.canon ? launch_scan_t<3, true>(c, sp, n) : launch_scan_t<3, false>(c, sp, n);
        default: return c->o.canon ? launch_scan_t<4, true>(c, sp, n) : launch_scan_t<4, false>(c, sp, n);
    }
}

}  // namespace

namespace {
template <class F>
void parallel_for(uint64_t n, F f) {
    const unsigned nt = pf_host_threads(32u);
    if (n < 4096 || nt == 1) { f(0, n); return; }
    std::vector<std::thread> th;
    const uint64_t chunk = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        const uint64_t a = t * chunk, b = std::min<uint64_t>(n, a + chunk);
        if (a < b) th.emplace_back([=] { f(a, b); });
    }
    for (auto& x : th) x.join();
}
void ensure_b64(pf_ctx* c) {
    // 24-char base64 of every pattern's digest (panfeed.py:176, 207), extended incrementally
    const size_t have = c->h_b64.size() / 24, want = c->h_pat_md5.size() / 16;
    c->h_b64.resize(want * 24);
    for (size_t p = have; p < want; p++) pf_b64_digest(c->h_pat_md5.data() + p * 16, c->h_b64.data() + p * 24);
}
}  // namespace


extern "C" {

const char* pf_last_error(void) { return g_err.c_str(); }
void pf_set_error_(const char* msg) { g_err = msg; }   /* for the other translation units of the library */
const char* pf_version(void) { return "panfeed_hip 0.1 (gfx950)"; }

int pf_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(PF_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

void pf_destroy(pf_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = {&c->d_maf_lo, &c->d_maf_hi, &c->pt_lo, &c->pt_val, &c->pt_first, &c->pt_counters, &c->pat_bits,
                      &c->pat_nan, &c->pat_n, &c->pat_md5, &c->tab_key, &c->tab_ord, &c->chunkbits, &c->chunkmask,
                      &c->slot_hash, &c->sorted_pair, &c->kept_prefix, &c->b_packed, &c->b_seg_word_off, &c->b_seg_len,
                      &c->b_seg_sample, &c->b_seg_ord, &c->b_cl_seg_off, &c->b_cl_nstr, &c->b_cl_npres, &c->b_cl_presab,
                      &c->b_cl_ordinal, &c->b_extra_ord, &c->b_extra_bits, &c->b_seg_strand_off, &c->cl_rec, &c->v_word_off, &c->v_len, &c->v_sample, &c->v_ord,
                      &c->seg_distinct, &c->v_nseg, &c->v_nstr, &c->v_mode, &c->v_dense, &c->extra_off, &c->extra_dense,
                      &c->bm4, &c->bm2, &c->mrows, &c->slot_out, &c->it_is_extra,
                      &c->cmask_lo, &c->cmask_hi, &c->it_compact, &c->cl_overflow, &c->cl_kmer_off, &c->cl_kmer_cnt, &c->cl_unique, &c->cl_pattern,
                      &c->cl_first, &c->cursor, &c->strand_bits, &c->it_cluster, &c->it_part, &c->it_nparts,
                      &c->it_nslots, &c->it_slice, &c->it_sib0, &c->it_nsib, &c->it_extra_first, &c->it_count,
                      &c->it_unique, &c->it_kept, &c->work_scan, &c->work_extra, &c->work_fin, &c->work_fin2, &c->work_fin3, &c->work_fin5, &c->work_rows, &c->sub_cluster, &c->sub_item0,
                      &c->sub_nitems, &c->it_binned, &c->bin_lists, &c->q_key, &c->q_ord, &c->q_bit, &c->q_off};
    for (DevBuf* b : bufs) b->release();
    for (Arena* a : c->arenas) { a->key.release(); a->pid.release(); a->first.release(); delete a; }
    for (auto& e : c->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; i++) { if (c->stage_pins[i]) (void)hipHostFree(c->stage_pins[i]); c->stage_devs[i].release(); if (c->ev_stage[i]) (void)hipEventDestroy(c->ev_stage[i]); }
    for (auto e : c->ev_part) if (e) (void)hipEventDestroy(e);

    for (auto& d : c->dplan) { d.block.release(); d.it_count.release(); d.out.release(); if (d.pin_out) (void)hipHostFree(d.pin_out); }
    c->dp_const.release(); c->plan_room.release(); c->plan_arena.release();
    for (auto& u : c->upool) {
        u.word_off.release(); u.len.release(); u.sample.release(); u.ord.release(); u.bits.release(); u.list.release();
        if (u.pin) (void)hipHostFree(u.pin);
    }
    c->v_bits.release(); c->view_off.release();
    if (c->pin_dedup) (void)hipHostFree(c->pin_dedup);
    if (c->pin_small) (void)hipHostFree(c->pin_small);
    if (c->pin_ovf) (void)hipHostFree(c->pin_ovf);
    c->scan_desc.release(); c->pat_b64.release(); c->txt_dev.release(); c->txt_meta.release();
    c->rp_order.release(); c->rp_rlen.release(); c->rp_rowoff.release(); c->wide_list.release();
    for (int i = 0; i < 2; i++) if (c->txt_pins[i]) (void)hipHostFree(c->txt_pins[i]); c->md5_list.release();
    for (int i = 0; i < 2; i++) if (c->kt_pins[i]) (void)hipHostFree(c->kt_pins[i]);
    for (DevBuf* b : {&c->kt_seqs, &c->kt_tiles, &c->kt_prefix, &c->kt_tbytes, &c->kt_toff, &c->kt_text}) b->release();
    c->g_store.release(); c->b_literal.release(); c->g_src_off.release(); c->g_src_start.release(); c->g_src_flags.release();
    c->mg_lo.release(); c->mg_cnt.release();
    if (c->ev_t0) (void)hipEventDestroy(c->ev_t0);
    if (c->ev_t1) (void)hipEventDestroy(c->ev_t1);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int pf_create(pf_ctx** out, int device, const pf_opts* o) {
    if (!out || !o) return fail(PF_ERR_ARG, "pf_create: null argument");
    *out = nullptr;
    if (o->klength < 1 || o->klength > PF_MAX_K)
        return fail(PF_ERR_ARG, "klength %u unsupported (1..%d)", o->klength, PF_MAX_K);
    if (o->max_strains < 1 || o->max_strains > pf::MAX_CHUNKS * 32)
        return fail(PF_ERR_ARG, "max_strains %u unsupported (1..%u)", o->max_strains, pf::MAX_CHUNKS * 32);
    if (!o->maf_lo || !o->maf_hi) return fail(PF_ERR_ARG, "maf_lo / maf_hi tables are required");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(PF_ERR_ARG, "device %d not present (%d devices)", device, ndev);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PF_ERR_ARG, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);

    pf_ctx* c = new pf_ctx();
    c->device = device;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->o = *o;
    c->KW = (int)((2 * o->klength + 62) / 63);       // 63 key bits per word: k <= 31 one word ... k <= 126 four
    c->NS = pf::nslots_max(c->KW);
    c->W = (o->max_strains + 31) / 32;
    c->max_items = o->max_items ? o->max_items : 2048;
    {
        // work items of one launch = scratch slices resident at once; keep them within half of the free HBM
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b) {
            const uint64_t fit = (free_b / 2) / slice_bytes(c);
            if (c->max_items > fit) c->max_items = (uint32_t)std::max<uint64_t>(fit, 64);
        }
    }
    c->maf_lo.assign(o->maf_lo, o->maf_lo + o->max_strains + 1);
    c->maf_hi.assign(o->maf_hi, o->maf_hi + o->max_strains + 1);
    c->o.maf_lo = c->maf_lo.data();
    c->o.maf_hi = c->maf_hi.data();
    int rc = PF_OK;
    auto guard = [&](int r) { if (r != PF_OK && rc == PF_OK) rc = r; return r == PF_OK; };
    do {
        hipError_t e = hipStreamCreate(&c->stream);
        if (e != hipSuccess) { rc = fail(PF_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); break; }
        if (hipStreamCreate(&c->side) != hipSuccess || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) { rc = fail(PF_ERR_HIP, "hipStreamCreate (side) failed"); break; }
        bool ev_ok = true;
        for (auto& ev : c->ev_part) ev_ok = ev_ok && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
        for (auto& ev : c->ev_stage) ev_ok = ev_ok && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
        if (!ev_ok ||
            hipHostMalloc((void**)&c->pin_small, 512, hipHostMallocDefault) != hipSuccess ||
            hipEventCreate(&c->ev_t0) != hipSuccess || hipEventCreate(&c->ev_t1) != hipSuccess) {
            rc = fail(PF_ERR_HIP, "hipEventCreate failed"); break;
        }
        if (!guard(scan_attr(c))) break;
        if (!guard(upload_vec(c, c->d_maf_lo, c->maf_lo))) break;
        if (!guard(upload_vec(c, c->d_maf_hi, c->maf_hi))) break;
        uint64_t cap = o->pattern_capacity ? o->pattern_capacity : (1ull << 24);
        uint64_t p2 = 1024;
        while (p2 < cap) p2 <<= 1;
        if (!guard(alloc_patterns(c, p2))) break;
        if (!guard(reset_patterns(c))) break;
        if (!guard(c->cursor.ensure(64)) || !guard(alloc_scratch(c, c->max_items))) break;
        e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { rc = fail(PF_ERR_HIP, "pf_create sync: %s", hipGetErrorString(e)); break; }
    } while (0);
    if (rc != PF_OK) { std::string keep = g_err; pf_destroy(c); g_err = keep; return rc; }
    *out = c;
    return PF_OK;
}

int pf_reset_patterns(pf_ctx* c) {
    if (!c) return fail(PF_ERR_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
    PFCHK(reset_patterns(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PF_OK;
}

int pf_dev_alloc(pf_ctx* c, uint64_t bytes, void** dptr) {
    if (!c || !dptr) return fail(PF_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMalloc(dptr, bytes ? bytes : 16));
    return PF_OK;
}
int pf_dev_free(pf_ctx* c, void* dptr) {
    if (!c) return fail(PF_ERR_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipFree(dptr));
    return PF_OK;
}
int pf_dev_upload(pf_ctx* c, void* dptr, const void* src, uint64_t bytes) {
    if (!c) return fail(PF_ERR_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy(dptr, src, bytes, hipMemcpyHostToDevice));
    return PF_OK;
}
int pf_dev_download(pf_ctx* c, void* dst, const void* dptr, uint64_t bytes) {
    if (!c) return fail(PF_ERR_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(dst, dptr, bytes, hipMemcpyDeviceToHost));
    return PF_OK;
}

int pf_synth_expand(pf_ctx* c, const uint64_t* allele_words, const uint64_t* allele_word_off,
                    const uint32_t* seg_allele, const uint64_t* seg_word_off, const uint32_t* seg_len,
                    uint32_t n_segs, uint64_t* packed) {
    if (!c) return fail(PF_ERR_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
    if (!n_segs) return PF_OK;
    hipLaunchKernelGGL(pf::synth_expand_kernel, dim3(2048), dim3(256), 0, c->stream, allele_words, allele_word_off,
                       seg_allele, seg_word_off, seg_len, n_segs, packed);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return PF_OK;
}

uint64_t pf_pack_acgt(const char* seq, uint32_t len, uint64_t* dst) {
    const uint64_t nw = 2ull * ((len + 63) / 64);
    for (uint64_t i = 0; i < nw; i++) dst[i] = 0;
    for (uint32_t i = 0; i < len; i++) {
        uint64_t code;
        switch (seq[i]) {
            case 'A': code = 0; break;
            case 'C': code = 1; break;
            case 'G': code = 2; break;
            default: code = 3; break;
        }
        dst[i >> 5] |= code << (62 - 2 * (i & 31));
    }
    return nw;
}

void pf_b64_digest(const uint8_t d[16], char out[24]) {
    static const char* T = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    int o = 0;
    for (int i = 0; i < 15; i += 3) {
        uint32_t v = ((uint32_t)d[i] << 16) | ((uint32_t)d[i + 1] << 8) | d[i + 2];
        out[o++] = T[(v >> 18) & 63]; out[o++] = T[(v >> 12) & 63]; out[o++] = T[(v >> 6) & 63]; out[o++] = T[v & 63];
    }
    uint32_t v = (uint32_t)d[15] << 16;
    out[o++] = T[(v >> 18) & 63]; out[o++] = T[(v >> 12) & 63]; out[o++] = '='; out[o++] = '=';
}

// ---------------------------------------------------------------------------------------------
namespace {
constexpr int PF_RETRY_PATTERNS = 1;   // internal: the batch ran out of pattern ids, *need = ids it asked for

int submit_once(pf_ctx* c, const pf_batch* b, const pf_gather* gth, pf_result* counters, uint64_t* need, bool rerun) {
    c->have_batch = false;
    // whatever way this call ends, nothing it queued is still reading the caller's arrays or the pinned staging
    // blocks afterwards (the successful path has waited already; an error return may come with work in flight)
    struct Drain { hipStream_t s, s2; ~Drain() { (void)hipStreamSynchronize(s2); (void)hipStreamSynchronize(s); } } drain{c->stream, c->side};
    const uint32_t C = b->n_clusters, NSEG = b->n_segs, W = c->W, NS = c->NS, KW = (uint32_t)c->KW;
    if (b->n_segs && (!b->packed || !b->seg_word_off || !b->seg_len || !b->seg_sample || !b->seg_ord_base))
        return fail(PF_ERR_ARG, "segment arrays missing");
    if (C && (!b->cluster_seg_off || !b->cluster_nstrains || !b->cluster_npresab || !b->cluster_presab ||
              !b->cluster_ordinal))
        return fail(PF_ERR_ARG, "cluster arrays missing");
    if (b->n_extra && (!b->extra_cluster || !b->extra_ord || !b->extra_bits))
        return fail(PF_ERR_ARG, "extra arrays missing");
    for (auto& e : c->events) { c->ev_pool.push_back(e.a); c->ev_pool.push_back(e.b); }
    c->events.clear();
    c->timing = pf_timing{};
    const bool dbg = getenv("PF_DEBUG_TIMING") != nullptr;
    auto t_host0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[pf_submit] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_host0).count());
        t_host0 = t;
    };
    HIPCHK(hipEventRecord(c->ev_t0, c->stream));

    // ---- batch arrays on the device
    pf_batch d = *b;
    std::vector<uint32_t> h_extra_cluster;
    if (gth && b->on_device) return fail(PF_ERR_ARG, "pf_submit_gather takes host arrays");
    const uint64_t total_words = gth ? gth->n_words : b->n_words;
    if (!b->on_device) {
        // validate what the kernels index with (host copies are at hand)
        for (uint32_t i = 0; i < C; i++) {
            if (b->cluster_seg_off[i] > b->cluster_seg_off[i + 1] || b->cluster_seg_off[i + 1] > NSEG)
                return fail(PF_ERR_ARG, "cluster_seg_off not monotone / out of range at %u", i);
            if (b->cluster_nstrains[i] > c->o.max_strains || b->cluster_npresab[i] > c->o.max_strains)
                return fail(PF_ERR_ARG, "cluster %u has more strains than max_strains", i);
            // init_presabs_vector, panfeed.py:19: a boolean mask must have the vector's length (numpy IndexError)
            if (c->o.consider_missing && b->cluster_nstrains[i] != b->cluster_npresab[i])
                return fail(PF_ERR_ARG, "cluster %u: consider_missing needs len(clusterpresab) == number of strains "
                            "(%u != %u)", i, b->cluster_npresab[i], b->cluster_nstrains[i]);
        }
        if (C && b->cluster_seg_off[0] != 0) return fail(PF_ERR_ARG, "cluster_seg_off[0] must be 0");
        const uint32_t pad_words = KW <= 2 ? 2u : 4u;     // a lane reads KW + 1 words from its window's first word
        for (uint32_t s = 0; s < NSEG; s++) {
            const uint64_t nw = 2ull * ((b->seg_len[s] + 63) / 64);
            if ((b->seg_word_off[s] & 1) || b->seg_word_off[s] + nw + pad_words > total_words)
                return fail(PF_ERR_ARG, "segment %u: misaligned or outside packed[] (needs %u words of tail padding)", s, pad_words);
        }
        for (uint32_t i = 0; i < C; i++)
            for (uint32_t s = b->cluster_seg_off[i]; s < b->cluster_seg_off[i + 1]; s++) {
                if (b->seg_sample[s] >= b->cluster_nstrains[i])
                    return fail(PF_ERR_ARG, "segment %u: sample column %u >= n_strains %u", s, b->seg_sample[s],
                                b->cluster_nstrains[i]);
                if (s > b->cluster_seg_off[i] && b->seg_sample[s] < b->seg_sample[s - 1])
                    return fail(PF_ERR_ARG, "segments of cluster %u are not sorted by sample", i);
            }
        if (!gth) PFCHK(upload(c, c->b_packed, b->packed, (size_t)b->n_words, &d.packed));
        PFCHK(upload(c, c->b_seg_word_off, b->seg_word_off, NSEG, &d.seg_word_off));
        PFCHK(upload(c, c->b_seg_len, b->seg_len, NSEG, &d.seg_len));
        if (gth) {
            // the packed input is produced on the device: segments copied (or reverse-complemented) out of the
            // resident genomes, plus the few the host packed itself (b->packed = the literal words)
            if (NSEG && (!gth->src_off || !gth->src_start || !gth->src_flags)) return fail(PF_ERR_ARG, "gather arrays missing");
            for (uint32_t s = 0; s < NSEG; s++) {
                const uint64_t nw = 2ull * ((b->seg_len[s] + 63) / 64);
                const uint64_t last = (uint64_t)gth->src_start[s] + b->seg_len[s];         // bases
                if (gth->src_flags[s] & 1u) {
                    if (gth->src_off[s] + nw + 1 > b->n_words || gth->src_start[s] != 0 || (gth->src_flags[s] & 2u))
                        return fail(PF_ERR_ARG, "segment %u: literal source outside packed[]", s);
                } else if (gth->src_off[s] + (last + 31) / 32 + 1 > c->g_words) {
                    return fail(PF_ERR_ARG, "segment %u: source range outside the resident genomes", s);
                }
            }
            PFCHK(c->b_packed.ensure((size_t)std::max<uint64_t>(total_words, 4) * 8));
            d.packed = c->b_packed.as<uint64_t>();
            const uint64_t* lit; const uint64_t* so; const uint32_t* ss; const uint32_t* sf;
            PFCHK(upload(c, c->b_literal, b->packed, (size_t)b->n_words, &lit));
            PFCHK(upload(c, c->g_src_off, gth->src_off, NSEG, &so));
            PFCHK(upload(c, c->g_src_start, gth->src_start, NSEG, &ss));
            PFCHK(upload(c, c->g_src_flags, gth->src_flags, NSEG, &sf));
            if (total_words >= 4)
                HIPCHK(hipMemsetAsync(c->b_packed.as<uint64_t>() + (total_words - 4), 0, 32, c->stream));
            if (NSEG) {
                pf::GatherParams gp{};
                gp.store = c->g_store.as<uint64_t>(); gp.literal = lit; gp.src_off = so; gp.src_start = ss; gp.src_flags = sf;
                gp.seg_word_off = d.seg_word_off; gp.seg_len = d.seg_len; gp.packed = c->b_packed.as<uint64_t>(); gp.n_segs = NSEG;
                hipLaunchKernelGGL(pf::gather_segments_kernel, dim3((NSEG + 15) / 16), dim3(256), 0, c->stream, gp);
                HIPCHK(hipGetLastError());
            }
        }
        PFCHK(upload(c, c->b_seg_sample, b->seg_sample, NSEG, &d.seg_sample));
        PFCHK(upload(c, c->b_seg_ord, b->seg_ord_base, NSEG, &d.seg_ord_base));
        PFCHK(upload(c, c->b_cl_seg_off, b->cluster_seg_off, (size_t)C + 1, &d.cluster_seg_off));
        PFCHK(upload(c, c->b_cl_nstr, b->cluster_nstrains, C, &d.cluster_nstrains));
        PFCHK(upload(c, c->b_cl_npres, b->cluster_npresab, C, &d.cluster_npresab));
        PFCHK(upload(c, c->b_cl_presab, b->cluster_presab, (size_t)C * W, &d.cluster_presab));
        PFCHK(upload(c, c->b_cl_ordinal, b->cluster_ordinal, C, &d.cluster_ordinal));
        PFCHK(upload(c, c->b_extra_ord, b->extra_ord, b->n_extra, &d.extra_ord));
        PFCHK(upload(c, c->b_extra_bits, b->extra_bits, (size_t)b->n_extra * W, &d.extra_bits));
        if (b->seg_strand_off) PFCHK(upload(c, c->b_seg_strand_off, b->seg_strand_off, NSEG, &d.seg_strand_off));
        if (b->n_extra) h_extra_cluster.assign(b->extra_cluster, b->extra_cluster + b->n_extra);
    }
    // extras per cluster (CSR).  A batch that is in device memory already has its list checked and counted there
    // (extra_csr_kernel); the counts come back with the first dedup results -- reading the list back and walking it here
    // was 1 ms in front of the first kernel with SURVEY 8d's share of 'N's (1.5 M rows per 50 000 clusters).
    const bool ex_on_device = b->on_device && b->n_extra;
    if (ex_on_device && !C) return fail(PF_ERR_ARG, "extra_cluster must be non-decreasing and < n_clusters");
    std::vector<uint32_t> ex_first(C + 1, 0);
    if (!ex_on_device) {
        for (uint32_t e = 0; e < b->n_extra; e++) {
            if (h_extra_cluster[e] >= C || (e && h_extra_cluster[e] < h_extra_cluster[e - 1]))
                return fail(PF_ERR_ARG, "extra_cluster must be non-decreasing and < n_clusters");
        }
        for (uint32_t e = 0; e < b->n_extra; e++) ex_first[h_extra_cluster[e] + 1]++;
        for (uint32_t i = 0; i < C; i++) ex_first[i + 1] += ex_first[i];
        PFCHK(upload_vec(c, c->extra_off, ex_first));
    }

    // ---- per batch outputs
    const size_t C1 = std::max(C, 1u), NSEG1 = std::max(NSEG, 1u), NEX1 = std::max(b->n_extra, 1u);
    PFCHK(c->cl_overflow.ensure(C1 * 4));
    PFCHK(c->cl_kmer_off.ensure(C1 * 8));
    PFCHK(c->cl_kmer_cnt.ensure(C1 * 4));
    PFCHK(c->cl_unique.ensure(C1 * 4));
    PFCHK(c->cl_pattern.ensure(C1 * 4));
    PFCHK(c->cl_first.ensure(C1 * 8));
    PFCHK(c->cl_rec.ensure(C1 * sizeof(pf::ClusterRec)));
    PFCHK(c->v_word_off.ensure(NSEG1 * 8));
    PFCHK(c->v_len.ensure(NSEG1 * 4));
    PFCHK(c->v_sample.ensure(NSEG1 * 4));
    PFCHK(c->v_ord.ensure(NSEG1 * 4));
    PFCHK(c->v_bits.ensure(NSEG1 * 4));
    PFCHK(c->view_off.ensure(C1 * 4));
    PFCHK(c->seg_distinct.ensure(NSEG1 * 4));
    PFCHK(c->v_nseg.ensure(C1 * 4));
    PFCHK(c->v_nstr.ensure(C1 * 4));
    PFCHK(c->v_mode.ensure(C1 * 4));
    PFCHK(c->v_dense.ensure(C1 * 4));
    PFCHK(c->extra_dense.ensure(NEX1 * 4));
    // (cl_overflow / cl_kmer_cnt / cl_unique / cl_pattern get their start values from the dedup kernel)
    HIPCHK(hipMemsetAsync(c->cursor.p, 0, 64, c->stream));

    // ---- identical segments -> scan view (mode 1) or the caller's list as it is (mode 0)
    // The dedup kernel's per-cluster results come back into pinned memory, in two halves for a large batch: the
    // host builds and launches the first half's work items while the GPU is still on the second half's dedup, and the
    // second half's while the first half's scan runs -- otherwise the GPU idles for the ~1.2 ms that takes.
    const size_t C8 = ((size_t)C + 1) & ~(size_t)1;
    {
        const size_t need = C8 * sizeof(pf::ClusterRec) + 64 + ((size_t)C + 2) * 4;
        if (need > c->pin_dedup_cap) {
            if (c->pin_dedup) (void)hipHostFree(c->pin_dedup);
            c->pin_dedup = nullptr; c->pin_dedup_cap = 0;
            hipError_t e = hipHostMalloc(&c->pin_dedup, need + need / 4, hipHostMallocDefault);
            if (e != hipSuccess) return fail(PF_ERR_OOM, "hipHostMalloc(%zu) failed: %s", need, hipGetErrorString(e));
            c->pin_dedup_cap = need + need / 4;
        }
    }
    // what the host needs of the dedup pass per cluster, one 40-byte record each (pf::ClusterRec, written by
    // cluster_ninst_kernel): ONE copy per part brings them over -- six small copies in a row were 40 us of the part's
    // critical path
    pf::ClusterRec* rec = reinterpret_cast<pf::ClusterRec*>(c->pin_dedup);
    uint32_t* h_exfirst = reinterpret_cast<uint32_t*>(rec + C8);   // [C + 1] + the "bad list" flag (device-side CSR only)
    if (ex_on_device) {
        PFCHK(c->extra_off.ensure(((size_t)C + 2) * 4));
        uint32_t* exo = c->extra_off.as<uint32_t>();
        HIPCHK(hipMemsetAsync(exo + C + 1, 0, 4, c->stream));
        hipLaunchKernelGGL(pf::extra_csr_kernel, dim3(std::min<uint32_t>((std::max(C + 1, b->n_extra) + 255) / 256, 2048u)), dim3(256), 0,
                           c->stream, b->extra_cluster, b->n_extra, C, exo, exo + C + 1);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h_exfirst, exo, ((size_t)C + 2) * 4, hipMemcpyDeviceToHost, c->stream));
    }
    // A large batch goes through in parts, all queued without a host sync in between: the host builds and launches a
    // part's work items while the GPU is on earlier parts (otherwise it idles for the ~1.2 ms that takes).  Two parts,
    // a quarter first: enough GPU work to hide building the rest.  (More, equal parts with the MD5 of part i on a second
    // stream beside part i + 1's finish kernels or part i + 2's dedup were measured and lose: DESIGN.md section 6.)
    uint32_t P = C >= 8192 ? 2u : 1u;
    uint32_t part_end[pf_ctx::MAX_PARTS];
    for (uint32_t q = 0; q < P; q++) part_end[q] = q + 1 == P ? C : (uint32_t)((uint64_t)C * (q + 1) / (2 * P));
    pf::DedupParams dp{};
    if (C) {
        dp.packed = d.packed; dp.seg_word_off = d.seg_word_off; dp.seg_len = d.seg_len;
        dp.seg_sample = d.seg_sample; dp.seg_ord_base = d.seg_ord_base;
        dp.cluster_seg_off = d.cluster_seg_off; dp.cluster_nstrains = d.cluster_nstrains;
        dp.extra_off = c->extra_off.as<uint32_t>(); dp.extra_ord = d.extra_ord;
        dp.v_word_off = c->v_word_off.as<uint64_t>(); dp.v_len = c->v_len.as<uint32_t>();
        dp.v_sample = c->v_sample.as<uint32_t>(); dp.v_ord = c->v_ord.as<uint32_t>();
        dp.seg_distinct = c->seg_distinct.as<uint32_t>();
        dp.v_bits = c->v_bits.as<uint32_t>(); dp.view_off = c->view_off.as<uint32_t>();
        dp.cl_overflow = c->cl_overflow.as<uint32_t>(); dp.cl_kmer_cnt = c->cl_kmer_cnt.as<uint32_t>();
        dp.cl_unique = c->cl_unique.as<uint32_t>(); dp.cl_pattern = c->cl_pattern.as<uint32_t>();
        dp.v_nseg = c->v_nseg.as<uint32_t>(); dp.v_nstr = c->v_nstr.as<uint32_t>();
        dp.v_mode = c->v_mode.as<uint32_t>(); dp.v_dense = c->v_dense.as<uint32_t>();
        dp.extra_dense = c->extra_dense.as<uint32_t>();
        dp.k = c->o.klength; dp.W = W; dp.canon = c->o.canon;
        dp.enable = (c->o.flags & PF_FLAG_NO_DEDUP) ? 0u : 1u;
    }
    // ---- the device plan (plan_kernel): the simple clusters' work items laid out behind the part's dedup, a 40-byte
    // summary on its way to pinned memory.  The estimate's learned line is what this context knew when the batch came in
    // (the host's own estimate for the rest of the part reads the same sums: they change at the end of a submit only).
    const bool use_plan = (c->o.flags & PF_FLAG_DEVICE_PLAN) && C > 0 && c->max_items < (1u << 22);
    const double share = 1.0 - std::pow(0.99, (double)c->o.klength) + 0.06;
    double reg_a = 0.0, reg_b = 0.0, reg_half_sd = 0.0;
    const bool reg_ready = c->reg_n >= 16;
    if (reg_ready) {
        const double n = c->reg_n, den = n * c->reg_xx - c->reg_x * c->reg_x;
        reg_a = c->reg_y / n;
        if (den > 1e-6 * n * c->reg_xx) { reg_b = (n * c->reg_xy - c->reg_x * c->reg_y) / den; reg_a = (c->reg_y - reg_b * c->reg_x) / n; }
        const double ss = std::max(0.0, c->reg_yy - reg_a * c->reg_y - reg_b * c->reg_xy);     // residual sum of squares
        // (half a residual standard deviation on top: with `room` at 0.9 of the table's limit that left no
        // cluster of the headline workload, with or without 'N's, to overflow; 0 left 6, 1 to 3 standard
        // deviations cost 0.5 % to 5 % in surplus partitions -- profiles/r02/partition_margin_experiment.txt)
        reg_half_sd = 0.5 * std::sqrt(ss / std::max(1.0, n - 2.0));
    }
    struct DPtrs { uint32_t *it_cluster, *it_nslots, *w_scan, *w_fin, *w_fin2, *w_fin5, *unit_cluster, *unit_base, *blk; };
    auto dplan_ptrs = [&](const pf_ctx::DPlan& dp) {
        uint32_t* b = dp.block.as<uint32_t>();
        const size_t n = dp.n;
        return DPtrs{b, b + n, b + 2 * n, b + 3 * n, b + 4 * n, b + 5 * n, b + 6 * n, b + 7 * n, b + 8 * n};
    };
    auto launch_plan = [&](uint32_t h, uint32_t c0, uint32_t c1) -> int {
        pf_ctx::DPlan& dp = c->dplan[h];
        dp.n = c1 - c0;
        const uint32_t nblk = (dp.n + pf::PLAN_THREADS - 1) / pf::PLAN_THREADS;
        PFCHK(dp.block.ensure(((size_t)dp.n * 8 + (size_t)nblk * pf::PLAN_BLK_WORDS) * 4));
        PFCHK(dp.it_count.ensure((size_t)dp.n * 4));
        PFCHK(dp.out.ensure(sizeof(pf::PlanOut)));
        if (!dp.pin_out) {
            hipError_t e = hipHostMalloc((void**)&dp.pin_out, 64, hipHostMallocDefault);
            if (e != hipSuccess) return fail(PF_ERR_OOM, "hipHostMalloc(64) failed: %s", hipGetErrorString(e));
        }
        const DPtrs q = dplan_ptrs(dp);
        pf::PlanParams pp{};
        pp.rec = c->cl_rec.as<pf::ClusterRec>(); pp.plan_room = c->plan_room.as<uint32_t>(); pp.plan_arena = c->plan_arena.as<uint32_t>();
        pp.c0 = c0; pp.c1 = c1;
        pp.NS = NS; pp.max_items = c->max_items;
        pp.it_cluster = q.it_cluster; pp.it_nslots = q.it_nslots; pp.w_scan = q.w_scan; pp.w_fin = q.w_fin; pp.w_fin2 = q.w_fin2;
        pp.w_fin5 = q.w_fin5; pp.unit_cluster = q.unit_cluster; pp.unit_base = q.unit_base; pp.blk = q.blk;
        pp.out = dp.out.as<pf::PlanOut>();
        hipLaunchKernelGGL(pf::plan_count_kernel, dim3(nblk), dim3(pf::PLAN_THREADS), 0, c->stream, pp);
        hipLaunchKernelGGL(pf::plan_scan_kernel, dim3(1), dim3(256), 0, c->stream, pp, nblk);
        hipLaunchKernelGGL(pf::plan_scatter_kernel, dim3(nblk), dim3(pf::PLAN_THREADS), 0, c->stream, pp);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(dp.pin_out, dp.out.p, sizeof(pf::PlanOut), hipMemcpyDeviceToHost, c->stream));
        return PF_OK;
    };
    if (use_plan && C > c->dp_const_n) {
        // zeros | ones | 0, 1, 2, ...: item_part / extra_first / is_extra / binned, item_nparts / nsib / compact, slice / sib0
        const uint32_t n = C + C / 4 + 64;
        std::vector<uint32_t> h(3 * (size_t)n, 0u);
        for (uint32_t i = 0; i < n; i++) { h[n + i] = 1u; h[2 * (size_t)n + i] = i; }
        PFCHK(c->dp_const.ensure(h.size() * 4));
        HIPCHK(hipMemcpy(c->dp_const.p, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        c->dp_const_n = n;
    }
    // what cluster_ninst_kernel works out per cluster for plan_kernel (nothing when the host builds every item)
    pf::PlanClassify pcls{};
    if (use_plan) {
        PFCHK(c->plan_room.ensure((size_t)C * 4));
        PFCHK(c->plan_arena.ensure((size_t)C * 4));
        pcls.extra_off = c->extra_off.as<uint32_t>(); pcls.plan_room = c->plan_room.as<uint32_t>(); pcls.plan_arena = c->plan_arena.as<uint32_t>();
        pcls.mult = c->o.canon ? 1u : 2u; pcls.NS = NS; pcls.W = W; pcls.unit_view = (c->o.flags & PF_FLAG_NO_UNIT_DEDUP) ? 0u : 1u;
        pcls.reg_ready = reg_ready ? 1u : 0u; pcls.share = share; pcls.reg_a = reg_a; pcls.reg_b = reg_b; pcls.reg_half_sd = reg_half_sd;
    }
    // the dedup of part h and its results on their way to pinned memory (ev_part[h]); all parts are queued up front
    auto launch_dedup_part = [&](uint32_t h) -> int {
        const uint32_t c0 = h ? part_end[h - 1] : 0, c1 = part_end[h], n = c1 - c0;
        if (n) {
            dp.cluster_base = c0;
            PFCHK(mark_begin(c, 3));
            hipLaunchKernelGGL(pf::cluster_dedup_kernel<pf::DedupSmall>, dim3(n), dim3(pf::DEDUP_THREADS), 0, c->stream, dp);
            HIPCHK(hipGetLastError());
            PFCHK(mark_end(c));
            hipLaunchKernelGGL(pf::cluster_ninst_kernel, dim3((n + 3) / 4), dim3(256), 0, c->stream, d.cluster_seg_off,
                               d.seg_len, c->v_len.as<uint32_t>(), c->v_nseg.as<uint32_t>(), c->o.klength, c0, c1,
                               (const uint32_t*)nullptr,
                               c->v_mode.as<uint32_t>(), c->v_dense.as<uint32_t>(), c->v_nstr.as<uint32_t>(), c->cl_rec.as<pf::ClusterRec>(), pcls);
            HIPCHK(hipGetLastError());
            if (use_plan) PFCHK(launch_plan(h, c0, c1));
            HIPCHK(hipMemcpyAsync(rec + c0, c->cl_rec.as<pf::ClusterRec>() + c0, (size_t)n * sizeof(pf::ClusterRec), hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(hipEventRecord(c->ev_part[h], c->stream));
        return PF_OK;
    };
    if (C)
        for (uint32_t h = 0; h < P; h++) PFCHK(launch_dedup_part(h));
    // ---- strand bits of target-strain segments (canonical mode)
    c->n_strand_words = (b->seg_strand_off && c->o.canon) ? b->n_strand_words : 0;
    if (c->n_strand_words && NSEG) {
        PFCHK(c->strand_bits.ensure((size_t)c->n_strand_words * 8));
        HIPCHK(hipMemsetAsync(c->strand_bits.p, 0, (size_t)c->n_strand_words * 8, c->stream));
        const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)NSEG + 3) / 4, 4096);
        auto strand = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, c->stream, d.packed, d.seg_word_off, d.seg_len,
                               d.seg_strand_off, NSEG, c->o.klength, c->strand_bits.as<uint64_t>());
        };
        switch (KW) {
            case 1: strand(pf::strand_bits_kernel<1>); break;
            case 2: strand(pf::strand_bits_kernel<2>); break;
            case 3: strand(pf::strand_bits_kernel<3>); break;
            default: strand(pf::strand_bits_kernel<4>); break;
        }
        HIPCHK(hipGetLastError());
    }
    lap("upload+dedup launch");
    const uint64_t mult = c->o.canon ? 1 : 2;
    uint64_t total_inst = 0;

    c->pid0 = c->n_patterns;
    if (!rerun) c->n_submits++;
    c->cluster_arena.assign(C, 0);

    std::vector<uint32_t> nparts(C, 1);
    std::vector<uint32_t> todo;
    // a deduplicated cluster whose distinct sequences alone carry far more windows than one table holds will
    // overflow it: start it with two key partitions instead of paying for a failed first scan (a wrong guess
    // only costs time: an overflow still triggers the doubling retry)
    // Estimate of the distinct windows of D near-identical sequences of average length L = vinst / D: the first
    // contributes all of its windows, every further one the share a 1 % divergence touches (1 - 0.99^k: 27 % of the
    // 31-mers, 40 % of the 51-mers) plus a margin.
    // what the host does with a part's dedup results once they have arrived (ev_part[h])
    std::vector<uint32_t> wide_list;
    auto prep_half = [&](int h) -> int {
        const uint32_t c0 = h ? part_end[h - 1] : 0, c1 = part_end[h];
        // clusters the small dedup class gave up on for lack of room (more than 64 distinct sequences, a sample-set
        // matrix or an ordinal bitmap that does not fit): the wide class on those alone, then their counts again
        wide_list.clear();
        for (uint32_t i = c0; i < c1; i++) if (rec[i].mode & pf::MODE_RETRY_WIDE) wide_list.push_back(i);
        if (!wide_list.empty()) {
            const uint32_t nw = (uint32_t)wide_list.size();
            PFCHK(c->wide_list.ensure((size_t)nw * 4));
            HIPCHK(hipMemcpyAsync(c->wide_list.p, wide_list.data(), (size_t)nw * 4, hipMemcpyHostToDevice, c->stream));
            pf::DedupParams dw = dp;
            dw.cluster_base = 0; dw.cluster_list = c->wide_list.as<uint32_t>();
            PFCHK(mark_begin(c, 3));
            hipLaunchKernelGGL(pf::cluster_dedup_kernel<pf::DedupWide>, dim3(nw), dim3(pf::DEDUP_THREADS), 0, c->stream, dw);
            HIPCHK(hipGetLastError());
            PFCHK(mark_end(c));
            hipLaunchKernelGGL(pf::cluster_ninst_kernel, dim3((nw + 3) / 4), dim3(256), 0, c->stream, d.cluster_seg_off,
                               d.seg_len, c->v_len.as<uint32_t>(), c->v_nseg.as<uint32_t>(), c->o.klength, 0u, nw,
                               c->wide_list.as<uint32_t>(),
                               c->v_mode.as<uint32_t>(), c->v_dense.as<uint32_t>(), c->v_nstr.as<uint32_t>(), c->cl_rec.as<pf::ClusterRec>(),
                               pf::PlanClassify{});        // (the wide class's clusters are the host's: no plan bits)
            HIPCHK(hipGetLastError());
            const size_t n = c1 - c0;
            HIPCHK(hipMemcpyAsync(rec + c0, c->cl_rec.as<pf::ClusterRec>() + c0, n * sizeof(pf::ClusterRec), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            c->timing.n_wide_clusters += nw;
        }
        for (uint32_t i = c0; i < c1; i++) {
            rec[i].mode &= 3u;
            if (rec[i].ninst * mult >= 0xFFFFFFF0ull) return fail(PF_ERR_ARG, "cluster %u has too many k-mer instances", i);
            total_inst += rec[i].ninst * mult;
            c->timing.n_dedup_clusters += rec[i].mode ? 1u : 0u;
            if (rec[i].pad & pf::PLAN_PLANNED) continue;          // laid out by plan_kernel: one key partition
            if (rec[i].mode && rec[i].vnstr) {
                const double D = (double)rec[i].vnstr, L = (double)(rec[i].vinst * mult) / D;
                const double est = L * (1.0 + share * (D - 1.0));
                const double room = 0.9 * (double)pf::insert_limit(NS);
                // The estimate is right for SURVEY 8d's alleles, each with its own substitutions and flanks.  The many
                // alleles of a population descend from one another and share far more (tens of new k-mers each, not
                // hundreds): past 24 distinct sequences the cluster starts as ONE item instead, and if that overflows
                // the scan reports how far it came and the retry gets the partitions it needs.  A failed first attempt
                // costs 1/P of the P scans that follow; an over-partitioned cluster costs every surplus scan in full.
                // Once the context has scanned enough such clusters it knows what a further sequence brings in THIS
                // pangenome (the line through what it observed, plus a margin) and the first
                // attempt is sized by that.
                if (D >= 2.0 && reg_ready) {
                    const double g = std::max(0.0, reg_a + reg_b * L) + reg_half_sd;      // (the line: see launch_plan)
                    const double est2 = L + g * (D - 1.0);
                    // (an estimate never asks for more items than a sub-batch holds: the cluster then starts with what
                    // fits and an overflowing scan says how many partitions it really needs)
                    if (est2 > room)
                        nparts[i] = (uint32_t)std::min<double>(std::ceil(est2 / room), (double)std::min(4096u, std::max(1u, c->max_items / 2)));
                } else if (est > room) {
                    nparts[i] = D > 24.0 ? 1u : (uint32_t)std::min<double>(std::ceil(est / room), (double)std::min(64u, std::max(1u, c->max_items / 2)));
                }
            }
        }
        // ---- unit view: identical 64-window units among the distinct sequences of a cluster are scanned once
        // (unit_class_kernel).  A cluster's pieces number at most the units of its plain view: that is its room in
        // this part's pool.
        if (!(c->o.flags & PF_FLAG_NO_UNIT_DEDUP)) {
            pf_ctx::UPool& up = c->upool[2 * h + 1];
            const size_t need_pin = (size_t)(c1 - c0) * 8 + 64;
            if (need_pin > up.pin_cap) {
                if (up.pin) (void)hipHostFree(up.pin);
                up.pin = nullptr; up.pin_cap = 0;
                hipError_t e = hipHostMalloc((void**)&up.pin, need_pin + need_pin / 4, hipHostMallocDefault);
                if (e != hipSuccess) return fail(PF_ERR_OOM, "hipHostMalloc(%zu) failed: %s", need_pin, hipGetErrorString(e));
                up.pin_cap = need_pin + need_pin / 4;
            }
            // clusters of up to 64 distinct sequences first (one wave each, no table), then the wider ones
            uint32_t nu = 0, nsmall = 0;
            uint64_t room = 0;
            uint32_t* lc = up.pin;
            auto takes = [&](uint32_t i) { return !(rec[i].pad & pf::PLAN_PLANNED) && rec[i].mode && rec[i].vnstr >= 2 && rec[i].words && room + rec[i].words / 2 < 0x7FFFFFF0ull; };
            for (uint32_t i = c0; i < c1; i++)
                if (rec[i].vnstr <= pf::UNIT_SMALL_MAX_D && takes(i)) { lc[nu++] = i; room += rec[i].words / 2; }
            nsmall = nu;
            for (uint32_t i = c0; i < c1; i++)
                if (rec[i].vnstr > pf::UNIT_SMALL_MAX_D && takes(i)) { lc[nu++] = i; room += rec[i].words / 2; }
            uint32_t* lb = up.pin + nu;
            room = 0;
            for (uint32_t j = 0; j < nu; j++) { lb[j] = (uint32_t)room; room += rec[lc[j]].words / 2; }
            if (nu) {
                // (twice the room: the wide kernel parks a cluster's pieces in the second stretch before it orders them by chunk)
                const size_t R = (size_t)room + 1, R2 = 2 * R;
                if (R2 > 0xFFFFFFF0ull) return fail(PF_ERR_CAPACITY, "unit view of %zu pieces: submit fewer clusters at a time", R);
                PFCHK(up.word_off.ensure(R2 * 8)); PFCHK(up.len.ensure(R2 * 4)); PFCHK(up.sample.ensure(R2 * 4));
                PFCHK(up.ord.ensure(R2 * 4)); PFCHK(up.bits.ensure(R2 * 4)); PFCHK(up.list.ensure((size_t)nu * 8));
                HIPCHK(hipMemcpyAsync(up.list.p, up.pin, (size_t)nu * 8, hipMemcpyHostToDevice, c->stream));
                pf::UnitParams q{};
                q.packed = d.packed; q.cluster_seg_off = d.cluster_seg_off; q.v_nstr = c->v_nstr.as<uint32_t>();
                q.list_cluster = up.list.as<uint32_t>(); q.list_base = up.list.as<uint32_t>() + nu;
                q.v_word_off = c->v_word_off.as<uint64_t>(); q.v_len = c->v_len.as<uint32_t>(); q.v_ord = c->v_ord.as<uint32_t>();
                q.u_word_off = up.word_off.as<uint64_t>(); q.u_len = up.len.as<uint32_t>(); q.u_sample = up.sample.as<uint32_t>();
                q.u_ord = up.ord.as<uint32_t>(); q.u_bits = up.bits.as<uint32_t>();
                q.v_nseg = c->v_nseg.as<uint32_t>(); q.view_off = c->view_off.as<uint32_t>(); q.k = c->o.klength; q.tmp_off = (uint32_t)R;
                PFCHK(mark_begin(c, 3));
                if (nsmall) {
                    const dim3 g((nsmall + 3) / 4), b(256);
                    switch ((63 + c->o.klength + 31) / 32) {       // words a unit's 63 + k bases take
                        case 2: hipLaunchKernelGGL(pf::unit_class_small_kernel<2>, g, b, 0, c->stream, q, nsmall); break;
                        case 3: hipLaunchKernelGGL(pf::unit_class_small_kernel<3>, g, b, 0, c->stream, q, nsmall); break;
                        case 4: hipLaunchKernelGGL(pf::unit_class_small_kernel<4>, g, b, 0, c->stream, q, nsmall); break;
                        case 5: hipLaunchKernelGGL(pf::unit_class_small_kernel<5>, g, b, 0, c->stream, q, nsmall); break;
                        default: hipLaunchKernelGGL(pf::unit_class_small_kernel<6>, g, b, 0, c->stream, q, nsmall); break;
                    }
                    HIPCHK(hipGetLastError());
                }
                if (nu > nsmall) {
                    pf::UnitParams qw = q;
                    qw.list_cluster += nsmall; qw.list_base += nsmall;
                    hipLaunchKernelGGL(pf::unit_class_kernel, dim3(nu - nsmall), dim3(pf::UNIT_THREADS), 0, c->stream, qw);
                    HIPCHK(hipGetLastError());
                }
                PFCHK(mark_end(c));
            }
        }
        return PF_OK;
    };
    // ---- parameter blocks of the scan and the fused finish kernels, from a pass's item arrays (the host's staged
    // upload, or plan_kernel's block with the constant arrays standing in for what is the same for every simple cluster)
    struct ItemPtrs { const uint32_t *cluster, *part, *nparts, *nslots, *slice, *compact, *binned; uint32_t* count; };
    auto host_items = [&]() {
        return ItemPtrs{c->it_cluster.as<uint32_t>(), c->it_part.as<uint32_t>(), c->it_nparts.as<uint32_t>(), c->it_nslots.as<uint32_t>(),
                        c->it_slice.as<uint32_t>(), c->it_compact.as<uint32_t>(), c->it_binned.as<uint32_t>(), c->it_count.as<uint32_t>()};
    };
    auto scan_params = [&](const ItemPtrs& ip, const pf_ctx::UPool& up, const uint32_t* work) {
        pf::ScanParams sp{};
        sp.packed = d.packed; sp.seg_word_off = c->v_word_off.as<uint64_t>(); sp.seg_len = c->v_len.as<uint32_t>();
        sp.seg_sample = c->v_sample.as<uint32_t>(); sp.seg_ord_base = c->v_ord.as<uint32_t>();
        sp.seg_bits = c->v_bits.as<uint32_t>();
        sp.u_word_off = up.word_off.as<uint64_t>(); sp.u_len = up.len.as<uint32_t>(); sp.u_sample = up.sample.as<uint32_t>();
        sp.u_ord_base = up.ord.as<uint32_t>(); sp.u_bits = up.bits.as<uint32_t>();
        sp.cluster_seg_off = c->view_off.as<uint32_t>(); sp.cluster_vnseg = c->v_nseg.as<uint32_t>();
        sp.cluster_vnstr = c->v_nstr.as<uint32_t>();
        sp.item_cluster = ip.cluster; sp.item_part = ip.part; sp.item_nparts = ip.nparts; sp.item_nslots = ip.nslots;
        sp.item_scratch = ip.slice; sp.item_compact = ip.compact;
        sp.cmask_lo = c->cmask_lo.as<uint32_t>(); sp.cmask_hi = c->cmask_hi.as<uint32_t>();
        sp.tab_key = c->tab_key.as<uint64_t>(); sp.tab_ord = c->tab_ord.as<uint32_t>();
        sp.chunkbits = c->chunkbits.as<uint32_t>(); sp.chunkmask = c->chunkmask.as<uint32_t>();
        sp.item_count = ip.count; sp.cluster_overflow = c->cl_overflow.as<uint32_t>();
        sp.work = work;
        sp.k = c->o.klength; sp.W = W; sp.NS = NS;
        sp.item_binned = ip.binned;
        return sp;
    };
    auto finish_params = [&](const ItemPtrs& ip, Arena* ar) {
        pf::FinishParams fp{};
        fp.item_cluster = ip.cluster; fp.item_nslots = ip.nslots;
        fp.item_scratch = ip.slice; fp.cluster_overflow = c->cl_overflow.as<uint32_t>();
        fp.item_nparts = ip.nparts;
        fp.cluster_seg_off = d.cluster_seg_off; fp.seg_sample = d.seg_sample;
        fp.seg_distinct = c->seg_distinct.as<uint32_t>();
        fp.v_nstr = c->v_nstr.as<uint32_t>(); fp.v_dense = c->v_dense.as<uint32_t>();
        fp.cluster_nstrains = d.cluster_nstrains; fp.cluster_npresab = d.cluster_npresab;
        fp.cluster_presab = d.cluster_presab; fp.cluster_ordinal = d.cluster_ordinal;
        fp.maf_lo = c->d_maf_lo.as<uint32_t>(); fp.maf_hi = c->d_maf_hi.as<uint32_t>();
        fp.tab_key = c->tab_key.as<uint64_t>(); fp.tab_ord = c->tab_ord.as<uint32_t>();
        fp.cmask_lo = c->cmask_lo.as<uint32_t>(); fp.cmask_hi = c->cmask_hi.as<uint32_t>();
        fp.item_count = ip.count;
        fp.extra_off = c->extra_off.as<uint32_t>(); fp.extra_dense = c->extra_dense.as<uint32_t>();
        fp.extra_bits = d.extra_bits;
        fp.out_key = ar->key.as<uint64_t>(); fp.out_pid = ar->pid.as<uint32_t>();
        fp.cluster_kmer_off = c->cl_kmer_off.as<uint64_t>(); fp.cluster_kmer_cnt = c->cl_kmer_cnt.as<uint32_t>();
        fp.cluster_unique = c->cl_unique.as<uint32_t>(); fp.cluster_pattern = c->cl_pattern.as<uint32_t>();
        fp.cursor = c->cursor.as<uint64_t>(); fp.pt = c->pt;
        fp.pat_bits = c->pat_bits.as<uint32_t>();
        fp.pat_nan = c->o.consider_missing ? c->pat_nan.as<uint32_t>() : nullptr;
        fp.pat_n = c->pat_n.as<uint32_t>();
        fp.out_base = ar->base; fp.out_cap = ar->cap; fp.W = W; fp.NS = NS; fp.KW = KW;
        fp.consider_missing = c->o.consider_missing; fp.patfilt = c->o.patfilt; fp.multiple_files = c->o.multiple_files;
        return fp;
    };
    uint64_t arena_base = 0;
    struct Deferred { Arena* ar; uint32_t pin; };
    std::vector<Deferred> deferred;        // passes launched and not yet waited for (their cursor read-backs are queued)
    uint32_t arena_i = 0;                  // arenas used so far: one per launched pass, the device-planned ones included
    // ---- the device-planned clusters of part h: unit view, scan, fused finish -- launched from plan_kernel's 40-byte summary
    auto launch_planned = [&](uint32_t h) -> int {
        pf_ctx::DPlan& dp = c->dplan[h];
        const pf::PlanOut po = *dp.pin_out;
        const uint32_t n = po.n_items;
        if (!n) return PF_OK;
        if (n > c->max_items || n > dp.n || po.n_fin + po.n_fin2 + po.n_fin5 != n || po.n_unit > n)
            return fail(PF_ERR_STATE, "plan_kernel summary out of range (%u items of %u clusters)", n, dp.n);
        const DPtrs q = dplan_ptrs(dp);
        const uint32_t* zeros = c->dp_const.as<uint32_t>();
        const uint32_t* ones = zeros + c->dp_const_n;
        const uint32_t* iota = zeros + 2 * (size_t)c->dp_const_n;
        const ItemPtrs ip{q.it_cluster, zeros, ones, q.it_nslots, iota, ones, zeros, dp.it_count.as<uint32_t>()};
        while (c->arenas.size() <= arena_i) c->arenas.push_back(new Arena());
        Arena* ar = c->arenas[arena_i];
        ar->cap = std::max<uint64_t>(po.arena_cap, 1);
        ar->base = arena_base;
        PFCHK(ar->key.ensure((size_t)ar->cap * 8 * KW));
        PFCHK(ar->pid.ensure((size_t)ar->cap * 4));
        PFCHK(ar->first.ensure((size_t)ar->cap * 8));
        c->pin_small[arena_i & 15] = arena_base;
        HIPCHK(hipMemcpyAsync(c->cursor.p, &c->pin_small[arena_i & 15], 8, hipMemcpyHostToDevice, c->stream));
        pf_ctx::UPool& up = c->upool[2 * h];
        if (po.n_unit) {
            const size_t R = (size_t)po.unit_room + 1, R2 = 2 * R;
            if (R2 > 0xFFFFFFF0ull) return fail(PF_ERR_CAPACITY, "unit view of %zu pieces: submit fewer clusters at a time", R);
            PFCHK(up.word_off.ensure(R2 * 8)); PFCHK(up.len.ensure(R2 * 4)); PFCHK(up.sample.ensure(R2 * 4));
            PFCHK(up.ord.ensure(R2 * 4)); PFCHK(up.bits.ensure(R2 * 4));
            pf::UnitParams uq{};
            uq.packed = d.packed; uq.cluster_seg_off = d.cluster_seg_off; uq.v_nstr = c->v_nstr.as<uint32_t>();
            uq.list_cluster = q.unit_cluster; uq.list_base = q.unit_base;
            uq.v_word_off = c->v_word_off.as<uint64_t>(); uq.v_len = c->v_len.as<uint32_t>(); uq.v_ord = c->v_ord.as<uint32_t>();
            uq.u_word_off = up.word_off.as<uint64_t>(); uq.u_len = up.len.as<uint32_t>(); uq.u_sample = up.sample.as<uint32_t>();
            uq.u_ord = up.ord.as<uint32_t>(); uq.u_bits = up.bits.as<uint32_t>();
            uq.v_nseg = c->v_nseg.as<uint32_t>(); uq.view_off = c->view_off.as<uint32_t>(); uq.k = c->o.klength; uq.tmp_off = (uint32_t)R;
            PFCHK(mark_begin(c, 3));
            const dim3 g((po.n_unit + 3) / 4), b(256);
            switch ((63 + c->o.klength + 31) / 32) {       // words a unit's 63 + k bases take
                case 2: hipLaunchKernelGGL(pf::unit_class_small_kernel<2>, g, b, 0, c->stream, uq, po.n_unit); break;
                case 3: hipLaunchKernelGGL(pf::unit_class_small_kernel<3>, g, b, 0, c->stream, uq, po.n_unit); break;
                case 4: hipLaunchKernelGGL(pf::unit_class_small_kernel<4>, g, b, 0, c->stream, uq, po.n_unit); break;
                case 5: hipLaunchKernelGGL(pf::unit_class_small_kernel<5>, g, b, 0, c->stream, uq, po.n_unit); break;
                default: hipLaunchKernelGGL(pf::unit_class_small_kernel<6>, g, b, 0, c->stream, uq, po.n_unit); break;
            }
            HIPCHK(hipGetLastError());
            PFCHK(mark_end(c));
        }
        {
            const pf::ScanParams sp = scan_params(ip, up, q.w_scan);
            PFCHK(mark_begin(c, 0));
            PFCHK(launch_scan(c, sp, n));
            PFCHK(mark_end(c));
            c->timing.scan_launches++;
        }
        {
            pf::FinishParams fp = finish_params(ip, ar);
            PFCHK(mark_begin(c, 6));
            if (po.n_fin5) {
                fp.work = q.w_fin5;
                hipLaunchKernelGGL((pf::finish_kernel<pf::FinHuge, true>), dim3(po.n_fin5), dim3(pf::FinHuge::THREADS), 0, c->stream, fp);
                HIPCHK(hipGetLastError());
            }
            if (po.n_fin2) {
                fp.work = q.w_fin2;
                hipLaunchKernelGGL((pf::finish_kernel<pf::FinLarge, false>), dim3(po.n_fin2), dim3(pf::FinLarge::THREADS), 0, c->stream, fp);
                HIPCHK(hipGetLastError());
            }
            if (po.n_fin) {
                fp.work = q.w_fin;
                hipLaunchKernelGGL((pf::finish_kernel<pf::FinSmall, false>), dim3(po.n_fin), dim3(pf::FinSmall::THREADS), 0, c->stream, fp);
                HIPCHK(hipGetLastError());
            }
            PFCHK(mark_end(c));
        }
        c->timing.n_items += n;
        c->timing.n_device_planned += n;
        const uint32_t pin = 16 + (arena_i & 31);
        HIPCHK(hipMemcpyAsync(&c->pin_small[pin], c->cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
        deferred.push_back(Deferred{ar, pin});
        arena_base += ar->cap;
        arena_i++;
        return PF_OK;
    };
    uint32_t pass = 0;                     // host-planned passes: the parts' (what plan_kernel left of them), then the re-runs
    uint32_t cnt2[3] = {0, 0, 0};          // pattern counters {ids handed out, pool overflow, arena overflow}
    c->counters = pf_result{};
    const uint32_t lim_full = pf::insert_limit(NS);
    for (;;) {
        c->stage_slot = (int)(pass & 1);
        if (pass < P) {
            // this part's dedup results (queued with the others up front) have to be here; its clusters are the pass
            HIPCHK(hipEventSynchronize(c->ev_part[pass]));
            if (pass == 0) lap("first part's dedup results");
            if (pass == 0 && ex_on_device) {
                if (h_exfirst[C + 1]) {
                    HIPCHK(hipStreamSynchronize(c->stream));
                    return fail(PF_ERR_ARG, "extra_cluster must be non-decreasing and < n_clusters");
                }
                ex_first.assign(h_exfirst, h_exfirst + C + 1);
            }
            // the clusters plan_kernel laid out go first: the GPU starts on them while the rest of the part is built here
            const uint32_t planned_arena = arena_i;
            if (use_plan) PFCHK(launch_planned(pass));
            PFCHK(prep_half((int)pass));
            lap("  prep (records, unit view)");
            todo.clear();
            for (uint32_t ci = pass ? part_end[pass - 1] : 0; ci < part_end[pass]; ci++) {
                if (!(rec[ci].pad & pf::PLAN_PLANNED)) { todo.push_back(ci); continue; }
                c->timing.scan_packed_bytes += rec[ci].words * 8;
                c->cluster_arena[ci] = planned_arena;
            }
        }
        if (todo.empty() && pass >= P) break;
        // ---- items of this pass
        std::vector<Item>& items = c->hs_items;
        std::vector<uint8_t>& item_fused = c->hs_fused;      // 0 unfused, 1 fused small class, 2 fused large class
        items.clear(); item_fused.clear();
        items.reserve(todo.size() + 64); item_fused.reserve(todo.size() + 64);
        struct Sub { uint32_t item0, nitems, cl0, ncl, pool, bin0, nbin; uint64_t q_total; };
        // A cluster of three or more key partitions (a dedup view, keys of up to two words) has its windows sorted by
        // partition first (bin_kernel): its items then read their own windows instead of each walking the whole view.
        // The entry arrays belong to a sub-batch; a sub-batch ends where they would pass BIN_MAX_ENTRIES.
        constexpr uint32_t BIN_MIN_PARTS = 3;      // (at two, a cluster's own bin_kernel workgroup takes longer than the second walk it saves)
        constexpr uint64_t BIN_MAX_ENTRIES = 1ull << 28;
        std::vector<uint32_t>& v_binned = c->hs_binned;
        std::vector<uint32_t>&bin_cluster = c->hs_bin[0], &bin_item0 = c->hs_bin[1], &bin_nparts = c->hs_bin[2], &bin_base = c->hs_bin[3];
        v_binned.clear(); bin_cluster.clear(); bin_item0.clear(); bin_nparts.clear(); bin_base.clear();
        // (which unit-view pool a cluster's view lies in: its part's device-planned one or the host-planned one)
        auto part_of = [&](uint32_t ci) { uint32_t q = 0; while (q + 1 < P && ci >= part_end[q]) q++; return 2 * q + ((rec[ci].pad & pf::PLAN_PLANNED) ? 0u : 1u); };
        std::vector<Sub> subs;
        std::vector<uint32_t>&sub_cluster = c->hs_sub[0], &sub_item0 = c->hs_sub[1], &sub_nitems = c->hs_sub[2];
        sub_cluster.clear(); sub_item0.clear(); sub_nitems.clear();
        uint64_t arena_cap = 0;
        Sub cur{0, 0, 0, 0, todo.empty() ? 0u : part_of(todo[0]), 0, 0, 0};
        for (uint32_t ci : todo) {
            const uint32_t np = nparts[ci];
            const uint32_t nex = ex_first[ci + 1] - ex_first[ci];
            // a deduplicated cluster that is one work item (or a few key partitions) is finished by one fused kernel
            // (rows + emit in LDS); its slow-path rows (a few k-mers around an 'N') are folded in by that kernel
            const uint32_t mwords = rec[ci].vnstr * ((W + 3) & ~3u);
            uint8_t fused = 0;   // 1/2: single item, small/large class; 3: first of several partitions; 4: the others;
                                 // 5: single item, huge class
            if (rec[ci].mode == 1 && nex <= pf::FUSED_MAX_EXTRA && NS <= 9600) {
                const bool fits_large = rec[ci].dense < pf::FinLarge::DW * 32 - 1 && mwords <= pf::FinLarge::MR;
                const bool fits_huge = rec[ci].dense < pf::FinHuge::DW * 32 - 1 && mwords <= pf::FinHuge::MR;
                if (np == 1) {
                    if (rec[ci].dense < pf::FinSmall::DW * 32 - 1 && mwords <= pf::FinSmall::MR) fused = 1;
                    else if (fits_large) fused = 2;
                    else if (fits_huge) fused = 5;
                } else if (fits_large) fused = 3;
                // (several partitions of a cluster that large stay on the general path, one workgroup each: the one
                // workgroup of the fused kernel took 15.9 ms where rows + emit + pattern rows take 4.5, 2 000 clusters
                // of 60 SURVEY alleles)
            }
            const uint32_t nex_items = fused ? 0 : (nex + lim_full - 1) / lim_full;
            const uint32_t nit = np + nex_items;
            if (nit > c->max_items) {
                if (c->max_items == 0) return fail(PF_ERR_STATE, "the context lost its scratch in a failed enlargement");
                // (sub-batches already built for this pass hold at most the old max_items items each: still valid)
                PFCHK(grow_scratch(c, nit));
            }
            const uint64_t qn = rec[ci].vinst * mult;          // entries of the cluster's queue at most (its view's windows)
            const uint32_t vch = (rec[ci].vnstr + 31) / 32;
            const bool binned = !(c->o.flags & PF_FLAG_NO_KEY_BINNING) && np >= BIN_MIN_PARTS && KW <= 2 && rec[ci].mode != 0 &&
                                vch <= pf::BIN_CHUNKS && (uint64_t)np * vch <= pf::BIN_CELLS && qn && qn <= BIN_MAX_ENTRIES;
            // (a launch reads one unit-view pool: a re-run pass does not mix the pools' clusters in a sub-batch)
            if (cur.nitems + nit > c->max_items || (cur.nitems && part_of(ci) != cur.pool) ||
                (binned && cur.q_total + qn > BIN_MAX_ENTRIES)) {
                subs.push_back(cur);
                cur = Sub{(uint32_t)items.size(), 0, (uint32_t)sub_cluster.size(), 0, part_of(ci), (uint32_t)bin_cluster.size(), 0, 0};
            }
            if (!cur.nitems) cur.pool = part_of(ci);
            const uint32_t sib0 = (uint32_t)items.size();
            // table size: a cluster that cannot overflow a small table gets one (less flush traffic)
            uint32_t ns = NS;
            const uint64_t inst = rec[ci].vinst * mult;
            if (np == 1 && NS > 4096 + pf::INSERT_SLACK && inst <= pf::insert_limit(4096)) ns = 4096;
            else if (np == 1 && NS > 6144 + pf::INSERT_SLACK && inst <= pf::insert_limit(6144)) ns = 6144;
            for (uint32_t q = 0; q < np; q++) {
                items.push_back(Item{ci, q, np, ns, cur.nitems + q, sib0, nit, 0, 0});
                item_fused.push_back(fused == 3 && q > 0 ? 4 : fused);
            }
            v_binned.resize(items.size() + nex_items, 0);
            if (binned) {
                for (uint32_t q = 0; q < np; q++) v_binned[sib0 + q] = 1;
                bin_cluster.push_back(ci); bin_item0.push_back(sib0); bin_nparts.push_back(np);
                bin_base.push_back((uint32_t)cur.q_total);
                cur.q_total += qn; cur.nbin++;
                c->timing.n_binned_clusters++;
            }
            for (uint32_t q = 0; q < nex_items; q++) {
                const uint32_t first = ex_first[ci] + q * lim_full;
                const uint32_t cnt = std::min(lim_full, ex_first[ci + 1] - first);
                items.push_back(Item{ci, 0, 1, cnt, cur.nitems + np + q, sib0, nit, first, 1});
                item_fused.push_back(0);
            }
            for (uint32_t q = 0; q < np; q++) arena_cap += std::min<uint64_t>(pf::insert_limit(ns), inst);
            arena_cap += nex;
            if (!fused) {
                sub_cluster.push_back(ci);
                sub_item0.push_back(sib0);
                sub_nitems.push_back(nit);
                cur.ncl++;
            }
            cur.nitems += nit;
            c->cluster_arena[ci] = arena_i;
        }
        if (cur.nitems) subs.push_back(cur);

        lap("  items");
        // ---- arena of this pass
        while (c->arenas.size() <= arena_i) c->arenas.push_back(new Arena());
        Arena* ar = c->arenas[arena_i];
        ar->cap = std::max<uint64_t>(arena_cap, 1);
        ar->base = arena_base;
        PFCHK(ar->key.ensure((size_t)ar->cap * 8 * KW));
        PFCHK(ar->pid.ensure((size_t)ar->cap * 4));
        PFCHK(ar->first.ensure((size_t)ar->cap * 8));

        // ---- item arrays
        const size_t NI = items.size();
        std::vector<uint32_t>&v_cluster = c->hs_v[0], &v_part = c->hs_v[1], &v_nparts = c->hs_v[2], &v_nslots = c->hs_v[3],
            &v_slice = c->hs_v[4], &v_sib0 = c->hs_v[5], &v_nsib = c->hs_v[6], &v_exfirst = c->hs_v[7], &v_isex = c->hs_v[8],
            &v_compact = c->hs_v[9], &w_scan = c->hs_w[0], &w_extra = c->hs_w[1], &w_fin = c->hs_w[2], &w_fin2 = c->hs_w[3],
            &w_fin3 = c->hs_w[4], &w_rows = c->hs_w[5], &w_fin5 = c->hs_w[6];
        for (auto& v : c->hs_v) v.resize(NI);
        v_binned.resize(NI, 0);
        for (auto& v : c->hs_w) { v.clear(); v.reserve(NI); }
        for (size_t i = 0; i < NI; i++) {
            v_cluster[i] = items[i].cluster; v_part[i] = items[i].part; v_nparts[i] = items[i].nparts;
            v_nslots[i] = items[i].nslots; v_slice[i] = items[i].slice; v_sib0[i] = items[i].sib0;
            v_nsib[i] = items[i].nsib; v_exfirst[i] = items[i].extra_first; v_isex[i] = items[i].is_extra;
            v_compact[i] = item_fused[i] ? 1 : 0;
        }
        lap("  arena + item columns");
        PFCHK(c->it_count.ensure(std::max<size_t>(NI, 1) * 4));
        PFCHK(c->it_unique.ensure(std::max<size_t>(NI, 1) * 4));
        PFCHK(c->it_kept.ensure(std::max<size_t>(NI, 1) * 4));
        // work lists per sub-batch, concatenated; scan items heaviest first (the grid drains evenly)
        std::vector<uint32_t> scan_off(subs.size() + 1, 0), extra_off(subs.size() + 1, 0), fin_off(subs.size() + 1, 0),
            fin2_off(subs.size() + 1, 0), fin3_off(subs.size() + 1, 0), fin5_off(subs.size() + 1, 0), rows_off(subs.size() + 1, 0);
        // heaviest items first inside each launch (the grid then drains evenly): a coarse O(n) order by
        // log2(scan instances) is enough
        auto wclass = [&](uint32_t it) -> int {
            const uint64_t w = rec[items[it].cluster].vinst;
            return w ? 63 - __builtin_clzll(w) : 0;
        };
        std::vector<uint32_t> tmp_scan, tmp_fin, tmp_fin2;
        auto append_by_weight = [&](std::vector<uint32_t>& src, std::vector<uint32_t>& dst) {
            if (src.size() > 64) {
                size_t cnt[65] = {0};
                for (uint32_t it : src) cnt[64 - wclass(it)]++;
                size_t run = dst.size();
                for (int b = 0; b < 65; b++) { const size_t n = cnt[b]; cnt[b] = run; run += n; }
                dst.resize(run);
                for (uint32_t it : src) dst[cnt[64 - wclass(it)]++] = it;
            } else {
                dst.insert(dst.end(), src.begin(), src.end());
            }
            src.clear();
        };
        for (size_t s = 0; s < subs.size(); s++) {
            for (uint32_t i = subs[s].item0; i < subs[s].item0 + subs[s].nitems; i++) {
                if (items[i].is_extra) w_extra.push_back(i); else tmp_scan.push_back(i);
                if (item_fused[i] == 1) tmp_fin.push_back(i);
                else if (item_fused[i] == 2) tmp_fin2.push_back(i);
                else if (item_fused[i] == 3) w_fin3.push_back(i);
                else if (item_fused[i] == 5) w_fin5.push_back(i);
                else if (item_fused[i] == 0) w_rows.push_back(i);
            }
            append_by_weight(tmp_scan, w_scan);
            append_by_weight(tmp_fin, w_fin);
            append_by_weight(tmp_fin2, w_fin2);
            scan_off[s + 1] = (uint32_t)w_scan.size();
            extra_off[s + 1] = (uint32_t)w_extra.size();
            fin_off[s + 1] = (uint32_t)w_fin.size();
            fin2_off[s + 1] = (uint32_t)w_fin2.size();
            fin3_off[s + 1] = (uint32_t)w_fin3.size();
            fin5_off[s + 1] = (uint32_t)w_fin5.size();
            rows_off[s + 1] = (uint32_t)w_rows.size();
        }
        // the four lists of the binned clusters travel as one block (cluster | first item | partitions | first entry)
        const size_t NB = bin_cluster.size();
        if (NB) {
            std::vector<uint32_t>& all = c->hs_bin[0];
            all.insert(all.end(), bin_item0.begin(), bin_item0.end());
            all.insert(all.end(), bin_nparts.begin(), bin_nparts.end());
            all.insert(all.end(), bin_base.begin(), bin_base.end());
            PFCHK(c->q_off.ensure(NI * (pf::BIN_CHUNKS + 1) * 4));
        }
        {
            std::vector<std::pair<DevBuf*, const std::vector<uint32_t>*>> arrs = {
                {&c->bin_lists, &c->hs_bin[0]},
                {&c->it_cluster, &v_cluster}, {&c->it_part, &v_part}, {&c->it_nparts, &v_nparts},
                {&c->it_nslots, &v_nslots}, {&c->it_slice, &v_slice}, {&c->it_sib0, &v_sib0}, {&c->it_nsib, &v_nsib},
                {&c->it_extra_first, &v_exfirst}, {&c->it_is_extra, &v_isex}, {&c->it_compact, &v_compact},
                {&c->it_binned, &v_binned},
                {&c->sub_cluster, &sub_cluster},
                {&c->sub_item0, &sub_item0}, {&c->sub_nitems, &sub_nitems}, {&c->work_scan, &w_scan},
                {&c->work_extra, &w_extra}, {&c->work_fin, &w_fin}, {&c->work_fin2, &w_fin2}, {&c->work_fin3, &w_fin3}, {&c->work_fin5, &w_fin5}, {&c->work_rows, &w_rows}};
            lap("  work lists");
            PFCHK(staged_upload(c, arrs));
        }
        // the cursor's next free index restarts at this arena's base
        {
            c->pin_small[arena_i & 15] = arena_base;
            HIPCHK(hipMemcpyAsync(c->cursor.p, &c->pin_small[arena_i & 15], 8, hipMemcpyHostToDevice, c->stream));
        }

        lap("upload items");
        for (size_t s = 0; s < subs.size(); s++) {
            const Sub& sb = subs[s];
            const uint32_t n_scan = scan_off[s + 1] - scan_off[s], n_extra_items = extra_off[s + 1] - extra_off[s];
            if (n_extra_items) {
                pf::ExtraParams ep{};
                ep.extra_ord = c->extra_dense.as<uint32_t>(); ep.extra_bits = d.extra_bits;
                ep.item_first = c->it_extra_first.as<uint32_t>(); ep.item_nslots = c->it_nslots.as<uint32_t>();
                ep.item_scratch = c->it_slice.as<uint32_t>();
                ep.tab_key = c->tab_key.as<uint64_t>(); ep.tab_ord = c->tab_ord.as<uint32_t>();
                ep.chunkbits = c->chunkbits.as<uint32_t>(); ep.chunkmask = c->chunkmask.as<uint32_t>();
                ep.item_count = c->it_count.as<uint32_t>();
                ep.work = c->work_extra.as<uint32_t>() + extra_off[s];
                ep.W = W; ep.NS = NS; ep.KW = KW;
                PFCHK(mark_begin(c, 2));
                hipLaunchKernelGGL(pf::extra_fill_kernel, dim3(n_extra_items), dim3(256), 0, c->stream, ep);
                HIPCHK(hipGetLastError());
                PFCHK(mark_end(c));
            }
            if (n_scan) {
                pf::ScanParams sp = scan_params(host_items(), c->upool[sb.pool], c->work_scan.as<uint32_t>() + scan_off[s]);
                PFCHK(mark_begin(c, 0));
                if (sb.nbin) {
                    const size_t qcap = (size_t)sb.q_total + 64;
                    PFCHK(c->q_key.ensure(qcap * 8 * KW)); PFCHK(c->q_ord.ensure(qcap * 4)); PFCHK(c->q_bit.ensure(qcap * 4));
                    sp.q_key = c->q_key.as<uint64_t>(); sp.q_ord = c->q_ord.as<uint32_t>(); sp.q_bit = c->q_bit.as<uint32_t>();
                    sp.q_stride = qcap; sp.q_off = c->q_off.as<uint32_t>();
                    sp.bin_cluster = c->bin_lists.as<uint32_t>() + sb.bin0; sp.bin_item0 = c->bin_lists.as<uint32_t>() + NB + sb.bin0;
                    sp.bin_nparts = c->bin_lists.as<uint32_t>() + 2 * NB + sb.bin0; sp.bin_base = c->bin_lists.as<uint32_t>() + 3 * NB + sb.bin0;
                    PFCHK(launch_bin(c, sp, sb.nbin));
                }
                PFCHK(launch_scan(c, sp, n_scan));
                PFCHK(mark_end(c));
                c->timing.scan_launches++;
            }
            const uint32_t n_fin = fin_off[s + 1] - fin_off[s], n_fin2 = fin2_off[s + 1] - fin2_off[s],
                           n_fin3 = fin3_off[s + 1] - fin3_off[s], n_fin5 = fin5_off[s + 1] - fin5_off[s],
                           n_rows = rows_off[s + 1] - rows_off[s];
            // A launch whose fused-finish workgroups do not fill the GPU while general-path items wait behind them (a batch
            // of many-allele clusters with a few dozen simple ones: two 1 024-thread workgroups took 0.4 ms each with the
            // other 250 CUs idle; any batch of a few hundred clusters): the finish kernels go to the context's second
            // stream and run BESIDE rows / emit / pattern rows -- they share nothing but atomically claimed output room
            // and the run-global pattern table.  (Not for full launches: two latency-bound kernels that each fill the GPU
            // take each other's wave slots -- five such pairings lost in rounds 2-3.)
            const uint32_t n_fused_wg = n_fin + n_fin2 + n_fin3 + n_fin5;
            const bool beside = n_fused_wg && n_rows && n_fused_wg <= 2u * (uint32_t)c->n_cu;
            if (n_fused_wg) {
                pf::FinishParams fp = finish_params(host_items(), ar);
                hipStream_t fs = c->stream;
                if (beside) {
                    HIPCHK(hipEventRecord(c->ev_fork, c->stream));
                    HIPCHK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
                    fs = c->side;
                    c->timing.n_side_launches++;
                }
                PFCHK(mark_begin(c, 6, fs));      // (on the side stream too: finish_ms must not leave these launches out)
                if (n_fin5) {   // the heaviest clusters first
                    fp.work = c->work_fin5.as<uint32_t>() + fin5_off[s];
                    hipLaunchKernelGGL((pf::finish_kernel<pf::FinHuge, true>), dim3(n_fin5), dim3(pf::FinHuge::THREADS), 0, fs, fp);
                    HIPCHK(hipGetLastError());
                }
                if (n_fin3) {
                    fp.work = c->work_fin3.as<uint32_t>() + fin3_off[s];
                    hipLaunchKernelGGL((pf::finish_kernel<pf::FinLargeM, true>), dim3(n_fin3), dim3(pf::FinLargeM::THREADS), 0, fs, fp);
                    HIPCHK(hipGetLastError());
                }
                if (n_fin2) {
                    fp.work = c->work_fin2.as<uint32_t>() + fin2_off[s];
                    hipLaunchKernelGGL((pf::finish_kernel<pf::FinLarge, false>), dim3(n_fin2), dim3(pf::FinLarge::THREADS), 0, fs, fp);
                    HIPCHK(hipGetLastError());
                }
                if (n_fin) {
                    fp.work = c->work_fin.as<uint32_t>() + fin_off[s];
                    hipLaunchKernelGGL((pf::finish_kernel<pf::FinSmall, false>), dim3(n_fin), dim3(pf::FinSmall::THREADS), 0, fs, fp);
                    HIPCHK(hipGetLastError());
                }
                PFCHK(mark_end(c, fs));
                if (beside) HIPCHK(hipEventRecord(c->ev_join, c->side));
            }
            if (!n_rows) continue;
            pf::RowsParams rp{};
            rp.item_cluster = c->it_cluster.as<uint32_t>(); rp.item_nslots = c->it_nslots.as<uint32_t>();
            rp.item_scratch = c->it_slice.as<uint32_t>(); rp.item_count = c->it_count.as<uint32_t>();
            rp.item_is_extra = c->it_is_extra.as<uint32_t>();
            rp.cluster_overflow = c->cl_overflow.as<uint32_t>();
            rp.cluster_seg_off = d.cluster_seg_off; rp.seg_sample = d.seg_sample;
            rp.seg_distinct = c->seg_distinct.as<uint32_t>();
            rp.v_mode = c->v_mode.as<uint32_t>(); rp.v_nstr = c->v_nstr.as<uint32_t>(); rp.v_dense = c->v_dense.as<uint32_t>();
            rp.cluster_nstrains = d.cluster_nstrains; rp.cluster_npresab = d.cluster_npresab;
            rp.cluster_presab = d.cluster_presab; rp.cluster_ordinal = d.cluster_ordinal;
            rp.maf_lo = c->d_maf_lo.as<uint32_t>(); rp.maf_hi = c->d_maf_hi.as<uint32_t>();
            rp.tab_ord = c->tab_ord.as<uint32_t>(); rp.chunkbits = c->chunkbits.as<uint32_t>();
            rp.chunkmask = c->chunkmask.as<uint32_t>();
            rp.slot_hash = c->slot_hash.as<uint4>(); rp.sorted_pair = c->sorted_pair.as<uint64_t>();
            rp.kept_prefix = c->kept_prefix.as<uint32_t>();
            rp.bm4 = c->bm4.as<uint4>(); rp.bm2 = c->bm2.as<uint2>(); rp.item_nsib = c->it_nsib.as<uint32_t>();
            rp.mrows = c->mrows.as<uint32_t>();
            rp.item_unique = c->it_unique.as<uint32_t>(); rp.item_kept = c->it_kept.as<uint32_t>();
            rp.work = c->work_rows.as<uint32_t>() + rows_off[s]; rp.W = W; rp.NS = NS;
            rp.consider_missing = c->o.consider_missing; rp.patfilt = c->o.patfilt; rp.multiple_files = c->o.multiple_files;
            PFCHK(mark_begin(c, 1));
            hipLaunchKernelGGL(pf::rows_kernel, dim3(n_rows), dim3(pf::ROWS_THREADS), 0, c->stream, rp);
            HIPCHK(hipGetLastError());
            PFCHK(mark_end(c));

            pf::BaseParams bp{};
            bp.sub_cluster = c->sub_cluster.as<uint32_t>() + sb.cl0;
            bp.cluster_item0 = c->sub_item0.as<uint32_t>() + sb.cl0;
            bp.cluster_nitems = c->sub_nitems.as<uint32_t>() + sb.cl0;
            bp.item_kept = c->it_kept.as<uint32_t>(); bp.item_unique = c->it_unique.as<uint32_t>();
            bp.cluster_overflow = c->cl_overflow.as<uint32_t>();
            bp.cluster_kmer_off = c->cl_kmer_off.as<uint64_t>(); bp.cluster_kmer_cnt = c->cl_kmer_cnt.as<uint32_t>();
            bp.cluster_unique = c->cl_unique.as<uint32_t>(); bp.cursor = c->cursor.as<uint64_t>();
            bp.n = sb.ncl;
            PFCHK(mark_begin(c, 2));
            hipLaunchKernelGGL(pf::cluster_base_kernel, dim3(1), dim3(1024), 0, c->stream, bp);
            HIPCHK(hipGetLastError());
            if (sb.nitems > sb.ncl) {     // some cluster of this sub-batch has several items
                pf::BitmapMergeParams bm{};
                bm.sub_cluster = bp.sub_cluster; bm.cluster_item0 = bp.cluster_item0; bm.cluster_nitems = bp.cluster_nitems;
                bm.item_scratch = c->it_slice.as<uint32_t>(); bm.cluster_overflow = bp.cluster_overflow;
                bm.v_mode = c->v_mode.as<uint32_t>(); bm.v_dense = c->v_dense.as<uint32_t>();
                bm.item_fused = c->it_compact.as<uint32_t>();
                bm.bm4 = c->bm4.as<uint4>(); bm.bm2 = c->bm2.as<uint2>();
                hipLaunchKernelGGL(pf::bitmap_merge_kernel, dim3(sb.ncl), dim3(256), 0, c->stream, bm);
                HIPCHK(hipGetLastError());
            }

            pf::EmitParams em{};
            em.item_cluster = c->it_cluster.as<uint32_t>(); em.item_scratch = c->it_slice.as<uint32_t>();
            em.item_unique = c->it_unique.as<uint32_t>(); em.item_nslots = c->it_nslots.as<uint32_t>();
            em.item_sib0 = c->it_sib0.as<uint32_t>();
            em.item_nsib = c->it_nsib.as<uint32_t>(); em.cluster_overflow = c->cl_overflow.as<uint32_t>();
            em.v_mode = c->v_mode.as<uint32_t>(); em.v_dense = c->v_dense.as<uint32_t>();
            em.cluster_nstrains = d.cluster_nstrains; em.cluster_npresab = d.cluster_npresab;
            em.cluster_presab = d.cluster_presab; em.cluster_ordinal = d.cluster_ordinal;
            em.cluster_kmer_off = c->cl_kmer_off.as<uint64_t>();
            em.tab_key = c->tab_key.as<uint64_t>(); em.tab_ord = c->tab_ord.as<uint32_t>();
            em.slot_hash = c->slot_hash.as<uint4>();
            em.sorted_pair = c->sorted_pair.as<uint64_t>(); em.kept_prefix = c->kept_prefix.as<uint32_t>(); em.kept_prefix_rw = c->kept_prefix.as<uint32_t>();
            em.bm4 = c->bm4.as<uint4>();
            em.slot_out = c->slot_out.as<uint32_t>();
            em.out_key = ar->key.as<uint64_t>(); em.out_pid = ar->pid.as<uint32_t>(); em.out_first = ar->first.as<uint64_t>();
            em.cluster_pattern = c->cl_pattern.as<uint32_t>(); em.cluster_first = c->cl_first.as<uint64_t>();
            em.pt = c->pt; em.out_base = ar->base; em.out_cap = ar->cap;
            em.work = c->work_rows.as<uint32_t>() + rows_off[s]; em.W = W; em.NS = NS; em.KW = KW;
            em.consider_missing = c->o.consider_missing; em.multiple_files = c->o.multiple_files;
            hipLaunchKernelGGL(pf::emit_kernel, dim3(n_rows), dim3(pf::EMIT_THREADS), 0, c->stream, em);
            HIPCHK(hipGetLastError());
            PFCHK(mark_end(c));
            PFCHK(mark_begin(c, 4));

            pf::PatRowsParams pr{};
            pr.item_cluster = em.item_cluster; pr.item_scratch = em.item_scratch; pr.item_unique = em.item_unique;
            pr.item_nslots = em.item_nslots; pr.item_is_extra = c->it_is_extra.as<uint32_t>();
            pr.item_sib0 = em.item_sib0; pr.item_nsib = em.item_nsib; pr.cluster_overflow = em.cluster_overflow;
            pr.v_mode = em.v_mode; pr.v_nstr = c->v_nstr.as<uint32_t>(); pr.v_dense = em.v_dense;
            pr.cluster_seg_off = d.cluster_seg_off; pr.seg_sample = d.seg_sample; pr.seg_distinct = c->seg_distinct.as<uint32_t>();
            pr.cluster_nstrains = d.cluster_nstrains; pr.cluster_npresab = d.cluster_npresab;
            pr.cluster_presab = d.cluster_presab; pr.cluster_kmer_off = em.cluster_kmer_off;
            pr.sorted_pair = em.sorted_pair; pr.kept_prefix = em.kept_prefix;
            pr.chunkbits = c->chunkbits.as<uint32_t>(); pr.chunkmask = c->chunkmask.as<uint32_t>();
            pr.slot_out = c->slot_out.as<uint32_t>(); pr.mrows = c->mrows.as<uint32_t>();
            pr.out_pid = em.out_pid; pr.out_first = em.out_first;
            pr.cluster_pattern = em.cluster_pattern; pr.cluster_first = em.cluster_first;
            pr.pat_first_seen = c->pt.first_seen;
            pr.pat_bits = c->pat_bits.as<uint32_t>();
            pr.pat_nan = c->o.consider_missing ? c->pat_nan.as<uint32_t>() : nullptr;
            pr.pat_n = c->pat_n.as<uint32_t>();
            pr.out_base = ar->base; pr.out_cap = ar->cap; pr.pool = c->pt.pool;
            pr.work = em.work; pr.W = W; pr.NS = NS; pr.consider_missing = c->o.consider_missing;
            hipLaunchKernelGGL(pf::pattern_rows_kernel, dim3(n_rows), dim3(pf::PR_THREADS), 0, c->stream, pr);
            HIPCHK(hipGetLastError());
            PFCHK(mark_end(c));
            if (beside) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));   // the next sub-batch takes the scratch slices over
        }
        c->timing.n_items += (uint32_t)NI;
        for (uint32_t ci : todo) c->timing.scan_packed_bytes += rec[ci].words * 8 * nparts[ci];

        lap("launch pass");
        if (pass + 1 < P) {
            // no wait: the next part's pass is built now and goes in behind this one
            const uint32_t pin = 16 + (arena_i & 31);
            HIPCHK(hipMemcpyAsync(&c->pin_small[pin], c->cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
            deferred.push_back(Deferred{ar, pin});
            arena_base += ar->cap;
            arena_i++;
            pass++;
            continue;
        }
        // ---- who overflowed?
        if ((size_t)C * 4 + 64 > c->pin_ovf_cap) {
            if (c->pin_ovf) (void)hipHostFree(c->pin_ovf);
            c->pin_ovf = nullptr; c->pin_ovf_cap = 0;
            const size_t want = ((size_t)C * 4 + 64) * 5 / 4;
            hipError_t e = hipHostMalloc((void**)&c->pin_ovf, want, hipHostMallocDefault);
            if (e != hipSuccess) return fail(PF_ERR_OOM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            c->pin_ovf_cap = want;
        }
        uint32_t* const ovf = c->pin_ovf;
        uint64_t* const cur3 = &c->pin_small[48];
        uint32_t* const cnt_pin = reinterpret_cast<uint32_t*>(&c->pin_small[52]);
        // clusters of this pass the key-partition estimate can learn from: their items' key counts come along
        bool learn = false;
        // (8 192 clusters settle the line; after that every 16th submit still looks, at half the old weight, so that a
        // pangenome whose later clusters differ from its first is followed -- the read-back is not free)
        // (not in the re-run of a batch after the pattern table grew: its clusters have been counted)
        uint32_t learn_planned = 0;            // ... and plan_kernel's items of the last part (their clusters, their key counts)
        if (!rerun && (c->reg_n < 8192 || (c->n_submits & 15) == 0)) {
            for (uint32_t ci : todo) if (rec[ci].mode && rec[ci].vnstr >= 2) { learn = true; break; }
            if (use_plan && pass + 1 == P) learn_planned = std::min<uint32_t>(c->dplan[pass].pin_out->n_items, 4096u);
        }
        if (learn) {
            c->hs_count.resize(NI);
            HIPCHK(hipMemcpyAsync(c->hs_count.data(), c->it_count.p, NI * 4, hipMemcpyDeviceToHost, c->stream));
        }
        if (learn_planned) {
            c->hs_plan.resize(2 * (size_t)learn_planned);
            HIPCHK(hipMemcpyAsync(c->hs_plan.data(), dplan_ptrs(c->dplan[pass]).it_cluster, (size_t)learn_planned * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(c->hs_plan.data() + learn_planned, c->dplan[pass].it_count.p, (size_t)learn_planned * 4, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(hipMemcpyAsync(ovf, c->cl_overflow.p, (size_t)C * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(cur3, c->cursor.p, 24, hipMemcpyDeviceToHost, c->stream));
        // (the pattern counters come along: when this was the last pass the MD5 launch needs no round trip of its own)
        HIPCHK(hipMemcpyAsync(cnt_pin, c->pt_counters.p, 12, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        cnt2[0] = cnt_pin[0]; cnt2[1] = cnt_pin[1]; cnt2[2] = cnt_pin[2];
        c->counters.n_unique = cur3[1]; c->counters.n_kept = cur3[2];
        lap("sync pass");
        ar->used = cur3[0] - ar->base;
        if (ar->used > ar->cap) return fail(PF_ERR_CAPACITY, "output arena overflow (%llu > %llu)",
                                            (unsigned long long)ar->used, (unsigned long long)ar->cap);
        arena_base += ar->cap;
        arena_i++;
        if (!deferred.empty()) {                   // the earlier passes finished before this one
            for (const Deferred& df : deferred) {
                df.ar->used = c->pin_small[df.pin] - df.ar->base;
                if (df.ar->used > df.ar->cap)
                    return fail(PF_ERR_CAPACITY, "output arena overflow (%llu > %llu)", (unsigned long long)df.ar->used,
                                (unsigned long long)df.ar->cap);
            }
            deferred.clear();
            todo.resize(C);                        // every cluster has been through its first pass now
            std::iota(todo.begin(), todo.end(), 0u);
        }
        if (learn || learn_planned) {
            if (c->reg_n >= 8192) {
                c->reg_n *= 0.5; c->reg_x *= 0.5; c->reg_y *= 0.5; c->reg_xx *= 0.5; c->reg_xy *= 0.5; c->reg_yy *= 0.5;
            }
            uint32_t looked = 0;               // (the GPU waits while this runs: a few thousand clusters say enough)
            for (uint32_t i = 0; i < learn_planned && looked < 4096; i++) {
                const uint32_t ci = c->hs_plan[i];
                if (ci >= C || ovf[ci] || !rec[ci].mode || rec[ci].vnstr < 2 || !rec[ci].vinst) continue;
                looked++;
                const double D = (double)rec[ci].vnstr, L = (double)(rec[ci].vinst * mult) / D;
                const double g = std::max(0.0, ((double)c->hs_plan[learn_planned + i] - L) / (D - 1.0));
                c->reg_n += 1; c->reg_x += L; c->reg_y += g; c->reg_xx += L * L; c->reg_xy += L * g; c->reg_yy += g * g;
            }
            if (learn)
            for (size_t i = 0; i < NI && looked < 4096; i++) {
                const Item& it = items[i];
                if (it.is_extra || it.part != 0) continue;
                const uint32_t ci = it.cluster;
                if (ovf[ci] || !rec[ci].mode || rec[ci].vnstr < 2 || !rec[ci].vinst) continue;
                looked++;
                uint64_t keys = 0;
                for (uint32_t q = 0; q < it.nparts; q++) keys += c->hs_count[i + q];
                const double D = (double)rec[ci].vnstr, L = (double)(rec[ci].vinst * mult) / D;
                const double g = std::max(0.0, ((double)keys - L) / (D - 1.0));
                c->reg_n += 1; c->reg_x += L; c->reg_y += g; c->reg_xx += L * L; c->reg_xy += L * g; c->reg_yy += g * g;
            }
        }
        std::vector<uint32_t> next;
        for (uint32_t ci : todo)
            if (ovf[ci]) {
                next.push_back(ci);
                // ovf = 64 * (the item's units / the units scanned when its table was full): that many times the keys
                // of one partition are to be expected (an overestimate: a cluster's first sequence brings more new keys
                // than its later ones), and key hashing spreads them evenly: a small margin is enough
                // (a view of distinct sequences only: among the copies of an every-copy cluster new keys stop coming
                // early and nothing can be extrapolated -- those double, as does a cluster that asks for more than 16x)
                double want = std::ceil((double)nparts[ci] * ((double)ovf[ci] / 64.0) * 1.06);
                if (!rec[ci].mode || want > 16.0 * nparts[ci]) want = 2.0 * nparts[ci];
                nparts[ci] = (uint32_t)std::min<double>(std::max<double>(want, (double)nparts[ci] + 1.0), 65537.0);
                if (nparts[ci] > 65536) return fail(PF_ERR_CAPACITY, "cluster %u does not fit 65536 key partitions", ci);
            }
        if (!next.empty()) {
            c->timing.n_retried += (uint32_t)next.size();
            HIPCHK(hipMemsetAsync(c->cl_overflow.p, 0, C1 * 4, c->stream));
        }
        todo.swap(next);
        pass++;
        if (todo.empty()) break;
    }
    for (size_t a = arena_i; a < c->arenas.size(); a++) c->arenas[a]->used = 0;   // arenas of an earlier, longer batch
    c->n_passes = arena_i;

    // ---- MD5 of the patterns this batch created (cnt2: read with the last pass's results)
    if (!C) {
        HIPCHK(hipMemcpyAsync(cnt2, c->pt_counters.p, 12, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (cnt2[2]) return fail(PF_ERR_CAPACITY, "output arena overflow inside a kernel");
    if (cnt2[1] || cnt2[0] > c->pt.pool) { *need = cnt2[0]; return PF_RETRY_PATTERNS; }
    const uint32_t pid1 = cnt2[0];
    if (pid1 > c->pid0) {
        pf::Md5Params mp{};
        mp.pat_bits = c->pat_bits.as<uint32_t>();
        mp.pat_nan = c->o.consider_missing ? c->pat_nan.as<uint32_t>() : nullptr;
        mp.pat_n = c->pat_n.as<uint32_t>(); mp.pat_md5 = c->pat_md5.as<uint8_t>();
        mp.pid0 = c->pid0; mp.pid1 = pid1; mp.W = W; mp.range = nullptr;
        // int64 rows are the clusters' own rows: the int pass goes by cluster (cl_pattern) and runs BESIDE the float pass, on
        // the side stream, instead of behind it
        mp.cluster_pattern = c->cl_pattern.as<uint32_t>(); mp.n_clusters = C;
        PFCHK(mark_begin(c, 5));
        // A small launch is spread over the chip: the float pass's workgroups (four waves, one per SIMD) number a few per
        // CU, and the dispatcher fills CUs with up to eight before it moves on -- a SIMD gets through its rows at one rate
        // however many waves share it, so the pass took as long as the FULLEST SIMD (a rank's share of configs[3]: 5 waves
        // per SIMD on average, 8 on two thirds of the CUs, none on the rest).  Dynamic LDS the kernel never touches caps the
        // workgroups per CU at what an even spread needs.
        const uint32_t n_float = (pid1 - c->pid0 + pf::MD5_THREADS - 1) / pf::MD5_THREADS;
        const uint32_t per_cu = (n_float + (uint32_t)c->n_cu - 1) / (uint32_t)c->n_cu;
        uint32_t lds_cap = 0;
        if (per_cu < 8) lds_cap = std::min<uint32_t>(64u << 10, ((160u << 10) / std::max(per_cu, 1u)) & ~1023u);
        const dim3 g_float(n_float);
        const dim3 g_int(std::min<uint32_t>((C + pf::MD5_THREADS - 1) / pf::MD5_THREADS, 1024u));
        HIPCHK(hipEventRecord(c->ev_fork, c->stream));
        HIPCHK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
        if (mp.pat_nan) {
            hipLaunchKernelGGL((pf::md5_kernel<false, true>), g_int, dim3(pf::MD5_THREADS), 0, c->side, mp);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL((pf::md5_kernel<true, true>), g_float, dim3(pf::MD5_THREADS), lds_cap, c->stream, mp);
        } else {
            hipLaunchKernelGGL((pf::md5_kernel<false, false>), g_int, dim3(pf::MD5_THREADS), 0, c->side, mp);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL((pf::md5_kernel<true, false>), g_float, dim3(pf::MD5_THREADS), lds_cap, c->stream, mp);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c->ev_join, c->side));
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
        PFCHK(mark_end(c));
    }
    c->n_patterns = pid1;
    const uint64_t n_unique_total = c->counters.n_unique, n_kept_total = c->counters.n_kept;   // the last pass's cursor
    HIPCHK(hipEventRecord(c->ev_t1, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));

    lap("md5 + final sync");
    // ---- timing
    HIPCHK(hipEventElapsedTime(&c->timing.total_ms, c->ev_t0, c->ev_t1));
    for (auto& e : c->events) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
        if (e.cat == 0) c->timing.scan_ms += ms;
        else if (e.cat == 1) c->timing.rows_ms += ms;
        else if (e.cat == 3) c->timing.dedup_ms += ms;
        else if (e.cat == 4) c->timing.patrows_ms += ms;
        else if (e.cat == 5) c->timing.md5_ms += ms;
        else if (e.cat == 6) c->timing.finish_ms += ms;
        else c->timing.emit_ms += ms;
    }

    c->n_clusters = C;
    c->last = d; c->last_nseg = NSEG;
    if (!c->n_strand_words) c->last.seg_strand_off = nullptr;
    c->kt_bytes = 0; c->kt_pref_valid = false;
    c->counters = pf_result{};
    c->counters.n_instances = total_inst;
    c->counters.n_unique = n_unique_total;
    c->counters.n_kept = n_kept_total;
    c->counters.n_new_patterns = pid1 - c->pid0;
    c->counters.n_patterns = pid1;
    c->counters.W = W;
    c->counters.key_words = KW;
    c->have_batch = true;
    c->h_strand_fresh = false;
    if (counters) *counters = c->counters;
    return PF_OK;
}
}  // namespace

int pf_submit(pf_ctx* c, const pf_batch* b, pf_result* counters) {
    if (!c || !b) return fail(PF_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    const pf_gather* gth = c->pending_gather;
    c->pending_gather = nullptr;
    if (c->o.multiple_files) {
        // the pattern set starts empty in every cluster (panfeed.py:165) and ids are salted by the cluster ordinal:
        // nothing of an earlier batch can ever be matched again
        PFCHK(reset_patterns(c));
    } else if (c->pt_stale) {
        return fail(PF_ERR_STATE, "the pattern table holds entries of a batch that failed while it was being enlarged; "
                                  "pf_reset_patterns (a new run) first");
    } else if ((uint64_t)c->n_patterns * 2 > c->pt.pool && c->pregrow_failed_pool != c->pt.pool) {
        // ahead of need: a re-run costs a whole batch.  No batch has failed here, so a growth that does not succeed
        // (out of memory with both tables resident) leaves a table that is whole: carry on with it -- and do not try
        // again at this pool size (every try is an allocation, a fill and a free of the larger table): the next
        // growth is the one a batch that really runs out of ids asks for.
        const int rc = grow_patterns(c, (uint64_t)c->n_patterns * 2);
        if (rc == PF_ERR_OOM) { c->pregrow_failed_pool = c->pt.pool; g_err.clear(); }
        else if (rc != PF_OK) return rc;
    }
    const uint64_t submits0 = c->n_submits;
    for (int attempt = 0;; attempt++) {
        uint64_t need = 0;
        const int rc = submit_once(c, b, gth, counters, &need, attempt > 0);
        if (rc != PF_RETRY_PATTERNS) return rc;
        int rg = attempt >= 8 ? fail(PF_ERR_CAPACITY, "pattern table still too small after %d enlargements", attempt)
                              : grow_patterns(c, std::max<uint64_t>(need, (uint64_t)c->pt.pool + 1));
        if (rg != PF_OK) {
            // the failed batch's patterns are still in the table; the batch itself does not count as submitted
            c->pt_stale = true;
            c->n_submits = submits0;
            c->timing = pf_timing{};
            return rg;
        }
    }
}

int pf_debug_limit_pattern_slots(pf_ctx* c, uint64_t max_slots) {
    if (!c) return fail(PF_ERR_ARG, "null context");
    c->pt_slot_limit = max_slots;
    c->pregrow_failed_pool = 0;
    return PF_OK;
}

int pf_debug_limit_alloc(uint64_t max_bytes, uint64_t stats[2]) {
    if (stats) {
        stats[0] = g_alloc_max_request.exchange(0);
        stats[1] = g_alloc_exact_retries.exchange(0);
    }
    g_alloc_limit.store(max_bytes);
    return PF_OK;
}

int pf_get_timing(pf_ctx* c, pf_timing* t) {
    if (!c || !t) return fail(PF_ERR_ARG, "null argument");
    *t = c->timing;
    t->n_scratch_grown = c->n_scratch_grown;
    return PF_OK;
}

namespace {
// used_strand bits of the last submit's target windows (strand_bits_kernel) to the host, once per batch
int fetch_strand_bits(pf_ctx* c) {
    if (c->h_strand_fresh) return PF_OK;
    c->h_strand.resize((size_t)c->n_strand_words);
    if (c->n_strand_words)
        HIPCHK(hipMemcpy(c->h_strand.data(), c->strand_bits.p, (size_t)c->n_strand_words * 8, hipMemcpyDeviceToHost));
    c->h_strand_fresh = true;
    return PF_OK;
}
}  // namespace

int pf_fetch(pf_ctx* c, pf_result* res) {
    if (!c || !res) return fail(PF_ERR_ARG, "null argument");
    if (!c->have_batch) return fail(PF_ERR_STATE, "pf_fetch without a successful pf_submit");
    HIPCHK(hipSetDevice(c->device));
    const uint32_t C = c->n_clusters, W = c->W, KW = (uint32_t)c->KW;
    c->h_kmer_off.resize(C); c->h_kmer_cnt.resize(C); c->h_cl_pattern.resize(C); c->h_cl_unique.resize(C);
    if (C) {
        HIPCHK(hipMemcpy(c->h_kmer_off.data(), c->cl_kmer_off.p, (size_t)C * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(c->h_kmer_cnt.data(), c->cl_kmer_cnt.p, (size_t)C * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(c->h_cl_pattern.data(), c->cl_pattern.p, (size_t)C * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(c->h_cl_unique.data(), c->cl_unique.p, (size_t)C * 4, hipMemcpyDeviceToHost));
    }
    // concatenate the used prefixes of the arenas; remap cluster offsets
    uint64_t total = 0;
    std::vector<uint64_t> host_base(c->arenas.size(), 0);
    for (size_t a = 0; a < c->arenas.size(); a++) { host_base[a] = total; total += c->arenas[a]->used; }
    c->h_kmer_key.resize((size_t)total * KW);
    c->h_kmer_pid.resize((size_t)total);
    for (size_t a = 0; a < c->arenas.size(); a++) {
        Arena* ar = c->arenas[a];
        if (!ar->used) continue;
        HIPCHK(hipMemcpy(c->h_kmer_key.data() + host_base[a] * KW, ar->key.p, (size_t)ar->used * 8 * KW, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(c->h_kmer_pid.data() + host_base[a], ar->pid.p, (size_t)ar->used * 4, hipMemcpyDeviceToHost));
    }
    for (uint32_t i = 0; i < C; i++) {
        const uint32_t a = c->cluster_arena[i];
        if (c->h_kmer_cnt[i]) c->h_kmer_off[i] = c->h_kmer_off[i] - c->arenas[a]->base + host_base[a];
        else c->h_kmer_off[i] = 0;
    }
    // pattern pool: extend the host mirror by the patterns of this batch
    const uint32_t p0 = c->pid0, p1 = c->n_patterns;
    c->h_pat_bits.resize((size_t)p1 * W); c->h_pat_n.resize(p1); c->h_pat_md5.resize((size_t)p1 * 16);
    c->h_first_seen.resize(p1);
    if (c->o.consider_missing) c->h_pat_nan.resize((size_t)p1 * W);
    if (p1 > p0) {
        const size_t n = p1 - p0;
        HIPCHK(hipMemcpy(c->h_pat_bits.data() + (size_t)p0 * W, c->pat_bits.as<uint32_t>() + (size_t)p0 * W, n * W * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(c->h_pat_n.data() + p0, c->pat_n.as<uint32_t>() + p0, n * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(c->h_pat_md5.data() + (size_t)p0 * 16, c->pat_md5.as<uint8_t>() + (size_t)p0 * 16, n * 16, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(c->h_first_seen.data() + p0, c->pt.first_seen + p0, n * 8, hipMemcpyDeviceToHost));
        if (c->o.consider_missing)
            HIPCHK(hipMemcpy(c->h_pat_nan.data() + (size_t)p0 * W, c->pat_nan.as<uint32_t>() + (size_t)p0 * W, n * W * 4, hipMemcpyDeviceToHost));
    }
    c->h_new_pid.resize(p1 - p0);
    std::iota(c->h_new_pid.begin(), c->h_new_pid.end(), p0);
    std::sort(c->h_new_pid.begin(), c->h_new_pid.end(),
              [&](uint32_t x, uint32_t y) { return c->h_first_seen[x] < c->h_first_seen[y]; });
    PFCHK(fetch_strand_bits(c));

    *res = c->counters;
    res->cluster_kmer_off = c->h_kmer_off.data();
    res->cluster_kmer_cnt = c->h_kmer_cnt.data();
    res->cluster_pattern = c->h_cl_pattern.data();
    res->cluster_unique = c->h_cl_unique.data();
    res->kmer_key = c->h_kmer_key.data();
    res->kmer_pattern = c->h_kmer_pid.data();
    res->new_pattern_id = c->h_new_pid.data();
    res->n_patterns = p1;
    res->pattern_md5 = c->h_pat_md5.data();
    res->pattern_bits = c->h_pat_bits.data();
    res->pattern_nan = c->o.consider_missing ? c->h_pat_nan.data() : nullptr;
    res->pattern_n = c->h_pat_n.data();
    res->pattern_first_seen = c->h_first_seen.data();
    res->strand_bits = c->n_strand_words ? c->h_strand.data() : nullptr;
    return PF_OK;
}

int pf_export_patterns(pf_ctx* c, uint64_t* n, const uint8_t** md5, const uint64_t** first_seen) {
    if (!c || !n || !md5 || !first_seen) return fail(PF_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    const uint32_t p1 = c->n_patterns;
    c->h_pat_md5.resize((size_t)p1 * 16);
    c->h_first_seen.resize(p1);
    if (p1) {
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemcpy(c->h_pat_md5.data(), c->pat_md5.p, (size_t)p1 * 16, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(c->h_first_seen.data(), c->pt.first_seen, (size_t)p1 * 8, hipMemcpyDeviceToHost));
    }
    *n = p1;
    *md5 = c->h_pat_md5.data();
    *first_seen = c->h_first_seen.data();
    return PF_OK;
}

int pf_export_patterns_dev(pf_ctx* c, uint64_t cap, void* d_md5, void* d_first_seen, uint64_t* n) {
    if (!c || !n) return fail(PF_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    const uint64_t m = std::min<uint64_t>(cap, c->n_patterns);
    if (m) {
        if (!d_md5 || !d_first_seen) return fail(PF_ERR_ARG, "null destination");
        HIPCHK(hipMemcpyAsync(d_md5, c->pat_md5.p, (size_t)m * 16, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(d_first_seen, c->pt.first_seen, (size_t)m * 8, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    *n = m;
    return PF_OK;
}

int pf_render_kmers_to_hashes(pf_ctx* c, const char* const* names, const char* const* extra_keys, char** out,
                              uint64_t* nbytes, uint64_t* cluster_end) {
    if (!c || !names || !out || !nbytes) return fail(PF_ERR_ARG, "null argument");
    if (!c->have_batch || c->h_kmer_off.size() != c->n_clusters) return fail(PF_ERR_STATE, "pf_render_* needs pf_fetch first");
    ensure_b64(c);
    const uint32_t C = c->n_clusters, k = c->o.klength, KW = (uint32_t)c->KW;
    std::vector<uint64_t> off(C + 1, 0);
    std::vector<uint32_t> nlen(C);
    for (uint32_t i = 0; i < C; i++) {
        nlen[i] = (uint32_t)strlen(names[i]);
        off[i + 1] = off[i] + (nlen[i] + 2 + 24 + 1) + (uint64_t)c->h_kmer_cnt[i] * (nlen[i] + 1 + k + 1 + 24 + 1);
    }
    char* buf = (char*)malloc(off[C] + 1);
    if (!buf) return fail(PF_ERR_OOM, "malloc(%llu) failed", (unsigned long long)off[C]);
    const char* b64 = c->h_b64.data();
    const uint64_t npat = c->h_b64.size() / 24;
    bool bad = false;
    parallel_for(C, [&](uint64_t a, uint64_t b) {
        for (uint64_t i = a; i < b; i++) {
            char* w = buf + off[i];
            const uint32_t L = nlen[i];
            const uint32_t cp = c->h_cl_pattern[i];
            if (cp >= npat) { bad = true; continue; }
            memcpy(w, names[i], L); w += L; *w++ = '\t'; *w++ = '\t';
            memcpy(w, b64 + (size_t)cp * 24, 24); w += 24; *w++ = '\n';
            const uint64_t o = c->h_kmer_off[i];
            for (uint32_t j = 0; j < c->h_kmer_cnt[i]; j++) {
                memcpy(w, names[i], L); w += L; *w++ = '\t';
                const uint64_t* key = c->h_kmer_key.data() + (o + j) * KW;
                if (key[0] >> 63) {
                    if (!extra_keys) { bad = true; memset(w, '?', k); }
                    else memcpy(w, extra_keys[(uint32_t)key[0]], k);
                } else {
                    // 2k-bit value, first base most significant, in KW words of 63 bits: bit b lives in word
                    // KW - 1 - b / 63 at bit b % 63
                    for (uint32_t q = 0; q < k; q++) {
                        const uint32_t b0 = 2 * (k - 1 - q), b1 = b0 + 1;
                        const uint32_t code = (uint32_t)((key[KW - 1 - b0 / 63] >> (b0 % 63)) & 1) |
                                              ((uint32_t)((key[KW - 1 - b1 / 63] >> (b1 % 63)) & 1) << 1);
                        w[q] = "ACGT"[code];
                    }
                }
                w += k; *w++ = '\t';
                const uint32_t pid = c->h_kmer_pid[o + j];
                if (pid >= npat) { bad = true; memset(w, '?', 24); }
                else memcpy(w, b64 + (size_t)pid * 24, 24);
                w += 24; *w++ = '\n';
            }
        }
    });
    if (bad) { free(buf); return fail(PF_ERR_STATE, "pf_render_kmers_to_hashes: inconsistent result (pattern id / extra key)"); }
    buf[off[C]] = 0;
    if (cluster_end) for (uint32_t i = 0; i < C; i++) cluster_end[i] = off[i + 1];
    *out = buf;
    *nbytes = off[C];
    return PF_OK;
}

int pf_render_hashes_to_patterns(pf_ctx* c, char** out, uint64_t* nbytes) {
    if (!c || !out || !nbytes) return fail(PF_ERR_ARG, "null argument");
    if (!c->have_batch || c->h_new_pid.size() != c->n_patterns - c->pid0) return fail(PF_ERR_STATE, "pf_render_* needs pf_fetch first");
    ensure_b64(c);
    const uint32_t W = c->W;
    const size_t P = c->h_new_pid.size();
    const bool miss = c->o.consider_missing != 0;
    std::vector<uint64_t> off(P + 1, 0);
    for (size_t i = 0; i < P; i++) {
        const uint32_t pid = c->h_new_pid[i];
        const uint32_t nk = c->h_pat_n[pid], n = nk & 0x7FFFFFFFu;
        uint32_t nn = 0;
        if (miss && !(nk >> 31))
            for (uint32_t w = 0; w < W; w++) nn += __builtin_popcount(c->h_pat_nan[(size_t)pid * W + w]);
        off[i + 1] = off[i] + 24 + n + (n - nn) + 1;
    }
    char* buf = (char*)malloc(off[P] + 1);
    if (!buf) return fail(PF_ERR_OOM, "malloc(%llu) failed", (unsigned long long)off[P]);
    parallel_for(P, [&](uint64_t a, uint64_t b) {
        for (uint64_t i = a; i < b; i++) {
            char* w = buf + off[i];
            const uint32_t pid = c->h_new_pid[i];
            const uint32_t nk = c->h_pat_n[pid], n = nk & 0x7FFFFFFFu;
            const bool use_nan = miss && !(nk >> 31);
            const uint32_t* bits = c->h_pat_bits.data() + (size_t)pid * W;
            const uint32_t* nan = use_nan ? c->h_pat_nan.data() + (size_t)pid * W : nullptr;
            memcpy(w, c->h_b64.data() + (size_t)pid * 24, 24); w += 24;
            for (uint32_t e = 0; e < n; e++) {
                *w++ = '\t';
                if (nan && ((nan[e >> 5] >> (e & 31)) & 1)) continue;          // '' for NaN (panfeed.py:220)
                *w++ = ((bits[e >> 5] >> (e & 31)) & 1) ? '1' : '0';
            }
            *w++ = '\n';
        }
    });
    buf[off[P]] = 0;
    *out = buf;
    *nbytes = off[P];
    return PF_OK;
}

namespace {
inline char* put_i64(char* w, long long v) {
    char tmp[24];
    int n = 0;
    unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *w++ = '-';
    while (n) *w++ = tmp[--n];
    return w;
}
inline size_t len_i64(long long v) { char t[24]; return (size_t)(put_i64(t, v) - t); }
}  // namespace

namespace {
int render_kmers_tsv_host(pf_ctx* c, const pf_target_seq* seqs, uint32_t n, const uint32_t* seg_strand_off, char** out,
                          uint64_t* nbytes, std::vector<uint64_t>* sizes_out) {
    if (!c || !out || !nbytes || (n && !seqs)) return fail(PF_ERR_ARG, "null argument");
    if (!c->have_batch) return fail(PF_ERR_STATE, "pf_render_kmers_tsv without a successful pf_submit");
    HIPCHK(hipSetDevice(c->device));
    PFCHK(fetch_strand_bits(c));          // all this renderer needs from the device (pf_fetch is not required)
    const uint32_t k = c->o.klength;
    const bool canon = c->o.canon != 0;
    // pass 1 (parallel): which strand every window of a target sequence uses (from the device's strand bits), and with
    // that the exact size of the sequence's rows -- no worst-case sizing, no compaction afterwards
    std::vector<uint64_t> off((size_t)n + 1, 0), fl_off((size_t)n + 1, 0);
    for (uint32_t i = 0; i < n; i++) {
        const long long nk = (long long)seqs[i].len - k + 1;
        fl_off[i + 1] = fl_off[i] + (canon && nk > 0 ? (uint64_t)nk : 0);
    }
    std::vector<uint8_t> rcflags(fl_off[n], 2);     // 2 = unknown, 0 forward, 1 reverse complement is canonical
    std::vector<uint64_t> size(n, 0);
    std::atomic<bool> bad{false};
    parallel_for(n, [&](uint64_t a, uint64_t b) {
        for (uint64_t i = a; i < b; i++) {
            const pf_target_seq& s = seqs[i];
            const long long nk = (long long)s.len - k + 1;
            if (nk <= 0) continue;
            uint8_t* rcflag = canon ? rcflags.data() + fl_off[i] : nullptr;
            if (canon) {
                for (uint32_t j = 0; j < s.n_segs; j++) {
                    if (!seg_strand_off || c->h_strand.empty()) { bad = true; break; }
                    const uint32_t so = seg_strand_off[s.seg_index[j]];
                    for (uint32_t q = 0; q < s.seg_nwin[j]; q++) {
                        const size_t word = (size_t)so + (q >> 6);
                        if (so == 0xFFFFFFFFu || word >= c->h_strand.size() || s.seg_start[j] + q >= (uint64_t)nk) { bad = true; break; }
                        rcflag[s.seg_start[j] + q] = (uint8_t)((c->h_strand[word] >> (q & 63)) & 1);
                    }
                }
            }
            const size_t head = strlen(s.cluster) + strlen(s.strain) + strlen(s.id) + strlen(s.chromosome) + 4 +
                                len_i64(s.strand) + 1;
            uint64_t bytes = 0;
            uint32_t ai = 0;
            for (long long pos = 0; pos < nk; pos++) {
                long long ts, te;
                if (s.strand > 0) { ts = s.start + pos; te = s.start + pos + k; }
                else { te = s.end - pos; ts = s.end - pos - k; }
                const size_t mid = len_i64(ts) + len_i64(te) + len_i64(pos - s.offset) + len_i64(pos + k - s.offset) + 4;
                if (canon) {
                    // strand column: the window's own (+-1), or what the caller worked out for a non-ACGT window
                    size_t us;
                    while (ai < s.n_ambig && s.ambig_pos[ai] < (uint64_t)pos) ai++;
                    if (ai < s.n_ambig && s.ambig_pos[ai] == (uint64_t)pos) us = len_i64(s.ambig_used[ai]);
                    else { if (rcflag[pos] > 1) bad = true; us = rcflag[pos] == 1 ? 2 : 1; }
                    bytes += head + mid + us + 1 + k + 1;
                } else {
                    bytes += 2 * (head + mid + k + 2) + len_i64(s.strand) + len_i64(-(long long)s.strand);
                }
            }
            size[i] = bytes;
        }
    });
    if (bad) return fail(PF_ERR_STATE, "pf_render_kmers_tsv: strand bits missing for a target window");
    for (uint32_t i = 0; i < n; i++) off[i + 1] = off[i] + size[i];
    char* buf = (char*)malloc(off[n] + 1);
    if (!buf) return fail(PF_ERR_OOM, "malloc(%llu) failed", (unsigned long long)off[n]);
    std::atomic<bool> bad2{false};
    // pass 2 (parallel): the rows, each sequence at its exact place
    parallel_for(n, [&](uint64_t a, uint64_t b) {
        for (uint64_t i = a; i < b; i++) {
            const pf_target_seq& s = seqs[i];
            const long long nk = (long long)s.len - k + 1;
            char* w = buf + off[i];
            if (nk > 0) {
                const uint8_t* rcflag = canon ? rcflags.data() + fl_off[i] : nullptr;
                uint32_t ai = 0;
                for (long long pos = 0; pos < nk; pos++) {
                    long long ts, te;
                    if (s.strand > 0) { ts = s.start + pos; te = s.start + pos + k; }       // panfeed.py:91-94
                    else { te = s.end - pos; ts = s.end - pos - k; }                         // panfeed.py:96-99
                    for (int rep = 0; rep < (canon ? 1 : 2); rep++) {
                        size_t l;
                        l = strlen(s.cluster); memcpy(w, s.cluster, l); w += l; *w++ = '\t';
                        l = strlen(s.strain); memcpy(w, s.strain, l); w += l; *w++ = '\t';
                        l = strlen(s.id); memcpy(w, s.id, l); w += l; *w++ = '\t';
                        l = strlen(s.chromosome); memcpy(w, s.chromosome, l); w += l; *w++ = '\t';
                        w = put_i64(w, s.strand); *w++ = '\t';
                        w = put_i64(w, ts); *w++ = '\t';
                        w = put_i64(w, te); *w++ = '\t';
                        w = put_i64(w, pos - s.offset); *w++ = '\t';                        // panfeed.py:101
                        w = put_i64(w, pos + k - s.offset); *w++ = '\t';                    // panfeed.py:102
                        if (canon) {
                            while (ai < s.n_ambig && s.ambig_pos[ai] < (uint64_t)pos) ai++;
                            if (ai < s.n_ambig && s.ambig_pos[ai] == (uint64_t)pos) {
                                w = put_i64(w, s.ambig_used[ai]); *w++ = '\t';
                                memcpy(w, s.ambig_key[ai], k); w += k;
                            } else {
                                const uint8_t rc = rcflag[(size_t)pos];
                                w = put_i64(w, rc == 1 ? -1 : 1); *w++ = '\t';
                                if (rc == 1) for (uint32_t q = 0; q < k; q++) w[q] = s.compsequence[pos + k - 1 - q];
                                else memcpy(w, s.sequence + pos, k);
                                w += k;
                            }
                        } else {
                            w = put_i64(w, rep == 0 ? s.strand : -(long long)s.strand); *w++ = '\t';   // panfeed.py:106-107
                            if (rep == 0) memcpy(w, s.sequence + pos, k);
                            else for (uint32_t q = 0; q < k; q++) w[q] = s.compsequence[pos + k - 1 - q];
                            w += k;
                        }
                        *w++ = '\n';
                    }
                }
            }
            if ((uint64_t)(w - (buf + off[i])) != size[i]) bad2 = true;
        }
    });
    if (bad2) { free(buf); return fail(PF_ERR_STATE, "pf_render_kmers_tsv: row sizes of the two passes differ"); }
    buf[off[n]] = 0;
    *out = buf;
    *nbytes = off[n];
    if (sizes_out) sizes_out->swap(size);
    return PF_OK;
}
}  // namespace

int pf_render_kmers_tsv(pf_ctx* c, const pf_target_seq* seqs, uint32_t n, const uint32_t* seg_strand_off, char** out,
                        uint64_t* nbytes) {
    return render_kmers_tsv_host(c, seqs, n, seg_strand_off, out, nbytes, nullptr);
}

// The same rows written by the GPU (kt_len_kernel / kt_text_kernel) for the sequences that are pure A/C/G/T -- one
// segment of the batch covering every window -- and by the host renderer above for the others (a target sequence with
// an 'N', a row too long for the kernel's tile), which are copied to their places in the device text: the text of all
// n sequences, in order, stays in device memory and is handed out block by block (pf_device_text_chunk).
int pf_render_kmers_tsv_device(pf_ctx* c, const pf_target_seq* seqs, uint32_t n, uint64_t* nbytes) {
    if (!c || !nbytes || (n && !seqs)) return fail(PF_ERR_ARG, "null argument");
    if (!c->have_batch) return fail(PF_ERR_STATE, "pf_render_kmers_tsv_device without a successful pf_submit");
    HIPCHK(hipSetDevice(c->device));
    c->kt_bytes = 0; c->kt_pref_valid = false;
    const uint32_t k = c->o.klength;
    const bool canon = c->o.canon != 0;
    const uint32_t reps = canon ? 1u : 2u;
    if (canon && n && (!c->last.seg_strand_off || !c->n_strand_words))
        for (uint32_t i = 0; i < n; i++) if ((long long)seqs[i].len - k + 1 > 0 && seqs[i].n_segs) return fail(PF_ERR_STATE, "the last pf_submit carried no strand bits for target segments");
    // ---- which sequences the device writes; their descriptors, the prefix block, the tiles
    std::vector<pf::KtSeq> ks;
    std::vector<uint2> tiles;
    std::vector<uint32_t> host_idx;              // sequences left to the host renderer
    std::vector<uint8_t> on_dev(n, 0);
    std::string prefix;
    auto digits = [](long long v) { return len_i64(v); };
    for (uint32_t i = 0; i < n; i++) {
        const pf_target_seq& s = seqs[i];
        const long long nk = (long long)s.len - k + 1;
        if (nk <= 0) continue;                   // no window, no row (panfeed.py:59,64)
        bool dev = s.n_segs == 1 && s.n_ambig == 0 && s.seg_start[0] == 0 && s.seg_nwin[0] == (uint64_t)nk &&
                   s.seg_index[0] < c->last_nseg && (uint64_t)nk * reps < 0xFFFFFF00ull;
        size_t plen = 0;
        if (dev) {
            plen = strlen(s.cluster) + strlen(s.strain) + strlen(s.id) + strlen(s.chromosome) + 5 + digits(s.strand);
            // the longest row this sequence can have must fit the tile KT_ROWS times over
            const long long far = s.strand > 0 ? s.start + nk + k : s.end - nk - k;
            const size_t num = std::max(digits(s.strand > 0 ? s.start : s.end), digits(far));
            const size_t gnum = std::max(digits(-s.offset), digits(nk + k - s.offset));
            const size_t rowmax = plen + 2 * num + 2 * gnum + std::max(digits(s.strand), digits(-(long long)s.strand)) + 5 + k + 1;
            if (rowmax * pf::KT_ROWS > pf::KT_TILE || prefix.size() + plen > 0x7FFFFFF0u) dev = false;
        }
        if (!dev) { host_idx.push_back(i); continue; }
        on_dev[i] = 1;
        pf::KtSeq q{};
        q.base = s.strand > 0 ? s.start : s.end; q.offset = s.offset; q.strand = s.strand;
        q.seg = s.seg_index[0]; q.nk = (uint32_t)nk; q.prefix_off = (uint32_t)prefix.size(); q.prefix_len = (uint32_t)plen;
        prefix += s.cluster; prefix += '\t'; prefix += s.strain; prefix += '\t'; prefix += s.id; prefix += '\t';
        prefix += s.chromosome; prefix += '\t';
        { char t[24]; prefix.append(t, put_i64(t, s.strand) - t); }
        prefix += '\t';
        const uint32_t rows = (uint32_t)nk * reps, si = (uint32_t)ks.size();
        for (uint32_t r0 = 0; r0 < rows; r0 += pf::KT_ROWS) tiles.push_back(make_uint2(si, r0));
        ks.push_back(q);
    }
    // ---- the host's share, rendered in one go (its sequences' sizes come back with it)
    char* htext = nullptr;
    uint64_t hbytes = 0;
    std::vector<uint64_t> hsizes;
    struct FreeText { char*& p; ~FreeText() { free(p); } } free_htext{htext};
    if (!host_idx.empty()) {
        std::vector<pf_target_seq> hs(host_idx.size());
        for (size_t j = 0; j < host_idx.size(); j++) hs[j] = seqs[host_idx[j]];
        // (the batch's strand offsets, host side: the caller's array went to the device with the batch)
        std::vector<uint32_t> sso;
        if (canon && c->last.seg_strand_off && c->last_nseg) {
            sso.resize(c->last_nseg);
            HIPCHK(hipMemcpy(sso.data(), c->last.seg_strand_off, (size_t)c->last_nseg * 4, hipMemcpyDeviceToHost));
        }
        PFCHK(render_kmers_tsv_host(c, hs.data(), (uint32_t)hs.size(), sso.empty() ? nullptr : sso.data(), &htext, &hbytes, &hsizes));
    }
    // ---- tile sizes
    const uint32_t NT = (uint32_t)tiles.size();
    pf::KtParams kp{};
    std::vector<uint32_t> tbytes(NT);
    if (NT) {
        PFCHK(c->kt_seqs.ensure(ks.size() * sizeof(pf::KtSeq)));
        PFCHK(c->kt_tiles.ensure((size_t)NT * 8));
        PFCHK(c->kt_prefix.ensure(prefix.size() + 16));
        PFCHK(c->kt_tbytes.ensure((size_t)NT * 4));
        PFCHK(c->kt_toff.ensure((size_t)NT * 8));
        HIPCHK(hipMemcpyAsync(c->kt_seqs.p, ks.data(), ks.size() * sizeof(pf::KtSeq), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->kt_tiles.p, tiles.data(), (size_t)NT * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->kt_prefix.p, prefix.data(), prefix.size(), hipMemcpyHostToDevice, c->stream));
        kp.seqs = c->kt_seqs.as<pf::KtSeq>(); kp.tiles = c->kt_tiles.as<uint2>(); kp.prefix = c->kt_prefix.as<char>();
        kp.packed = c->last.packed; kp.seg_word_off = c->last.seg_word_off; kp.seg_strand_off = c->last.seg_strand_off;
        kp.strand_bits = c->strand_bits.as<uint64_t>();
        kp.tile_bytes = c->kt_tbytes.as<uint32_t>(); kp.tile_off = c->kt_toff.as<uint64_t>();
        kp.k = k; kp.canon = canon ? 1u : 0u;
        hipLaunchKernelGGL(pf::kt_len_kernel, dim3(NT), dim3(pf::KT_ROWS), 0, c->stream, kp);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(tbytes.data(), c->kt_tbytes.p, (size_t)NT * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    // ---- offsets, in the order of the sequences
    std::vector<uint64_t> toff(NT), hoff(host_idx.size());
    uint64_t total = 0;
    {
        size_t ti = 0, hi = 0;
        uint32_t si = 0;
        for (uint32_t i = 0; i < n; i++) {
            if (on_dev[i]) {
                while (ti < NT && tiles[ti].x == si) { toff[ti] = total; total += tbytes[ti]; ti++; }
                si++;
            } else if (hi < host_idx.size() && host_idx[hi] == i) {
                hoff[hi] = total; total += hsizes[hi]; hi++;
            }
        }
    }
    PFCHK(c->kt_text.ensure((size_t)total + 64));
    if (NT) {
        HIPCHK(hipMemcpyAsync(c->kt_toff.p, toff.data(), (size_t)NT * 8, hipMemcpyHostToDevice, c->stream));
        kp.text = c->kt_text.as<char>();
        hipLaunchKernelGGL(pf::kt_text_kernel, dim3(NT), dim3(pf::KT_ROWS), 0, c->stream, kp);
        HIPCHK(hipGetLastError());
    }
    {
        uint64_t at = 0;
        for (size_t j = 0; j < host_idx.size(); j++) {
            if (hsizes[j]) HIPCHK(hipMemcpyAsync(c->kt_text.as<char>() + hoff[j], htext + at, (size_t)hsizes[j], hipMemcpyHostToDevice, c->stream));
            at += hsizes[j];
        }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    c->kt_bytes = total;
    c->kt_host_seqs = (uint32_t)host_idx.size();
    *nbytes = total;
    return PF_OK;
}

// Bytes [offset, offset + n) of the text the last pf_render_kmers_tsv_device left on the device, n = min(max_bytes, what
// remains), in pinned host memory; the block after it is already on its way when the call returns.
int pf_device_text_chunk(pf_ctx* c, uint64_t offset, uint64_t max_bytes, const char** ptr, uint64_t* nbytes) {
    if (!c || !ptr || !nbytes || !max_bytes) return fail(PF_ERR_ARG, "null argument");
    if (offset > c->kt_bytes) return fail(PF_ERR_ARG, "offset beyond the text");
    HIPCHK(hipSetDevice(c->device));
    auto ensure_pin = [&](int slot) -> int {
        if (c->kt_pin_caps[slot] >= max_bytes) return PF_OK;
        if (c->kt_pins[slot]) (void)hipHostFree(c->kt_pins[slot]);
        c->kt_pins[slot] = nullptr; c->kt_pin_caps[slot] = 0;
        hipError_t e = hipHostMalloc((void**)&c->kt_pins[slot], max_bytes, hipHostMallocDefault);
        if (e != hipSuccess) return fail(PF_ERR_OOM, "hipHostMalloc(%llu) failed: %s", (unsigned long long)max_bytes, hipGetErrorString(e));
        c->kt_pin_caps[slot] = max_bytes;
        return PF_OK;
    };
    const uint64_t n = std::min<uint64_t>(max_bytes, c->kt_bytes - offset);
    int slot;
    if (c->kt_pref_valid && c->kt_pref_off == offset && c->kt_pref_n == n) {
        slot = c->kt_pref_slot;                       // requested by the call before: wait for it
        HIPCHK(hipStreamSynchronize(c->side));
    } else {
        HIPCHK(hipStreamSynchronize(c->side));        // (a block in flight that nobody asked for)
        slot = 0;
        PFCHK(ensure_pin(slot));
        if (n) HIPCHK(hipMemcpyAsync(c->kt_pins[slot], c->kt_text.as<char>() + offset, n, hipMemcpyDeviceToHost, c->side));
        HIPCHK(hipStreamSynchronize(c->side));
    }
    c->kt_pref_valid = false;
    const uint64_t next = offset + n;
    if (n && next < c->kt_bytes) {
        const int ns = slot ^ 1;
        PFCHK(ensure_pin(ns));
        const uint64_t nn = std::min<uint64_t>(max_bytes, c->kt_bytes - next);
        HIPCHK(hipMemcpyAsync(c->kt_pins[ns], c->kt_text.as<char>() + next, nn, hipMemcpyDeviceToHost, c->side));
        c->kt_pref_valid = true; c->kt_pref_off = next; c->kt_pref_n = nn; c->kt_pref_slot = ns;
    }
    *ptr = c->kt_pins[slot];
    *nbytes = n;
    return PF_OK;
}

void pf_free_text(char* p) { free(p); }

namespace {
int merge_impl(pf_ctx* c, const void* d_gathered, uint64_t n_total, uint64_t my_first, uint64_t my_count,
               const void* d_slot_counts, uint64_t slot_rows, void* d_keep, uint64_t* n_global) {
    HIPCHK(hipSetDevice(c->device));
    *n_global = 0;
    if (!n_total) return PF_OK;
    if (!d_gathered || (my_count && !d_keep) || my_first + my_count > n_total) return fail(PF_ERR_ARG, "pf_merge_patterns: bad range");
    uint64_t cap = 1024;
    while (cap < 2 * n_total) cap <<= 1;
    PFCHK(c->mg_lo.ensure(cap * 32));
    PFCHK(c->mg_cnt.ensure(8));
    PFCHK(fill_u64(c, c->mg_lo.p, pf::EMPTY64, cap * 4));
    HIPCHK(hipMemsetAsync(c->mg_cnt.p, 0, 8, c->stream));
    pf::MergeParams mp{};
    mp.gathered = (const uint64_t*)d_gathered; mp.n = n_total;
    mp.tab = c->mg_lo.as<uint64_t>();
    mp.cap = cap; mp.my_first = my_first; mp.my_count = my_count; mp.keep = (uint8_t*)d_keep;
    mp.slot_counts = (const int64_t*)d_slot_counts; mp.slot_rows = d_slot_counts ? slot_rows : 0;
    mp.n_global = c->mg_cnt.as<unsigned long long>();
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_total + 255) / 256, 8192);
    hipLaunchKernelGGL(pf::merge_insert_kernel, dim3(blocks), dim3(256), 0, c->stream, mp);
    HIPCHK(hipGetLastError());
    if (my_count) {
        const uint32_t b2 = (uint32_t)std::min<uint64_t>((my_count + 255) / 256, 8192);
        hipLaunchKernelGGL(pf::merge_lookup_kernel, dim3(b2), dim3(256), 0, c->stream, mp);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(n_global, c->mg_cnt.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PF_OK;
}
}  // namespace

int pf_merge_patterns(pf_ctx* c, const void* d_gathered, uint64_t n_total, uint64_t my_first, uint64_t my_count,
                      void* d_keep, uint64_t* n_global) {
    if (!c || !n_global) return fail(PF_ERR_ARG, "null argument");
    return merge_impl(c, d_gathered, n_total, my_first, my_count, nullptr, 0, d_keep, n_global);
}

int pf_merge_patterns_padded(pf_ctx* c, const void* d_gathered, uint64_t world, uint64_t slot_rows,
                             const void* d_slot_counts, uint64_t rank, uint64_t my_count, void* d_keep,
                             uint64_t* n_global) {
    if (!c || !n_global || !d_slot_counts || rank >= world || my_count > slot_rows) return fail(PF_ERR_ARG, "pf_merge_patterns_padded: bad argument");
    return merge_impl(c, d_gathered, world * slot_rows, rank * slot_rows, my_count, d_slot_counts, slot_rows, d_keep, n_global);
}

int pf_result_checksum(pf_ctx* c, uint64_t out[3]) {
    if (!c || !out) return fail(PF_ERR_ARG, "null argument");
    if (!c->have_batch) return fail(PF_ERR_STATE, "pf_result_checksum without a successful pf_submit");
    HIPCHK(hipSetDevice(c->device));
    const uint32_t C = c->n_clusters, KW = (uint32_t)c->KW;
    out[0] = out[1] = out[2] = 0;
    if (!C) return PF_OK;
    // where each cluster's k-mers stand: arena of the cluster + offset inside it
    std::vector<uint64_t> off(C);
    HIPCHK(hipMemcpy(off.data(), c->cl_kmer_off.p, (size_t)C * 8, hipMemcpyDeviceToHost));
    std::vector<const uint64_t*> kp(C);
    std::vector<const uint32_t*> pp(C);
    for (uint32_t i = 0; i < C; i++) {
        const Arena* ar = c->arenas[c->cluster_arena[i]];
        const uint64_t local = off[i] >= ar->base ? off[i] - ar->base : 0;      // clusters without k-mers: never read
        kp[i] = ar->key.as<uint64_t>() + local * KW;
        pp[i] = ar->pid.as<uint32_t>() + local;
    }
    DevBuf d_kp, d_pp, d_acc;
    PFCHK(d_kp.ensure((size_t)C * 8)); PFCHK(d_pp.ensure((size_t)C * 8)); PFCHK(d_acc.ensure(24));
    int rc = PF_OK;
    if (hipMemcpy(d_kp.p, kp.data(), (size_t)C * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_pp.p, pp.data(), (size_t)C * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemsetAsync(d_acc.p, 0, 24, c->stream) != hipSuccess) rc = fail(PF_ERR_HIP, "pf_result_checksum: copy failed");
    if (rc == PF_OK) {
        hipLaunchKernelGGL(pf::result_checksum_kernel, dim3(C), dim3(256), 0, c->stream,
                           (const uint64_t* const*)d_kp.p, (const uint32_t* const*)d_pp.p, c->cl_kmer_cnt.as<uint32_t>(),
                           c->cl_unique.as<uint32_t>(), c->cl_pattern.as<uint32_t>(), c->pat_md5.as<uint8_t>(), KW,
                           (unsigned long long*)d_acc.p);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess ||
            hipMemcpy(out, d_acc.p, 24, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(PF_ERR_HIP, "pf_result_checksum: kernel failed");
    }
    d_kp.release(); d_pp.release(); d_acc.release();
    return rc;
}

int pf_pattern_count(pf_ctx* c, uint64_t* n) {
    if (!c || !n) return fail(PF_ERR_ARG, "null argument");
    *n = c->n_patterns;
    return PF_OK;
}

int pf_submit_gather(pf_ctx* c, const pf_batch* b, const pf_gather* g, pf_result* r) {
    if (!c || !b || !g) return fail(PF_ERR_ARG, "pf_submit_gather: null argument");
    c->pending_gather = g;
    const int rc = pf_submit(c, b, r);
    c->pending_gather = nullptr;
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------------------
// One-pass ingest: the reader's sink (pf_ingest.h).  A ring of blocks, each a pinned host block with a device twin: a reader
// thread read()s a genome's file straight into a pinned block, parses the GFF lines there, measures the contigs without
// copying a base, and gives the block back with one piece per contig; the block goes to its twin (one copy over PCIe) and
// genome_pack_text_kernel de-wraps, upper-cases and packs the pieces into the genome store, while other threads are still
// reading other files.  The host never touches a base of a pure-A/C/G/T contig: rounds 1-4 made three passes over every
// genome on the host (read, upper-casing copy into contig strings, copy into pinned blocks) before the same pack.
namespace {
struct IngestSlot {
    char* pin = nullptr; size_t cap = 0;
    bool own = false;         // its blocks are its own (a file larger than the ring's slots), not parts of the ring's two blocks
    DevBuf dev, dpieces;
    pf::TextPiece* pin_pieces = nullptr; size_t pieces_cap = 0;
    hipEvent_t ev = nullptr;
    bool held = false;        // a reader thread is filling it
    bool inflight = false;    // its copy / kernel may not have finished (ev)
};
struct Ingest {
    pf_ctx* c = nullptr;
    const pf_pangenome_opts* o = nullptr;
    pf_ctx* (*get_ctx)(void*) = nullptr;
    void* user = nullptr;
    std::once_flag once;
    bool setup_ok = false;
    char* ring_pin = nullptr;
    DevBuf ring_dev;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<IngestSlot> slots;
    uint64_t store_cap = 0;
    std::atomic<uint64_t> store_used{0};
    std::string err;
    int rc = PF_OK;
    uint64_t bytes_up = 0;
    int failed(int code, const std::string& m) { if (rc == PF_OK) { rc = code; err = m; } return code; }   // (caller holds mu)
};
int ingest_ready(Ingest* I);
int ingest_acquire(void* self, size_t bytes, char** host, uint32_t* slot) {
    Ingest* I = (Ingest*)self;
    if (ingest_ready(I) != PF_OK) return I->rc != PF_OK ? I->rc : PF_ERR_STATE;
    if (hipSetDevice(I->c->device) != hipSuccess) return PF_ERR_HIP;
    std::unique_lock<std::mutex> lk(I->mu);
    for (;;) {
        if (I->rc != PF_OK) return I->rc;
        int pick = -1, waitable = -1;
        for (size_t i = 0; i < I->slots.size(); i++) {
            IngestSlot& s = I->slots[i];
            if (s.held) continue;
            if (s.inflight && hipEventQuery(s.ev) == hipSuccess) s.inflight = false;
            if (!s.inflight) { if (pick < 0 || (s.cap >= bytes && I->slots[pick].cap < bytes)) pick = (int)i; }
            else if (waitable < 0) waitable = (int)i;
        }
        if (pick >= 0) {
            IngestSlot& s = I->slots[pick];
            s.held = true;
            if (s.cap < bytes) {
                lk.unlock();                                   // (the slot is ours: nobody else looks at it while it is held)
                if (s.pin && s.own) { (void)hipHostUnregister(s.pin); free(s.pin); }
                if (!s.own) { s.dev.p = nullptr; s.dev.cap = 0; s.dev.view = false; }     // (a part of the ring's block: not ours to free)
                s.pin = nullptr; s.cap = 0; s.own = true;
                const size_t want = bytes + bytes / 4 + 4096;
                bool ok = posix_memalign((void**)&s.pin, 4096, want) == 0 && s.pin;
                if (ok && hipHostRegister(s.pin, want, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); free(s.pin); s.pin = nullptr; ok = false; }
                ok = ok && s.dev.ensure(want) == PF_OK;
                lk.lock();
                if (!ok) { s.held = false; I->cv.notify_all(); return I->failed(PF_ERR_OOM, "no pinned / device block of " + std::to_string(want) + " bytes for a genome's text"); }
                s.cap = want;
            }
            *host = s.pin; *slot = (uint32_t)pick;
            return PF_OK;
        }
        if (waitable >= 0) {                                    // every free block is still on its way to the device
            hipEvent_t ev = I->slots[waitable].ev;
            lk.unlock();
            (void)hipEventSynchronize(ev);
            lk.lock();
            continue;
        }
        I->cv.wait(lk);
    }
}
uint64_t ingest_claim(void* self, uint64_t nwords) {
    Ingest* I = (Ingest*)self;
    if (ingest_ready(I) != PF_OK) return UINT64_MAX;
    const uint64_t at = I->store_used.fetch_add(nwords);
    return at + nwords <= I->store_cap ? at : UINT64_MAX;
}
int ingest_submit(void* self, uint32_t slot, size_t text_bytes, const pf_ingest_piece* pieces, uint32_t n) {
    Ingest* I = (Ingest*)self;
    IngestSlot& s = I->slots[slot];
    std::unique_lock<std::mutex> lk(I->mu);
    auto done = [&](int rc) { s.held = false; I->cv.notify_all(); return rc; };
    if (!n || !text_bytes || I->rc != PF_OK) return done(I->rc);
    if (hipSetDevice(I->c->device) != hipSuccess) return done(I->failed(PF_ERR_HIP, "hipSetDevice failed"));
    // the pieces travel behind the text in the same block when there is room (one copy instead of two per file)
    const size_t tail = (text_bytes + 63) & ~(size_t)63;
    const bool inline_pieces = tail + (size_t)n * sizeof(pf::TextPiece) <= s.cap;
    pf::TextPiece* const pcs = inline_pieces ? reinterpret_cast<pf::TextPiece*>(s.pin + tail) : nullptr;
    if (!inline_pieces && n > s.pieces_cap) {
        if (s.pin_pieces) (void)hipHostFree(s.pin_pieces);
        s.pin_pieces = nullptr; s.pieces_cap = 0;
        const size_t want = (size_t)n + n / 2 + 64;
        if (hipHostMalloc((void**)&s.pin_pieces, want * sizeof(pf::TextPiece), hipHostMallocDefault) != hipSuccess)
            return done(I->failed(PF_ERR_OOM, "hipHostMalloc failed (ingest pieces)"));
        s.pieces_cap = want;
    }
    uint64_t lo = text_bytes, blocks = 0;
    for (uint32_t i = 0; i < n; i++) {
        pf::TextPiece& t = (inline_pieces ? pcs : s.pin_pieces)[i];
        t.text_off = pieces[i].text_off; t.dst_word = pieces[i].dst_word; t.nbases = (uint32_t)pieces[i].nbases;
        t.nwords = (uint32_t)(2 * ((pieces[i].nbases + 63) / 64) + 4);
        t.width = pieces[i].width; t.eol = pieces[i].eol; t.block0 = (uint32_t)blocks; t.pad = 0;
        blocks += (t.nwords + 255) / 256;
        lo = std::min<uint64_t>(lo, pieces[i].text_off);
        // the last letter's byte must lie inside the block
        const uint64_t lines = pieces[i].width && pieces[i].nbases ? (pieces[i].nbases - 1) / pieces[i].width : 0;
        if (pieces[i].text_off + pieces[i].nbases + lines * pieces[i].eol > text_bytes || pieces[i].dst_word + t.nwords > I->store_cap)
            return done(I->failed(PF_ERR_STATE, "ingest: a contig's letters lie outside its block"));
    }
    if (blocks > 0x7FFFFFFFull) return done(I->failed(PF_ERR_CAPACITY, "ingest: too many words in one file"));
    lo &= ~(uint64_t)63;
    hipStream_t st = I->c->stream;
    const pf::TextPiece* dpcs;
    if (getenv("PF_DEBUG_INGEST_SKIP_UPLOAD")) return done(PF_OK);        // (timing experiment: the reader alone; nothing reaches the store)
    if (inline_pieces) {
        if (hipMemcpyAsync((char*)s.dev.p + lo, s.pin + lo, tail + (size_t)n * sizeof(pf::TextPiece) - lo, hipMemcpyHostToDevice, st) != hipSuccess)
            return done(I->failed(PF_ERR_HIP, "ingest: upload of a genome's text failed"));
        dpcs = reinterpret_cast<const pf::TextPiece*>((char*)s.dev.p + tail);
    } else {
        if (s.dpieces.ensure((size_t)n * sizeof(pf::TextPiece)) != PF_OK) return done(I->failed(PF_ERR_OOM, "device block for ingest pieces"));
        if (hipMemcpyAsync((char*)s.dev.p + lo, s.pin + lo, text_bytes - lo, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(s.dpieces.p, s.pin_pieces, (size_t)n * sizeof(pf::TextPiece), hipMemcpyHostToDevice, st) != hipSuccess)
            return done(I->failed(PF_ERR_HIP, "ingest: upload of a genome's text failed"));
        dpcs = (const pf::TextPiece*)s.dpieces.p;
    }
    hipLaunchKernelGGL(pf::genome_pack_text_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, (const uint8_t*)s.dev.p,
                       dpcs, n, I->c->g_store.as<uint64_t>());
    if (hipGetLastError() != hipSuccess || hipEventRecord(s.ev, st) != hipSuccess) return done(I->failed(PF_ERR_HIP, "genome_pack_text_kernel launch failed"));
    s.inflight = true;
    I->bytes_up += text_bytes - lo;
    return done(PF_OK);
}
}  // namespace

namespace {
// what the first callback of the reader sets up (the context may still be in the making while the reader parses the table:
// it is asked for here, when the first genome needs it): the store, the events, the ring's two blocks
int ingest_setup(Ingest* I) {
    const pf_pangenome_opts* o = I->o;
    pf_ctx* c = I->get_ctx ? I->get_ctx(I->user) : nullptr;
    if (!c) return fail(PF_ERR_ARG, "pf_pangenome_open_device: no context");
    I->c = c;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    // the store's room: 2 bits per letter, and 2 * ceil(len / 64) + 4 words per contig -- the letters are at most the files'
    // bytes; half a byte of store per byte of text covers contigs down to ~200 letters on average (smaller ones: the
    // claim fails, PF_ERR_CAPACITY, and the caller takes pf_pangenome_open + pf_genomes_upload)
    uint64_t text_bytes = 0;
    size_t biggest = 0;
    for (uint32_t i = 0; i < o->n_genomes; i++) {
        const char* path = (o->fasta_paths && o->fasta_paths[i]) ? o->fasta_paths[i] : (o->gff_paths ? o->gff_paths[i] : nullptr);
        struct stat st;
        if (path && ::stat(path, &st) == 0 && S_ISREG(st.st_mode)) { text_bytes += (uint64_t)st.st_size; biggest = std::max<size_t>(biggest, (size_t)st.st_size); }
        else text_bytes += 64ull << 20;
    }
    I->store_cap = text_bytes / 16 + (8ull << 20);            // words
    c->g_store.release();
    c->g_words = 0;
    PFCHK(c->g_store.ensure((size_t)I->store_cap * 8));
    const unsigned nt = pf_host_threads(32u);
    // (a slot is held for one memcpy and handed to the copy engine, which empties it in a fifth of a millisecond: half as many
    // slots as reader threads, plus a few, are never all busy -- and 64 slots of 10 MB were 0.6 GB to allocate and page-lock
    // in front of the first upload)
    I->slots.resize(std::max<size_t>(4, std::min<size_t>((size_t)nt / 2 + 4, (size_t)o->n_genomes + 1)));
    for (auto& s : I->slots)
        if (hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) != hipSuccess) return fail(PF_ERR_HIP, "hipEventCreate failed");
    // the ring's blocks: ONE page-locked allocation and ONE device allocation, cut into slots that hold the largest file (an
    // allocation per slot was 2 x 32 calls into the driver, one after the other, in front of the first upload).  Ordinary
    // memory, page-locked where it lies (hipHostRegister): the reader only COPIES the FASTA text into these blocks -- it
    // parses in its threads' own buffers.
    const size_t slot_bytes = (std::min<size_t>(biggest, 256u << 20) + 64 + (64u << 10) + 4095) & ~(size_t)4095;   // (+ room for the pieces)
    const size_t total = slot_bytes * I->slots.size();
    if (posix_memalign((void**)&I->ring_pin, 4096, total) == 0 && I->ring_pin) {
        if (hipHostRegister(I->ring_pin, total, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); free(I->ring_pin); I->ring_pin = nullptr; }
    } else I->ring_pin = nullptr;
    if (I->ring_pin && I->ring_dev.ensure(total) == PF_OK) {
        for (size_t i = 0; i < I->slots.size(); i++) {
            IngestSlot& s = I->slots[i];
            s.pin = I->ring_pin + i * slot_bytes; s.cap = slot_bytes; s.own = false;
            s.dev.p = (char*)I->ring_dev.p + i * slot_bytes; s.dev.cap = 0; s.dev.view = true;
        }
    } else if (I->ring_pin) { (void)hipHostUnregister(I->ring_pin); free(I->ring_pin); I->ring_pin = nullptr; I->ring_dev.release(); }   // (slots then allocate their own)
    return PF_OK;
}
// every callback starts here: PF_OK once the set-up has succeeded (it runs once, on whichever reader thread comes first)
int ingest_ready(Ingest* I) {
    std::call_once(I->once, [I] {
        const int rc = ingest_setup(I);
        if (rc != PF_OK) { std::lock_guard<std::mutex> g(I->mu); I->failed(rc, pf_last_error()); }
        I->setup_ok = rc == PF_OK;
    });
    return I->setup_ok ? PF_OK : (I->rc != PF_OK ? I->rc : PF_ERR_STATE);
}
}  // namespace

int pf_pangenome_open_device_cb(const pf_pangenome_opts* o, pf_ctx* (*get_ctx)(void*), void* user, pf_pangenome** out) {
    if (!o || !get_ctx || !out) return fail(PF_ERR_ARG, "pf_pangenome_open_device: null argument");
    *out = nullptr;
    Ingest I;
    I.o = o; I.get_ctx = get_ctx; I.user = user;
    const bool dbg = getenv("PF_DEBUG_TIMING") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[open_device] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - T0).count());
        T0 = t;
    };
    pf_pangenome* P = nullptr;
    pf_ingest_sink sink{&I, ingest_acquire, ingest_claim, ingest_submit};
    int rc = pf_pangenome_open_sink(o, &sink, &P);           // (its error text is this thread's: the reader sets it here)
    if (I.rc != PF_OK) rc = fail(I.rc, "%s", I.err.c_str());
    lap("reader (table, files -> pieces)");
    if (rc == PF_OK && ingest_ready(&I) != PF_OK) rc = fail(I.rc != PF_OK ? I.rc : PF_ERR_STATE, "%s", I.err.c_str());   // (a pangenome without a genome)
    pf_ctx* c = I.c;
    if (c) {
        (void)hipSetDevice(c->device);
        if (hipStreamSynchronize(c->stream) != hipSuccess && rc == PF_OK) rc = fail(PF_ERR_HIP, "the genomes' upload failed");
    }
    lap("last uploads");
    if (I.ring_pin) { (void)hipHostUnregister(I.ring_pin); free(I.ring_pin); }
    I.ring_dev.release();
    for (auto& s : I.slots) {
        if (s.pin && s.own) { (void)hipHostUnregister(s.pin); free(s.pin); }
        if (!s.own) { s.dev.p = nullptr; s.dev.cap = 0; s.dev.view = false; }
        if (s.pin_pieces) (void)hipHostFree(s.pin_pieces);
        if (s.ev) (void)hipEventDestroy(s.ev);
        s.dev.release(); s.dpieces.release();
    }
    if (rc != PF_OK) {
        if (P) pf_pangenome_close(P);
        if (c) c->g_store.release();
        return rc;
    }
    c->g_words = std::min<uint64_t>(I.store_used.load(), I.store_cap);
    *out = P;
    return PF_OK;
}

int pf_pangenome_open_device(const pf_pangenome_opts* o, pf_ctx* c, pf_pangenome** out) {
    if (!c) return fail(PF_ERR_ARG, "pf_pangenome_open_device: null context");
    return pf_pangenome_open_device_cb(o, [](void* u) { return (pf_ctx*)u; }, c, out);
}

int pf_genomes_clear(pf_ctx* c) {
    if (!c) return fail(PF_ERR_ARG, "pf_genomes_clear: null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->g_store.release();
    c->g_words = 0;
    return PF_OK;
}

int pf_genomes_upload(pf_ctx* c, uint32_t n, const char* const* ascii, const uint64_t* len, uint64_t* word_off) {
    if (!c || (n && (!ascii || !len || !word_off))) return fail(PF_ERR_ARG, "pf_genomes_upload: null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (len[i] >= 0xFFFFFF00ull) return fail(PF_ERR_ARG, "contig %u is too long for 32-bit coordinates", i);
        word_off[i] = total;
        total += 2 * ((len[i] + 63) / 64) + 4;
    }
    c->g_store.release();
    PFCHK(c->g_store.ensure((size_t)std::max<uint64_t>(total, 2) * 8));
    c->g_words = total;
    // staged in blocks: pinned host block -> device ASCII block -> 2-bit words.  Two sets of staging buffers: the host
    // threads copy block i + 1 into its pinned block while block i is on its way to the device and being packed there.
    const size_t BLOCK = 64u << 20;
    struct Src { const char* p; size_t n, at; };
    struct Blk { std::vector<pf::PackPiece> pieces; std::vector<Src> src; size_t fill = 0; uint32_t blocks = 0; };
    std::vector<Blk> blks(1);
    for (uint32_t i = 0; i < n; i++) {
        uint64_t done = 0;
        const uint64_t L = len[i];
        do {
            if (BLOCK - blks.back().fill < 64) blks.emplace_back();
            Blk& bk = blks.back();
            const uint64_t room = (BLOCK - bk.fill - 32) / 32 * 32;               // bases this block still takes
            const uint64_t take = std::min<uint64_t>(L - done, room);
            const bool last = done + take == L;
            bk.src.push_back(Src{ascii[i] + done, (size_t)take, bk.fill});
            const size_t padded = (take + 31) / 32 * 32;
            pf::PackPiece pc{};
            pc.ascii_off = bk.fill; pc.dst_word = word_off[i] + done / 32; pc.nbases = (uint32_t)take;
            const uint64_t contig_words = 2 * ((L + 63) / 64) + 4;
            pc.nwords = (uint32_t)(last ? contig_words - done / 32 : take / 32);
            pc.block0 = bk.blocks;
            bk.blocks += (pc.nwords + 255) / 256;
            if (pc.nwords) bk.pieces.push_back(pc);
            bk.fill += padded;
            done += take;
            if (!last) blks.emplace_back();
        } while (done < L);
    }
    char* pin[2] = {nullptr, nullptr};
    DevBuf dasc[2], dpieces[2];
    hipEvent_t ev[2] = {nullptr, nullptr};
    int rc = PF_OK;
    do {
        for (int q = 0; q < 2 && rc == PF_OK; q++) {
            if (hipHostMalloc((void**)&pin[q], BLOCK, hipHostMallocDefault) != hipSuccess) { rc = fail(PF_ERR_OOM, "hipHostMalloc(%zu) failed", BLOCK); break; }
            if ((rc = dasc[q].ensure(BLOCK)) != PF_OK) break;
            if (hipEventCreateWithFlags(&ev[q], hipEventDisableTiming) != hipSuccess) { rc = fail(PF_ERR_HIP, "hipEventCreate failed"); break; }
        }
        if (rc != PF_OK) break;
        for (size_t bi = 0; bi < blks.size() && rc == PF_OK; bi++) {
            Blk& bk = blks[bi];
            if (bk.pieces.empty()) continue;
            const int q = (int)(bi & 1);
            if (hipEventSynchronize(ev[q]) != hipSuccess) { rc = fail(PF_ERR_HIP, "hipEventSynchronize failed"); break; }   // the slot's last block has left it
            // the block's pieces into the pinned block, on the host threads (padding bases are 'A')
            char* dst = pin[q];
            parallel_for(bk.fill, [&](uint64_t a, uint64_t e) {          // every thread takes a byte range of the block
                for (const Src& sp : bk.src) {
                    const uint64_t lo = std::max<uint64_t>(a, sp.at), hi = std::min<uint64_t>(e, sp.at + sp.n);
                    if (lo < hi) memcpy(dst + lo, sp.p + (lo - sp.at), hi - lo);
                }
            });
            for (const Src& sp : bk.src) { const size_t padded = (sp.n + 31) / 32 * 32; memset(dst + sp.at + sp.n, 'A', padded - sp.n); }
            if ((rc = dpieces[q].ensure(bk.pieces.size() * sizeof(pf::PackPiece))) != PF_OK) break;
            if (hipMemcpyAsync(dasc[q].p, pin[q], bk.fill, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                hipMemcpyAsync(dpieces[q].p, bk.pieces.data(), bk.pieces.size() * sizeof(pf::PackPiece), hipMemcpyHostToDevice, c->stream) != hipSuccess) {
                rc = fail(PF_ERR_HIP, "genome upload failed"); break;
            }
            hipLaunchKernelGGL(pf::genome_pack_kernel, dim3(bk.blocks), dim3(256), 0, c->stream, (const uint8_t*)dasc[q].p,
                               (const pf::PackPiece*)dpieces[q].p, (uint32_t)bk.pieces.size(), c->g_store.as<uint64_t>());
            if (hipGetLastError() != hipSuccess || hipEventRecord(ev[q], c->stream) != hipSuccess) { rc = fail(PF_ERR_HIP, "genome_pack_kernel launch failed"); break; }
        }
    } while (0);
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == PF_OK) rc = fail(PF_ERR_HIP, "genome upload failed");
    for (int q = 0; q < 2; q++) {
        if (pin[q]) (void)hipHostFree(pin[q]);
        if (ev[q]) (void)hipEventDestroy(ev[q]);
        dasc[q].release(); dpieces[q].release();
    }
    return rc;
}

namespace {
// base64 of every digest up to n_patterns, on the device (incremental)
int ensure_b64_dev(pf_ctx* c) {
    if (!c->pat_b64.p) PFCHK(c->pat_b64.ensure((size_t)c->pt.pool * 24));
    const uint32_t p1 = c->n_patterns;
    if (c->b64_done < p1) {
        hipLaunchKernelGGL(pf::b64_kernel, dim3((p1 - c->b64_done + 255) / 256), dim3(256), 0, c->stream,
                           c->pat_md5.as<uint8_t>(), c->b64_done, p1, c->pat_b64.as<char>());
        HIPCHK(hipGetLastError());
        c->b64_done = p1;
    }
    return PF_OK;
}
// pinned host block `slot` of the rendered text, at least `total` bytes
int text_pin(pf_ctx* c, size_t total) {
    c->txt_slot ^= 1;
    char*& pin = c->txt_pins[c->txt_slot];
    size_t& cap = c->txt_pin_caps[c->txt_slot];
    if (total > cap) {
        if (pin) (void)hipHostFree(pin);
        pin = nullptr; cap = 0;
        const size_t want = total + total / 4;
        if (hipHostMalloc((void**)&pin, want, hipHostMallocDefault) != hipSuccess) return fail(PF_ERR_OOM, "hipHostMalloc(%zu) failed", want);
        cap = want;
    }
    return PF_OK;
}
}  // namespace

int pf_render_pattern_rows(pf_ctx* c, const uint32_t* pids, uint64_t n, const char** text, uint64_t* nbytes) {
    if (!c || !text || !nbytes || (n && !pids)) return fail(PF_ERR_ARG, "pf_render_pattern_rows: null argument");
    HIPCHK(hipSetDevice(c->device));
    *text = nullptr; *nbytes = 0;
    if (!n) return PF_OK;
    if (n > 0x7FFFFFFFull) return fail(PF_ERR_ARG, "pf_render_pattern_rows: too many rows in one call");
    for (uint64_t i = 0; i < n; i++)
        if (pids[i] >= c->n_patterns) return fail(PF_ERR_ARG, "pf_render_pattern_rows: pattern id %u out of range (%u patterns)", pids[i], c->n_patterns);
    hipStream_t st = c->stream;
    const uint32_t P = (uint32_t)n, W = c->W;
    PFCHK(ensure_b64_dev(c));
    PFCHK(c->rp_order.ensure((size_t)P * 4));
    PFCHK(c->rp_rlen.ensure((size_t)P * 4));
    PFCHK(c->rp_rowoff.ensure(((size_t)P + 1) * 8));
    HIPCHK(hipMemcpyAsync(c->rp_order.p, pids, (size_t)P * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(pf::hp_rowlen_kernel, dim3((P + 255) / 256), dim3(256), 0, st, c->pat_n.as<uint32_t>(),
                       c->o.consider_missing ? c->pat_nan.as<uint32_t>() : (const uint32_t*)nullptr, W, 0u,
                       c->rp_order.as<uint32_t>(), P, c->rp_rlen.as<uint32_t>());
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> rlen(P);
    HIPCHK(hipMemcpyAsync(rlen.data(), c->rp_rlen.p, (size_t)P * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<uint64_t> row_off((size_t)P + 1, 0);
    for (uint32_t i = 0; i < P; i++) row_off[i + 1] = row_off[i] + rlen[i];
    const uint64_t total = row_off[P];
    PFCHK(c->txt_dev.ensure(total + 16));
    PFCHK(text_pin(c, total + 16));
    HIPCHK(hipMemcpyAsync(c->rp_rowoff.p, row_off.data(), ((size_t)P + 1) * 8, hipMemcpyHostToDevice, st));
    pf::HpTextParams hpp{};
    hpp.order = c->rp_order.as<uint32_t>(); hpp.row_off = c->rp_rowoff.as<uint64_t>();
    hpp.pat_bits = c->pat_bits.as<uint32_t>();
    hpp.pat_nan = c->o.consider_missing ? c->pat_nan.as<uint32_t>() : nullptr;
    hpp.pat_n = c->pat_n.as<uint32_t>(); hpp.b64 = c->pat_b64.as<char>();
    hpp.text = c->txt_dev.as<char>(); hpp.n = P; hpp.W = W;
    hipLaunchKernelGGL(pf::hp_text_kernel, dim3(P), dim3(256), 0, st, hpp);
    HIPCHK(hipGetLastError());
    char* pin = c->txt_pins[c->txt_slot];
    HIPCHK(hipMemcpyAsync(pin, c->txt_dev.p, total, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *text = pin; *nbytes = total;
    return PF_OK;
}

int pf_render_device(pf_ctx* c, const char* const* names, const char* extra_keys, uint64_t n_extra,
                     const char** kh, uint64_t* kh_bytes, const char** hp, uint64_t* hp_bytes) {
    return pf_render_device_ex(c, names, extra_keys, n_extra, 0, kh, kh_bytes, hp, hp_bytes);
}

int pf_render_device_ex(pf_ctx* c, const char* const* names, const char* extra_keys, uint64_t n_extra, uint32_t flags,
                        const char** kh, uint64_t* kh_bytes, const char** hp, uint64_t* hp_bytes) {
    if (!c || !kh || !kh_bytes || !hp || !hp_bytes) return fail(PF_ERR_ARG, "pf_render_device: null argument");
    const bool want_hp = !(flags & PF_RENDER_NO_PATTERN_ROWS);
    if (!c->have_batch) return fail(PF_ERR_STATE, "pf_render_device needs a successful pf_submit");
    if (c->o.multiple_files) return fail(PF_ERR_ARG, "pf_render_device writes one pair of texts per batch; use the host renderers under multiple_files");
    HIPCHK(hipSetDevice(c->device));
    const uint32_t C = c->n_clusters, W = c->W, KW = (uint32_t)c->KW, k = c->o.klength;
    if (C && !names) return fail(PF_ERR_ARG, "pf_render_device: cluster names missing");
    if (c->n_passes > pf::TEXT_MAX_ARENAS) return fail(PF_ERR_CAPACITY, "pf_render_device: too many passes (%u)", c->n_passes);
    hipStream_t st = c->stream;
    // ---- small per-cluster / per-pattern arrays to the host: counts and the first-seen order
    std::vector<uint64_t> koff(C);
    std::vector<uint32_t> kcnt(C);
    if (C) {
        HIPCHK(hipMemcpyAsync(koff.data(), c->cl_kmer_off.p, (size_t)C * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(kcnt.data(), c->cl_kmer_cnt.p, (size_t)C * 4, hipMemcpyDeviceToHost, st));
    }
    const uint32_t p0 = c->pid0, p1 = c->n_patterns, P = want_hp ? p1 - p0 : 0;
    std::vector<uint64_t> fs(P);
    std::vector<uint32_t> rlen(P);
    DevBuf d_rlen;
    if (P) {
        PFCHK(d_rlen.ensure((size_t)P * 4));
        hipLaunchKernelGGL(pf::hp_rowlen_kernel, dim3((P + 255) / 256), dim3(256), 0, st, c->pat_n.as<uint32_t>(),
                           c->o.consider_missing ? c->pat_nan.as<uint32_t>() : (const uint32_t*)nullptr, W, p0,
                           (const uint32_t*)nullptr, P, d_rlen.as<uint32_t>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(fs.data(), c->pt.first_seen + p0, (size_t)P * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(rlen.data(), d_rlen.p, (size_t)P * 4, hipMemcpyDeviceToHost, st));
    }
    PFCHK(ensure_b64_dev(c));
    HIPCHK(hipStreamSynchronize(st));
    d_rlen.release();
    // ---- kmers_to_hashes layout: rows per cluster, workgroups per cluster
    std::vector<uint64_t> text_off(C + 1, 0);
    std::vector<uint32_t> name_off(C + 1, 0), arena_of(C), blk_cluster, blk_row0;
    std::string blob;
    uint32_t rows_per_block = 256;
    for (uint32_t i = 0; i < C; i++) {
        const uint32_t L = (uint32_t)strlen(names[i]);
        blob.append(names[i], L);
        name_off[i + 1] = (uint32_t)blob.size();
        const uint64_t head = L + 2 + 24 + 1, rowlen = (uint64_t)L + 1 + k + 1 + 24 + 1;
        if (head + rowlen > pf::TEXT_TILE) return fail(PF_ERR_ARG, "cluster name too long for the text kernel (%u bytes)", L);
        rows_per_block = std::min<uint32_t>(rows_per_block, (uint32_t)((pf::TEXT_TILE - head) / rowlen));
        text_off[i + 1] = text_off[i] + head + (uint64_t)kcnt[i] * rowlen;
        const uint32_t a = c->cluster_arena[i];
        arena_of[i] = a;
        koff[i] = kcnt[i] ? koff[i] - c->arenas[a]->base : 0;
    }
    for (uint32_t i = 0; i < C; i++)
        for (uint32_t r = 0; r < kcnt[i] + 1; r += rows_per_block) { blk_cluster.push_back(i); blk_row0.push_back(r); }
    // ---- hashes_to_patterns layout: new patterns in first-seen order
    std::vector<uint32_t> order(P);
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return fs[x] < fs[y]; });
    std::vector<uint64_t> row_off(P + 1, 0);
    for (uint32_t i = 0; i < P; i++) { row_off[i + 1] = row_off[i] + rlen[order[i]]; order[i] += p0; }
    const uint64_t kh_n = text_off[C], hp_n = row_off[P];
    const uint64_t hp_at = (kh_n + 255) & ~(uint64_t)255;          // the second text starts 256-byte aligned
    const uint64_t total = hp_at + hp_n + 16;
    PFCHK(c->txt_dev.ensure(total));
    PFCHK(text_pin(c, total));
    char* txt_pin = c->txt_pins[c->txt_slot];
    // ---- one block of tables for both kernels
    auto pad8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
    size_t o = 0;
    const size_t o_text = o; o += pad8((size_t)(C + 1) * 8);
    const size_t o_koff = o; o += pad8((size_t)C * 8);
    const size_t o_rowoff = o; o += pad8((size_t)(P + 1) * 8);
    const size_t o_name = o; o += pad8((size_t)(C + 1) * 4);
    const size_t o_kcnt = o; o += pad8((size_t)C * 4);
    const size_t o_arena = o; o += pad8((size_t)C * 4);
    const size_t o_bc = o; o += pad8(blk_cluster.size() * 4);
    const size_t o_br = o; o += pad8(blk_row0.size() * 4);
    const size_t o_order = o; o += pad8((size_t)P * 4);
    const size_t o_blob = o; o += pad8(blob.size() + 1);
    const size_t o_extra = o; o += pad8((size_t)n_extra * k + 1);
    std::vector<char> meta(o, 0);
    memcpy(&meta[o_text], text_off.data(), (size_t)(C + 1) * 8);
    if (C) memcpy(&meta[o_koff], koff.data(), (size_t)C * 8);
    memcpy(&meta[o_rowoff], row_off.data(), (size_t)(P + 1) * 8);
    memcpy(&meta[o_name], name_off.data(), (size_t)(C + 1) * 4);
    if (C) { memcpy(&meta[o_kcnt], kcnt.data(), (size_t)C * 4); memcpy(&meta[o_arena], arena_of.data(), (size_t)C * 4); }
    if (!blk_cluster.empty()) { memcpy(&meta[o_bc], blk_cluster.data(), blk_cluster.size() * 4); memcpy(&meta[o_br], blk_row0.data(), blk_row0.size() * 4); }
    if (P) memcpy(&meta[o_order], order.data(), (size_t)P * 4);
    if (!blob.empty()) memcpy(&meta[o_blob], blob.data(), blob.size());
    if (n_extra && extra_keys) memcpy(&meta[o_extra], extra_keys, (size_t)n_extra * k);
    PFCHK(c->txt_meta.ensure(o));
    HIPCHK(hipMemcpyAsync(c->txt_meta.p, meta.data(), o, hipMemcpyHostToDevice, st));
    const char* dm = c->txt_meta.as<char>();
    if (!blk_cluster.empty()) {
        pf::KhTextParams kp{};
        kp.text_off = (const uint64_t*)(dm + o_text); kp.name_off = (const uint32_t*)(dm + o_name); kp.names = dm + o_blob;
        kp.kmer_off = (const uint64_t*)(dm + o_koff); kp.kmer_cnt = (const uint32_t*)(dm + o_kcnt);
        kp.cluster_pattern = c->cl_pattern.as<uint32_t>(); kp.cluster_arena = (const uint32_t*)(dm + o_arena);
        kp.block_cluster = (const uint32_t*)(dm + o_bc); kp.block_row0 = (const uint32_t*)(dm + o_br);
        for (size_t a = 0; a < c->n_passes; a++) { kp.arena_key[a] = c->arenas[a]->key.as<uint64_t>(); kp.arena_pid[a] = c->arenas[a]->pid.as<uint32_t>(); }
        kp.b64 = c->pat_b64.as<char>(); kp.extra_keys = dm + o_extra; kp.text = c->txt_dev.as<char>();
        kp.k = k; kp.KW = KW; kp.rows_per_block = rows_per_block;
        hipLaunchKernelGGL(pf::kh_text_kernel, dim3((uint32_t)blk_cluster.size()), dim3(256), 0, st, kp);
        HIPCHK(hipGetLastError());
    }
    if (P) {
        pf::HpTextParams hpp{};
        hpp.order = (const uint32_t*)(dm + o_order); hpp.row_off = (const uint64_t*)(dm + o_rowoff);
        hpp.pat_bits = c->pat_bits.as<uint32_t>();
        hpp.pat_nan = c->o.consider_missing ? c->pat_nan.as<uint32_t>() : nullptr;
        hpp.pat_n = c->pat_n.as<uint32_t>(); hpp.b64 = c->pat_b64.as<char>();
        hpp.text = c->txt_dev.as<char>() + hp_at; hpp.n = P; hpp.W = W;
        hipLaunchKernelGGL(pf::hp_text_kernel, dim3(P), dim3(256), 0, st, hpp);
        HIPCHK(hipGetLastError());
    }
    if (kh_n) HIPCHK(hipMemcpyAsync(txt_pin, c->txt_dev.p, kh_n, hipMemcpyDeviceToHost, st));
    if (hp_n) HIPCHK(hipMemcpyAsync(txt_pin + hp_at, c->txt_dev.as<char>() + hp_at, hp_n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *kh = txt_pin; *kh_bytes = kh_n;
    *hp = txt_pin + hp_at; *hp_bytes = hp_n;
    return PF_OK;
}

#ifdef PF_PROF
/* profiling builds only (not part of the ABI): cycles per kernel phase, see PF_PROF_STAMP in pf_kernels.h */
int pf_debug_prof(uint64_t* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(pf::pf_prof), sizeof(uint64_t) * 64) != hipSuccess) return PF_ERR_HIP;
    if (reset) {
        static const uint64_t zero[64] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(pf::pf_prof), zero, sizeof zero) != hipSuccess) return PF_ERR_HIP;
    }
    return PF_OK;
}
#endif

}  // extern "C"
