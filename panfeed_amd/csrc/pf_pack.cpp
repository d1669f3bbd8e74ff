// pf_pack.cpp -- host side of the device boundary in native code: reference-shaped sequences -> the packed
// segment arrays of include/panfeed_hip.h, plus the slow path for windows that contain a non-ACGT base
// (grouped with the reference's own string semantics, /root/reference/panfeed/panfeed.py:64-88).
// No GPU involved; panfeed_amd/packing.py holds the same logic in numpy and tests compare the two array for array.
#include "pf_host.h"
#include <sched.h>
#include <cstdio>
#include <mutex>
#include "../../include/panfeed_hip.h"

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <immintrin.h>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

extern "C" const char* pf_last_error(void);
extern "C" void pf_set_error_(const char* msg);   // pf_api.hip

namespace {

int pk_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    pf_set_error_(buf);
    return code;
}

struct Extra { uint32_t ord; std::vector<uint32_t> bits; std::string key; };

struct ClusterOut {
    std::vector<uint64_t> words;                 // packed segments of the cluster, in final (sorted) order
    std::vector<uint32_t> seg_woff, seg_len, seg_sample, seg_ord, seg_strand_nw;   // woff: words into the cluster's
                                                 // share of the device buffer
    std::vector<uint32_t> seg_lit;               // literal segments: words into `words`; by reference: 0xFFFFFFFF
    std::vector<uint64_t> seg_src_off;           // by reference: the range in the resident genomes
    std::vector<uint32_t> seg_src_start, seg_src_flags;
    uint32_t dev_words = 0;
    std::vector<uint8_t> seg_wants_strand;
    std::vector<Extra> extras;
    // per target sequence (iteration order)
    std::vector<uint32_t> t_seq, t_seg_off, t_seg_local, t_seg_start, t_seg_nwin, t_amb_off, t_amb_pos;
    std::vector<int8_t> t_amb_used;
    std::string t_amb_keys;
    uint64_t n_inst = 0;
    std::string error;
};

struct CodeLut {
    int8_t t[256];
    // for the branch-free check of a whole sequence: u[seq byte] ^ v[comp byte] is 0 exactly when the first is A/C/G/T
    // and the second its complement
    uint8_t u[256], v[256];
    CodeLut() {
        for (int i = 0; i < 256; i++) { t[i] = -1; u[i] = 4; v[i] = 8; }
        t['A'] = 0; t['C'] = 1; t['G'] = 2; t['T'] = 3;
        u['A'] = 0; u['C'] = 1; u['G'] = 2; u['T'] = 3;
        v['T'] = 0; v['G'] = 1; v['C'] = 2; v['A'] = 3;
    }
};
const CodeLut g_lut;
inline int code_of(unsigned char ch) { return g_lut.t[ch]; }   // table lookup: no data-dependent branches

// The two per-base passes, 32 bases at a time where the host has AVX2 (a lookup per base through a serial shift-or chain
// was ~3 cycles a base, and the packer's threads are what `Seqinfo` records -> files waits for).
static const bool g_avx2 = __builtin_cpu_supports("avx2");
// every base of seq[0, n) one of A/C/G/T and comp its complement?  (n a multiple of 32)
__attribute__((target("avx2"))) static bool clean32_avx2(const char* seq, const char* comp, uint32_t n) {
    const __m256i A = _mm256_set1_epi8('A'), C = _mm256_set1_epi8('C'), G = _mm256_set1_epi8('G'), T = _mm256_set1_epi8('T');
    __m256i all = _mm256_set1_epi8((char)0xFF);
    for (uint32_t i = 0; i < n; i += 32) {
        const __m256i s = _mm256_loadu_si256((const __m256i*)(seq + i)), c = _mm256_loadu_si256((const __m256i*)(comp + i));
        const __m256i ok = _mm256_or_si256(
            _mm256_or_si256(_mm256_and_si256(_mm256_cmpeq_epi8(s, A), _mm256_cmpeq_epi8(c, T)),
                            _mm256_and_si256(_mm256_cmpeq_epi8(s, C), _mm256_cmpeq_epi8(c, G))),
            _mm256_or_si256(_mm256_and_si256(_mm256_cmpeq_epi8(s, G), _mm256_cmpeq_epi8(c, C)),
                            _mm256_and_si256(_mm256_cmpeq_epi8(s, T), _mm256_cmpeq_epi8(c, A))));
        all = _mm256_and_si256(all, ok);
    }
    return (uint32_t)_mm256_movemask_epi8(all) == 0xFFFFFFFFu;
}
// 32 bases (A/C/G/T only) -> one word, first base in bits 63:62, A0 C1 G2 T3
__attribute__((target("avx2"))) static inline uint64_t pack32_avx2(const char* p) {
    const __m256i c = _mm256_loadu_si256((const __m256i*)p);
    const __m256i x = _mm256_and_si256(_mm256_srli_epi16(c, 1), _mm256_set1_epi8(3));             // A0 C1 T2 G3
    const __m256i code = _mm256_xor_si256(x, _mm256_and_si256(_mm256_srli_epi16(x, 1), _mm256_set1_epi8(1)));   // G <-> T
    const __m256i p16 = _mm256_maddubs_epi16(code, _mm256_set1_epi16(0x0104));     // two bases: first * 4 + second
    const __m256i p32 = _mm256_madd_epi16(p16, _mm256_set1_epi32(0x00010010));     // four bases in the low byte of a dword
    const __m256i w16 = _mm256_packus_epi32(p32, p32);
    const __m256i w8 = _mm256_packus_epi16(w16, w16);                              // bytes 0-3 of each half: its four dwords
    const uint32_t lo = (uint32_t)_mm256_extract_epi32(w8, 0), hi = (uint32_t)_mm256_extract_epi32(w8, 4);
    return ((uint64_t)__builtin_bswap32(lo) << 32) | __builtin_bswap32(hi);
}
__attribute__((target("avx2"))) static void pack_words_avx2(const char* seq, uint32_t nwords, uint64_t* wp) {
    for (uint32_t w = 0; w < nwords; w++) wp[w] = pack32_avx2(seq + 32 * (size_t)w);
}

void pack_cluster(const pf_pack_in* in, uint32_t ci, ClusterOut& o) {
    const uint32_t k = in->klength, W = in->W;
    const uint32_t s0 = in->cluster_seq_off[ci], s1 = in->cluster_seq_off[ci + 1];
    struct Seg { uint32_t col, order, a, len, ord, seq; bool strand; };
    std::vector<Seg> segs;
    std::unordered_map<std::string, uint32_t> amb_index;      // key -> index in o.extras (insertion order kept)
    uint64_t ord_base = 0;
    uint32_t order = 0;
    std::string rev(k, 'A');
    {   // room for the common case (every sequence one segment) in one go: growing the word vector by doubling is a dozen
        // mmap / page-fault / munmap rounds per cluster, and those take a process-wide lock -- the packer's threads then
        // run one after the other
        uint64_t nw = 0;
        for (uint32_t q = s0; q < s1; q++) nw += 2 * (((uint64_t)in->seq_len[q] + 63) / 64);
        o.words.reserve((size_t)nw + 16);
        const size_t ns = (size_t)(s1 - s0) + 16;
        segs.reserve(ns);
        o.seg_woff.reserve(ns); o.seg_len.reserve(ns); o.seg_sample.reserve(ns); o.seg_ord.reserve(ns);
        o.seg_wants_strand.reserve(ns); o.seg_strand_nw.reserve(ns); o.seg_lit.reserve(ns);
        o.seg_src_off.reserve(ns); o.seg_src_start.reserve(ns); o.seg_src_flags.reserve(ns);
    }
    for (uint32_t q = s0; q < s1; q++) {
        const char* seq = in->seq[q];
        const char* comp = in->comp[q];
        const uint32_t L = in->seq_len[q];
        const uint32_t col = in->seq_col[q];
        const bool target = in->seq_target && in->seq_target[q];
        const uint64_t num_kmer = L >= k ? (uint64_t)L - k + 1 : 0;              // panfeed.py:59,64
        const bool byref = in->seq_flags && (in->seq_flags[q] & 1u);
        if (byref && target) { o.error = "a target strain's sequence cannot be given by reference"; return; }
        if (!byref && (!seq || !comp)) { o.error = "sequence text missing"; return; }
        std::vector<uint32_t> bad;
        if (!byref) {
            // nearly every sequence is pure A/C/G/T with the right complement: one branch-free pass says so; only the
            // others are looked at base by base (where the non-ACGT letters are, whether the complement is wrong)
            unsigned acc = 0;
            uint32_t i = 0;
            if (g_avx2 && L >= 32) {
                i = L & ~31u;
                acc = clean32_avx2(seq, comp, i) ? 0u : 1u;
            }
            for (; i < L; i++) acc |= (unsigned)(g_lut.u[(unsigned char)seq[i]] ^ g_lut.v[(unsigned char)comp[i]]);
            if (acc) {
                for (uint32_t i = 0; i < L; i++) {
                    const int cs = code_of((unsigned char)seq[i]);
                    if (cs < 0) bad.push_back(i);
                    else if (code_of((unsigned char)comp[i]) != 3 - cs) {
                        o.error = "compsequence is not the complement of sequence";
                        return;
                    }
                }
            }
        }
        if (target) {
            o.t_seq.push_back(q);
            o.t_seg_off.push_back((uint32_t)o.t_seg_local.size());
            o.t_amb_off.push_back((uint32_t)o.t_amb_pos.size());
        }
        // maximal A/C/G/T runs -> device segments
        uint32_t a = 0;
        for (size_t bi = 0; bi <= bad.size(); bi++) {
            const uint32_t b = bi < bad.size() ? bad[bi] : L;
            if (b >= a && b - a >= k) {
                const bool strand = target && in->canon && in->want_strand;
                segs.push_back(Seg{col, order, a, b - a, (uint32_t)(ord_base + a), q, strand});
                if (target) {
                    o.t_seg_local.push_back(order);
                    o.t_seg_start.push_back(a);
                    o.t_seg_nwin.push_back(b - a - k + 1);
                }
                order++;
            }
            a = b + 1;
        }
        // windows touching a non-ACGT base: the reference's own string semantics
        if (!bad.empty() && num_kmer > 0) {
            std::vector<uint8_t> touched(num_kmer, 0);
            for (uint32_t p : bad) {
                const uint64_t lo = p + 1 >= k ? (uint64_t)p - k + 1 : 0, hi = std::min<uint64_t>(num_kmer, (uint64_t)p + 1);
                for (uint64_t x = lo; x < hi; x++) touched[x] = 1;
            }
            for (uint64_t pos = 0; pos < num_kmer; pos++) {
                if (!touched[pos]) continue;
                const char* spec = seq + pos;                                   // panfeed.py:65
                for (uint32_t j = 0; j < k; j++) rev[j] = comp[pos + k - 1 - j]; // panfeed.py:67
                auto add = [&](const char* key, uint32_t ord) {
                    std::string ks(key, k);
                    auto it = amb_index.find(ks);
                    uint32_t e;
                    if (it == amb_index.end()) {
                        e = (uint32_t)o.extras.size();
                        amb_index.emplace(ks, e);
                        o.extras.push_back(Extra{ord, std::vector<uint32_t>(W, 0), ks});
                    } else e = it->second;
                    o.extras[e].bits[col >> 5] |= 1u << (col & 31);
                };
                if (in->canon) {
                    const bool fwd = memcmp(spec, rev.data(), k) <= 0;          // panfeed.py:70-75
                    add(fwd ? spec : rev.data(), (uint32_t)(ord_base + pos));
                    if (target) {
                        o.t_amb_pos.push_back((uint32_t)pos);
                        o.t_amb_used.push_back(fwd ? 1 : -1);
                        o.t_amb_keys.append(fwd ? spec : rev.data(), k);
                    }
                } else {                                                        // panfeed.py:82-88
                    add(spec, (uint32_t)(2 * (ord_base + pos)));
                    add(rev.data(), (uint32_t)(2 * (ord_base + pos) + 1));
                }
            }
        }
        ord_base += num_kmer;
    }
    o.n_inst = ord_base * (in->canon ? 1 : 2);
    if (ord_base * 2 >= 0xFFFFFFF0ull) { o.error = "too many k-mer instances for 32-bit ordinals"; return; }
    // segments sorted by sample column (stable); local order -> final position for the strand bits
    std::stable_sort(segs.begin(), segs.end(), [](const Seg& x, const Seg& y) { return x.col < y.col; });
    std::vector<uint32_t> where(segs.size());
    for (size_t j = 0; j < segs.size(); j++) {
        const Seg& g = segs[j];
        where[g.order] = (uint32_t)j;
        const uint32_t nw = 2 * ((g.len + 63) / 64);
        const uint32_t w0 = (uint32_t)o.words.size();
        const bool byref = in->seq_flags && (in->seq_flags[g.seq] & 1u);
        o.seg_woff.push_back(o.dev_words);
        o.dev_words += nw;
        o.seg_len.push_back(g.len);
        o.seg_sample.push_back(g.col);
        o.seg_ord.push_back(g.ord);
        o.seg_wants_strand.push_back(g.strand ? 1 : 0);
        o.seg_strand_nw.push_back(g.strand ? (g.len - k + 1 + 63) / 64 : 0);
        if (byref) {                 // the whole sequence is one segment (pure A/C/G/T): gathered on the device
            o.seg_lit.push_back(0xFFFFFFFFu);
            o.seg_src_off.push_back(in->seq_src_off[g.seq]);
            o.seg_src_start.push_back(in->seq_src_start[g.seq]);
            o.seg_src_flags.push_back(in->seq_flags[g.seq] & 2u);
            continue;
        }
        o.seg_lit.push_back(w0);
        o.seg_src_off.push_back(0); o.seg_src_start.push_back(0); o.seg_src_flags.push_back(1u);
        o.words.resize(w0 + nw, 0);
        const char* seq = in->seq[g.seq] + g.a;
        uint64_t* wp = o.words.data() + w0;
        uint32_t i0 = 0;
        if (g_avx2) {
            i0 = g.len & ~31u;
            pack_words_avx2(seq, i0 >> 5, wp);
        }
        for (; i0 + 32 <= g.len; i0 += 32) {                 // whole words: 32 lookups, no bounds inside
            uint64_t w = 0;
#pragma GCC unroll 8
            for (uint32_t i = 0; i < 32; i++) w = (w << 2) | (uint64_t)g_lut.u[(unsigned char)seq[i0 + i]];
            wp[i0 >> 5] = w;
        }
        if (i0 < g.len) {
            uint64_t w = 0;
            const uint32_t m = g.len - i0;
            for (uint32_t i = 0; i < m; i++) w = (w << 2) | (uint64_t)g_lut.u[(unsigned char)seq[i0 + i]];
            wp[i0 >> 5] = w << (2 * (32 - m));
        }
    }
    for (auto& x : o.t_seg_local) x = where[x];
}

}  // namespace

struct pf_packed {
    std::vector<uint64_t> packed;
    std::vector<uint64_t> seg_word_off;
    std::vector<uint32_t> seg_len, seg_sample, seg_ord_base, seg_strand_off, cluster_seg_off;
    std::vector<uint32_t> extra_cluster, extra_ord, extra_bits;
    std::string extra_keys;                       // klength bytes each
    std::vector<uint32_t> t_seq, t_seg_off, t_seg_index, t_seg_start, t_seg_nwin, t_amb_off, t_amb_pos;
    std::vector<int8_t> t_amb_used;
    std::string t_amb_keys;
    std::vector<uint64_t> cluster_ninst;
    uint64_t n_strand_words = 0, n_instances = 0;
    bool gather = false;
    uint64_t n_words_dev = 0;
    std::vector<uint64_t> g_src_off;
    std::vector<uint32_t> g_src_start, g_src_flags;
};

extern "C" {

int pf_pack_records(const pf_pack_in* in, pf_packed** out) {
    if (!in || !out) return pk_fail(PF_ERR_ARG, "pf_pack_records: null argument");
    *out = nullptr;
    if (in->klength < 1) return pk_fail(PF_ERR_ARG, "pf_pack_records: klength must be >= 1");
    const uint32_t C = in->n_clusters;
    std::vector<ClusterOut> outs(C);
    const auto T_0 = std::chrono::steady_clock::now();
    unsigned nt = pf_host_threads(32u);
    if (C < 8) nt = 1;
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t] { for (uint32_t ci = t; ci < C; ci += nt) pack_cluster(in, ci, outs[ci]); });
        for (auto& x : th) x.join();
    }
    const auto T_par = std::chrono::steady_clock::now();

    for (uint32_t ci = 0; ci < C; ci++)
        if (!outs[ci].error.empty()) return pk_fail(PF_ERR_ARG, "cluster %u: %s", ci, outs[ci].error.c_str());
    pf_packed* p = new pf_packed();
    p->cluster_seg_off.assign(1, 0);
    p->gather = in->seq_flags != nullptr;
    if (p->gather && (!in->seq_src_off || !in->seq_src_start)) { delete p; return pk_fail(PF_ERR_ARG, "pf_pack_records: seq_src arrays missing"); }
    uint64_t woff = 0, strand_words = 0, lit_off = 0;
    {   // the merged arrays' sizes are known: no growth by doubling while they are filled
        size_t nseg = 0, nword = 4, nex = 0;
        for (uint32_t ci = 0; ci < C; ci++) { nseg += outs[ci].seg_len.size(); nword += outs[ci].words.size(); nex += outs[ci].extras.size(); }
        p->seg_word_off.reserve(nseg); p->seg_len.reserve(nseg); p->seg_sample.reserve(nseg); p->seg_ord_base.reserve(nseg);
        p->seg_strand_off.reserve(nseg); p->cluster_seg_off.reserve((size_t)C + 1); p->packed.reserve(nword);
        p->cluster_ninst.reserve(C);
        p->extra_cluster.reserve(nex); p->extra_ord.reserve(nex); p->extra_bits.reserve(nex * in->W);
        if (p->gather) { p->g_src_off.reserve(nseg); p->g_src_start.reserve(nseg); p->g_src_flags.reserve(nseg); }
    }
    for (uint32_t ci = 0; ci < C; ci++) {
        ClusterOut& o = outs[ci];
        const uint32_t seg_base = (uint32_t)p->seg_len.size();
        for (size_t j = 0; j < o.seg_len.size(); j++) {
            p->seg_word_off.push_back(woff + o.seg_woff[j]);
            if (p->gather) {
                p->g_src_off.push_back(o.seg_lit[j] != 0xFFFFFFFFu ? lit_off + o.seg_lit[j] : o.seg_src_off[j]);
                p->g_src_start.push_back(o.seg_src_start[j]);
                p->g_src_flags.push_back(o.seg_src_flags[j]);
            }
            p->seg_len.push_back(o.seg_len[j]);
            p->seg_sample.push_back(o.seg_sample[j]);
            p->seg_ord_base.push_back(o.seg_ord[j]);
        }
        // strand-bit offsets follow the order in which the sequences were visited (as packing.py assigns them):
        // walk the target sequences' segments in iteration order
        std::vector<uint32_t> soff(o.seg_len.size(), 0xFFFFFFFFu);
        for (size_t t = 0; t < o.t_seq.size(); t++) {
            const uint32_t a = o.t_seg_off[t], b = t + 1 < o.t_seq.size() ? o.t_seg_off[t + 1] : (uint32_t)o.t_seg_local.size();
            for (uint32_t j = a; j < b; j++) {
                const uint32_t sj = o.t_seg_local[j];
                if (o.seg_wants_strand[sj]) { soff[sj] = (uint32_t)strand_words; strand_words += o.seg_strand_nw[sj]; }
            }
        }
        p->seg_strand_off.insert(p->seg_strand_off.end(), soff.begin(), soff.end());
        p->packed.insert(p->packed.end(), o.words.begin(), o.words.end());
        woff += o.dev_words;
        lit_off += o.words.size();
        p->cluster_seg_off.push_back((uint32_t)p->seg_len.size());
        for (auto& e : o.extras) {
            p->extra_cluster.push_back(ci);
            p->extra_ord.push_back(e.ord);
            p->extra_bits.insert(p->extra_bits.end(), e.bits.begin(), e.bits.end());
            p->extra_keys += e.key;
        }
        const uint32_t tsb = (uint32_t)p->t_seg_index.size(), tab = (uint32_t)p->t_amb_pos.size();
        for (size_t t = 0; t < o.t_seq.size(); t++) {
            p->t_seq.push_back(o.t_seq[t]);
            p->t_seg_off.push_back(tsb + o.t_seg_off[t]);
            p->t_amb_off.push_back(tab + o.t_amb_off[t]);
        }
        for (size_t j = 0; j < o.t_seg_local.size(); j++) {
            p->t_seg_index.push_back(seg_base + o.t_seg_local[j]);
            p->t_seg_start.push_back(o.t_seg_start[j]);
            p->t_seg_nwin.push_back(o.t_seg_nwin[j]);
        }
        p->t_amb_pos.insert(p->t_amb_pos.end(), o.t_amb_pos.begin(), o.t_amb_pos.end());
        p->t_amb_used.insert(p->t_amb_used.end(), o.t_amb_used.begin(), o.t_amb_used.end());
        p->t_amb_keys += o.t_amb_keys;
        p->cluster_ninst.push_back(o.n_inst);
        p->n_instances += o.n_inst;
    }
    p->t_seg_off.push_back((uint32_t)p->t_seg_index.size());
    p->t_amb_off.push_back((uint32_t)p->t_amb_pos.size());
    for (int i = 0; i < 4; i++) p->packed.push_back(0);   // 32 bytes of tail padding (a window of k <= 126 bases
                                                          // reaches up to four words past its own)
    p->n_words_dev = p->gather ? woff + 4 : 0;
    p->n_strand_words = strand_words;
    if (getenv("PF_DEBUG_TIMING"))
        fprintf(stderr, "[pf_pack_records] clusters on %u threads %.4f s, merge %.4f s\n", nt, (T_par - T_0).count() / 1e9,
                (std::chrono::steady_clock::now() - T_par).count() / 1e9);
    *out = p;
    return PF_OK;
}

void pf_packed_free(pf_packed* p) { delete p; }

int pf_packed_view(const pf_packed* p, pf_packed_view_t* v) {
    if (!p || !v) return pk_fail(PF_ERR_ARG, "pf_packed_view: null argument");
    v->n_segs = (uint32_t)p->seg_len.size();
    v->n_words = p->packed.size();
    v->packed = p->packed.data();
    v->seg_word_off = p->seg_word_off.data();
    v->seg_len = p->seg_len.data();
    v->seg_sample = p->seg_sample.data();
    v->seg_ord_base = p->seg_ord_base.data();
    v->seg_strand_off = p->seg_strand_off.data();
    v->cluster_seg_off = p->cluster_seg_off.data();
    v->n_extra = (uint32_t)p->extra_ord.size();
    v->extra_cluster = p->extra_cluster.data();
    v->extra_ord = p->extra_ord.data();
    v->extra_bits = p->extra_bits.data();
    v->extra_keys = p->extra_keys.data();
    v->n_targets = (uint32_t)p->t_seq.size();
    v->target_seq = p->t_seq.data();
    v->target_seg_off = p->t_seg_off.data();
    v->target_seg_index = p->t_seg_index.data();
    v->target_seg_start = p->t_seg_start.data();
    v->target_seg_nwin = p->t_seg_nwin.data();
    v->target_ambig_off = p->t_amb_off.data();
    v->target_ambig_pos = p->t_amb_pos.data();
    v->target_ambig_used = p->t_amb_used.data();
    v->target_ambig_keys = p->t_amb_keys.data();
    v->cluster_ninst = p->cluster_ninst.data();
    v->n_strand_words = p->n_strand_words;
    v->n_instances = p->n_instances;
    v->n_words_dev = p->n_words_dev;
    v->gather_src_off = p->gather ? p->g_src_off.data() : nullptr;
    v->gather_src_start = p->gather ? p->g_src_start.data() : nullptr;
    v->gather_src_flags = p->gather ? p->g_src_flags.data() : nullptr;
    return PF_OK;
}

// Binding helper for CPython callers (panfeed_amd/packing.py): the UTF-8 address of every `str` of an object array, through
// the interpreter's own PyUnicode_AsUTF8 handed in as a function pointer (this library does not link against libpython).
// One C loop instead of one ctypes call per string; the caller holds the GIL (ctypes.PyDLL).  out[i] = 0 where the API
// returned NULL (not a str / not encodable): the caller falls back to copying.
int pf_py_str_addresses(void* const* objs, uint64_t n, const char* (*as_utf8)(void*), uint64_t* out) {
    if ((n && (!objs || !out)) || !as_utf8) return pk_fail(PF_ERR_ARG, "pf_py_str_addresses: null argument");
    for (uint64_t i = 0; i < n; i++) out[i] = (uint64_t)(uintptr_t)as_utf8(objs[i]);
    return PF_OK;
}

// Binding helper for CPython callers: two str attributes of every object of a Python list (Seqinfo.sequence and
// .compsequence, classes.py:11-18) -> the address of their UTF-8 bytes and their length, in one C loop through the
// interpreter's own API (function pointers: this library does not link against libpython; the caller holds the GIL).
// The attribute objects are kept referenced in `held` (2 per object) until pf_py_release gives them back: an attribute that
// is computed on access stays alive as long as its bytes are used.  flags[i]: bit 0 set = both are str of equal length, all
// ASCII (their UTF-8 bytes are their characters); anything else leaves the object to the caller's general path.
int pf_py_seqinfo_columns(void* list, uint64_t n, void* attr_seq, void* attr_comp, const pf_py_api* api,
                          uint64_t* a_seq, uint64_t* a_comp, uint32_t* len, uint8_t* flags, void** held) {
    if (!list || !attr_seq || !attr_comp || !api || (n && (!a_seq || !a_comp || !len || !flags || !held)))
        return pk_fail(PF_ERR_ARG, "pf_py_seqinfo_columns: null argument");
    for (uint64_t i = 0; i < n; i++) {
        a_seq[i] = a_comp[i] = 0; len[i] = 0; flags[i] = 0; held[2 * i] = held[2 * i + 1] = nullptr;
        void* o = api->list_get_item(list, (long long)i);
        if (!o) { api->err_clear(); continue; }
        void* s = api->get_attr(o, attr_seq);
        void* c = s ? api->get_attr(o, attr_comp) : nullptr;
        held[2 * i] = s; held[2 * i + 1] = c;
        if (!s || !c) { api->err_clear(); continue; }
        long long ns = 0, nc = 0;
        const char* ps = api->as_utf8_and_size(s, &ns);
        const char* pc = ps ? api->as_utf8_and_size(c, &nc) : nullptr;
        if (!ps || !pc) { api->err_clear(); continue; }
        const long long ls = api->get_length(s), lc = api->get_length(c);
        if (ls < 0 || lc < 0) { api->err_clear(); continue; }
        len[i] = (uint32_t)ls;
        if (ls == lc && ls == ns && lc == nc && ls <= 0xFFFFFFFFll) {
            a_seq[i] = (uint64_t)(uintptr_t)ps; a_comp[i] = (uint64_t)(uintptr_t)pc; flags[i] = 1;
        }
    }
    return PF_OK;
}
int pf_py_release(void** held, uint64_t n, const pf_py_api* api) {
    if (!api || (n && !held)) return pk_fail(PF_ERR_ARG, "pf_py_release: null argument");
    for (uint64_t i = 0; i < n; i++) if (held[i]) { api->dec_ref(held[i]); held[i] = nullptr; }
    return PF_OK;
}

}  // extern "C"

// ---- pf_host.h ------------------------------------------------------------------------------------------------------------
static unsigned usable_cpus() {
    unsigned n = std::thread::hardware_concurrency();
    if (!n) n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int a = CPU_COUNT(&set);
        if (a > 0) n = std::min<unsigned>(n, (unsigned)a);
    }
    // cgroup v2: "<quota|max> <period>"; v1: two files
    long long quota = -1, period = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm') quota = atoll(q);
        fclose(f);
    } else {
        if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lld", &quota) != 1) quota = -1; fclose(fq); }
        if (FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lld", &period) != 1) period = 0; fclose(fp); }
    }
    // twice the quota: the sections that use this are bound by memory latency as much as by instructions (the reader's
    // 8 M look-ups in per-genome hash maps: 0.50 s with 32 threads on a 16-CPU quota, 0.65 s with 16), and more threads
    // than CPUs keep more misses in flight; what the bound prevents is 32 threads on a 2-CPU quota
    if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, 2 * ((quota + period - 1) / period)));
    return std::max(1u, n);
}
unsigned pf_host_threads(unsigned cap) {
    static const unsigned n = usable_cpus();
    return std::max(1u, std::min(n, cap));
}
