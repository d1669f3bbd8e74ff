// pf_input.cpp -- native restatement of the serial producer in front of the hot path:
//   parse_gff            /root/reference/panfeed/input.py:274-332
//   iter_gene_clusters   /root/reference/panfeed/input.py:335-468
// plus what they read: the panaroo gene_presence_absence.csv (input.py:188-191), GFF3 files and FASTA
// (embedded after ##FASTA, input.py:103-108, or separate files).  Output: the Seqinfo records of a batch of
// clusters as flat arrays whose sequence pointers go straight into pf_pack_records -- no per-base work in Python.
//
// PARITY UNPINNED at the pyfaidx boundary: pyfaidx is not in /root/reference and not installed; what it would return
// is restated from its documented behaviour (record name = header up to the first whitespace, newline-free
// sequence, sequence_always_upper, Python slice clipping, `-seq` = reverse complement with the IUPAC table below).
#include "pf_host.h"
#include "../../include/panfeed_hip.h"
#include "pf_ingest.h"

#include <immintrin.h>
#include <algorithm>
#include <atomic>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

extern "C" void pf_set_error_(const char* msg);

namespace {

int in_fail(int code, const std::string& msg) { pf_set_error_(msg.c_str()); return code; }

struct Contig;
struct Feature { std::string id, chrom; long long start, end; int strand; const Contig* ctg = nullptr; };

struct Contig {
    std::string seq;                 // upper case; with pf_pangenome_open_device only where text is needed (has_text)
    std::vector<uint32_t> bad;       // positions of bytes other than A/C/G/T, ascending
    uint64_t word_off = 0;           // in the device genome store (pf_pangenome_set_store / the ingest sink)
    uint64_t len = 0;                // letters
    bool has_text = true;            // seq is the whole contig; false: only `spans` (what the reader will cut out of text)
    struct Span { uint64_t start; std::string text; };
    std::vector<Span> spans;         // ascending, disjoint
};

// contig[a:b] as text: from the whole contig, or from the span that was kept for it
inline bool contig_text(const Contig& c, long long a, long long b, std::string& out) {
    if (c.has_text) { out = c.seq.substr((size_t)a, (size_t)(b - a)); return true; }
    if (b <= a) { out.clear(); return true; }
    auto it = std::upper_bound(c.spans.begin(), c.spans.end(), (uint64_t)a, [](uint64_t v, const Contig::Span& s) { return v < s.start; });
    if (it == c.spans.begin()) return false;
    --it;
    if ((uint64_t)b > it->start + it->text.size()) return false;
    out = it->text.substr((size_t)((uint64_t)a - it->start), (size_t)(b - a));
    return true;
}

// contig[a:b] iter_gene_clusters cuts for a feature (input.py:413-446), with Python's slice clipping: the half-open range of
// letters, and the coordinates the Seqinfo reports
struct SliceOf { long long a, b, seq_start, seq_end, offset; };
inline SliceOf slice_of(const Feature& f, long long up, long long down, bool dsc, long long n) {
    SliceOf r;
    const long long offset = (f.strand > 0 && f.start - 1 - up < 0) ? f.start - 1 : up;        // :415-418
    const long long offset_d = (f.strand < 0 && f.start - 1 - down < 0) ? f.start - 1 : down;  // :421-424
    long long a, b;
    if (!dsc) {                                   // :427-436
        if (f.strand > 0) { a = f.start - 1 - offset; b = f.end + offset_d; r.seq_start = f.start - offset; r.seq_end = f.end + offset_d; }
        else { a = f.start - 1 - offset_d; b = f.end + offset; r.seq_start = f.start - offset_d; r.seq_end = f.end + offset; }
    } else {                                         // :437-446
        if (f.strand > 0) { a = f.start - 1 - offset; b = f.start + offset_d; r.seq_start = f.start - offset; r.seq_end = f.start + offset_d; }
        else { a = f.end - 1 - offset_d; b = f.end + offset; r.seq_start = f.end - offset_d; r.seq_end = f.end + offset; }
    }
    // Python slice clipping of contig[a:b]
    if (a < 0) a = std::max(0LL, a + n);
    if (b < 0) b = std::max(0LL, b + n);
    a = std::min(a, n); b = std::min(b, n);
    if (b < a) b = a;
    r.a = a; r.b = b; r.offset = offset;
    return r;
}

struct Genome {
    std::unordered_map<std::string, Contig> contigs;            // name -> sequence
    std::unordered_map<std::string, Feature> features;          // ID -> feature (later lines overwrite, as a dict)
    std::string warnings;
    std::string error;
};

// the whole file in one string: one read() into a buffer of the file's size (a stream copy through a stringbuf was three
// passes over every genome)
bool read_file(const std::string& path, std::string& out) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    size_t want = 0;
    if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) want = (size_t)st.st_size;
    out.clear();
    size_t have = 0;
    for (;;) {
        if (out.size() - have < (1u << 16)) out.resize(std::max(out.size() * 2, std::max<size_t>(want + 1, 1u << 16)));
        const ssize_t r = ::read(fd, &out[have], out.size() - have);
        if (r < 0) { ::close(fd); return false; }
        if (r == 0) break;
        have += (size_t)r;
    }
    ::close(fd);
    out.resize(have);
    return true;
}

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

struct FastaLut {
    unsigned char up[256], bad[256];       // letter -> upper case (a-z only, as before); 1 for anything but A/C/G/T after that
    FastaLut() {
        for (int i = 0; i < 256; i++) {
            unsigned char c = (unsigned char)i;
            if (c >= 'a' && c <= 'z') c = (unsigned char)(c - 32);
            up[i] = c;
            bad[i] = (c != 'A' && c != 'C' && c != 'G' && c != 'T') ? 1 : 0;
        }
    }
};
const FastaLut g_fasta;

// FASTA text -> contigs (pyfaidx: name up to first whitespace, lines joined, upper case)
void parse_fasta(const char* p, const char* end, Genome& g) {
    Contig* cur = nullptr;
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', end - p);
        const char* le = nl ? nl : end;
        const char* e = le;
        while (e > p && (e[-1] == '\r')) e--;
        if (p < e && *p == '>') {
            const char* q = p + 1;
            while (q < e && !is_space(*q)) q++;
            std::string name(p + 1, q);
            auto ins = g.contigs.emplace(name, Contig());
            cur = ins.second ? &ins.first->second : nullptr;      // a repeated name keeps its first record
        } else if (cur) {
            // upper-cased copy through a table, 8 bytes at a time; the positions of letters other than A/C/G/T are
            // looked for only in a stretch that holds one (a genome is ~all A/C/G/T: this loop is what opening a
            // pangenome spends its time in)
            const size_t o = cur->seq.size(), n = (size_t)(e - p);
            cur->len = o + n;
            cur->seq.resize(o + n);
            char* dst = &cur->seq[o];
            const unsigned char* src = (const unsigned char*)p;
            size_t i = 0;
            for (; i + 8 <= n; i += 8) {
                unsigned bad = 0;
                for (int j = 0; j < 8; j++) { const unsigned char c = src[i + j]; dst[i + j] = (char)g_fasta.up[c]; bad |= g_fasta.bad[c]; }
                if (bad)
                    for (int j = 0; j < 8; j++) if (g_fasta.bad[src[i + j]]) cur->bad.push_back((uint32_t)(o + i + j));
            }
            for (; i < n; i++) {
                const unsigned char c = src[i];
                dst[i] = (char)g_fasta.up[c];
                if (g_fasta.bad[c]) cur->bad.push_back((uint32_t)(o + i));
            }
        }
        p = nl ? nl + 1 : end;
    }
}

// ---- one-pass ingest (pf_pangenome_open_device): where a record's letters lie, without copying one -----------------------
// every byte of [p, p + n) one of A C G T a c g t?
__attribute__((target("avx2"))) static bool all_acgt_avx2(const unsigned char* p, size_t n) {
    const __m256i lower = _mm256_set1_epi8(0x20), a = _mm256_set1_epi8('a'), c = _mm256_set1_epi8('c'), g = _mm256_set1_epi8('g'),
                  t = _mm256_set1_epi8('t');
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i x = _mm256_or_si256(_mm256_loadu_si256((const __m256i*)(p + i)), lower);
        const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(x, a), _mm256_cmpeq_epi8(x, c)),
                                           _mm256_or_si256(_mm256_cmpeq_epi8(x, g), _mm256_cmpeq_epi8(x, t)));
        if ((uint32_t)_mm256_movemask_epi8(ok) != 0xFFFFFFFFu) return false;
    }
    unsigned bad = 0;
    for (; i < n; i++) bad |= g_fasta.bad[p[i]];
    return !bad;
}
static bool all_acgt(const unsigned char* p, size_t n) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return all_acgt_avx2(p, n);
    unsigned bad = 0;
    for (size_t i = 0; i < n; i++) bad |= g_fasta.bad[p[i]];
    return !bad;
}

// One vector pass over a record's bytes: how many '\n' and '\r' it holds, and where the bytes lie that are neither a line
// end nor one of A C G T a c g t (offsets from p, ascending -- a genome has few).
__attribute__((target("avx2"))) static void scan_region_avx2(const unsigned char* p, size_t n, size_t* n_nl, size_t* n_cr,
                                                             std::vector<uint32_t>& other) {
    const __m256i lower = _mm256_set1_epi8(0x20), a = _mm256_set1_epi8('a'), c = _mm256_set1_epi8('c'), g = _mm256_set1_epi8('g'),
                  t = _mm256_set1_epi8('t'), nl = _mm256_set1_epi8('\n'), cr = _mm256_set1_epi8('\r');
    size_t i = 0, cnl = 0, ccr = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i v = _mm256_loadu_si256((const __m256i*)(p + i));
        const __m256i x = _mm256_or_si256(v, lower);
        const __m256i let = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(x, a), _mm256_cmpeq_epi8(x, c)),
                                            _mm256_or_si256(_mm256_cmpeq_epi8(x, g), _mm256_cmpeq_epi8(x, t)));
        const uint32_t m_nl = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, nl));
        const uint32_t m_cr = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, cr));
        uint32_t rest = ~((uint32_t)_mm256_movemask_epi8(let) | m_nl | m_cr);
        while (rest) { other.push_back((uint32_t)(i + (size_t)__builtin_ctz(rest))); rest &= rest - 1; }
        cnl += (size_t)__builtin_popcount(m_nl); ccr += (size_t)__builtin_popcount(m_cr);
    }
    for (; i < n; i++) {
        const unsigned char ch = p[i];
        if (ch == '\n') cnl++;
        else if (ch == '\r') ccr++;
        else if (g_fasta.bad[ch]) other.push_back((uint32_t)i);
    }
    *n_nl = cnl; *n_cr = ccr;
}
// The common record -- every line but the last `width` letters and one line end ("\n" or "\r\n") -- recognised in ONE vector
// pass over its bytes plus one look per line at where the line ends must be; `other` receives the LETTER positions (line
// ends taken out) of what is not A/C/G/T.  Anything else (lines of other lengths, blank lines) is left to the exact
// line-by-line walk.
static bool fast_uniform(const char* b, const char* e, uint64_t* nbases, uint32_t* width, uint32_t* eol, std::vector<uint32_t>& other) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    other.clear();
    if (!avx2 || b >= e || (size_t)(e - b) >= 0xFFFFFF00u) return false;
    const char* nl = (const char*)memchr(b, '\n', (size_t)(e - b));
    size_t n_nl = 0, n_cr = 0;
    scan_region_avx2((const unsigned char*)b, (size_t)(e - b), &n_nl, &n_cr, other);
    if (!nl) {                                          // one line, no line end
        if (n_cr) return false;
        *nbases = (uint64_t)(e - b); *width = (uint32_t)(e - b); *eol = 1;
        return true;
    }
    const size_t gap = (nl > b && nl[-1] == '\r') ? 2 : 1;
    const size_t w = (size_t)(nl - b) + 1 - gap;
    if (!w) return false;
    const size_t stride = w + gap, body = (size_t)(e - b), n_full = body / stride, rem = body - n_full * stride;
    // the line ends where they must be ...
    for (size_t k = 0; k < n_full; k++) {
        const char* q = b + k * stride + w;
        if (gap == 2 ? (q[0] != '\r' || q[1] != '\n') : q[0] != '\n') return false;
    }
    // ... and nowhere else: the last, shorter line may carry one line end of its own
    size_t last = rem, tail_nl = 0, tail_cr = 0;
    if (rem) {
        const char* q = b + n_full * stride;
        if (q[rem - 1] == '\n') { last--; tail_nl = 1; if (gap == 2) { if (last && q[last - 1] == '\r') { last--; tail_cr = 1; } else return false; } }
    }
    if (n_nl != n_full + tail_nl || n_cr != (gap == 2 ? n_full : 0) + tail_cr) return false;
    *nbases = (uint64_t)n_full * w + last; *width = (uint32_t)w; *eol = (uint32_t)gap;
    for (auto& o : other) o = (uint32_t)(o - (o / stride) * gap);          // byte offset -> letter position
    return true;
}

// The FASTA text [p, end) of one genome, lying in a block the sink gave (`block` = its first byte): contigs are named and
// measured as parse_fasta names and measures them (header up to the first whitespace; the lines joined, a line's trailing
// '\r's dropped; a repeated name keeps its first record), but their letters stay where they are -- a contig becomes one
// piece {first letter, letters, letters per line, bytes between lines}.  A record whose lines are not wrapped evenly (every
// line but the last of one length and one line end) is joined in place first.  Text is built, as parse_fasta builds it,
// only for contigs with a letter other than A/C/G/T (the reader cuts such sequences out of text) and, with want_text, for
// all of a target strain's.
bool scan_fasta(char* block, char* p, char* end, Genome& g, bool want_text, pf_ingest_sink* sink,
                std::vector<pf_ingest_piece>& pieces, long long up, long long down, bool dsc) {
    // the genome's features by contig: which stretches of a contig with other letters the reader will ask for as text
    std::unordered_map<std::string, std::vector<const Feature*>> by_chrom;
    for (auto& fv : g.features) by_chrom[fv.second.chrom].push_back(&fv.second);
    const std::string* cur_name = nullptr;
    std::vector<uint32_t> other;
    Contig* cur = nullptr;
    char* rec_begin = nullptr;            // first byte behind the current record's header line
    auto finish = [&](char* rec_end) -> bool {
        if (!cur) return true;
        // ---- the record's lines: letters, wrapping, other letters
        uint64_t nbases = 0;
        uint32_t width = 0, eol = 1;
        bool started = false, ragged = false, prev_full = true, blank_after = false, bad = false;
        char* first = rec_begin;
        const bool quick = fast_uniform(rec_begin, rec_end, &nbases, &width, &eol, other);
        if (quick && !other.empty() && !want_text) {
            // Other letters in an evenly wrapped record: their positions, and text ONLY for the slices the reader will cut
            // around them (a feature's contig[a:b] that holds one: it goes to the packer as text) -- not the whole contig.
            if (nbases >= 0xFFFFFF00ull) { g.error = "a contig is too long for 32-bit coordinates"; return false; }
            cur->len = nbases;
            cur->has_text = false;
            cur->bad = other;
            std::vector<std::pair<uint64_t, uint64_t>> need;
            auto fit = by_chrom.find(*cur_name);
            if (fit != by_chrom.end())
                for (const Feature* f : fit->second) {
                    const SliceOf so = slice_of(*f, up, down, dsc, (long long)nbases);
                    if (so.b <= so.a) continue;
                    auto it = std::lower_bound(cur->bad.begin(), cur->bad.end(), (uint32_t)so.a);
                    if (it != cur->bad.end() && (long long)*it < so.b) need.emplace_back((uint64_t)so.a, (uint64_t)so.b);
                }
            std::sort(need.begin(), need.end());
            const size_t stride = (size_t)width + eol;
            for (size_t i = 0; i < need.size();) {
                uint64_t s0 = need[i].first, s1 = need[i].second;
                size_t j = i + 1;
                while (j < need.size() && need[j].first <= s1) { s1 = std::max(s1, need[j].second); j++; }
                Contig::Span sp;
                sp.start = s0;
                sp.text.resize((size_t)(s1 - s0));
                for (uint64_t q = s0; q < s1;) {                   // line by line: letters [q, end of its line or s1)
                    const uint64_t line = q / width, col = q - line * width;
                    const uint64_t take = std::min<uint64_t>(s1 - q, width - col);
                    const unsigned char* src = (const unsigned char*)rec_begin + line * stride + col;
                    char* dst = &sp.text[(size_t)(q - s0)];
                    for (uint64_t z = 0; z < take; z++) dst[z] = (char)g_fasta.up[src[z]];
                    q += take;
                }
                cur->spans.push_back(std::move(sp));
                i = j;
            }
            const uint64_t nwords = 2 * ((nbases + 63) / 64) + 4;
            const uint64_t w0 = sink->claim_words(sink->self, nwords);
            if (w0 == UINT64_MAX) { g.error = "\x01genome store full"; return false; }
            cur->word_off = w0;
            pf_ingest_piece pc{};
            pc.text_off = (uint64_t)(rec_begin - block); pc.nbases = nbases; pc.dst_word = w0; pc.width = nbases ? width : 0; pc.eol = eol;
            pieces.push_back(pc);
            return true;
        }
        if (quick && !other.empty()) bad = true;
        for (char* q = rec_begin; !quick && q < rec_end;) {
            char* nl = (char*)memchr(q, '\n', rec_end - q);
            char* le = nl ? nl : rec_end;
            char* e = le;
            while (e > q && e[-1] == '\r') e--;
            const size_t L = (size_t)(e - q);
            const size_t gap = (size_t)((nl ? nl + 1 : rec_end) - e);
            if (L) {
                if (blank_after) ragged = true;
                if (!started) { started = true; first = q; width = (uint32_t)std::min<size_t>(L, 0xFFFFFFFFu); eol = (uint32_t)gap; prev_full = true; if (L > 0xFFFFFFF0u) ragged = true; }
                else if (!prev_full || L > width) ragged = true;
                prev_full = L == width && gap == eol;
                nbases += L;
                if (!bad && !all_acgt((const unsigned char*)q, L)) bad = true;
            } else if (started) {
                blank_after = true;
            }
            q = nl ? nl + 1 : rec_end;
        }
        cur->len = nbases;
        cur->has_text = want_text || bad;
        if (cur->has_text) {
            // (parse_fasta's loop over this record's lines: upper-cased text + the positions of the other letters)
            Genome tmp;
            std::string hdr = ">x\n";
            auto ins = tmp.contigs.emplace("x", Contig());
            Contig* c2 = &ins.first->second;
            (void)hdr;
            c2->seq.reserve((size_t)nbases);
            for (char* q = rec_begin; q < rec_end;) {
                char* nl = (char*)memchr(q, '\n', rec_end - q);
                char* e = nl ? nl : rec_end;
                while (e > q && e[-1] == '\r') e--;
                const size_t o = c2->seq.size(), n = (size_t)(e - q);
                c2->seq.resize(o + n);
                for (size_t i2 = 0; i2 < n; i2++) {
                    const unsigned char ch = (unsigned char)q[i2];
                    c2->seq[o + i2] = (char)g_fasta.up[ch];
                    if (g_fasta.bad[ch]) c2->bad.push_back((uint32_t)(o + i2));
                }
                q = nl ? nl + 1 : rec_end;
            }
            cur->seq.swap(c2->seq);
            cur->bad.swap(c2->bad);
        }
        if (nbases >= 0xFFFFFF00ull) { g.error = "a contig is too long for 32-bit coordinates"; return false; }
        if (ragged) {                     // join the lines in place: the letters then lie back to back from `first` on
            char* dst = first;
            for (char* q = first; q < rec_end;) {
                char* nl = (char*)memchr(q, '\n', rec_end - q);
                char* e = nl ? nl : rec_end;
                while (e > q && e[-1] == '\r') e--;
                const size_t L = (size_t)(e - q);
                if (L && dst != q) memmove(dst, q, L);
                dst += L;
                q = nl ? nl + 1 : rec_end;
            }
            width = 0;
        }
        if (!nbases) width = 0;
        const uint64_t nwords = 2 * ((nbases + 63) / 64) + 4;
        const uint64_t w0 = sink->claim_words(sink->self, nwords);
        if (w0 == UINT64_MAX) { g.error = "\x01genome store full"; return false; }
        cur->word_off = w0;
        pf_ingest_piece pc{};
        pc.text_off = (uint64_t)(first - block); pc.nbases = nbases; pc.dst_word = w0; pc.width = width; pc.eol = eol;
        pieces.push_back(pc);
        return true;
    };
    while (p < end) {
        char* nl = (char*)memchr(p, '\n', end - p);
        char* le = nl ? nl : end;
        if (p < le && *p == '>') {
            if (!finish(p)) return false;
            char* e = le;
            while (e > p && e[-1] == '\r') e--;
            char* q = p + 1;
            while (q < e && !is_space(*q)) q++;
            std::string name(p + 1, q);
            auto ins = g.contigs.emplace(name, Contig());
            cur = ins.second ? &ins.first->second : nullptr;      // a repeated name keeps its first record
            cur_name = &ins.first->first;
            rec_begin = nl ? nl + 1 : end;
            // the record runs to the next header line: the next '>' that stands at the start of a line
            p = rec_begin;
            while (p < end) {
                char* gt = (char*)memchr(p, '>', end - p);
                if (!gt) { p = end; break; }
                if (gt == rec_begin || gt[-1] == '\n') { p = gt; break; }
                p = gt + 1;
            }
            continue;
        }
        p = nl ? nl + 1 : end;            // (text in front of the first header: nobody's)
    }
    return finish(end);
}

// a whole file into `buf` (a thread's own buffer, kept from file to file: no allocation, no page faults after the first
// files): *n bytes
bool read_file_buf(const std::string& path, std::vector<char>& buf, size_t* n, std::string& err) {
    int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) { err = "cannot read " + path; return false; }
    struct stat st;
    size_t want = 1u << 16;
    if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) want = (size_t)st.st_size;
    if (buf.size() < want + 64) buf.resize(want + want / 4 + 4096);
    size_t have = 0;
    for (;;) {
        if (buf.size() - have < 64) buf.resize(buf.size() * 2);
        const ssize_t r = ::read(fd, buf.data() + have, buf.size() - have);
        if (r < 0) { ::close(fd); err = "cannot read " + path; return false; }
        if (r == 0) break;
        have += (size_t)r;
    }
    ::close(fd);
    *n = have;
    return true;
}

// Python int(str): optional surrounding whitespace, a sign, decimal digits with single underscores between them
// ("1_0" is 10; "_1", "1_", "1__0" are errors); other bases and non-ASCII digits are not handled -> error
bool py_int(const std::string& s, long long* out) {
    size_t a = 0, b = s.size();
    while (a < b && is_space(s[a])) a++;
    while (b > a && is_space(s[b - 1])) b--;
    if (a == b) return false;
    bool neg = false;
    if (s[a] == '+' || s[a] == '-') { neg = s[a] == '-'; a++; }
    if (a == b) return false;
    long long v = 0;
    bool prev_digit = false;
    for (size_t i = a; i < b; i++) {
        if (s[i] == '_') {
            if (!prev_digit || i + 1 == b) return false;
            prev_digit = false;
            continue;
        }
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (s[i] - '0');
        prev_digit = true;
    }
    *out = neg ? -v : v;
    return true;
}

// repr() of a str, as it appears in int()'s ValueError text: single quotes (double when the text holds ' and no "),
// backslash escapes for \ \t \n \r and the quote, \xNN for other control bytes
std::string py_repr(const std::string& s) {
    const bool dq = s.find('\'') != std::string::npos && s.find('"') == std::string::npos;
    const char q = dq ? '"' : '\'';
    std::string r(1, q);
    for (unsigned char c : s) {
        if (c == (unsigned char)q || c == '\\') { r += '\\'; r += (char)c; }
        else if (c == '\t') r += "\\t";
        else if (c == '\n') r += "\\n";
        else if (c == '\r') r += "\\r";
        else if (c < 0x20 || c == 0x7f) { char b[8]; snprintf(b, sizeof b, "\\x%02x", c); r += b; }
        else r += (char)c;
    }
    r += q;
    return r;
}

// input.py:274-332 (feature_types = {'CDS'}).  Fields are looked at where they lie in the file's text: a line becomes
// strings only where the reference's result needs one (the feature's ID and contig; a warning's text).
void parse_gff(const char* text, size_t text_bytes, const std::string& file_name, Genome& g) {
    const char* p = text;
    const char* end = p + text_bytes;
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', end - p);
        const char* le = nl ? nl + 1 : end;                      // the line keeps its newline, as in `for line in gff`
        const char* ls = p;
        p = le;
        const char* a = ls;
        while (a < le && is_space(*a)) a++;                       // line.lstrip()
        if (le - a >= 7 && memcmp(a, "##FASTA", 7) == 0) break;   // input.py:286-288
        if (a < le && *a == '#') continue;                        // input.py:290-292
        // line.split('\t'): the first nine fields' bounds, and how many there are
        const char* fb[9]; const char* fe[9];
        size_t nf = 0;
        const char* q = ls;
        for (;;) {
            const char* t = (const char*)memchr(q, '\t', le - q);
            if (nf < 9) { fb[nf] = q; fe[nf] = t ? t : le; }
            nf++;
            if (!t) break;
            q = t + 1;
            if (nf >= 9) {                                        // field 8 runs to the next tab or the line's end
                break;
            }
        }
        if (nf >= 9) { const char* t = (const char*)memchr(fb[8], '\t', le - fb[8]); fe[8] = t ? t : le; }
        auto warn = [&](const char* what) {
            std::string l(ls, le);
            while (!l.empty() && is_space(l.back())) l.pop_back();
            g.warnings += std::string(what) + ", skipping line \"" + l + "\" from " + file_name + "\n";
        };
        if (nf < 3) { warn("list index out of range"); continue; }
        if (!(fe[2] - fb[2] == 3 && memcmp(fb[2], "CDS", 3) == 0)) continue;      // input.py:300-301
        // the reference evaluates entries[3], int(), entries[4], int(), entries[6], entries[8] in that order
        // (input.py:303-316): the first of them to fail names the warning
        long long st, en;
        {
            bool bad = false;
            for (int f = 3; f <= 4 && !bad; f++) {
                if (nf <= (size_t)f) { warn("list index out of range"); bad = true; break; }
                const std::string fs(fb[f], fe[f]);
                if (!py_int(fs, f == 3 ? &st : &en)) {    // ValueError text of int(entries[3]) / int(entries[4]), input.py:304-305
                    warn(("invalid literal for int() with base 10: " + py_repr(fs)).c_str()); bad = true;
                }
            }
            if (bad) continue;
        }
        if (nf < 9) { warn("list index out of range"); continue; }      // entries[6], entries[8]
        const int strand = (fe[6] - fb[6] == 1 && *fb[6] == '+') ? 1 : -1;       // input.py:309-312
        bool have = false;
        const char* idb = nullptr; const char* ide = nullptr;
        const char* q0 = fb[8];
        for (;;) {                                                // entries[8].split(';')
            const char* t = (const char*)memchr(q0, ';', fe[8] - q0);
            const char* ee = t ? t : fe[8];
            if (ee - q0 >= 2 && q0[0] == 'I' && q0[1] == 'D') {   // entry.startswith('ID') and '=' in entry (input.py:317)
                const char* eq = (const char*)memchr(q0, '=', ee - q0);
                if (eq) {
                    const char* eq2 = (const char*)memchr(eq + 1, '=', ee - (eq + 1));
                    idb = eq + 1; ide = eq2 ? eq2 : ee;           // split('=')[1]
                    have = true;
                }
            }
            if (!t) break;
            q0 = t + 1;
        }
        if (!have) continue;                                      // input.py:321-322
        std::string id(idb, ide);
        g.features[id] = Feature{id, std::string(fb[0], fe[0]), st, en, strand};     // input.py:325
    }
}

const char* COMP_FROM = "ACTGNactgnYRWSKMDVHBXyrwskmdvhbx";
const char* COMP_TO = "TGACNtgacnRYWSMKHBDVXryswmkhbdvx";
struct CompLut {
    unsigned char t[256];
    CompLut() { for (int i = 0; i < 256; i++) t[i] = (unsigned char)i; for (int i = 0; COMP_FROM[i]; i++) t[(unsigned char)COMP_FROM[i]] = (unsigned char)COMP_TO[i]; }
};
const CompLut g_comp;

// pandas' default NA strings (read_csv): a cell equal to one of these is NaN
const char* NA_STRINGS[] = {"", "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN", "<NA>",
                            "N/A", "NA", "NULL", "NaN", "None", "n/a", "nan", "null"};

// RFC-4180-ish CSV record splitter (quotes, doubled quotes, embedded newlines)
bool next_csv_record(const std::string& t, size_t& pos, std::vector<std::string>& out) {
    out.clear();
    if (pos >= t.size()) return false;
    std::string cell;
    bool inq = false, any = false;
    for (;;) {
        if (pos >= t.size()) { out.push_back(cell); return any || !out.empty(); }
        char c = t[pos++];
        any = true;
        if (inq) {
            if (c == '"') { if (pos < t.size() && t[pos] == '"') { cell += '"'; pos++; } else inq = false; }
            else cell += c;
        } else if (c == '"') inq = true;
        else if (c == ',') { out.push_back(cell); cell.clear(); }
        else if (c == '\n') { out.push_back(cell); return true; }
        else if (c == '\r') { /* swallow */ }
        else cell += c;
    }
}

}  // namespace

struct pf_pangenome {
    std::vector<std::string> strains;          // CSV column order (genepres.columns)
    std::vector<std::string> sorted_strains;
    std::vector<uint32_t> sorted_pos;          // strains[i] -> index in sorted_strains
    std::vector<std::string> cluster_names;
    std::vector<std::vector<std::string>> cells;   // [row][strain] "" = NaN
    std::unordered_map<std::string, Genome> genomes;
    std::unordered_set<std::string> targets, genes;
    bool have_genes = false;
    long long up = 0, down = 0;
    bool dsc = false, raise_missing = false;
    size_t next_row = 0;
    size_t end_row = (size_t)-1;               // one past the last table row of this reader's range
    std::string log;
    uint32_t W = 0;
    // genomes resident on the device: flat contig order handed to pf_genomes_upload, by-reference mode
    std::vector<const Genome*> strain_genome;  // per table strain: its genome data or null (input.py:384-387)
    std::vector<uint8_t> strain_target;
    std::vector<Contig*> flat;
    std::vector<const char*> flat_ptr;
    std::vector<uint64_t> flat_len;
    bool by_ref = false;
};

struct pf_records {
    // storage
    std::vector<std::string> seq_store, comp_store;
    std::vector<const char*> seq, comp, ids, chroms, cluster_name_ptr, strain_ptr;
    std::vector<uint32_t> seq_len, seq_col, seq_strain, cluster_seq_off, cluster_nstrains, cluster_npresab, cluster_presab,
        cluster_row, cluster_strain_off, strain_index;
    std::vector<uint8_t> seq_target;
    std::vector<int32_t> seq_strand;
    std::vector<int64_t> seq_start, seq_end, seq_offset;
    std::vector<uint64_t> seq_src_off;
    std::vector<uint32_t> seq_src_start, seq_flags;
};

namespace {
// one row of the table -> its records (thread-safe: reads the pangenome only)
struct RowOut {
    std::vector<std::string> seq, comp;          // literal sequences only (seq_flags bit 0 clear), in order
    std::vector<const char*> ids, chroms;
    std::vector<uint32_t> seq_len, seq_col, seq_strain, presab, dict, seq_src_start, seq_flags;
    std::vector<uint64_t> seq_src_off;
    std::vector<uint8_t> seq_target;
    std::vector<int32_t> seq_strand;
    std::vector<int64_t> seq_start, seq_end, seq_offset;
    std::string log, error;
};

void build_row(const pf_pangenome* P, size_t row, RowOut& R) {
    const size_t S = P->strains.size();
    const std::string& idx = P->cluster_names[row];
    const auto& cells = P->cells[row];
    R.presab.assign(P->W, 0);
    // dict insertion order: present strains with genome data (CSV order), then absent strains (sorted)
    std::vector<uint32_t>& dict = R.dict;                            // strain indices
    for (size_t s = 0; s < S; s++)
        if (!cells[s].empty()) {
            const uint32_t sp = P->sorted_pos[s];
            R.presab[sp >> 5] |= 1u << (sp & 31);                    // input.py:375-377
            if (P->strain_genome[s]) dict.push_back((uint32_t)s);   // input.py:384-387
        }
    const size_t npresent_dict = dict.size();
    {
        std::vector<uint32_t> absent;
        for (size_t s = 0; s < S; s++) if (cells[s].empty()) absent.push_back((uint32_t)s);
        // input.py:373 `strains.difference(present)`: sorted by pandas -- unless `present` is empty (a row without any
        // gene), where Index.difference hands the index back as it is, in table order (tests/golden/n1: grp_none)
        if (absent.size() < S)
            std::sort(absent.begin(), absent.end(), [&](uint32_t a, uint32_t b) { return P->sorted_pos[a] < P->sorted_pos[b]; });
        dict.insert(dict.end(), absent.begin(), absent.end());      // input.py:465-466
    }
    // column of each dict strain in sorted(cluster.keys())  (panfeed.py:47-49)
    std::vector<uint32_t> order(dict.size());
    for (size_t i = 0; i < dict.size(); i++) order[i] = (uint32_t)i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return P->sorted_pos[dict[a]] < P->sorted_pos[dict[b]]; });
    std::vector<uint32_t> col(dict.size());
    for (size_t i = 0; i < order.size(); i++) col[order[i]] = (uint32_t)i;

    {
        const size_t est = npresent_dict + npresent_dict / 16 + 4;
        R.ids.reserve(est); R.chroms.reserve(est); R.seq_len.reserve(est); R.seq_col.reserve(est); R.seq_strain.reserve(est);
        R.seq_src_start.reserve(est); R.seq_flags.reserve(est); R.seq_src_off.reserve(est); R.seq_target.reserve(est);
        R.seq_strand.reserve(est); R.seq_start.reserve(est); R.seq_end.reserve(est); R.seq_offset.reserve(est);
    }
    std::string gene;
    for (size_t di = 0; di < npresent_dict; di++) {
        const std::string& strain = P->strains[dict[di]];
        const Genome& g = *P->strain_genome[dict[di]];
        const bool is_target = P->strain_target[dict[di]] != 0;
        const std::string& genes = cells[dict[di]];
        size_t q0 = 0;
        for (;;) {                                                   // genes.split(';')  input.py:393
            size_t t = genes.find(';', q0);
            gene.assign(genes, q0, t == std::string::npos ? std::string::npos : t - q0);
            auto fit = g.features.find(gene);
            if (fit == g.features.end()) {                           // input.py:396-402
                R.log += "Could not find gene " + gene + " from " + idx + " in " + strain + "\n";
                if (P->raise_missing) { R.error = "Could not find gene " + gene + " from " + idx + " in " + strain; return; }
            } else {
                const Feature& f = fit->second;
                if (!f.ctg) {                                        // input.py:404-411
                    R.log += "Could not find chromosome " + f.chrom + " in " + strain + "\n";
                    if (P->raise_missing) { R.error = "Could not find chromosome " + f.chrom + " in " + strain; return; }
                } else {
                    const Contig& ctg = *f.ctg;
                    const SliceOf so = slice_of(f, P->up, P->down, P->dsc, (long long)ctg.len);
                    const long long a = so.a, b = so.b, seq_start = so.seq_start, seq_end = so.seq_end, offset = so.offset;
                    // resident genomes: a pure-ACGT range of a non-target strain goes by reference
                    bool ref = P->by_ref && !is_target;
                    if (ref) {
                        auto it = std::lower_bound(ctg.bad.begin(), ctg.bad.end(), (uint32_t)a);
                        if (it != ctg.bad.end() && (long long)*it < b) ref = false;
                    }
                    R.seq_len.push_back((uint32_t)(b - a));
                    if (ref) {
                        R.seq_src_off.push_back(ctg.word_off); R.seq_src_start.push_back((uint32_t)a);
                        R.seq_flags.push_back(1u | (f.strand < 0 ? 2u : 0u));
                    } else {
                        std::string seq;
                        if (!contig_text(ctg, a, b, seq)) { R.error = "internal: text of contig " + f.chrom + " of " + strain + " was not kept"; return; }
                        if (f.strand < 0) {                          // -sequences[...]: reverse complement
                            std::reverse(seq.begin(), seq.end());
                            for (auto& ch : seq) ch = (char)g_comp.t[(unsigned char)ch];
                        }
                        std::string comp(seq.size(), 'N');           // revseq = (-seq)[::-1] : the complement (:449-452)
                        for (size_t i = 0; i < seq.size(); i++) comp[i] = (char)g_comp.t[(unsigned char)seq[i]];
                        R.seq.push_back(std::move(seq));
                        R.comp.push_back(std::move(comp));
                        R.seq_src_off.push_back(0); R.seq_src_start.push_back(0); R.seq_flags.push_back(0);
                    }
                    R.ids.push_back(f.id.c_str());
                    R.chroms.push_back(f.chrom.c_str());
                    R.seq_col.push_back(col[di]);
                    R.seq_strain.push_back((uint32_t)di);
                    R.seq_target.push_back(is_target ? 1 : 0);
                    R.seq_strand.push_back(f.strand);
                    R.seq_start.push_back(seq_start);
                    R.seq_end.push_back(seq_end);
                    R.seq_offset.push_back(offset);
                }
            }
            if (t == std::string::npos) break;
            q0 = t + 1;
        }
    }
}
}  // namespace

// The closing threads stay joinable: the next asynchronous close, the next pf_pangenome_open and the process's exit join
// what is still running (a detached thread could be freeing memory while the interpreter goes away, and nobody could
// wait for it).  A thread that cannot be started (EAGAIN) makes this the synchronous close: nothing crosses the C ABI.
namespace {
struct Closers {
    std::mutex mu;
    std::vector<std::thread> th;
    void join_all() {
        std::vector<std::thread> mine;
        { std::lock_guard<std::mutex> g(mu); mine.swap(th); }
        for (auto& t : mine) if (t.joinable()) t.join();
    }
    ~Closers() { join_all(); }
};
Closers& closers() { static Closers c; return c; }
}  // namespace

extern "C" {

}  // extern "C"
namespace {
int open_impl(const pf_pangenome_opts* o, pf_ingest_sink* sink, pf_pangenome** out) {
    if (!o || !out || !o->presence_absence_csv) return in_fail(PF_ERR_ARG, "pf_pangenome_open: null argument");
    *out = nullptr;
    closers().join_all();              // a reader still being torn down in the background: its memory first
    std::string csv;
    if (!read_file(o->presence_absence_csv, csv)) return in_fail(PF_ERR_ARG, std::string("cannot read ") + o->presence_absence_csv);
    pf_pangenome* P = new pf_pangenome();
    P->up = o->upstream; P->down = o->downstream; P->dsc = o->downstream_start_codon != 0; P->raise_missing = o->raise_missing != 0;
    for (uint32_t i = 0; i < o->n_targets; i++) P->targets.insert(o->target_strains[i]);
    P->have_genes = o->gene_list != nullptr;
    for (uint32_t i = 0; i < o->n_genes; i++) P->genes.insert(o->gene_list[i]);
    // ---- panaroo table: index_col=0, drop 'Non-unique Gene name' and 'Annotation' (input.py:188-191)
    size_t pos = 0;
    std::vector<std::string> rec;
    if (!next_csv_record(csv, pos, rec)) { delete P; return in_fail(PF_ERR_ARG, "empty presence/absence table"); }
    std::vector<int> keep;     // header column -> strain index or -1
    bool d1 = false, d2 = false;
    for (size_t c = 1; c < rec.size(); c++) {
        if (rec[c] == "Non-unique Gene name") { d1 = true; keep.push_back(-1); }
        else if (rec[c] == "Annotation") { d2 = true; keep.push_back(-1); }
        else { keep.push_back((int)P->strains.size()); P->strains.push_back(rec[c]); }
    }
    if (!d1 || !d2) { delete P; return in_fail(PF_ERR_ARG, "presence/absence table lacks 'Non-unique Gene name' / 'Annotation' columns"); }
    std::unordered_set<std::string> na(std::begin(NA_STRINGS), std::end(NA_STRINGS));
    {
        // The table is one row per gene cluster with one cell per strain (16 MB per 1 000 clusters x 1 000 strains):
        // record starts are found in one pass that only tracks quotes, the records are split on all host threads.
        std::vector<size_t> starts;
        bool inq = false;
        size_t at = pos;
        const char* base = csv.data();
        const size_t n = csv.size();
        if (at < n) starts.push_back(at);
        while (at < n) {
            if (!inq) {
                const char* q = (const char*)memchr(base + at, '\n', n - at);
                const char* dq = (const char*)memchr(base + at, '"', (q ? (size_t)(q - base) : n) - at);
                if (dq) { inq = true; at = (size_t)(dq - base) + 1; continue; }
                if (!q) break;
                at = (size_t)(q - base) + 1;
                if (at < n) starts.push_back(at);
            } else {
                const char* dq = (const char*)memchr(base + at, '"', n - at);
                if (!dq) break;
                at = (size_t)(dq - base) + 1;
                if (at < n && base[at] == '"') at++;               // doubled quote inside a quoted cell
                else inq = false;
            }
        }
        const size_t nrec = starts.size();
        std::vector<std::string> names(nrec);
        std::vector<std::vector<std::string>> cells(nrec);
        std::vector<uint8_t> blank(nrec, 0);
        unsigned ntc = pf_host_threads(32u);
        if (nrec < 64) ntc = 1;
        std::atomic<size_t> nextrec{0};
        auto work = [&] {
            std::vector<std::string> r;
            for (size_t i; (i = nextrec.fetch_add(1)) < nrec;) {
                size_t ppos = starts[i];
                if (!next_csv_record(csv, ppos, r)) { blank[i] = 1; continue; }
                if (r.size() == 1 && r[0].empty()) { blank[i] = 1; continue; }      // blank line
                names[i] = r[0];
                std::vector<std::string> row(P->strains.size());
                for (size_t c = 1; c < r.size() && c - 1 < keep.size(); c++)
                    if (keep[c - 1] >= 0 && !na.count(r[c])) row[keep[c - 1]] = std::move(r[c]);
                cells[i] = std::move(row);
            }
        };
        std::vector<std::thread> thc;
        for (unsigned t = 1; t < ntc; t++) thc.emplace_back(work);
        work();
        for (auto& x : thc) x.join();
        for (size_t i = 0; i < nrec; i++) {
            if (blank[i]) continue;
            P->cluster_names.push_back(std::move(names[i]));
            P->cells.push_back(std::move(cells[i]));
        }
    }
    P->sorted_strains = P->strains;
    std::sort(P->sorted_strains.begin(), P->sorted_strains.end());
    std::unordered_map<std::string, uint32_t> sp;
    for (uint32_t i = 0; i < P->sorted_strains.size(); i++) sp[P->sorted_strains[i]] = i;
    for (auto& s : P->strains) P->sorted_pos.push_back(sp[s]);
    P->W = (uint32_t)(P->strains.size() + 31) / 32;
    // ---- genomes: GFF features + FASTA (embedded or separate), loaded in parallel
    std::vector<Genome> gs(o->n_genomes);
    std::atomic<long long> dbg_ns[4] = {};
    auto TA = std::chrono::steady_clock::now();
    unsigned nt = pf_host_threads(32u);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            for (uint32_t i = t; i < o->n_genomes; i += nt) {
                Genome& g = gs[i];
                if (sink) {
                    // ---- one pass: the file's text into a block of the sink, GFF lines parsed where they lie, the contigs'
                    // letters left where they lie (scan_fasta) and handed on as pieces
                    auto T0 = std::chrono::steady_clock::now();
                    const bool want_text = P->targets.count(o->genome_names[i]) != 0;
                    const bool separate = o->fasta_paths && o->fasta_paths[i];
                    // The file is read into THIS THREAD's buffer, parsed and measured there, and only the FASTA text is
                    // copied on into the sink's (page-locked) block: the CPU reads page-locked memory -- the driver's own
                    // or registered -- ten times slower than ordinary memory on this platform (GFF lines parsed in a
                    // pinned block: 18 thread-seconds against 1.5), and a buffer kept from file to file is in the caches.
                    static thread_local std::vector<char> tbuf;
                    size_t nb = 0;
                    char *fa0 = nullptr, *fa1 = nullptr;
                    if (separate) {
                        std::string text;
                        if (!read_file(o->gff_paths[i], text)) { g.error = std::string("cannot read ") + o->gff_paths[i]; continue; }
                        parse_gff(text.data(), text.size(), o->gff_paths[i], g);
                        if (!read_file_buf(o->fasta_paths[i], tbuf, &nb, g.error)) continue;
                        fa0 = tbuf.data(); fa1 = tbuf.data() + nb;
                    } else {
                        if (!read_file_buf(o->gff_paths[i], tbuf, &nb, g.error)) continue;
                        auto T1 = std::chrono::steady_clock::now();
                        parse_gff(tbuf.data(), nb, o->gff_paths[i], g);
                        dbg_ns[0] += (T1 - T0).count(); dbg_ns[1] += (std::chrono::steady_clock::now() - T1).count();
                        // open(gff).read().split("##FASTA")[1]   (input.py:103-108)
                        char* a = (char*)memmem(tbuf.data(), nb, "##FASTA", 7);
                        if (!a) { g.error = std::string("no ##FASTA section in ") + o->gff_paths[i]; continue; }
                        a += 7;
                        char* b = (char*)memmem(a, (size_t)(tbuf.data() + nb - a), "##FASTA", 7);
                        fa0 = a; fa1 = b ? b : tbuf.data() + nb;
                    }
                    auto T2 = std::chrono::steady_clock::now();
                    std::vector<pf_ingest_piece> pieces;
                    if (!scan_fasta(fa0, fa0, fa1, g, want_text, sink, pieces, P->up, P->down, P->dsc)) continue;   // (offsets relative to the FASTA text)
                    dbg_ns[2] += (std::chrono::steady_clock::now() - T2).count();
                    auto T3 = std::chrono::steady_clock::now();
                    if (!pieces.empty()) {
                        char* blk = nullptr; uint32_t slot = 0;
                        const size_t fb = (size_t)(fa1 - fa0);
                        if (sink->acquire(sink->self, fb + 64, &blk, &slot) != PF_OK) { g.error = "\x01no block for a genome's text"; continue; }
                        memcpy(blk, fa0, fb);
                        if (sink->submit(sink->self, slot, fb, pieces.data(), (uint32_t)pieces.size()) != PF_OK) { g.error = "\x01the genome store's upload failed"; continue; }
                    }
                    dbg_ns[3] += (std::chrono::steady_clock::now() - T3).count();
                    for (auto& fv : g.features) {
                        auto cit = g.contigs.find(fv.second.chrom);
                        fv.second.ctg = cit == g.contigs.end() ? nullptr : &cit->second;
                    }
                    continue;
                }
                std::string text;
                auto T0 = std::chrono::steady_clock::now();
                if (!read_file(o->gff_paths[i], text)) { g.error = std::string("cannot read ") + o->gff_paths[i]; continue; }
                auto T1 = std::chrono::steady_clock::now();
                parse_gff(text.data(), text.size(), o->gff_paths[i], g);
                auto T2 = std::chrono::steady_clock::now();
                dbg_ns[0] += (T1 - T0).count(); dbg_ns[1] += (T2 - T1).count();
                if (o->fasta_paths && o->fasta_paths[i]) {
                    std::string fa;
                    if (!read_file(o->fasta_paths[i], fa)) { g.error = std::string("cannot read ") + o->fasta_paths[i]; continue; }
                    parse_fasta(fa.data(), fa.data() + fa.size(), g);
                } else {
                    // open(gff).read().split("##FASTA")[1]   (input.py:103-108)
                    size_t a = text.find("##FASTA");
                    if (a == std::string::npos) { g.error = std::string("no ##FASTA section in ") + o->gff_paths[i]; continue; }
                    a += 7;
                    size_t b = text.find("##FASTA", a);
                    if (b == std::string::npos) b = text.size();
                    parse_fasta(text.data() + a, text.data() + b, g);
                }
                auto T3 = std::chrono::steady_clock::now();
                dbg_ns[2] += (T3 - T2).count();
                for (auto& fv : g.features) {                      // feature -> contig, once (the maps keep their nodes when moved)
                    auto cit = g.contigs.find(fv.second.chrom);
                    fv.second.ctg = cit == g.contigs.end() ? nullptr : &cit->second;
                }
            }
        });
    for (auto& x : th) x.join();
    if (getenv("PF_DEBUG_TIMING")) {
        auto TB = std::chrono::steady_clock::now();
        fprintf(stderr, "[open] genomes wall %.3f s; thread-seconds: read %.3f gff %.3f fasta %.3f to-sink %.3f\n", (TB - TA).count() / 1e9,
                dbg_ns[0] / 1e9, dbg_ns[1] / 1e9, dbg_ns[2] / 1e9, dbg_ns[3] / 1e9);
    }
    for (uint32_t i = 0; i < o->n_genomes; i++) {
        if (!gs[i].error.empty()) {
            std::string e = gs[i].error;
            delete P;
            // (\x01: not the input's fault -- the sink ran out of room: the caller may take the two-step way)
            if (e[0] == '\x01') return in_fail(PF_ERR_CAPACITY, e.substr(1));
            return in_fail(PF_ERR_ARG, e);
        }
        P->log += gs[i].warnings;
        P->genomes.emplace(o->genome_names[i], std::move(gs[i]));
    }
    for (auto& st : P->strains) {
        auto git = P->genomes.find(st);
        P->strain_genome.push_back(git == P->genomes.end() ? nullptr : &git->second);
        P->strain_target.push_back(P->targets.count(st) ? 1 : 0);
    }
    // input.py:338-348
    size_t missing = 0;
    for (auto& s : P->strains) if (!P->genomes.count(s)) missing++;
    if (missing) {
        P->log += "There are " + std::to_string(missing) + " strains present in the pangenome table but not in the GFF directory\n";
        if (P->raise_missing) { delete P; return in_fail(PF_ERR_ARG, "Missing " + std::to_string(missing) + " from the GFF directory"); }
    }
    if (sink) P->by_ref = true;            // the genomes are in the device's store: sequences go by reference from the start
    *out = P;
    return PF_OK;
}
}  // namespace

int pf_pangenome_open_sink(const void* opts, pf_ingest_sink* sink, pf_pangenome** out) {
    if (!sink || !sink->acquire || !sink->claim_words || !sink->submit) return in_fail(PF_ERR_ARG, "pf_pangenome_open_sink: null sink");
    return open_impl((const pf_pangenome_opts*)opts, sink, out);
}

extern "C" {

int pf_pangenome_open(const pf_pangenome_opts* o, pf_pangenome** out) { return open_impl(o, nullptr, out); }

// Test hook: pf_pangenome_open_device's reader side WITHOUT a device -- the sink is ordinary memory and the store is
// filled on the host by the addressing genome_pack_text_kernel uses (letter j = byte text_off + j + (j / width) * eol), so
// that what scan_fasta makes of a FASTA (names, lengths, wrapping, joined-in-place records, kept text) can be checked in
// the CPU test suite.  *store is malloc'd (pf_free_text), *nwords its used words.
int pf_debug_open_hostsink(const pf_pangenome_opts* o, pf_pangenome** out, uint64_t** store, uint64_t* nwords) {
    if (!o || !out || !store || !nwords) return in_fail(PF_ERR_ARG, "pf_debug_open_hostsink: null argument");
    struct Host {
        std::mutex mu;
        std::vector<std::unique_ptr<char[]>> blocks;
        std::vector<uint64_t> words;
        std::atomic<uint64_t> used{0};
    } H;
    uint64_t text_bytes = 0;
    for (uint32_t i = 0; i < o->n_genomes; i++) {
        const char* path = (o->fasta_paths && o->fasta_paths[i]) ? o->fasta_paths[i] : o->gff_paths[i];
        struct stat st;
        if (path && ::stat(path, &st) == 0) text_bytes += (uint64_t)st.st_size;
    }
    H.words.assign((size_t)(text_bytes / 4 + 4096), 0);
    pf_ingest_sink sink{};
    sink.self = &H;
    sink.acquire = [](void* self, size_t bytes, char** host, uint32_t* slot) -> int {
        Host* h = (Host*)self;
        std::lock_guard<std::mutex> g(h->mu);
        h->blocks.emplace_back(new char[bytes]);
        *host = h->blocks.back().get(); *slot = (uint32_t)(h->blocks.size() - 1);
        return PF_OK;
    };
    sink.claim_words = [](void* self, uint64_t n) -> uint64_t {
        Host* h = (Host*)self;
        const uint64_t at = h->used.fetch_add(n);
        return at + n <= h->words.size() ? at : UINT64_MAX;
    };
    sink.submit = [](void* self, uint32_t slot, size_t text_bytes2, const pf_ingest_piece* pieces, uint32_t n) -> int {
        Host* h = (Host*)self;
        const unsigned char* text;
        { std::lock_guard<std::mutex> g(h->mu); text = (const unsigned char*)h->blocks[slot].get(); }
        if (getenv("PF_DEBUG_INGEST_SKIP_UPLOAD")) return PF_OK;       // (timing experiment: the reader alone)
        for (uint32_t i = 0; i < n; i++) {
            const pf_ingest_piece& pc = pieces[i];
            for (uint64_t j = 0; j < pc.nbases; j++) {
                const uint64_t at = pc.text_off + j + (pc.width ? (j / pc.width) * pc.eol : 0);
                if (at >= text_bytes2) return PF_ERR_STATE;
                const unsigned ch = text[at];
                h->words[pc.dst_word + j / 32] |= (uint64_t)(((ch >> 1) ^ (ch >> 2)) & 3u) << (62 - 2 * (j & 31));
            }
        }
        return PF_OK;
    };
    const int rc = open_impl(o, &sink, out);
    if (rc != PF_OK) return rc;
    const uint64_t n = std::min<uint64_t>(H.used.load(), H.words.size());
    uint64_t* w = (uint64_t*)malloc((size_t)std::max<uint64_t>(n, 1) * 8);
    if (!w) { pf_pangenome_close(*out); *out = nullptr; return in_fail(PF_ERR_OOM, "malloc failed"); }
    memcpy(w, H.words.data(), (size_t)n * 8);
    *store = w; *nwords = n;
    return PF_OK;
}

// The reader holds millions of small heap objects (a feature map and the contigs per genome, a string per table cell):
// a plain `delete` walks them on one thread -- 0.16 s for 1 000 genomes x 1 000 clusters, as long as the run's whole GPU
// part.  They are emptied on all host threads first.
void pf_pangenome_close(pf_pangenome* P) {
    if (!P) return;
    std::vector<Genome*> gs;
    gs.reserve(P->genomes.size());
    for (auto& kv : P->genomes) gs.push_back(&kv.second);
    const size_t ng = gs.size(), nr = P->cells.size();
    unsigned nt = pf_host_threads(32u);
    if (ng + nr < 64) nt = 1;
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (size_t i; (i = next.fetch_add(1)) < ng + nr;) {
            if (i < ng) { Genome dead; std::swap(dead.contigs, gs[i]->contigs); std::swap(dead.features, gs[i]->features); }
            else { std::vector<std::string> dead; dead.swap(P->cells[i - ng]); }
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
    delete P;
}

// The same, without making the caller wait for it: the reader is handed to a thread of its own, which gives the memory
// back (the contigs alone are gigabytes of pages to unmap: 0.1 s at the end of a run that took 0.35 s).  For callers
// that are done with the run; pf_pangenome_close is the one that has returned everything when it returns.
void pf_pangenome_close_async(pf_pangenome* P) {
    if (!P) return;
    closers().join_all();
    // the vector's slot is made BEFORE the thread exists: once the thread runs it owns P, and nothing after its start may
    // throw (a joinable std::thread destroyed by unwinding is std::terminate; a second close would free P twice)
    std::unique_lock<std::mutex> g(closers().mu);
    try {
        closers().th.emplace_back();
    } catch (...) {
        g.unlock();
        pf_pangenome_close(P);
        return;
    }
    try {
        closers().th.back() = std::thread([P] { pf_pangenome_close(P); });
    } catch (...) {                      // the thread's constructor itself threw: nothing runs, P is still ours
        closers().th.pop_back();
        g.unlock();
        pf_pangenome_close(P);
    }
}

int pf_pangenome_info(pf_pangenome* P, pf_pangenome_info_t* info) {
    if (!P || !info) return in_fail(PF_ERR_ARG, "null argument");
    info->n_clusters = (uint32_t)P->cluster_names.size();
    info->n_strains = (uint32_t)P->strains.size();
    info->next_cluster = (uint32_t)P->next_row;
    return PF_OK;
}

const char* pf_pangenome_strain(pf_pangenome* P, uint32_t i, int sorted) {
    if (!P || i >= P->strains.size()) return nullptr;
    return sorted ? P->sorted_strains[i].c_str() : P->strains[i].c_str();
}

// the accumulated warnings (what the reference sends to logger.warning); cleared by the call
const char* pf_pangenome_take_log(pf_pangenome* P) {
    static thread_local std::string hold;
    if (!P) return "";
    hold.swap(P->log);
    P->log.clear();
    return hold.c_str();
}

void pf_records_free(pf_records* r) { delete r; }

// iter_gene_clusters for the next `max_clusters` rows of the table (input.py:352-468)
int pf_pangenome_next(pf_pangenome* P, uint32_t max_clusters, pf_records** out, pf_records_view_t* v) {
    if (!P || !out || !v) return in_fail(PF_ERR_ARG, "null argument");
    *out = nullptr;
    // rows of this call
    std::vector<size_t> rows;
    while (P->next_row < P->cluster_names.size() && P->next_row < P->end_row && rows.size() < max_clusters) {
        const size_t row = P->next_row++;
        if (P->have_genes && !P->genes.count(P->cluster_names[row])) continue;        // input.py:353-355
        rows.push_back(row);
    }
    const uint32_t made = (uint32_t)rows.size();
    std::vector<RowOut> ro(made);
    const bool dbg = getenv("PF_DEBUG_TIMING") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[pf_pangenome_next] %-12s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t0).count());
        t0 = t;
    };
    {
        unsigned nt = pf_host_threads(32u);
        if (made < 4) nt = 1;
        std::atomic<uint32_t> next{0};
        std::vector<std::thread> th;
        auto work = [&] { for (uint32_t i; (i = next.fetch_add(1)) < made;) build_row(P, rows[i], ro[i]); };
        for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
        work();
        for (auto& x : th) x.join();
    }
    lap("rows");
    for (uint32_t i = 0; i < made; i++) {
        P->log += ro[i].log;                                                           // in table order
        if (!ro[i].error.empty()) return in_fail(PF_ERR_ARG, ro[i].error);
    }
    pf_records* R = new pf_records();
    R->cluster_seq_off.push_back(0);
    R->cluster_strain_off.push_back(0);
    size_t nlit = 0;
    for (auto& r : ro) nlit += r.seq.size();
    R->seq_store.reserve(nlit); R->comp_store.reserve(nlit);
    {   // the merged columns' sizes are known: no growth by doubling while they are filled
        size_t nseq = 0, nstr = 0, npres = 0;
        for (auto& r : ro) { nseq += r.seq_len.size(); nstr += r.dict.size(); npres += r.presab.size(); }
        R->ids.reserve(nseq); R->chroms.reserve(nseq); R->seq_len.reserve(nseq); R->seq_col.reserve(nseq);
        R->seq_strain.reserve(nseq); R->seq_target.reserve(nseq); R->seq_strand.reserve(nseq); R->seq_start.reserve(nseq);
        R->seq_end.reserve(nseq); R->seq_offset.reserve(nseq); R->seq_src_off.reserve(nseq); R->seq_src_start.reserve(nseq);
        R->seq_flags.reserve(nseq); R->cluster_seq_off.reserve((size_t)made + 1); R->cluster_strain_off.reserve((size_t)made + 1);
        R->cluster_nstrains.reserve(made); R->cluster_npresab.reserve(made); R->cluster_row.reserve(made);
        R->cluster_presab.reserve(npres); R->strain_index.reserve(nstr);
    }
    auto app = [](auto& dst, const auto& src) { dst.insert(dst.end(), src.begin(), src.end()); };
    for (uint32_t i = 0; i < made; i++) {
        RowOut& r = ro[i];
        for (auto& x : r.seq) R->seq_store.push_back(std::move(x));
        for (auto& x : r.comp) R->comp_store.push_back(std::move(x));
        app(R->ids, r.ids); app(R->chroms, r.chroms);
        app(R->seq_len, r.seq_len); app(R->seq_col, r.seq_col); app(R->seq_strain, r.seq_strain);
        app(R->seq_target, r.seq_target); app(R->seq_strand, r.seq_strand);
        app(R->seq_start, r.seq_start); app(R->seq_end, r.seq_end); app(R->seq_offset, r.seq_offset);
        app(R->seq_src_off, r.seq_src_off); app(R->seq_src_start, r.seq_src_start); app(R->seq_flags, r.seq_flags);
        R->cluster_seq_off.push_back((uint32_t)R->seq_len.size());
        R->cluster_nstrains.push_back((uint32_t)r.dict.size());
        R->cluster_npresab.push_back((uint32_t)P->strains.size());
        app(R->cluster_presab, r.presab);
        R->cluster_row.push_back((uint32_t)rows[i]);
        app(R->strain_index, r.dict);
        R->cluster_strain_off.push_back((uint32_t)R->strain_index.size());
    }
    lap("merge");
    const size_t n = R->seq_len.size();
    R->seq.resize(n); R->comp.resize(n);
    for (size_t i = 0, li = 0; i < n; i++) {
        const bool ref = (R->seq_flags[i] & 1u) != 0;
        R->seq[i] = ref ? nullptr : R->seq_store[li].c_str();
        R->comp[i] = ref ? nullptr : R->comp_store[li].c_str();
        if (!ref) li++;
    }
    R->cluster_name_ptr.resize(made);
    for (uint32_t i = 0; i < made; i++) R->cluster_name_ptr[i] = P->cluster_names[R->cluster_row[i]].c_str();
    R->strain_ptr.resize(R->strain_index.size());
    for (size_t i = 0; i < R->strain_index.size(); i++) R->strain_ptr[i] = P->strains[R->strain_index[i]].c_str();
    v->n_clusters = made; v->n_seqs = (uint32_t)n; v->W = P->W; v->reserved = 0;
    v->seq = R->seq.data(); v->comp = R->comp.data(); v->id = R->ids.data(); v->chromosome = R->chroms.data();
    v->seq_len = R->seq_len.data(); v->seq_col = R->seq_col.data(); v->seq_strain = R->seq_strain.data();
    v->seq_target = R->seq_target.data(); v->seq_strand = R->seq_strand.data();
    v->seq_start = R->seq_start.data(); v->seq_end = R->seq_end.data(); v->seq_offset = R->seq_offset.data();
    v->cluster_seq_off = R->cluster_seq_off.data(); v->cluster_name = R->cluster_name_ptr.data();
    v->cluster_nstrains = R->cluster_nstrains.data(); v->cluster_npresab = R->cluster_npresab.data();
    v->cluster_presab = R->cluster_presab.data(); v->cluster_strain_off = R->cluster_strain_off.data();
    v->cluster_strain = R->strain_ptr.data();
    v->seq_src_off = P->by_ref ? R->seq_src_off.data() : nullptr;
    v->seq_src_start = P->by_ref ? R->seq_src_start.data() : nullptr;
    v->seq_flags = P->by_ref ? R->seq_flags.data() : nullptr;
    *out = R;
    return PF_OK;
}

int pf_pangenome_weights(pf_pangenome* P, uint32_t cap, uint32_t* weights, uint32_t* n_processed) {
    if (!P || !n_processed) return in_fail(PF_ERR_ARG, "pf_pangenome_weights: null argument");
    uint32_t n = 0;
    for (size_t row = 0; row < P->cluster_names.size(); row++) {
        if (P->have_genes && !P->genes.count(P->cluster_names[row])) continue;        // input.py:353-355
        if (weights && n < cap) {
            uint32_t w = 0;
            for (auto& cell : P->cells[row])
                if (!cell.empty()) w += 1 + (uint32_t)std::count(cell.begin(), cell.end(), ';');     // input.py:393
            weights[n] = w;
        }
        n++;
    }
    *n_processed = n;
    return PF_OK;
}

int pf_pangenome_set_range(pf_pangenome* P, uint32_t first, uint32_t count) {
    if (!P) return in_fail(PF_ERR_ARG, "pf_pangenome_set_range: null argument");
    // table rows of processed clusters `first` and `first + count`
    size_t seen = 0, row_first = P->cluster_names.size(), row_end = P->cluster_names.size();
    for (size_t row = 0; row < P->cluster_names.size(); row++) {
        if (P->have_genes && !P->genes.count(P->cluster_names[row])) continue;
        if (seen == first) row_first = row;
        if (seen == (size_t)first + count) { row_end = row; break; }
        seen++;
    }
    if (count == 0) row_end = row_first;
    P->next_row = row_first;
    P->end_row = row_end;
    return PF_OK;
}

int pf_pangenome_contigs(pf_pangenome* P, uint32_t* n, const char* const** ascii, const uint64_t** len) {
    if (!P || !n || !ascii || !len) return in_fail(PF_ERR_ARG, "pf_pangenome_contigs: null argument");
    if (P->flat.empty()) {
        // any fixed order will do; keep it deterministic: genomes by name, contigs by name
        std::vector<std::string> gnames;
        for (auto& kv : P->genomes) gnames.push_back(kv.first);
        std::sort(gnames.begin(), gnames.end());
        for (auto& gn : gnames) {
            Genome& g = P->genomes.find(gn)->second;
            std::vector<std::string> cn;
            for (auto& kv : g.contigs) cn.push_back(kv.first);
            std::sort(cn.begin(), cn.end());
            for (auto& c : cn) P->flat.push_back(&g.contigs.find(c)->second);
        }
        for (Contig* c : P->flat) {
            if (!c->has_text && c->len) {
                P->flat.clear(); P->flat_ptr.clear(); P->flat_len.clear();       // (a second call fails the same way)
                return in_fail(PF_ERR_STATE, "pf_pangenome_contigs: this reader's genomes went to the device as it was opened");
            }
            P->flat_ptr.push_back(c->seq.data()); P->flat_len.push_back(c->seq.size());
        }
    }
    *n = (uint32_t)P->flat.size();
    *ascii = P->flat_ptr.data();
    *len = P->flat_len.data();
    return PF_OK;
}

int pf_pangenome_set_store(pf_pangenome* P, const uint64_t* contig_word_off, uint32_t n) {
    if (!P || (n && !contig_word_off)) return in_fail(PF_ERR_ARG, "pf_pangenome_set_store: null argument");
    if (n != P->flat.size()) return in_fail(PF_ERR_ARG, "pf_pangenome_set_store: call pf_pangenome_contigs first (contig count differs)");
    for (uint32_t i = 0; i < n; i++) P->flat[i]->word_off = contig_word_off[i];
    P->by_ref = true;
    return PF_OK;
}

}  // extern "C"
