// pf_rowfilter.hip -- SURVEY 8f row N4: the row filter of the reference's downstream tools on the device.
//
//   panfeed-get-clusters  /root/reference/panfeed/get_clusters.py:89-94   rows of kmers_to_hashes.tsv whose
//                                                                          hashed_pattern is in a set of hashes
//   panfeed-get-kmers     /root/reference/panfeed/get_kmers.py:103-106    the same, and
//                         /root/reference/panfeed/get_kmers.py:131-134    rows of kmers.tsv whose cluster is in a set
//
// The reference streams the file through pandas in 100 000-row chunks and keeps `x[x[col].isin(keys)]`.  Here a block
// of the file's text goes to HBM as it is; every thread looks at 16 bytes, and for every line end it finds there it
// hashes that line's key field (the LAST field of the line for hashed_pattern -- 24 base64 characters -- or the FIRST
// field of the line that starts behind it for cluster), probes a device hash set of the keys' 64-bit hashes and appends
// the position to a list.  The host then checks every candidate against the exact key strings (so a 64-bit hash
// collision cannot add a row) and hands back the matching lines in file order.  Byte work, HBM/PCIe-bound; no parsing
// of the other fields.
#include <hip/hip_runtime.h>

#include "../../include/panfeed_hip.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_set>
#include <thread>
#include <vector>

extern "C" void pf_set_error_(const char* msg);

namespace {

int rf_fail(int code, const std::string& msg) { pf_set_error_(msg.c_str()); return code; }
#define RFCHK(expr)                                                                                       \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return rf_fail(e_ == hipErrorOutOfMemory ? PF_ERR_OOM : PF_ERR_HIP,                           \
                           std::string(#expr) + " failed: " + hipGetErrorString(e_));                     \
    } while (0)

constexpr uint64_t RF_EMPTY = 0;
constexpr uint32_t RF_MAX_FIELD = 4096;       // a key field longer than this never matches (nor do the keys)

__host__ __device__ inline uint64_t rf_hash_step(uint64_t h, unsigned char c) { return (h ^ c) * 0x100000001B3ull; }
__host__ __device__ inline uint64_t rf_hash_fin(uint64_t h) {
    h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
    return h ? h : 1;                                   // 0 marks an empty slot
}
inline uint64_t rf_hash(const char* s, size_t n) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (size_t i = 0; i < n; i++) h = rf_hash_step(h, (unsigned char)s[i]);
    return rf_hash_fin(h);
}

struct RfParams {
    const unsigned char* text;   // device copy of the block
    uint64_t n;                  // bytes of complete lines (text[n - 1] == '\n')
    const uint64_t* set;         // open addressing, RF_EMPTY = free
    uint64_t cap;                // power of two
    int first_field;             // 1: key = first field of the line; 0: last field
    uint64_t* out;               // candidate positions: first_field ? line start : position of the line's '\n'
    unsigned long long* count;
    uint64_t out_cap;
};

__device__ __forceinline__ bool rf_probe(const RfParams& p, uint64_t h) {
    uint64_t slot = h & (p.cap - 1);
    for (uint64_t probes = 0; probes < p.cap; probes++) {
        const uint64_t cur = p.set[slot];
        if (cur == h) return true;
        if (cur == RF_EMPTY) return false;
        slot = (slot + 1) & (p.cap - 1);
    }
    return false;
}

// one line whose key field is to be tested; `at` = position of the '\n' that ends it (last field) or of the
// '\n' in front of it (first field; -1 for the line at the start of the block)
__device__ __forceinline__ void rf_line(const RfParams& p, int64_t at) {
    uint64_t h = 0xCBF29CE484222325ull;
    uint64_t pos;
    if (p.first_field) {
        const uint64_t s = (uint64_t)(at + 1);
        if (s >= p.n) return;
        uint64_t e = s;
        while (e < p.n && e - s < RF_MAX_FIELD && p.text[e] != '\t' && p.text[e] != '\n') { h = rf_hash_step(h, p.text[e]); e++; }
        if (e - s >= RF_MAX_FIELD) return;
        pos = s;
    } else {
        if (at < 0) return;
        int64_t s = at;                                  // field = (s, at)
        while (s > 0 && at - s < (int64_t)RF_MAX_FIELD && p.text[s - 1] != '\t' && p.text[s - 1] != '\n') s--;
        if (at - s >= (int64_t)RF_MAX_FIELD) return;
        for (int64_t i = s; i < at; i++) h = rf_hash_step(h, p.text[i]);
        pos = (uint64_t)at;
    }
    if (!rf_probe(p, rf_hash_fin(h))) return;
    const unsigned long long k = atomicAdd(p.count, 1ull);
    if (k < p.out_cap) p.out[k] = pos;
}

__global__ __launch_bounds__(256) void rowfilter_kernel(RfParams p) {
    const uint64_t nvec = (p.n + 15) / 16;
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < nvec; v += (uint64_t)gridDim.x * blockDim.x) {
        // the block's buffer is padded to a multiple of 16 bytes with zeros
        const uint4 w = reinterpret_cast<const uint4*>(p.text)[v];
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
        if (v == 0 && p.first_field) rf_line(p, -1);     // the line that starts the block
#pragma unroll
        for (int q = 0; q < 4; q++) {
            // bytes equal to '\n' in this word
            const uint32_t x = ws[q] ^ 0x0A0A0A0Au;
            uint32_t m = (x - 0x01010101u) & ~x & 0x80808080u;
            while (m) {
                const int b = (__ffs((int)m) - 1) >> 3;
                m &= m - 1;
                const uint64_t at = v * 16 + q * 4 + b;
                if (at < p.n) rf_line(p, (int64_t)at);
            }
        }
    }
}

}  // namespace

struct pf_rowfilter {
    int device = 0;
    int first_field = 0;
    hipStream_t stream = nullptr;
    std::unordered_set<std::string> keys;
    uint64_t* d_set = nullptr;
    uint64_t cap = 0;
    unsigned char* d_text = nullptr;
    size_t text_cap = 0;
    char* pin = nullptr;
    size_t pin_cap = 0;
    uint64_t* d_out = nullptr;
    size_t out_cap = 0;
    unsigned long long* d_count = nullptr;
    std::vector<uint64_t> begin, end;          // result of the last scan
    uint64_t bytes_scanned = 0;
    float device_ms = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};

extern "C" {

void pf_rowfilter_destroy(pf_rowfilter* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->stream) (void)hipStreamSynchronize(f->stream);
    if (f->d_set) (void)hipFree(f->d_set);
    if (f->d_text) (void)hipFree(f->d_text);
    if (f->d_out) (void)hipFree(f->d_out);
    if (f->d_count) (void)hipFree(f->d_count);
    if (f->pin) (void)hipHostFree(f->pin);
    if (f->e0) (void)hipEventDestroy(f->e0);
    if (f->e1) (void)hipEventDestroy(f->e1);
    if (f->stream) (void)hipStreamDestroy(f->stream);
    delete f;
}

int pf_rowfilter_create(int device, int first_field, const char* const* keys, const uint32_t* key_len, uint64_t n_keys,
                        pf_rowfilter** out) {
    if (!out || (n_keys && (!keys || !key_len))) return rf_fail(PF_ERR_ARG, "pf_rowfilter_create: null argument");
    *out = nullptr;
    int ndev = 0;
    RFCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return rf_fail(PF_ERR_ARG, "pf_rowfilter_create: no such device");
    RFCHK(hipSetDevice(device));
    pf_rowfilter* f = new pf_rowfilter();
    f->device = device;
    f->first_field = first_field ? 1 : 0;
    std::vector<uint64_t> table;
    uint64_t cap = 1024;
    while (cap < 4 * (n_keys + 1)) cap <<= 1;
    table.assign(cap, RF_EMPTY);
    for (uint64_t i = 0; i < n_keys; i++) {
        if (key_len[i] >= RF_MAX_FIELD) continue;        // cannot match: the kernel gives up on fields this long
        auto ins = f->keys.emplace(keys[i], key_len[i]);
        if (!ins.second) continue;
        const uint64_t h = rf_hash(keys[i], key_len[i]);
        uint64_t slot = h & (cap - 1);
        while (table[slot] != RF_EMPTY && table[slot] != h) slot = (slot + 1) & (cap - 1);
        table[slot] = h;
    }
    f->cap = cap;
    int rc = PF_OK;
    do {
        if (hipStreamCreate(&f->stream) != hipSuccess || hipEventCreate(&f->e0) != hipSuccess || hipEventCreate(&f->e1) != hipSuccess ||
            hipMalloc((void**)&f->d_set, cap * 8) != hipSuccess || hipMalloc((void**)&f->d_count, 8) != hipSuccess) {
            rc = rf_fail(PF_ERR_HIP, "pf_rowfilter_create: device allocation failed");
            break;
        }
        if (hipMemcpy(f->d_set, table.data(), cap * 8, hipMemcpyHostToDevice) != hipSuccess)
            rc = rf_fail(PF_ERR_HIP, "pf_rowfilter_create: upload failed");
    } while (0);
    if (rc != PF_OK) { pf_rowfilter_destroy(f); return rc; }
    *out = f;
    return PF_OK;
}

int pf_rowfilter_scan(pf_rowfilter* f, const char* text, uint64_t nbytes, const uint64_t** line_begin,
                      const uint64_t** line_end, uint64_t* n_lines, uint64_t* consumed) {
    if (!f || !line_begin || !line_end || !n_lines || !consumed || (nbytes && !text))
        return rf_fail(PF_ERR_ARG, "pf_rowfilter_scan: null argument");
    RFCHK(hipSetDevice(f->device));
    f->begin.clear(); f->end.clear();
    *line_begin = nullptr; *line_end = nullptr; *n_lines = 0; *consumed = 0;
    // complete lines only: the caller carries the rest over to its next block
    uint64_t n = nbytes;
    while (n && text[n - 1] != '\n') n--;
    *consumed = n;
    if (!n || f->keys.empty()) return PF_OK;
    const size_t padded = (n + 15) / 16 * 16 + 16;
    if (padded > f->text_cap) {
        if (f->d_text) (void)hipFree(f->d_text);
        if (f->pin) (void)hipHostFree(f->pin);
        f->d_text = nullptr; f->pin = nullptr; f->text_cap = 0;
        const size_t want = padded + padded / 8;
        RFCHK(hipMalloc((void**)&f->d_text, want));
        RFCHK(hipHostMalloc((void**)&f->pin, want, hipHostMallocDefault));
        f->text_cap = want;
    }
    // room for the candidates: the filter keeps few rows, so the room is a guess (a position per 64 bytes of text, a
    // million at least) and the kernel is run again with what it asked for should the guess be too small -- counting
    // the lines of the block on the host to size it for the worst case cost more than the kernel itself
    const size_t guess = std::max<size_t>((size_t)1 << 20, (size_t)(n / 64));
    if (guess > f->out_cap) {
        if (f->d_out) (void)hipFree(f->d_out);
        f->d_out = nullptr; f->out_cap = 0;
        RFCHK(hipMalloc((void**)&f->d_out, guess * 8));
        f->out_cap = guess;
    }
    {   // the block into pinned memory on a few threads (one memcpy of 256 MB is slower than the rest of the call)
        const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(8, n >> 22));
        std::vector<std::thread> th;
        const uint64_t step = (n + nt - 1) / nt;
        for (unsigned t = 1; t < nt; t++) {
            const uint64_t a = t * step, b = std::min<uint64_t>(n, a + step);
            if (a < b) th.emplace_back([=] { memcpy(f->pin + a, text + a, (size_t)(b - a)); });
        }
        memcpy(f->pin, text, (size_t)std::min<uint64_t>(n, step));
        for (auto& t : th) t.join();
    }
    memset(f->pin + n, 0, padded - n);
    RFCHK(hipMemcpyAsync(f->d_text, f->pin, padded, hipMemcpyHostToDevice, f->stream));
    unsigned long long cnt = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        RFCHK(hipMemsetAsync(f->d_count, 0, 8, f->stream));
        RfParams p{};
        p.text = f->d_text; p.n = n; p.set = f->d_set; p.cap = f->cap; p.first_field = f->first_field;
        p.out = f->d_out; p.count = f->d_count; p.out_cap = f->out_cap;
        const uint64_t nvec = (n + 15) / 16;
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((nvec + 255) / 256, 256 * 16);
        RFCHK(hipEventRecord(f->e0, f->stream));
        hipLaunchKernelGGL(rowfilter_kernel, dim3(blocks), dim3(256), 0, f->stream, p);
        RFCHK(hipGetLastError());
        RFCHK(hipEventRecord(f->e1, f->stream));
        RFCHK(hipMemcpyAsync(&cnt, f->d_count, 8, hipMemcpyDeviceToHost, f->stream));
        RFCHK(hipStreamSynchronize(f->stream));
        float ms = 0;
        if (hipEventElapsedTime(&ms, f->e0, f->e1) == hipSuccess) f->device_ms += ms;
        if (cnt <= f->out_cap) break;
        // more candidates than room (the kernel counted them all and kept what fitted): again, with room for all
        (void)hipFree(f->d_out);
        f->d_out = nullptr; f->out_cap = 0;
        RFCHK(hipMalloc((void**)&f->d_out, ((size_t)cnt + (size_t)cnt / 8) * 8));
        f->out_cap = (size_t)cnt + (size_t)cnt / 8;
    }
    f->bytes_scanned += n;
    if (cnt > f->out_cap) return rf_fail(PF_ERR_STATE, "pf_rowfilter_scan: more candidates than room, twice");
    std::vector<uint64_t> pos((size_t)cnt);
    if (cnt) RFCHK(hipMemcpy(pos.data(), f->d_out, (size_t)cnt * 8, hipMemcpyDeviceToHost));
    std::sort(pos.begin(), pos.end());
    // exact check of every candidate (a 64-bit hash collision must not add a row), then the line's extent
    for (uint64_t q : pos) {
        uint64_t b, e;
        std::string key;
        if (f->first_field) {
            b = q;
            const char* nl = (const char*)memchr(text + b, '\n', (size_t)(n - b));
            e = nl ? (uint64_t)(nl - text) : n;
            uint64_t t = b;
            while (t < e && text[t] != '\t') t++;
            key.assign(text + b, (size_t)(t - b));
        } else {
            e = q;
            b = e;
            while (b > 0 && text[b - 1] != '\n') b--;
            uint64_t t = e;
            while (t > b && text[t - 1] != '\t') t--;
            key.assign(text + t, (size_t)(e - t));
        }
        if (!f->keys.count(key)) continue;
        f->begin.push_back(b);
        f->end.push_back(e + 1 <= n ? e + 1 : n);          // the '\n' is part of the line
    }
    *line_begin = f->begin.data(); *line_end = f->end.data(); *n_lines = f->begin.size();
    return PF_OK;
}

int pf_rowfilter_stats(pf_rowfilter* f, uint64_t* bytes_scanned, float* device_ms) {
    if (!f) return rf_fail(PF_ERR_ARG, "pf_rowfilter_stats: null argument");
    if (bytes_scanned) *bytes_scanned = f->bytes_scanned;
    if (device_ms) *device_ms = f->device_ms;
    return PF_OK;
}

}  // extern "C"
