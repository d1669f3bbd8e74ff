// Parallel gzip for the three output files.  Host code, no GPU involved.
//
// The reference writes its --compress outputs through gzip.open(..., "wt", compresslevel=9)
// (/root/reference/panfeed/input.py:235-259): one deflate stream, one core, ~20 MB/s -- minutes for the 10-20 GB of
// TSV of a large run.  Here a block of text is cut into chunks at line ends, every chunk is deflated (level 9 by
// default) as its own gzip member on its own thread, and the members are concatenated: a multi-member gzip file
// (RFC 1952 section 2.2), which gzip / zcat / Python's gzip / pandas read back as the same text.
#include "pf_host.h"
#include "../../include/panfeed_hip.h"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" void pf_set_error_(const char* msg);

extern "C" int pf_gzip_members(const char* data, uint64_t n, int level, uint64_t chunk_bytes, char** out, uint64_t* out_n) {
    if ((!data && n) || !out || !out_n) { pf_set_error_("pf_gzip_members: null argument"); return PF_ERR_ARG; }
    if (level < 0 || level > 9) { pf_set_error_("pf_gzip_members: level must be 0..9"); return PF_ERR_ARG; }
    if (chunk_bytes < 65536) chunk_bytes = 65536;
    // chunk boundaries at line ends, so that every member decompresses to whole lines
    std::vector<uint64_t> cut{0};
    while (cut.back() < n) {
        uint64_t e = std::min<uint64_t>(n, cut.back() + chunk_bytes);
        if (e < n) {
            const void* nl = memchr(data + e, '\n', (size_t)std::min<uint64_t>(n - e, 1u << 20));
            e = nl ? (uint64_t)((const char*)nl - data) + 1 : e;
        }
        cut.push_back(e);
    }
    const size_t nchunk = cut.size() - 1;
    std::vector<std::string> parts(nchunk);
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    auto work = [&] {
        for (size_t i; (i = next.fetch_add(1)) < nchunk;) {
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad = 1; continue; }
            const uint64_t len = cut[i + 1] - cut[i];
            std::string& o = parts[i];
            o.resize(deflateBound(&zs, (uLong)len) + 32);
            zs.next_in = (Bytef*)(data + cut[i]);
            zs.avail_in = (uInt)len;
            zs.next_out = (Bytef*)&o[0];
            zs.avail_out = (uInt)o.size();
            const int rc = deflate(&zs, Z_FINISH);
            if (rc != Z_STREAM_END) bad = 1;
            o.resize(zs.total_out);
            deflateEnd(&zs);
        }
    };
    if (chunk_bytes >= (1ull << 31)) { pf_set_error_("pf_gzip_members: chunk_bytes must stay below 2 GiB"); return PF_ERR_ARG; }
    unsigned nt = (unsigned)std::min<size_t>(pf_host_threads(64u), std::max<size_t>(nchunk, 1));
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
    if (bad) { pf_set_error_("pf_gzip_members: deflate failed"); return PF_ERR_STATE; }
    uint64_t total = 0;
    for (auto& p : parts) total += p.size();
    char* buf = (char*)malloc(total ? total : 1);
    if (!buf) { pf_set_error_("pf_gzip_members: out of memory"); return PF_ERR_OOM; }
    uint64_t o = 0;
    for (auto& p : parts) { memcpy(buf + o, p.data(), p.size()); o += p.size(); }
    *out = buf;
    *out_n = total;
    return PF_OK;
}
