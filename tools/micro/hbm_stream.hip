// Calibration: what does a plain streaming read of the packed input reach on this GPU?  16-byte loads per lane, grid-stride,
// for several grid sizes / waves per SIMD, and the same with 8 lanes per 128-byte piece walking "segments" the way the
// dedup pass does (8 lanes share a ~300-byte segment, 64 segments per 512-thread workgroup and trip).
//   hipcc --offload-arch=gfx950 -O3 -o hbm_stream hbm_stream.hip && ./hbm_stream [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef unsigned long long u64;

template <int UNROLL>
__global__ __launch_bounds__(512) void stream_sum(const ulonglong2* p, size_t n16, u64* out) {
    u64 acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        ulonglong2 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u].x ^ v[u].y;
    }
    for (; i < n16; i += stride) { const ulonglong2 v = p[i]; acc += v.x ^ v.y; }
    if (acc == 0x123456789abcdefull) out[0] = acc;
}

// one workgroup per "cluster" of nseg segments of seg16 16-byte pieces each (contiguous), 8 lanes per segment, one trip =
// 64 segments, up to 3 pieces per lane in flight -- the dedup pass's read pattern without its work
__global__ __launch_bounds__(512) void cluster_walk(const ulonglong2* p, uint32_t nseg, uint32_t seg16, u64* out) {
    const ulonglong2* base = p + (size_t)blockIdx.x * nseg * seg16;
    const uint32_t grp = threadIdx.x >> 3, gl = threadIdx.x & 7;
    u64 acc = 0;
    for (uint32_t s = grp; s < nseg; s += 64) {
        const ulonglong2* w = base + (size_t)s * seg16;
        ulonglong2 v[3];
#pragma unroll
        for (int q = 0; q < 3; q++) v[q] = gl + 8 * q < seg16 ? w[gl + 8 * q] : make_ulonglong2(0, 0);
#pragma unroll
        for (int q = 0; q < 3; q++) acc += v[q].x ^ v[q].y;
    }
    if (acc == 0x123456789abcdefull) out[0] = acc;
}

// the same walk with the dedup pass's constraints added one by one: LDS bytes per workgroup (occupancy), segment offsets and
// lengths read from arrays one trip ahead (META), the content hash (HASH), a prologue of dependent reads (PRO)
template <int LDSB, bool META, bool HASH, bool PRO>
__global__ __launch_bounds__(512) void cluster_walk2(const ulonglong2* p, const uint32_t* clu_off, const unsigned long long* seg_off,
                                                     const uint32_t* seg_len, uint32_t nseg, uint32_t seg16, u64* out) {
    __shared__ uint32_t pad[LDSB / 4];
    if (threadIdx.x == 0) pad[0] = 0;
    const uint32_t s0 = PRO ? clu_off[blockIdx.x] : blockIdx.x * nseg;
    const size_t wbase = PRO ? seg_off[s0] : (size_t)s0 * seg16 * 2;
    const ulonglong2* base = p + wbase / 2;
    const uint32_t grp = threadIdx.x >> 3, gl = threadIdx.x & 7;
    u64 acc = 0, tot = 0;
    uint32_t nlen = META ? seg_len[s0 + grp] : seg16 * 64;
    uint32_t nwo = META ? (uint32_t)(seg_off[s0 + grp] - wbase) : grp * seg16 * 2;
    for (uint32_t s = grp; s < nseg; s += 64) {
        const uint32_t pc = (nlen + 63) >> 6;
        const ulonglong2* w = base + (nwo >> 1);
        ulonglong2 v[3];
#pragma unroll
        for (int q = 0; q < 3; q++) v[q] = gl + 8 * q < pc ? w[gl + 8 * q] : make_ulonglong2(0, 0);
        const uint32_t sn = s + 64;
        if (META) { nlen = sn < nseg ? seg_len[s0 + sn] : 0; nwo = sn < nseg ? (uint32_t)(seg_off[s0 + sn] - wbase) : 0; }
        else { nwo = sn * seg16 * 2; }
        if (HASH) {
            acc = 0;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const uint32_t sl = 0x9E3779B1u * (4 * (gl + 8 * q) + 1), sh = 0x85EBCA77u * (4 * (gl + 8 * q) + 1);
                acc += (u64)((uint32_t)v[q].x ^ sl) * (u64)((uint32_t)(v[q].x >> 32) ^ sh) +
                       (u64)((uint32_t)v[q].y ^ (sl + 2 * 0x9E3779B1u)) * (u64)((uint32_t)(v[q].y >> 32) ^ (sh + 2 * 0x85EBCA77u));
            }
            for (int d = 1; d < 8; d <<= 1) acc += __shfl_xor(acc, d);
            acc ^= acc >> 30; acc *= 0xbf58476d1ce4e5b9ull; acc ^= acc >> 27; acc *= 0x94d049bb133111ebull; acc ^= acc >> 31;
            tot += acc;
        } else {
#pragma unroll
            for (int q = 0; q < 3; q++) tot += v[q].x ^ v[q].y;
        }
    }
    if (tot == 0x123456789abcdefull) out[0] = tot + pad[threadIdx.x & 1];
}

template <typename F>
static float time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 11.2;
    const size_t bytes = (size_t)(gib * (1ull << 30)) / 4864 * 4864;      // multiple of one 1000 x 304-byte "cluster"
    ulonglong2* buf; u64* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, bytes);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const size_t n16 = bytes / 16;
    printf("%.2f GB, %d CUs\n", bytes / 1e9, cus);
    for (int wg_per_cu : {2, 3, 4}) {
        const int grid = cus * wg_per_cu;
        float ms = time_ms([&] { stream_sum<1><<<grid, 512>>>(buf, n16, out); });
        printf("stream_sum unroll 1, %d x 512 threads per CU: %.3f ms  %.2f TB/s\n", wg_per_cu, ms, bytes / ms / 1e9);
        ms = time_ms([&] { stream_sum<2><<<grid, 512>>>(buf, n16, out); });
        printf("stream_sum unroll 2, %d x 512 threads per CU: %.3f ms  %.2f TB/s\n", wg_per_cu, ms, bytes / ms / 1e9);
        ms = time_ms([&] { stream_sum<4><<<grid, 512>>>(buf, n16, out); });
        printf("stream_sum unroll 4, %d x 512 threads per CU: %.3f ms  %.2f TB/s\n", wg_per_cu, ms, bytes / ms / 1e9);
    }
    {   // clusters of 1000 segments x 19 pieces (304 bytes): one workgroup each, as many as fit the buffer
        const uint32_t nseg = 1000, seg16 = 19;
        const uint32_t nclu = (uint32_t)(n16 / ((size_t)nseg * seg16));
        const float ms = time_ms([&] { cluster_walk<<<nclu, 512>>>(buf, nseg, seg16, out); });
        printf("cluster_walk %u clusters x %u segments x %u B, one 512-thread workgroup each: %.3f ms  %.2f TB/s\n", nclu, nseg,
               seg16 * 16, ms, (double)nclu * nseg * seg16 * 16 / ms / 1e9);
    }
    {
        const uint32_t nseg = 1000, seg16 = 19;
        const uint32_t nclu = (uint32_t)(n16 / ((size_t)nseg * seg16));
        uint32_t* clu_off; unsigned long long* seg_off; uint32_t* seg_len;
        (void)hipMalloc(&clu_off, (size_t)(nclu + 1) * 4); (void)hipMalloc(&seg_off, (size_t)nclu * nseg * 8); (void)hipMalloc(&seg_len, (size_t)nclu * nseg * 4);
        {
            uint32_t* h0 = (uint32_t*)malloc((size_t)(nclu + 1) * 4);
            unsigned long long* h1 = (unsigned long long*)malloc((size_t)nclu * nseg * 8);
            uint32_t* h2 = (uint32_t*)malloc((size_t)nclu * nseg * 4);
            for (uint32_t c = 0; c <= nclu; c++) h0[c] = c * nseg;
            for (size_t i = 0; i < (size_t)nclu * nseg; i++) { h1[i] = i * seg16 * 2; h2[i] = seg16 * 64 - 20; }
            (void)hipMemcpy(clu_off, h0, (size_t)(nclu + 1) * 4, hipMemcpyHostToDevice);
            (void)hipMemcpy(seg_off, h1, (size_t)nclu * nseg * 8, hipMemcpyHostToDevice);
            (void)hipMemcpy(seg_len, h2, (size_t)nclu * nseg * 4, hipMemcpyHostToDevice);
            free(h0); free(h1); free(h2);
        }
        const double gb = (double)nclu * nseg * seg16 * 16 / 1e9;
#define RUN(NAME, ...) { const float ms = time_ms([&] { cluster_walk2<__VA_ARGS__><<<nclu, 512>>>(buf, clu_off, seg_off, seg_len, nseg, seg16, out); }); \
        printf("cluster_walk2 %-44s %.3f ms  %.2f TB/s\n", NAME, ms, gb / ms); }
        RUN("no LDS, direct offsets", 16, false, false, false)
        RUN("46 KiB LDS (3 WG/CU)", 47104, false, false, false)
        RUN("46 KiB LDS, offsets from arrays 1 trip ahead", 47104, true, false, false)
        RUN("46 KiB LDS, arrays, content hash", 47104, true, true, false)
        RUN("46 KiB LDS, arrays, hash, dependent prologue", 47104, true, true, true)
        RUN("no LDS, arrays, hash, dependent prologue", 16, true, true, true)
        RUN("70 KiB LDS (2 WG/CU), arrays, hash, prologue", 71680, true, true, true)
        hipFree(clu_off); hipFree(seg_off); hipFree(seg_len);
    }
    hipFree(buf); hipFree(out);
    return 0;
}
