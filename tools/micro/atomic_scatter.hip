// What do scattered atomics cost on this GPU?  The run-global pattern table's inserts are one compare-and-swap on a random
// 8-byte word of a 270 MB table, then a minimum on a word per new id (contiguous across the wave's lanes) or per found id
// (scattered).  Measured here, each as N operations spread over a chip-filling grid, 8 independent operations per thread:
//   cas_agent      64-bit CAS at agent scope (what the kernels use), random slots of the whole table
//   cas_wg_own     64-bit CAS at workgroup scope, random slots of THE WORKGROUP'S XCD's eighth of the table (XCC_ID read
//                  from the hardware register): does an atomic that may stay in the XCD's L2 run faster?
//   min_agent_rand / min_agent_seq   64-bit minimum at agent scope, random words / consecutive words per wave
//   min_wg_own     64-bit minimum at workgroup scope, random words of the XCD's eighth
//   load_sc1       8-byte agent-scope loads of random slots (the probe in front of a CAS)
//   hipcc --offload-arch=gfx950 -O3 -o atomic_scatter atomic_scatter.hip && ./atomic_scatter [log2 slots = 25] [millions of ops = 8]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef unsigned long long u64;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ u64 mix64(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
__device__ __forceinline__ uint32_t xcc_id() {
    // hwreg(HW_REG_XCC_ID = 20), bits 3:0
    return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;
}

constexpr int U = 8;
enum Mode { CAS_AGENT, CAS_WG_OWN, MIN_AGENT_RAND, MIN_AGENT_SEQ, MIN_WG_OWN, LOAD_SC1 };

template <int MODE>
__global__ __launch_bounds__(256) void k(u64* tab, u64 mask, u64 n, u64 salt, u64* sink, uint32_t* xcc_seen) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x, stride = (u64)gridDim.x * blockDim.x;
    const uint32_t xcc = xcc_id() & 7u;
    if (threadIdx.x == 0) atomicAdd(&xcc_seen[xcc], 1u);
    const u64 part = (mask + 1) >> 3;
    u64 acc = 0;
    for (u64 i0 = gid; i0 < n; i0 += (u64)U * stride) {
        u64 r[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u64 i = i0 + (u64)u * stride;
            const u64 h = mix64(i ^ salt);
            u64 slot = h & mask;
            if (MODE == CAS_WG_OWN || MODE == MIN_WG_OWN) slot = (u64)xcc * part + (h & (part - 1));
            if (MODE == MIN_AGENT_SEQ) slot = i & mask;
            r[u] = 0;
            if (i >= n) continue;
            if (MODE == CAS_AGENT) {
                u64 exp = ~0ull;
                __hip_atomic_compare_exchange_strong(&tab[slot], &exp, h | 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                r[u] = exp;
            } else if (MODE == CAS_WG_OWN) {
                u64 exp = ~0ull;
                __hip_atomic_compare_exchange_strong(&tab[slot], &exp, h | 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                r[u] = exp;
            } else if (MODE == MIN_AGENT_RAND || MODE == MIN_AGENT_SEQ) {
                r[u] = __hip_atomic_fetch_min(&tab[slot], h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (MODE == MIN_WG_OWN) {
                r[u] = __hip_atomic_fetch_min(&tab[slot], h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                r[u] = __hip_atomic_load(&tab[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += r[u];
    }
    if (acc == 0x123456789abcdefull) sink[0] = acc;
}

template <int MODE>
int run(const char* name, u64* tab, u64 slots, u64 n, u64* sink, uint32_t* xcc_seen) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CHK(hipMemset(tab, 0xFF, slots * 8));
        CHK(hipMemset(xcc_seen, 0, 32));
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(a));
        hipLaunchKernelGGL(k<MODE>, dim3(4096), dim3(256), 0, 0, tab, slots - 1, n, 0x9E3779B97F4A7C15ull * (rep + 1), sink, xcc_seen);
        CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    uint32_t seen[8];
    CHK(hipMemcpy(seen, xcc_seen, 32, hipMemcpyDeviceToHost));
    printf("%-16s %8.3f ms  %7.2f G ops/s   (workgroups per XCC_ID: %u %u %u %u %u %u %u %u)\n", name, best, n / best * 1e-6,
           seen[0], seen[1], seen[2], seen[3], seen[4], seen[5], seen[6], seen[7]);
    return 0;
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 25;
    const u64 slots = 1ull << lg, n = (u64)((argc > 2 ? atof(argv[2]) : 8.0) * 1e6);
    u64 *tab, *sink; uint32_t* xs;
    CHK(hipMalloc(&tab, slots * 8)); CHK(hipMalloc(&sink, 8)); CHK(hipMalloc(&xs, 32));
    printf("table 2^%d slots = %.0f MB, %.1f M operations, 4096 x 256 threads, %d in flight per thread\n", lg, slots * 8 / 1e6, n / 1e6, U);
    if (run<LOAD_SC1>("load_sc1", tab, slots, n, sink, xs)) return 1;
    if (run<CAS_AGENT>("cas_agent", tab, slots, n, sink, xs)) return 1;
    if (run<CAS_WG_OWN>("cas_wg_own", tab, slots, n, sink, xs)) return 1;
    if (run<MIN_AGENT_RAND>("min_agent_rand", tab, slots, n, sink, xs)) return 1;
    if (run<MIN_AGENT_SEQ>("min_agent_seq", tab, slots, n, sink, xs)) return 1;
    if (run<MIN_WG_OWN>("min_wg_own", tab, slots, n, sink, xs)) return 1;
    return 0;
}
