// Micro-benchmark: how fast does one SIMD issue MD5-shaped dependent integer chains, by waves per SIMD and by the
// number of independent chains a wave interleaves?  (Why md5_kernel sits at 0.46 of the vector issue peak.)
//   hipcc --offload-arch=gfx950 -O3 -o valu_chain valu_chain.hip && ./valu_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t rotl(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }

// 16 MD5 round-1 steps with literal constants (the shape of md5_block), on C independent states
template <int C>
__global__ __launch_bounds__(256) void md5ish(uint32_t* out, uint32_t iters, uint32_t mval) {
    uint32_t a[C], b[C], c[C], d[C];
#pragma unroll
    for (int i = 0; i < C; i++) { a[i] = threadIdx.x + i; b[i] = blockIdx.x * 7 + i; c[i] = 0x98badcfeu + i; d[i] = 0x10325476u ^ i; }
    for (uint32_t it = 0; it < iters; it++) {
#define STEP(A, B, Cc, D, K, S, M) _Pragma("unroll") for (int i = 0; i < C; i++) { A[i] += (D[i] ^ (B[i] & (Cc[i] ^ D[i]))) + K + M; A[i] = B[i] + rotl(A[i], S); }
        STEP(a, b, c, d, 0xd76aa478u, 7, 0) STEP(d, a, b, c, 0xe8c7b756u, 12, mval) STEP(c, d, a, b, 0x242070dbu, 17, 0) STEP(b, c, d, a, 0xc1bdceeeu, 22, mval)
        STEP(a, b, c, d, 0xf57c0fafu, 7, 0) STEP(d, a, b, c, 0x4787c62au, 12, mval) STEP(c, d, a, b, 0xa8304613u, 17, 0) STEP(b, c, d, a, 0xfd469501u, 22, mval)
        STEP(a, b, c, d, 0x698098d8u, 7, 0) STEP(d, a, b, c, 0x8b44f7afu, 12, mval) STEP(c, d, a, b, 0xffff5bb1u, 17, 0) STEP(b, c, d, a, 0x895cd7beu, 22, mval)
        STEP(a, b, c, d, 0x6b901122u, 7, 0) STEP(d, a, b, c, 0xfd987193u, 12, mval) STEP(c, d, a, b, 0xa679438eu, 17, 0) STEP(b, c, d, a, 0x49b40821u, 22, mval)
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < C; i++) r ^= a[i] ^ b[i] ^ c[i] ^ d[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// the same steps with the three-operand add split into two two-operand adds (kept apart by empty asm statements): does
// a stream of mostly 2-cycle instructions around the one 4-cycle rotate issue faster than bitop3 / add3 / alignbit / add?
template <int C>
__global__ __launch_bounds__(256) void md5ish_split(uint32_t* out, uint32_t iters, uint32_t mval) {
    uint32_t a[C], b[C], c[C], d[C];
#pragma unroll
    for (int i = 0; i < C; i++) { a[i] = threadIdx.x + i; b[i] = blockIdx.x * 7 + i; c[i] = 0x98badcfeu + i; d[i] = 0x10325476u ^ i; }
    for (uint32_t it = 0; it < iters; it++) {
#define STEP2(A, B, Cc, D, K, S, M) _Pragma("unroll") for (int i = 0; i < C; i++) { uint32_t f = __builtin_amdgcn_bitop3_b32(B[i], Cc[i], D[i], 0xCA); \
            uint32_t t = A[i] + (K + M); asm volatile("" : "+v"(t)); t += f; asm volatile("" : "+v"(t)); A[i] = B[i] + rotl(t, S); }
        STEP2(a, b, c, d, 0xd76aa478u, 7, 0) STEP2(d, a, b, c, 0xe8c7b756u, 12, mval) STEP2(c, d, a, b, 0x242070dbu, 17, 0) STEP2(b, c, d, a, 0xc1bdceeeu, 22, mval)
        STEP2(a, b, c, d, 0xf57c0fafu, 7, 0) STEP2(d, a, b, c, 0x4787c62au, 12, mval) STEP2(c, d, a, b, 0xa8304613u, 17, 0) STEP2(b, c, d, a, 0xfd469501u, 22, mval)
        STEP2(a, b, c, d, 0x698098d8u, 7, 0) STEP2(d, a, b, c, 0x8b44f7afu, 12, mval) STEP2(c, d, a, b, 0xffff5bb1u, 17, 0) STEP2(b, c, d, a, 0x895cd7beu, 22, mval)
        STEP2(a, b, c, d, 0x6b901122u, 7, 0) STEP2(d, a, b, c, 0xfd987193u, 12, mval) STEP2(c, d, a, b, 0xa679438eu, 17, 0) STEP2(b, c, d, a, 0x49b40821u, 22, mval)
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < C; i++) r ^= a[i] ^ b[i] ^ c[i] ^ d[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// plain dependent v_add_u32 chain (VOP2, 4-byte encodings) on C independent accumulators
template <int C>
__global__ __launch_bounds__(256) void addchain(uint32_t* out, uint32_t iters, uint32_t inc) {
    uint32_t a[C];
#pragma unroll
    for (int i = 0; i < C; i++) a[i] = threadIdx.x + i;
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 64; r++)
#pragma unroll
            for (int i = 0; i < C; i++) a[i] = (a[i] ^ inc) + (a[i] >> 3);      // 3 dependent ops
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < C; i++) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// f32 FMA chain and a pure VOP2 integer chain (v_add_u32 / v_xor_b32, 4-byte encodings), same shape
template <int C>
__global__ __launch_bounds__(256) void fmachain(uint32_t* out, uint32_t iters, float inc) {
    float a[C];
#pragma unroll
    for (int i = 0; i < C; i++) a[i] = threadIdx.x + i;
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 128; r++)
#pragma unroll
            for (int i = 0; i < C; i++) a[i] = __builtin_fmaf(a[i], inc, 0.5f);
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < C; i++) r += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(r);
}
template <int C>
__global__ __launch_bounds__(256) void vop2chain(uint32_t* out, uint32_t iters, uint32_t inc) {
    uint32_t a[C];
#pragma unroll
    for (int i = 0; i < C; i++) a[i] = threadIdx.x + i;
    const uint32_t t = threadIdx.x * 2654435761u;
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 64; r++)
#pragma unroll
            for (int i = 0; i < C; i++) { a[i] += t; a[i] ^= inc; }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < C; i++) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
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
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    uint32_t* out;
    CK(hipMalloc(&out, 256 * 8192 * 4));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    const uint32_t iters = 4000;
    // waves per SIMD w: one 256-thread workgroup = 1 wave on each of the 4 SIMDs of a CU; w workgroups per CU
    for (int w : {1, 2, 4, 8}) {
        const int grid = cus * w;
        auto report = [&](const char* name, int chains, float ms, double ops_per_iter_per_chain) {
            const double instr = (double)iters * ops_per_iter_per_chain * chains * w;      // wave-instructions per SIMD
            const double cyc = ms * 1e-3 * 2.4e9;
            printf("%-10s waves/SIMD %d chains %d: %8.3f ms  %.2f cycles per wave-instruction per SIMD (2.4 GHz), %.1f cycles per step per wave\n",
                   name, w, chains, ms, cyc / instr, cyc / (iters * 16.0));
        };
        // per md5ish iteration and chain: 16 steps x (bitop3, add3, alignbit, add) = 64 (K + m is folded on the scalar side); addchain: 64 x (xad, lshr) = 128
        report("md5ish", 1, time_ms([&] { md5ish<1><<<grid, 256>>>(out, iters, 0x3FF00000u); }), 64);
        report("md5ish", 2, time_ms([&] { md5ish<2><<<grid, 256>>>(out, iters, 0x3FF00000u); }), 64);
        report("md5ish", 4, time_ms([&] { md5ish<4><<<grid, 256>>>(out, iters, 0x3FF00000u); }), 64);
        report("md5split", 1, time_ms([&] { md5ish_split<1><<<grid, 256>>>(out, iters, 0x3FF00000u); }), 80);
        report("md5split", 2, time_ms([&] { md5ish_split<2><<<grid, 256>>>(out, iters, 0x3FF00000u); }), 80);
        report("addchain", 1, time_ms([&] { addchain<1><<<grid, 256>>>(out, iters, 0x9E3779B9u); }), 128);
        report("addchain", 2, time_ms([&] { addchain<2><<<grid, 256>>>(out, iters, 0x9E3779B9u); }), 128);
        report("addchain", 4, time_ms([&] { addchain<4><<<grid, 256>>>(out, iters, 0x9E3779B9u); }), 128);
        report("fma_f32", 1, time_ms([&] { fmachain<1><<<grid, 256>>>(out, iters, 1.0001f); }), 128);
        report("fma_f32", 4, time_ms([&] { fmachain<4><<<grid, 256>>>(out, iters, 1.0001f); }), 128);
        report("vop2_int", 1, time_ms([&] { vop2chain<1><<<grid, 256>>>(out, iters, 0x9E3779B9u); }), 128);
        report("vop2_int", 4, time_ms([&] { vop2chain<4><<<grid, 256>>>(out, iters, 0x9E3779B9u); }), 128);
    }
    hipFree(out);
    return 0;
}
