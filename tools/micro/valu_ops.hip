// Micro-benchmark: issue cost of the integer vector instructions the panfeed kernels are made of, on one SIMD of
// gfx950: cycles per wave-instruction with 1 and 8 waves per SIMD, one dependent chain and four independent ones.
//   hipcc --offload-arch=gfx950 -O3 -o valu_ops valu_ops.hip && ./valu_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

// 32-bit accumulators: %0 accumulator, %1 / %2 other vector registers, %3 a scalar
#define KERNEL32(NAME, TXT)                                                                                             \
    __global__ __launch_bounds__(256) void k1_##NAME(uint32_t* out, uint32_t iters, uint32_t x, uint32_t s) {           \
        __shared__ uint32_t lds_buf[256 * 4]; lds_buf[threadIdx.x] = x;                                                \
        uint32_t a0 = threadIdx.x, xv = x ^ threadIdx.x, yv = threadIdx.x * 16;                                         \
        for (uint32_t it = 0; it < iters; it++) {                                                                       \
            REP32(asm volatile(TXT : "+v"(a0) : "v"(xv), "v"(yv), "s"(s) : "vcc", "scc", "s20", "s21", "v40", "v41", "v42", "v43", "v44", "v45", "memory");                                      \
                  asm volatile(TXT : "+v"(a0) : "v"(xv), "v"(yv), "s"(s) : "vcc", "scc", "s20", "s21", "v40", "v41", "v42", "v43", "v44", "v45", "memory");                                      \
                  asm volatile(TXT : "+v"(a0) : "v"(xv), "v"(yv), "s"(s) : "vcc", "scc", "s20", "s21", "v40", "v41", "v42", "v43", "v44", "v45", "memory");                                      \
                  asm volatile(TXT : "+v"(a0) : "v"(xv), "v"(yv), "s"(s) : "vcc", "scc", "s20", "s21", "v40", "v41", "v42", "v43", "v44", "v45", "memory");)                                     \
        }                                                                                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + lds_buf[(threadIdx.x * 7) & 1023];                                   \
    }                                                                                                                   \
    __global__ __launch_bounds__(256) void k4_##NAME(uint32_t* out, uint32_t iters, uint32_t x, uint32_t s) {           \
        __shared__ uint32_t lds_buf[256 * 4]; lds_buf[threadIdx.x] = x;                                                \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, xv = x ^ threadIdx.x, yv = threadIdx.x * 16;  \
        for (uint32_t it = 0; it < iters; it++) {                                                                       \
            REP32(asm volatile(TXT : "+v"(a0) : "v"(xv), "v"(yv), "s"(s) : "vcc", "scc", "s20", "s21", "v40", "v41", "v42", "v43", "v44", "v45", "memory");                                      \
                  asm volatile(TXT : "+v"(a1) : "v"(xv), "v"(yv), "s"(s) : "vcc", "scc", "s20", "s21", "v40", "v41", "v42", "v43", "v44", "v45", "memory");                                      \
                  asm volatile(TXT : "+v"(a2) : "v"(xv), "v"(yv), "s"(s) : "vcc", "scc", "s20", "s21", "v40", "v41", "v42", "v43", "v44", "v45", "memory");                                      \
                  asm volatile(TXT : "+v"(a3) : "v"(xv), "v"(yv), "s"(s) : "vcc", "scc", "s20", "s21", "v40", "v41", "v42", "v43", "v44", "v45", "memory");)                                     \
        }                                                                                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
        out[blockIdx.x * 256 + threadIdx.x] = (a0 ^ a1 ^ a2 ^ a3) + lds_buf[(threadIdx.x * 7) & 1023];                  \
    }
// 64-bit accumulators (register pairs)
#define KERNEL64(NAME, TXT)                                                                                             \
    __global__ __launch_bounds__(256) void k1_##NAME(uint32_t* out, uint32_t iters, uint32_t x, uint32_t s) {           \
        uint64_t a0 = threadIdx.x; uint32_t xv = x ^ threadIdx.x, yv = x + threadIdx.x * 3;                             \
        for (uint32_t it = 0; it < iters; it++) {                                                                       \
            REP32(asm volatile(TXT : "+v"(a0) : "v"(xv), "v"(yv), "s"(s) : "vcc", "scc", "s20", "s21", "v40", "v41", "v42", "v43", "v44", "v45", "memory");                                      \
                  asm volatile(TXT : "+v"(a0) : "v"(xv), "v"(yv), "s"(s) : "vcc", "scc", "s20", "s21", "v40", "v41", "v42", "v43", "v44", "v45", "memory");                                      \
                  asm volatile(TXT : "+v"(a0) : "v"(xv), "v"(yv), "s"(s) : "vcc", "scc", "s20", "s21", "v40", "v41", "v42", "v43", "v44", "v45", "memory");                                      \
                  asm volatile(TXT : "+v"(a0) : "v"(xv), "v"(yv), "s"(s) : "vcc", "scc", "s20", "s21", "v40", "v41", "v42", "v43", "v44", "v45", "memory");)                                     \
        }                                                                                                               \
        out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ (a0 >> 32));                                              \
    }                                                                                                                   \
    __global__ __launch_bounds__(256) void k4_##NAME(uint32_t* out, uint32_t iters, uint32_t x, uint32_t s) {           \
        uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3; uint32_t xv = x ^ threadIdx.x, yv = x + threadIdx.x * 3; \
        for (uint32_t it = 0; it < iters; it++) {                                                                       \
            REP32(asm volatile(TXT : "+v"(a0) : "v"(xv), "v"(yv), "s"(s) : "vcc", "scc", "s20", "s21", "v40", "v41", "v42", "v43", "v44", "v45", "memory");                                      \
                  asm volatile(TXT : "+v"(a1) : "v"(xv), "v"(yv), "s"(s) : "vcc", "scc", "s20", "s21", "v40", "v41", "v42", "v43", "v44", "v45", "memory");                                      \
                  asm volatile(TXT : "+v"(a2) : "v"(xv), "v"(yv), "s"(s) : "vcc", "scc", "s20", "s21", "v40", "v41", "v42", "v43", "v44", "v45", "memory");                                      \
                  asm volatile(TXT : "+v"(a3) : "v"(xv), "v"(yv), "s"(s) : "vcc", "scc", "s20", "s21", "v40", "v41", "v42", "v43", "v44", "v45", "memory");)                                     \
        }                                                                                                               \
        const uint64_t r = a0 ^ a1 ^ a2 ^ a3;                                                                           \
        out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(r ^ (r >> 32));                                                \
    }

KERNEL32(add_u32, "v_add_u32 %0, %0, %1")
KERNEL32(xor_b32, "v_xor_b32 %0, %0, %1")
KERNEL32(and_b32, "v_and_b32 %0, %0, %1")
KERNEL32(mov_b32, "v_mov_b32 %0, %1")
KERNEL32(not_b32, "v_not_b32 %0, %0")
KERNEL32(bfrev_b32, "v_bfrev_b32 %0, %0")
KERNEL32(lshlrev_b32, "v_lshlrev_b32 %0, 1, %0")
KERNEL32(lshrrev_b32_v, "v_lshrrev_b32 %0, %1, %0")
KERNEL32(min_u32, "v_min_u32 %0, %0, %1")
KERNEL32(sub_u32, "v_sub_u32 %0, %0, %1")
KERNEL32(bcnt_u32, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL32(ffbh_u32, "v_ffbh_u32 %0, %0")
KERNEL32(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(cmp_lt_u32, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL32(cmp_e64_sgpr, "v_cmp_lt_u32 s[20:21], %0, %1")
KERNEL32(add_co_u32, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL32(addc_co_u32, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL32(alignbit, "v_alignbit_b32 %0, %0, %0, 7")
KERNEL32(bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
KERNEL32(add3_u32, "v_add3_u32 %0, %0, %1, %2")
KERNEL32(add3_lit, "v_add3_u32 %0, %0, %1, %3")
KERNEL32(add_lit, "v_add_u32 %0, 0x12345678, %0")
KERNEL32(xad_u32, "v_xad_u32 %0, %0, %1, %2")
KERNEL32(lshl_add_u32, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL32(lshl_or_b32, "v_lshl_or_b32 %0, %0, 2, %1")
KERNEL32(and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(or3_b32, "v_or3_b32 %0, %0, %1, %2")
KERNEL32(bfe_u32, "v_bfe_u32 %0, %0, 3, 8")
KERNEL32(bfe_i32, "v_bfe_i32 %0, %0, 3, 1")
KERNEL32(bfi_b32, "v_bfi_b32 %0, %1, %0, %2")
KERNEL32(perm_b32, "v_perm_b32 %0, %0, %1, %2")
KERNEL32(mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
KERNEL32(mul_lo_sgpr, "v_mul_lo_u32 %0, %0, %3")
KERNEL32(readfirstlane, "v_readfirstlane_b32 s20, %0")
KERNEL32(readlane, "v_readlane_b32 s20, %0, 5")
KERNEL32(writelane, "v_writelane_b32 %0, %3, 5")
KERNEL32(mbcnt_lo, "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL32(mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(add_dpp, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(cmp_sdwa, "v_cmp_eq_u32_sdwa vcc, %0, %1 src0_sel:WORD_1 src1_sel:DWORD")
KERNEL32(pk_add_u16, "v_pk_add_u16 %0, %0, %1")
KERNEL32(cndmask_e64, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL32(cmp_cndmask, "v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
KERNEL32(cmp64_cndmask, "v_cmp_lt_u32 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]")
KERNEL32(cmp_x_cndmask, "v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc")
KERNEL32(or_b32, "v_or_b32 %0, %0, %1")
KERNEL32(max_u32, "v_max_u32 %0, %0, %1")
KERNEL32(lshrrev_b32_c, "v_lshrrev_b32 %0, 3, %0")
KERNEL32(lshlrev_b32_v, "v_lshlrev_b32 %0, %1, %0")
KERNEL32(ashrrev_i32, "v_ashrrev_i32 %0, 3, %0")
KERNEL32(add_sgpr, "v_add_u32 %0, %3, %0")
KERNEL32(and_lit, "v_and_b32 %0, 0x55555555, %0")
KERNEL32(subrev_u32, "v_subrev_u32 %0, %0, %1")
KERNEL32(mul_lo_lit, "v_mul_lo_u32 %0, %0, %3")
KERNEL32(salu_mov, "s_mov_b32 s20, s21")
KERNEL32(s_nop, "s_nop 0")
KERNEL32(ds_read_b32, "ds_read_b32 %1, %2")
KERNEL32(ds_read_b64, "ds_read_b64 v[40:41], %2")
KERNEL32(ds_read_b128, "ds_read_b128 v[40:43], %2")
KERNEL32(ds_write_b32, "ds_write_b32 %2, %1")
KERNEL32(ds_write_b64, "ds_write_b64 %2, v[40:41]")
KERNEL32(ds_or_b32, "ds_or_b32 %2, %1")
KERNEL32(ds_min_u32, "ds_min_u32 %2, %1")
KERNEL32(ds_add_rtn, "ds_add_rtn_u32 %1, %2, %1")
KERNEL32(ds_cmpst_b64, "ds_cmpst_rtn_b64 v[40:41], %2, v[42:43], v[44:45]")
KERNEL32(ds_bpermute, "ds_bpermute_b32 %1, %2, %1")
KERNEL32(ds_swizzle, "ds_swizzle_b32 %1, %1 offset:0x041F")
KERNEL32(pair_add_smov, "v_add_u32 %0, %0, %1\n\ts_mov_b32 s20, s21")
KERNEL32(pair_add3_smov, "v_add3_u32 %0, %0, %1, %2\n\ts_mov_b32 s20, s21")
KERNEL32(pair_add3_2smov, "v_add3_u32 %0, %0, %1, %2\n\ts_mov_b32 s20, s21\n\ts_mov_b32 s21, s20")
KERNEL32(pair_add3_smov_dep, "v_add3_u32 %0, %0, %1, %2\n\ts_add_u32 s20, s20, s21")
KERNEL32(pair_add3_sor64, "v_add3_u32 %0, %0, %1, %2\n\ts_or_b64 s[20:21], s[20:21], exec")
KERNEL32(pair_add3_dsread, "v_add3_u32 %0, %0, %1, %2\n\tds_read_b32 v40, %2")
KERNEL64(lshlrev_b64, "v_lshlrev_b64 %0, 1, %0")
KERNEL64(lshrrev_b64, "v_lshrrev_b64 %0, 1, %0")
KERNEL64(lshrrev_b64_v, "v_lshrrev_b64 %0, %1, %0")
KERNEL64(lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %0")
KERNEL64(mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL64(cmp_lt_u64, "v_cmp_lt_u64 vcc, %0, %0")
KERNEL64(cmp_eq_u64, "v_cmp_eq_u64 vcc, %0, %0")
KERNEL64(mov_b64, "v_mov_b64 %0, %0")
KERNEL64(pk_mov, "v_pk_mov_b32 %0, %0, %0")

typedef void (*kern_t)(uint32_t*, uint32_t, uint32_t, uint32_t);

static float time_ms(kern_t k, int grid, uint32_t* out, uint32_t iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, iters, 12345u, 0x9E3779B1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, iters, 12345u, 0x9E3779B1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

int main() {
    uint32_t* out;
    if (hipMalloc(&out, 256 * 4096 * 4) != hipSuccess) return 1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount;
    const uint32_t iters = 2000;
    printf("%d CUs; cycles per wave-instruction per SIMD at a nominal 2.4 GHz (128 instructions x %u iterations per wave)\n", cus, iters);
    printf("%-16s %10s %10s %10s %10s\n", "instruction", "1w/1chain", "1w/4chain", "8w/1chain", "8w/4chain");
    struct { const char* name; kern_t k1, k4; } ops[] = {
#define OP(N) {#N, k1_##N, k4_##N},
        OP(add_u32) OP(xor_b32) OP(and_b32) OP(mov_b32) OP(not_b32) OP(bfrev_b32) OP(lshlrev_b32) OP(lshrrev_b32_v) OP(min_u32) OP(sub_u32)
        OP(bcnt_u32) OP(ffbh_u32) OP(cndmask) OP(cmp_lt_u32) OP(cmp_e64_sgpr) OP(add_co_u32) OP(addc_co_u32) OP(alignbit) OP(bitop3) OP(add3_u32)
        OP(add3_lit) OP(add_lit) OP(xad_u32) OP(lshl_add_u32) OP(lshl_or_b32) OP(and_or_b32) OP(or3_b32) OP(bfe_u32) OP(bfe_i32) OP(bfi_b32) OP(perm_b32)
        OP(mul_lo_u32) OP(mul_hi_u32) OP(mul_u32_u24) OP(mad_u32_u24) OP(mul_lo_sgpr) OP(readfirstlane) OP(readlane) OP(writelane) OP(mbcnt_lo)
        OP(mov_dpp) OP(add_dpp) OP(cmp_sdwa) OP(pk_add_u16)
        OP(cndmask_e64) OP(cmp_cndmask) OP(cmp64_cndmask) OP(cmp_x_cndmask) OP(or_b32) OP(max_u32) OP(lshrrev_b32_c) OP(lshlrev_b32_v) OP(ashrrev_i32)
        OP(add_sgpr) OP(and_lit) OP(subrev_u32) OP(mul_lo_lit) OP(salu_mov) OP(s_nop)
        OP(ds_read_b32) OP(ds_read_b64) OP(ds_read_b128) OP(ds_write_b32) OP(ds_write_b64) OP(ds_or_b32) OP(ds_min_u32) OP(ds_add_rtn) OP(ds_cmpst_b64)
        OP(ds_bpermute) OP(ds_swizzle)
        OP(pair_add_smov) OP(pair_add3_smov) OP(pair_add3_2smov) OP(pair_add3_smov_dep) OP(pair_add3_sor64) OP(pair_add3_dsread)
        OP(lshlrev_b64) OP(lshrrev_b64) OP(lshrrev_b64_v) OP(lshl_add_u64) OP(mad_u64_u32) OP(cmp_lt_u64) OP(cmp_eq_u64) OP(mov_b64) OP(pk_mov)
    };
    for (auto& op : ops) {
        double c[4];
        int i = 0;
        for (int w : {1, 8})
            for (kern_t k : {op.k1, op.k4}) {
                const float ms = time_ms(k, cus * w, out, iters);
                c[i++] = ms * 1e-3 * 2.4e9 / ((double)iters * 128 * w);
            }
        printf("%-16s %10.2f %10.2f %10.2f %10.2f\n", op.name, c[0], c[1], c[2], c[3]);
        fflush(stdout);
    }
    hipFree(out);
    return 0;
}
