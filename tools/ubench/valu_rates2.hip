// Micro-benchmark (round 2): issue rate of the instructions the Fermat / Goldilocks / lazy-Shoup NTT arithmetic is
// built from (shifts, sub-dword SDWA operands, DPP operands, carry chains, 24-bit multiplies, 64-bit shifts).
// Each kernel runs ITER iterations of 8 independent dependency chains of ONE instruction per lane.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates2 valu_rates2.hip ; run: ./valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 2048

#define CHAIN8(ASMSTR, CLOB)                                          \
    asm volatile(ASMSTR : "+v"(a0) : "v"(b), "v"(c) : CLOB);          \
    asm volatile(ASMSTR : "+v"(a1) : "v"(b), "v"(c) : CLOB);          \
    asm volatile(ASMSTR : "+v"(a2) : "v"(b), "v"(c) : CLOB);          \
    asm volatile(ASMSTR : "+v"(a3) : "v"(b), "v"(c) : CLOB);          \
    asm volatile(ASMSTR : "+v"(a4) : "v"(b), "v"(c) : CLOB);          \
    asm volatile(ASMSTR : "+v"(a5) : "v"(b), "v"(c) : CLOB);          \
    asm volatile(ASMSTR : "+v"(a6) : "v"(b), "v"(c) : CLOB);          \
    asm volatile(ASMSTR : "+v"(a7) : "v"(b), "v"(c) : CLOB);

#define KERNEL32(NAME, ASMSTR)                                                                        \
    __global__ void NAME(unsigned *out, unsigned seed)                                                \
    {                                                                                                 \
        unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;          \
        unsigned a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;              \
        unsigned b = seed * 2654435761u + threadIdx.x, c = b ^ 0x5a5a5;                               \
        for (int i = 0; i < ITER; i++) { CHAIN8(ASMSTR, "vcc") }                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;           \
    }

#define KERNEL64(NAME, ASMSTR)                                                                        \
    __global__ void NAME(unsigned *out, unsigned seed)                                                \
    {                                                                                                 \
        unsigned long long a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3; \
        unsigned long long a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;    \
        unsigned b = seed * 2654435761u + threadIdx.x, c = b ^ 0x5a5a5;                               \
        for (int i = 0; i < ITER; i++) { CHAIN8(ASMSTR, "vcc") }                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7); \
    }

KERNEL32(k_add, "v_add_u32 %0, %0, %1")
KERNEL32(k_sub, "v_sub_u32 %0, %0, %1")
KERNEL32(k_and, "v_and_b32 %0, %0, %1")
KERNEL32(k_lshl, "v_lshlrev_b32 %0, 3, %0")
KERNEL32(k_lshlv, "v_lshlrev_b32 %0, %1, %0")
KERNEL32(k_lshr, "v_lshrrev_b32 %0, 3, %0")
KERNEL32(k_ashr, "v_ashrrev_i32 %0, 3, %0")
KERNEL32(k_max, "v_max_u32 %0, %0, %1")
KERNEL32(k_mini, "v_min_i32 %0, %0, %1")
KERNEL32(k_mov, "v_mov_b32 %0, %1")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cmp, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL32(k_addco, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL32(k_addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL32(k_subco, "v_sub_co_u32 %0, vcc, %0, %1")
KERNEL32(k_subb, "v_subb_co_u32 %0, vcc, %0, %1, vcc")
KERNEL32(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL32(k_add_lshl, "v_add_lshl_u32 %0, %0, %1, 3")
KERNEL32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(k_xad, "v_xad_u32 %0, %0, %1, %2")
KERNEL32(k_bfi, "v_bfi_b32 %0, %0, %1, %2")
KERNEL32(k_mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
KERNEL32(k_mul_i24, "v_mul_i32_i24 %0, %0, %1")
KERNEL32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mul_hi_i, "v_mul_hi_i32 %0, %0, %1")
KERNEL32(k_sub_sdwa, "v_sub_u32_sdwa %0, %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1")
KERNEL32(k_add_sdwa_b, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
KERNEL32(k_add_dpp_row, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_add_dpp_quad, "v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL32(k_mov_dpp_bcast, "v_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xf bank_mask:0xf")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 24")
KERNEL32(k_alignbitv, "v_alignbit_b32 %0, %0, %1, %2")
KERNEL32(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
KERNEL32(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL32(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
KERNEL32(k_pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
KERNEL32(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
KERNEL32(k_cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
KERNEL32(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL32(k_rndne_f32, "v_rndne_f32 %0, %0")
KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_readlane, "v_readlane_b32 s20, %0, 3\n v_add_u32 %0, s20, %0")
KERNEL32(k_swizzle, "ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,16)\n s_waitcnt lgkmcnt(0)")

KERNEL64(k_lshl64, "v_lshlrev_b64 %0, 3, %0")
KERNEL64(k_lshl64v, "v_lshlrev_b64 %0, %1, %0")
KERNEL64(k_lshr64, "v_lshrrev_b64 %0, 3, %0")
KERNEL64(k_ashr64, "v_ashrrev_i64 %0, 3, %0")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %0")
KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL64(k_mad_i64_i32, "v_mad_i64_i32 %0, vcc, %1, %2, %0")
KERNEL64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %0")
KERNEL64(k_add_f64, "v_add_f64 %0, %0, %0")
KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %0, %0")

// 64-bit add with carry written as a pair (what a Goldilocks butterfly is made of)
__global__ void k_add64_pair(unsigned *out, unsigned seed)
{
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    unsigned a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    unsigned b = seed * 2654435761u + threadIdx.x, c = b ^ 0x5a5a5;
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a0), "+v"(a1) : "v"(b), "v"(c) : "vcc");
        asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a4), "+v"(a5) : "v"(b), "v"(c) : "vcc");
        asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

// LDS store / load rates in the exchange patterns of the register NTT (b32, conflict-free, 16 ops per wait)
template <int W>
__global__ void k_lds_rw(unsigned *out, unsigned seed)
{
    extern __shared__ unsigned lds[];
    unsigned v[16];
    for (int j = 0; j < 16; j++) v[j] = threadIdx.x * 17 + j + seed;
    unsigned acc = 0;
    for (int i = 0; i < ITER / 16; i++) {
        if (W & 1) {
#pragma unroll
            for (int j = 0; j < 16; j++) lds[j * 1024 + threadIdx.x] = v[j] + i;
        }
        __syncthreads();
        if (W & 2) {
#pragma unroll
            for (int j = 0; j < 16; j++) acc += lds[j * 1024 + ((threadIdx.x + 64 * j) & 1023)];
        }
        __syncthreads();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

struct Entry { const char *name; void (*fn)(unsigned *, unsigned); double ops_per_iter; size_t lds; int threads; int blocks; };

int main()
{
    unsigned *d;
    hipMalloc(&d, sizeof(unsigned) * 2048 * 1024);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
#define E32(N, K) {N, K, 8, 0, 256, 2048}
    std::vector<Entry> es = {
        E32("v_add_u32", k_add), E32("v_sub_u32", k_sub), E32("v_and_b32", k_and), E32("v_lshlrev_b32 imm", k_lshl),
        E32("v_lshlrev_b32 vgpr", k_lshlv), E32("v_lshrrev_b32", k_lshr), E32("v_ashrrev_i32", k_ashr), E32("v_max_u32", k_max),
        E32("v_min_i32", k_mini), E32("v_mov_b32", k_mov), E32("v_cndmask_b32", k_cndmask), E32("v_cmp_lt_u32", k_cmp),
        E32("v_add_co_u32", k_addco), E32("v_addc_co_u32", k_addc), E32("v_sub_co_u32", k_subco), E32("v_subb_co_u32", k_subb),
        E32("v_add3_u32", k_add3), E32("v_lshl_add_u32", k_lshl_add), E32("v_add_lshl_u32", k_add_lshl), E32("v_and_or_b32", k_and_or),
        E32("v_xad_u32", k_xad), E32("v_bfi_b32", k_bfi), E32("v_mad_i32_i24", k_mad_i24), E32("v_mul_i32_i24", k_mul_i24),
        E32("v_mul_lo_u32", k_mul_lo), E32("v_mul_hi_u32", k_mul_hi), E32("v_mul_hi_i32", k_mul_hi_i),
        E32("v_sub_u32_sdwa W0-sext(W1)", k_sub_sdwa), E32("v_add_u32_sdwa src1 W1", k_add_sdwa_b),
        E32("v_add_u32_dpp row_shr:1", k_add_dpp_row), E32("v_add_u32_dpp quad_perm", k_add_dpp_quad), E32("v_mov_b32_dpp row_bcast15", k_mov_dpp_bcast),
        E32("v_alignbit_b32 imm", k_alignbit), E32("v_alignbit_b32 vgpr", k_alignbitv),
        E32("v_pk_add_u16", k_pk_add_u16), E32("v_pk_mul_lo_u16", k_pk_mul_lo_u16), E32("v_pk_mad_u16", k_pk_mad_u16), E32("v_pk_sub_i16", k_pk_sub_i16),
        E32("v_cvt_f32_u32", k_cvt_f32_u32), E32("v_cvt_u32_f32", k_cvt_u32_f32), E32("v_mul_f32", k_mul_f32), E32("v_rndne_f32", k_rndne_f32), E32("v_fma_f32", k_fma_f32),
        E32("v_lshlrev_b64 imm", k_lshl64), E32("v_lshlrev_b64 vgpr", k_lshl64v), E32("v_lshrrev_b64", k_lshr64), E32("v_ashrrev_i64", k_ashr64),
        E32("v_lshl_add_u64", k_lshl_add_u64), E32("v_mad_u64_u32", k_mad_u64_u32), E32("v_mad_i64_i32", k_mad_i64_i32),
        E32("v_pk_add_f32", k_pk_add_f32), E32("v_add_f64", k_add_f64), E32("v_fma_f64", k_fma_f64),
        {"add_co+addc pair (per instr)", k_add64_pair, 8, 0, 256, 2048},
        {"lds write b32 only (per op)", k_lds_rw<1>, 1, 65536, 1024, 512}, {"lds read b32 only (per op)", k_lds_rw<2>, 1, 65536, 1024, 512},
        {"lds write+read b32 (per pair)", k_lds_rw<3>, 1, 65536, 1024, 512},
    };
    for (auto &e : es) {
        if (e.lds) hipFuncSetAttribute((const void *)e.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipLaunchKernelGGL(e.fn, dim3(e.blocks), dim3(e.threads), e.lds, 0, d, 1u);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%-34s FAILED\n", e.name); (void)hipGetLastError(); continue; }
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, 0);
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(e.fn, dim3(e.blocks), dim3(e.threads), e.lds, 0, d, (unsigned)r);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        double lane_ops = (double)e.blocks * e.threads * ITER * e.ops_per_iter;
        double per_cu_clk = lane_ops / (ms * 1e-3) / prop.multiProcessorCount / (prop.clockRate * 1e3);
        printf("%-34s %8.3f ms  %8.2f Tlane-op/s  %6.1f lane-ops/clk/CU\n", e.name, ms, lane_ops / (ms * 1e-3) / 1e12, per_cu_clk);
    }
    return 0;
}
