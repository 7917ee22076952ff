// Micro-benchmark: issue rate of the integer / fp instructions the finite-field kernels are built from (gfx950).
// Each kernel runs ITER iterations of 8 independent dependency chains of ONE instruction per lane.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define ITER 2048

#define KERNEL2(NAME, ASMSTR)                                                                         \
    __global__ void NAME(unsigned *out, unsigned seed)                                                \
    {                                                                                                 \
        unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;          \
        unsigned a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;              \
        unsigned b = seed * 2654435761u + threadIdx.x;                                                \
        for (int i = 0; i < ITER; i++) {                                                              \
            asm volatile(ASMSTR : "+v"(a0) : "v"(b));                                                 \
            asm volatile(ASMSTR : "+v"(a1) : "v"(b));                                                 \
            asm volatile(ASMSTR : "+v"(a2) : "v"(b));                                                 \
            asm volatile(ASMSTR : "+v"(a3) : "v"(b));                                                 \
            asm volatile(ASMSTR : "+v"(a4) : "v"(b));                                                 \
            asm volatile(ASMSTR : "+v"(a5) : "v"(b));                                                 \
            asm volatile(ASMSTR : "+v"(a6) : "v"(b));                                                 \
            asm volatile(ASMSTR : "+v"(a7) : "v"(b));                                                 \
        }                                                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;           \
    }

KERNEL2(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL2(k_min_u32, "v_min_u32 %0, %0, %1")
KERNEL2(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL2(k_mul_hi_u24, "v_mul_hi_u32_u24 %0, %0, %1")
KERNEL2(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL2(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
KERNEL2(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %1")
KERNEL2(k_alignbit, "v_alignbit_b32 %0, %0, %1, 24")
KERNEL2(k_perm, "v_perm_b32 %0, %0, %1, %1")
KERNEL2(k_bfe, "v_bfe_u32 %0, %0, 8, 8")
KERNEL2(k_lshl_or, "v_lshl_or_b32 %0, %0, 8, %1")
KERNEL2(k_fma_f32, "v_fma_f32 %0, %0, %1, %1")
KERNEL2(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL2(k_bpermute, "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)")

#define KERNEL64(NAME, ASMSTR)                                                                        \
    __global__ void NAME(unsigned *out, unsigned seed)                                                \
    {                                                                                                 \
        unsigned long long a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3; \
        unsigned long long a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;    \
        unsigned long long b = seed * 2654435761ull + threadIdx.x;                                    \
        for (int i = 0; i < ITER; i++) {                                                              \
            asm volatile(ASMSTR : "+v"(a0) : "v"(b));                                                 \
            asm volatile(ASMSTR : "+v"(a1) : "v"(b));                                                 \
            asm volatile(ASMSTR : "+v"(a2) : "v"(b));                                                 \
            asm volatile(ASMSTR : "+v"(a3) : "v"(b));                                                 \
            asm volatile(ASMSTR : "+v"(a4) : "v"(b));                                                 \
            asm volatile(ASMSTR : "+v"(a5) : "v"(b));                                                 \
            asm volatile(ASMSTR : "+v"(a6) : "v"(b));                                                 \
            asm volatile(ASMSTR : "+v"(a7) : "v"(b));                                                 \
        }                                                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7); \
    }
KERNEL64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %1")
KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %1, %1")
KERNEL64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %1")

__global__ void k_mad_u64_u32(unsigned *out, unsigned seed)
{
    unsigned long long a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15;
    unsigned b = seed * 2654435761u + threadIdx.x, c = b ^ 0x5555;
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a0) : "v"(b), "v"(c) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a1) : "v"(b), "v"(c) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a2) : "v"(b), "v"(c) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a3) : "v"(b), "v"(c) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a4) : "v"(b), "v"(c) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a5) : "v"(b), "v"(c) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a6) : "v"(b), "v"(c) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}

// LDS byte gathers: random (data-dependent) addresses into a 64 KiB table vs a 32x replicated conflict-free layout
__global__ void k_lds_gather_u8(unsigned *out, unsigned seed)
{
    extern __shared__ unsigned char lds[];
    for (int i = threadIdx.x; i < 65536; i += blockDim.x) lds[i] = (unsigned char)(i * 37 + seed);
    __syncthreads();
    unsigned a0 = (threadIdx.x * 2654435761u + seed) & 0xffff, a1 = (a0 * 3 + 1) & 0xffff, a2 = (a0 * 5 + 2) & 0xffff, a3 = (a0 * 7 + 3) & 0xffff;
    unsigned acc = 0;
    for (int i = 0; i < ITER; i++) {
        unsigned r0 = lds[a0], r1 = lds[a1], r2 = lds[a2], r3 = lds[a3];
        acc += r0 + r1 + r2 + r3;
        a0 = (a0 * 1664525u + 1013904223u + r0) & 0xffff; a1 = (a1 * 1664525u + 12345u + r1) & 0xffff;
        a2 = (a2 * 22695477u + 1u + r2) & 0xffff; a3 = (a3 * 1103515245u + 12345u + r3) & 0xffff;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_lds_gather_rep32(unsigned *out, unsigned seed)
{
    extern __shared__ unsigned char ldsraw[];
    unsigned *rep = (unsigned *)ldsraw;
    for (int i = threadIdx.x; i < 256 * 32; i += blockDim.x) rep[i] = (i >> 5) * 37 + seed;
    __syncthreads();
    const unsigned *my = rep + (threadIdx.x & 31);
    unsigned a0 = (threadIdx.x * 2654435761u + seed) & 0xff, a1 = (a0 * 3 + 1) & 0xff, a2 = (a0 * 5 + 2) & 0xff, a3 = (a0 * 7 + 3) & 0xff;
    unsigned acc = 0;
    for (int i = 0; i < ITER; i++) {
        unsigned r0 = my[a0 << 5], r1 = my[a1 << 5], r2 = my[a2 << 5], r3 = my[a3 << 5];
        acc += r0 + r1 + r2 + r3;
        a0 = (a0 * 1664525u + 1013904223u + r0) & 0xff; a1 = (a1 * 1664525u + 12345u + r1) & 0xff;
        a2 = (a2 * 22695477u + 1u + r2) & 0xff; a3 = (a3 * 1103515245u + 12345u + r3) & 0xff;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

struct Entry { const char *name; void (*fn)(unsigned *, unsigned); double ops_per_iter; size_t lds; };

int main()
{
    unsigned *d;
    const int blocks = 256 * 8, threads = 256;
    hipMalloc(&d, sizeof(unsigned) * blocks * 1024);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    std::vector<Entry> es = {
        {"v_add_u32", k_add_u32, 8, 0}, {"v_min_u32", k_min_u32, 8, 0}, {"v_xor_b32", k_xor, 8, 0},
        {"v_mul_u32_u24", k_mul_u24, 8, 0}, {"v_mul_hi_u32_u24", k_mul_hi_u24, 8, 0}, {"v_mad_u32_u24", k_mad_u24, 8, 0},
        {"v_mul_lo_u32", k_mul_lo_u32, 8, 0}, {"v_mul_hi_u32", k_mul_hi_u32, 8, 0}, {"v_mad_u64_u32", k_mad_u64_u32, 8, 0},
        {"v_alignbit_b32", k_alignbit, 8, 0}, {"v_perm_b32", k_perm, 8, 0}, {"v_bfe_u32", k_bfe, 8, 0}, {"v_lshl_or_b32", k_lshl_or, 8, 0},
        {"v_lshl_add_u64", k_lshl_add_u64, 8, 0},
        {"v_fma_f32", k_fma_f32, 8, 0}, {"v_pk_fma_f32", k_pk_fma_f32, 8, 0}, {"v_pk_add_f32", k_pk_add_f32, 8, 0}, {"v_fma_f64", k_fma_f64, 8, 0},
        {"ds_bpermute_b32(+wait)", k_bpermute, 8, 0},
        {"lds u8 gather 64KiB random", k_lds_gather_u8, 4, 65536}, {"lds b32 gather 32x-replicated", k_lds_gather_rep32, 4, 32768},
    };
    for (auto &e : es) {
        int th = e.lds ? 1024 : threads;
        int bl = e.lds ? 256 * 2 : blocks;
        if (e.lds) hipFuncSetAttribute((const void *)e.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipLaunchKernelGGL(e.fn, dim3(bl), dim3(th), e.lds, 0, d, 1u);
        hipDeviceSynchronize();
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, 0);
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(e.fn, dim3(bl), dim3(th), e.lds, 0, d, (unsigned)r);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        double lane_ops = (double)bl * th * ITER * e.ops_per_iter;
        double per_cu_clk = lane_ops / (ms * 1e-3) / prop.multiProcessorCount / (prop.clockRate * 1e3);
        printf("%-34s %8.3f ms  %8.2f Tlane-op/s  %6.1f lane-ops/clk/CU\n", e.name, ms, lane_ops / (ms * 1e-3) / 1e12, per_cu_clk);
    }
    return 0;
}
