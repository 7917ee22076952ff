// Micro-benchmark: which launch structure streams 2 reads + 1 write (out = a ^ b, uint8, 1e8 elements) fastest on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef long long i64;

template <int THREADS, int UNROLL, bool CONTIG>
__global__ __launch_bounds__(THREADS) void k_gridstride(const u32x4 *a, const u32x4 *b, u32x4 *o, i64 nvec)
{
    if (!CONTIG) {
        const i64 stride = (i64)gridDim.x * THREADS;
        i64 i = (i64)blockIdx.x * THREADS + threadIdx.x;
        for (; i + (UNROLL - 1) * stride < nvec; i += UNROLL * stride) {
            u32x4 x[UNROLL], y[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) { x[u] = a[i + u * stride]; y[u] = b[i + u * stride]; }
#pragma unroll
            for (int u = 0; u < UNROLL; u++) o[i + u * stride] = x[u] ^ y[u];
        }
        for (; i < nvec; i += stride) o[i] = a[i] ^ b[i];
    } else {
        // each workgroup walks contiguous chunks of THREADS*UNROLL vectors
        const i64 chunk = (i64)THREADS * UNROLL;
        for (i64 base = (i64)blockIdx.x * chunk; base < nvec; base += (i64)gridDim.x * chunk) {
            u32x4 x[UNROLL], y[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const i64 i = base + u * THREADS + threadIdx.x;
                if (i < nvec) { x[u] = a[i]; y[u] = b[i]; }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const i64 i = base + u * THREADS + threadIdx.x;
                if (i < nvec) o[i] = x[u] ^ y[u];
            }
        }
    }
}

// persistent workgroups that claim chunks from a global counter (dynamic balance); the last one out resets it
template <int THREADS, int UNROLL>
__global__ __launch_bounds__(THREADS) void k_dyn(const u32x4 *a, const u32x4 *b, u32x4 *o, i64 nvec, unsigned *ctr)
{
    __shared__ unsigned s_chunk;
    const i64 chunk = (i64)THREADS * UNROLL;
    const unsigned nchunks = (unsigned)((nvec + chunk - 1) / chunk);
    unsigned c = blockIdx.x;   // first chunk is static
    while (c < nchunks) {
        const i64 base = (i64)c * chunk;
        if (threadIdx.x == 0) s_chunk = atomicAdd(ctr, 1u) + gridDim.x;   // prefetch the next claim
        u32x4 x[UNROLL], y[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const i64 i = base + u * THREADS + threadIdx.x;
            if (i < nvec) { x[u] = a[i]; y[u] = b[i]; }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const i64 i = base + u * THREADS + threadIdx.x;
            if (i < nvec) o[i] = x[u] ^ y[u];
        }
        __syncthreads();
        c = s_chunk;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const unsigned done = atomicAdd(ctr + 1, 1u);
        if (done == gridDim.x - 1) { ctr[0] = 0; ctr[1] = 0; __threadfence(); }
    }
}

typedef unsigned int u32;
__device__ __forceinline__ u32 lookup4(const unsigned char *lds, u32 aw, u32 bw)
{
    u32 i0 = __builtin_amdgcn_perm(aw, bw, 0x0c0c0400u);
    u32 i1 = __builtin_amdgcn_perm(aw, bw, 0x0c0c0501u);
    u32 i2 = __builtin_amdgcn_perm(aw, bw, 0x0c0c0602u);
    u32 i3 = __builtin_amdgcn_perm(aw, bw, 0x0c0c0703u);
    u32 r0 = lds[i0], r1 = lds[i1], r2 = lds[i2], r3 = lds[i3];
    return r0 | (r1 << 8) | (r2 << 16) | (r3 << 24);
}
__device__ __forceinline__ u32x4 lookup16(const unsigned char *lds, u32x4 x, u32x4 y)
{
    u32x4 r;
    r.x = lookup4(lds, x.x, y.x); r.y = lookup4(lds, x.y, y.y);
    r.z = lookup4(lds, x.z, y.z); r.w = lookup4(lds, x.w, y.w);
    return r;
}
// library structure
template <int UNROLL>
__global__ __launch_bounds__(1024) void k_tab(const unsigned char *table, const u32x4 *a, const u32x4 *b, u32x4 *o, i64 nvec)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 4096; i += 1024) ((uint4 *)lds)[i] = ((const uint4 *)table)[i];
    __syncthreads();
    const i64 stride = (i64)gridDim.x * 1024;
    i64 i = (i64)blockIdx.x * 1024 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < nvec; i += UNROLL * stride) {
        u32x4 x[UNROLL], y[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { x[u] = a[i + u * stride]; y[u] = b[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) o[i + u * stride] = lookup16(lds, x[u], y[u]);
    }
    for (; i < nvec; i += stride) o[i] = lookup16(lds, a[i], b[i]);
}
// software-pipelined: DEPTH vectors per operand requested before the table is staged, refilled as consumed
template <int DEPTH>
__global__ __launch_bounds__(1024) void k_tab_pipe(const unsigned char *table, const u32x4 *a, const u32x4 *b, u32x4 *o, i64 nvec)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const i64 stride = (i64)gridDim.x * 1024;
    i64 i = (i64)blockIdx.x * 1024 + threadIdx.x;
    u32x4 x[DEPTH], y[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; u++) {
        const i64 j = i + u * stride;
        if (j < nvec) { x[u] = a[j]; y[u] = b[j]; }
    }
    for (int t = threadIdx.x; t < 4096; t += 1024) ((uint4 *)lds)[t] = ((const uint4 *)table)[t];
    __syncthreads();
    for (; i < nvec; i += DEPTH * stride) {
#pragma unroll
        for (int u = 0; u < DEPTH; u++) {
            const i64 j = i + u * stride;
            if (j < nvec) {
                const u32x4 cx = x[u], cy = y[u];
                const i64 jn = j + DEPTH * stride;
                if (jn < nvec) { x[u] = a[jn]; y[u] = b[jn]; }
                o[j] = lookup16(lds, cx, cy);
            }
        }
    }
}
// bit-serial packed-byte GF(2^8) multiply, no table: r = sum_k b_k * (a x^k mod poly)
__device__ __forceinline__ u32 gf256_mul4(u32 a, u32 b, u32 red)
{
    u32 r = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const u32 m = ((b >> k) & 0x01010101u) * 0xffu;
        r ^= a & m;
        if (k < 7) {
            const u32 hi = (a >> 7) & 0x01010101u;
            a = ((a << 1) & 0xfefefefeu) ^ (hi * red);
        }
    }
    return r;
}
template <int THREADS, bool FLAT>
__global__ __launch_bounds__(THREADS) void k_bits(const u32x4 *a, const u32x4 *b, u32x4 *o, i64 nvec, u32 red)
{
    const i64 stride = FLAT ? nvec : (i64)gridDim.x * THREADS;
    for (i64 i = (i64)blockIdx.x * THREADS + threadIdx.x; i < nvec; i += stride) {
        const u32x4 x = a[i], y = b[i];
        u32x4 r;
        r.x = gf256_mul4(x.x, y.x, red); r.y = gf256_mul4(x.y, y.y, red);
        r.z = gf256_mul4(x.z, y.z, red); r.w = gf256_mul4(x.w, y.w, red);
        o[i] = r;
    }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_flat(const u32x4 *a, const u32x4 *b, u32x4 *o, i64 nvec)
{
    const i64 i = (i64)blockIdx.x * THREADS + threadIdx.x;
    if (i < nvec) o[i] = a[i] ^ b[i];
}

// persistent, but the workgroup -> chunk assignment rotates every iteration (breaks any fixed XCD <-> HBM-channel affinity)
template <int THREADS, int ROT>
__global__ __launch_bounds__(THREADS) void k_rotate(const u32x4 *a, const u32x4 *b, u32x4 *o, i64 nvec)
{
    const int G = gridDim.x;
    int slot = blockIdx.x;
    for (i64 base = 0; base < nvec; base += (i64)G * THREADS) {
        const i64 i = base + (i64)slot * THREADS + threadIdx.x;
        if (i < nvec) o[i] = a[i] ^ b[i];
        slot += ROT;
        if (slot >= G) slot -= G;
    }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_copy(const u32x4 *a, u32x4 *o, i64 nvec)
{
    const i64 stride = (i64)gridDim.x * THREADS;
    for (i64 i = (i64)blockIdx.x * THREADS + threadIdx.x; i < nvec; i += stride) o[i] = a[i];
}

template <typename F>
float timeit(F f)
{
    f(); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; i++) f();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms / 20 < best) best = ms / 20;
    }
    return best;
}

int main(int argc, char **argv)
{
    const i64 n = argc > 1 ? atoll(argv[1]) : 100000000, nvec = n / 16;
    u32x4 *a, *b, *o;
    hipMalloc(&a, n); hipMalloc(&b, n); hipMalloc(&o, n);
    hipMemset(a, 1, n); hipMemset(b, 2, n);
    const int cus = 256;
    unsigned *ctr; hipMalloc(&ctr, 8); hipMemset(ctr, 0, 8);
#define RUN(name, bytes, ...) { float ms = timeit([&]() { __VA_ARGS__; }); printf("%-46s %7.2f us  %6.3f TB/s\n", name, ms * 1e3, (bytes) * (double)n / 1e8 / ms / 1e9); }
    // GF(2^8)/0x11d product table and random operands
    unsigned char *htab = (unsigned char *)malloc(65536), *dtab;
    for (int x = 0; x < 256; x++) for (int y = 0; y < 256; y++) {
        unsigned r = 0, aa = x;
        for (int k = 0; k < 8; k++) { if (y >> k & 1) r ^= aa; aa <<= 1; if (aa & 0x100) aa ^= 0x11d; }
        htab[x * 256 + y] = (unsigned char)r;
    }
    hipMalloc(&dtab, 65536); hipMemcpy(dtab, htab, 65536, hipMemcpyHostToDevice);
    {
        unsigned char *h = (unsigned char *)malloc(n);
        unsigned long long st = 88172645463325252ull;
        for (i64 i = 0; i < n; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h[i] = (unsigned char)(st >> 24); }
        hipMemcpy(a, h, n, hipMemcpyHostToDevice);
        for (i64 i = 0; i < n; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h[i] = (unsigned char)(st >> 24); }
        hipMemcpy(b, h, n, hipMemcpyHostToDevice);
        free(h);
    }
    hipFuncSetAttribute((const void *)k_tab<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)k_tab<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)k_tab_pipe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)k_tab_pipe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)k_tab_pipe<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    u32x4 *o2; hipMalloc(&o2, n);
    RUN("TAB unroll2 2/CU (library)", 3e8, hipLaunchKernelGGL((k_tab<2>), dim3(cus * 2), dim3(1024), 65536, 0, dtab, a, b, o, nvec))
    RUN("TAB unroll1 2/CU", 3e8, hipLaunchKernelGGL((k_tab<1>), dim3(cus * 2), dim3(1024), 65536, 0, dtab, a, b, o, nvec))
    RUN("TAB unroll2 1/CU", 3e8, hipLaunchKernelGGL((k_tab<2>), dim3(cus * 1), dim3(1024), 65536, 0, dtab, a, b, o, nvec))
    RUN("TAB pipelined depth1 2/CU", 3e8, hipLaunchKernelGGL((k_tab_pipe<1>), dim3(cus * 2), dim3(1024), 65536, 0, dtab, a, b, o2, nvec))
    RUN("TAB pipelined depth2 2/CU", 3e8, hipLaunchKernelGGL((k_tab_pipe<2>), dim3(cus * 2), dim3(1024), 65536, 0, dtab, a, b, o2, nvec))
    RUN("TAB pipelined depth3 2/CU", 3e8, hipLaunchKernelGGL((k_tab_pipe<3>), dim3(cus * 2), dim3(1024), 65536, 0, dtab, a, b, o2, nvec))
    RUN("TAB pipelined depth2 1/CU", 3e8, hipLaunchKernelGGL((k_tab_pipe<2>), dim3(cus * 1), dim3(1024), 65536, 0, dtab, a, b, o2, nvec))
    RUN("BITS flat 256thr", 3e8, hipLaunchKernelGGL((k_bits<256, true>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, a, b, o2, nvec, 0x1du))
    RUN("BITS flat 1024thr", 3e8, hipLaunchKernelGGL((k_bits<1024, true>), dim3((unsigned)((nvec + 1023) / 1024)), dim3(1024), 0, 0, a, b, o2, nvec, 0x1du))
    RUN("BITS gridstride 1024thr 2/CU", 3e8, hipLaunchKernelGGL((k_bits<1024, false>), dim3(cus * 2), dim3(1024), 0, 0, a, b, o2, nvec, 0x1du))
    RUN("BITS gridstride 256thr 8/CU", 3e8, hipLaunchKernelGGL((k_bits<256, false>), dim3(cus * 8), dim3(256), 0, 0, a, b, o2, nvec, 0x1du))
    {
        unsigned char *h1 = (unsigned char *)malloc(n), *h2 = (unsigned char *)malloc(n);
        hipMemcpy(h1, o, n, hipMemcpyDeviceToHost); hipMemcpy(h2, o2, n, hipMemcpyDeviceToHost);
        i64 bad = 0; for (i64 i = 0; i < n; i++) bad += h1[i] != h2[i];
        printf("table vs bit-serial mismatches: %lld\n", bad);
    }
    RUN("gridstride 1024thr 2/CU unroll2 (library)", 3e8, hipLaunchKernelGGL((k_gridstride<1024, 2, false>), dim3(cus * 2), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("gridstride 1024thr 2/CU unroll4", 3e8, hipLaunchKernelGGL((k_gridstride<1024, 4, false>), dim3(cus * 2), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("contig     1024thr 2/CU unroll2", 3e8, hipLaunchKernelGGL((k_gridstride<1024, 2, true>), dim3(cus * 2), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("contig     1024thr 2/CU unroll4", 3e8, hipLaunchKernelGGL((k_gridstride<1024, 4, true>), dim3(cus * 2), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("gridstride 256thr 8/CU unroll2", 3e8, hipLaunchKernelGGL((k_gridstride<256, 2, false>), dim3(cus * 8), dim3(256), 0, 0, a, b, o, nvec))
    RUN("gridstride 256thr 8/CU unroll1", 3e8, hipLaunchKernelGGL((k_gridstride<256, 1, false>), dim3(cus * 8), dim3(256), 0, 0, a, b, o, nvec))
    RUN("gridstride 256thr 4/CU unroll4", 3e8, hipLaunchKernelGGL((k_gridstride<256, 4, false>), dim3(cus * 4), dim3(256), 0, 0, a, b, o, nvec))
    RUN("contig     256thr 8/CU unroll4", 3e8, hipLaunchKernelGGL((k_gridstride<256, 4, true>), dim3(cus * 8), dim3(256), 0, 0, a, b, o, nvec))
    RUN("gridstride 512thr 4/CU unroll2", 3e8, hipLaunchKernelGGL((k_gridstride<512, 2, false>), dim3(cus * 4), dim3(512), 0, 0, a, b, o, nvec))
    RUN("gridstride 1024thr 1/CU unroll2", 3e8, hipLaunchKernelGGL((k_gridstride<1024, 2, false>), dim3(cus * 1), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("gridstride 1024thr 4/CU unroll2 (oversub)", 3e8, hipLaunchKernelGGL((k_gridstride<1024, 2, false>), dim3(cus * 4), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("gridstride 1024thr 2/CU unroll1", 3e8, hipLaunchKernelGGL((k_gridstride<1024, 1, false>), dim3(cus * 2), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("dyn 1024thr 2/CU chunk 16K", 3e8, hipLaunchKernelGGL((k_dyn<1024, 1>), dim3(cus * 2), dim3(1024), 0, 0, a, b, o, nvec, ctr))
    RUN("dyn 1024thr 2/CU chunk 32K", 3e8, hipLaunchKernelGGL((k_dyn<1024, 2>), dim3(cus * 2), dim3(1024), 0, 0, a, b, o, nvec, ctr))
    RUN("dyn 1024thr 2/CU chunk 64K", 3e8, hipLaunchKernelGGL((k_dyn<1024, 4>), dim3(cus * 2), dim3(1024), 0, 0, a, b, o, nvec, ctr))
    RUN("dyn 256thr 8/CU chunk 4K", 3e8, hipLaunchKernelGGL((k_dyn<256, 1>), dim3(cus * 8), dim3(256), 0, 0, a, b, o, nvec, ctr))
    RUN("dyn 256thr 8/CU chunk 8K", 3e8, hipLaunchKernelGGL((k_dyn<256, 2>), dim3(cus * 8), dim3(256), 0, 0, a, b, o, nvec, ctr))
    RUN("dyn 512thr 4/CU chunk 8K", 3e8, hipLaunchKernelGGL((k_dyn<512, 1>), dim3(cus * 4), dim3(512), 0, 0, a, b, o, nvec, ctr))
    RUN("rotate+1  1024thr 2/CU", 3e8, hipLaunchKernelGGL((k_rotate<1024, 1>), dim3(cus * 2), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("rotate+3  1024thr 2/CU", 3e8, hipLaunchKernelGGL((k_rotate<1024, 3>), dim3(cus * 2), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("rotate+37 1024thr 2/CU", 3e8, hipLaunchKernelGGL((k_rotate<1024, 37>), dim3(cus * 2), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("rotate+1  256thr 8/CU", 3e8, hipLaunchKernelGGL((k_rotate<256, 1>), dim3(cus * 8), dim3(256), 0, 0, a, b, o, nvec))
    RUN("flat 256thr one vector per thread", 3e8, hipLaunchKernelGGL((k_flat<256>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, a, b, o, nvec))
    RUN("flat 1024thr one vector per thread", 3e8, hipLaunchKernelGGL((k_flat<1024>), dim3((unsigned)((nvec + 1023) / 1024)), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("copy gridstride 256thr 8/CU", 2e8, hipLaunchKernelGGL((k_copy<256>), dim3(cus * 8), dim3(256), 0, 0, a, o, nvec))
    RUN("copy gridstride 1024thr 2/CU", 2e8, hipLaunchKernelGGL((k_copy<1024>), dim3(cus * 2), dim3(1024), 0, 0, a, o, nvec))
    RUN("hipMemcpyDtoD", 2e8, hipMemcpyAsync(o, a, n, hipMemcpyDeviceToDevice, 0))
    return 0;
}
