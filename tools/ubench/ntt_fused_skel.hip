// Skeleton (no arithmetic) of an XCD-fused 2^20-point four-step transform: the 32 workgroups that land on one XCD
// (workgroup b runs on XCD b % 8) own one whole transform at a time.  Pass 1 reads the input from HBM and writes the
// intermediate to a 4 MiB scratch that only this XCD touches (its own L2), an XCD-wide barrier, pass 2 reads the scratch and
// writes the output.  Question answered: does the intermediate stay on chip, i.e. does the pair run in the time of ONE
// round trip through HBM?   Compare tools/ubench/ntt_access (two separate launches: 0.20-0.22 ms for 64 transforms).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/ntt_fused_skel tools/ubench/ntt_fused_skel.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
typedef long long i64;

__device__ __forceinline__ bool xcd_barrier(unsigned *ctr, unsigned target, int *err)
{ // all 32 workgroups of this XCD; bounded spin so that a scheduling surprise ends in an error flag, not a hang
    __syncthreads();
    if (threadIdx.x == 0) {
        // no release fence: an agent-scope release writes the whole L2 back (the XCDs' L2s are not coherent with each other);
        // __syncthreads() has waited for this workgroup's stores to reach the L2, which is all the other CUs of the XCD need
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 22)) { *err = 1; break; }
        }
    }
    __syncthreads();
    return true;
}

template <int AUXIO, int AUXSC>
__global__ __launch_bounds__(1024) void k_fused(const u32 *__restrict__ in, u32 *__restrict__ out, u32 *scratch, unsigned *counters,
                                                int per_xcd, int *err)
{
    extern __shared__ unsigned char smem[]; // only there to hold the workgroup count per CU at one
    const int xcd = blockIdx.x & 7, member = blockIdx.x >> 3;
    u32 *sc = scratch + (i64)xcd * (1 << 20);
    unsigned *ctr = counters + xcd * 64;
    const int tid = threadIdx.x;
    const int ca = tid & 31, r = tid >> 5;  // pass 1: 32 columns x 1024 rows, column runs fastest
    const int cl = tid >> 5, rr = tid & 31; // pass 2 load: 32 rows, position runs fastest
    const int c = tid & 31, ka = tid >> 5;  // pass 2 store: row runs fastest
    const int off1 = (r * 1024 + member * 32 + ca) * 4;
    const int ioff2 = ((member * 32 + cl) * 1024 + rr) * 4, ooff2 = (ka * 1024 + member * 32 + c) * 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)sc, 0, 0xffffffffu, 0x00020000);
    u32 v[32];
    {
        const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (i64)(xcd * per_xcd) * (1 << 20)), 0, 0xffffffffu, 0x00020000);
#pragma unroll
        for (int a = 0; a < 32; a++) v[a] = __builtin_amdgcn_raw_buffer_load_b32(ri, off1, a * 32 * 4096, AUXIO);
    }
    for (int it = 0; it < per_xcd; it++) {
        const i64 b = (i64)xcd * per_xcd + it;
        if (it) xcd_barrier(ctr, 32u * (2 * it), err); // everyone is done reading the scratch of the previous transform
#pragma unroll
        for (int a = 0; a < 32; a++) __builtin_amdgcn_raw_buffer_store_b32(v[a] + 1u, rs, off1, a * 32 * 4096, AUXSC);
        xcd_barrier(ctr, 32u * (2 * it + 1), err);
        u32 w[32];
#pragma unroll
        for (int a = 0; a < 32; a++) w[a] = __builtin_amdgcn_raw_buffer_load_b32(rs, ioff2, a * 32 * 4, AUXSC);
        if (it + 1 < per_xcd) { // next transform's input: in flight while the others finish with the scratch
            const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (b + 1) * (1 << 20)), 0, 0xffffffffu, 0x00020000);
#pragma unroll
            for (int a = 0; a < 32; a++) v[a] = __builtin_amdgcn_raw_buffer_load_b32(ri, off1, a * 32 * 4096, AUXIO);
        }
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(out + b * (1 << 20)), 0, 0xffffffffu, 0x00020000);
#pragma unroll
        for (int a = 0; a < 32; a++) __builtin_amdgcn_raw_buffer_store_b32(w[a] + 1u, ro, ooff2, a * 32 * 4096, AUXIO);
    }
}

int main(int argc, char **argv)
{
    const int batch = argc > 1 ? atoi(argv[1]) : 64;
    const i64 n = (i64)batch << 20;
    u32 *a, *o, *sc;
    unsigned *ctr;
    int *err;
    hipMalloc(&a, n * 4); hipMalloc(&o, n * 4); hipMalloc(&sc, (size_t)8 << 22); hipMalloc(&ctr, 8 * 64 * 4); hipMalloc(&err, 4);
    hipMemset(a, 1, n * 4); hipMemset(o, 0, n * 4); hipMemset(err, 0, 4);
    auto run = [&](auto kern, const char *name) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9f;
        for (int rep = 0; rep < 12; rep++) {
            hipMemsetAsync(ctr, 0, 8 * 64 * 4, 0);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 100 * 1024, 0, a, o, sc, ctr, batch / 8, err);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2 && ms < best) best = ms;
        }
        int herr = 0;
        hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
        printf("%-44s %8.4f ms  = %5.3f of the 8 B/point roofline at 8 TB/s%s\n", name, best, (2.0 * n * 4 / 8e9) / best, herr ? "  [BARRIER TIMEOUT]" : "");
        fflush(stdout);
    };
    run(k_fused<0, 0>, "fused skeleton, plain accesses (L1 may be stale)");
    run(k_fused<0, 16>, "fused skeleton, plain i/o, sc1 scratch");
    run(k_fused<2, 16>, "fused skeleton, nt i/o, sc1 scratch");
    run(k_fused<2, 17>, "fused skeleton, nt i/o, sc0 sc1 scratch");
    run(k_fused<2, 1>, "fused skeleton, nt i/o, sc0 scratch");
    // spot check of the data path: out = in + 2 everywhere (in bytes 0x01010101)
    u32 h[4];
    hipMemcpy(h, o + 12345, 16, hipMemcpyDeviceToHost);
    printf("spot check: %08x (expect 01010103)\n", h[0]);
    return 0;
}
