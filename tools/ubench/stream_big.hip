// 2 reads + 1 write streaming (out = a ^ b) on 1e9-byte arrays (beyond the Infinity Cache): plain vs nontemporal accesses,
// persistent grid-stride vs flat launch.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef long long i64;
template <bool NTL, bool NTS>
__global__ __launch_bounds__(1024) void k_gs(const u32x4 *a, const u32x4 *b, u32x4 *o, i64 nvec)
{
    const i64 stride = (i64)gridDim.x * 1024;
    for (i64 i = (i64)blockIdx.x * 1024 + threadIdx.x; i < nvec; i += stride) {
        const u32x4 x = NTL ? __builtin_nontemporal_load(a + i) : a[i];
        const u32x4 y = NTL ? __builtin_nontemporal_load(b + i) : b[i];
        const u32x4 r = x ^ y;
        if (NTS) __builtin_nontemporal_store(r, o + i); else o[i] = r;
    }
}
template <int THREADS, bool NT>
__global__ __launch_bounds__(THREADS) void k_gs2(const u32x4 *a, const u32x4 *b, u32x4 *o, i64 nvec)
{
    const i64 stride = (i64)gridDim.x * THREADS;
    for (i64 i = (i64)blockIdx.x * THREADS + threadIdx.x; i < nvec; i += stride) {
        const u32x4 x = NT ? __builtin_nontemporal_load(a + i) : a[i];
        const u32x4 y = NT ? __builtin_nontemporal_load(b + i) : b[i];
        const u32x4 r = x ^ y;
        if (NT) __builtin_nontemporal_store(r, o + i); else o[i] = r;
    }
}
// persistent workgroups, but each takes the NEXT unclaimed 16 KiB block in launch order of a virtual flat grid: block index
// = iteration * gridDim.x + blockIdx.x is what grid-stride does already; here the blocks of one workgroup are adjacent
template <bool NT, int RUN>
__global__ __launch_bounds__(1024) void k_runs(const u32x4 *a, const u32x4 *b, u32x4 *o, i64 nvec)
{
    const i64 nblk = (nvec + 1023) / 1024;
    for (i64 blk0 = (i64)blockIdx.x * RUN; blk0 < nblk; blk0 += (i64)gridDim.x * RUN) {
#pragma unroll
        for (int r = 0; r < RUN; r++) {
            const i64 i = (blk0 + r) * 1024 + threadIdx.x;
            if (i < nvec) {
                const u32x4 x = NT ? __builtin_nontemporal_load(a + i) : a[i];
                const u32x4 y = NT ? __builtin_nontemporal_load(b + i) : b[i];
                const u32x4 v = x ^ y;
                if (NT) __builtin_nontemporal_store(v, o + i); else o[i] = v;
            }
        }
    }
}
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_flat(const u32x4 *a, const u32x4 *b, u32x4 *o, i64 nvec)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i < nvec) {
        const u32x4 x = NTL ? __builtin_nontemporal_load(a + i) : a[i];
        const u32x4 y = NTL ? __builtin_nontemporal_load(b + i) : b[i];
        const u32x4 r = x ^ y;
        if (NTS) __builtin_nontemporal_store(r, o + i); else o[i] = r;
    }
}
template <typename F>
float timeit(F f)
{
    f(); (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 10; i++) f();
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}
int main()
{
    const i64 n = 1000000000, nvec = n / 16;
    u32x4 *a, *b, *o;
    (void)hipMalloc(&a, n); (void)hipMalloc(&b, n); (void)hipMalloc(&o, n);
    (void)hipMemset(a, 1, n); (void)hipMemset(b, 2, n);
#define RUN(name, ...) { float ms = timeit([&]() { __VA_ARGS__; }); printf("%-44s %8.1f us  %6.3f TB/s\n", name, ms * 1e3, 3.0 * n / ms / 1e9); }
    RUN("gridstride plain", hipLaunchKernelGGL((k_gs<false, false>), dim3(512), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("gridstride nt loads", hipLaunchKernelGGL((k_gs<true, false>), dim3(512), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("gridstride nt stores", hipLaunchKernelGGL((k_gs<false, true>), dim3(512), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("gridstride nt loads+stores", hipLaunchKernelGGL((k_gs<true, true>), dim3(512), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("gridstride 256thr x 2048 plain", hipLaunchKernelGGL((k_gs2<256, false>), dim3(2048), dim3(256), 0, 0, a, b, o, nvec))
    RUN("gridstride 256thr x 2048 nt", hipLaunchKernelGGL((k_gs2<256, true>), dim3(2048), dim3(256), 0, 0, a, b, o, nvec))
    RUN("gridstride 1024thr x 256 (1/CU) nt", hipLaunchKernelGGL((k_gs2<1024, true>), dim3(256), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("gridstride 1024thr x 1024 (oversub) nt", hipLaunchKernelGGL((k_gs2<1024, true>), dim3(1024), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("gridstride 1024thr x 4096 (oversub) nt", hipLaunchKernelGGL((k_gs2<1024, true>), dim3(4096), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("runs of 4 blocks, 512 WGs nt", hipLaunchKernelGGL((k_runs<true, 4>), dim3(512), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("runs of 4 blocks, 4096 WGs nt", hipLaunchKernelGGL((k_runs<true, 4>), dim3(4096), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("flat 1024thr nt", hipLaunchKernelGGL((k_gs2<1024, true>), dim3((unsigned)((nvec + 1023) / 1024)), dim3(1024), 0, 0, a, b, o, nvec))
    RUN("flat plain", hipLaunchKernelGGL((k_flat<false, false>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, a, b, o, nvec))
    RUN("flat nt loads+stores", hipLaunchKernelGGL((k_flat<true, true>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, a, b, o, nvec))
    RUN("flat nt stores", hipLaunchKernelGGL((k_flat<false, true>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, a, b, o, nvec))
    { float ms = timeit([&]() { (void)hipMemcpyAsync(o, a, n, hipMemcpyDeviceToDevice, 0); }); printf("%-44s %8.1f us  %6.3f TB/s (2 B per byte)\n", "hipMemcpyDtoD", ms * 1e3, 2.0 * n / ms / 1e9); }
    return 0;
}
