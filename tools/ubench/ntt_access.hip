// Memory-access skeletons of the two passes of a 2^20-point four-step transform (1024 x 1024, 32-bit elements, batch 64)
// with NO arithmetic: what the memory system gives the access pattern itself.
//   pass 1: tile = C adjacent columns x 1024 rows (row stride 4 KiB): loads and stores are C*4-byte segments
//   pass 2: tile = C adjacent rows read contiguously, stored transposed (C*4-byte segments at 4 KiB stride)
// Each thread moves 32 elements, like the transform kernels.  Variants: C (segment width), workgroup size, XCD-aware tile
// order, non-temporal accesses, and the pair run on sub-batches small enough for the intermediate to stay in the 256 MiB
// Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/ntt_access tools/ubench/ntt_access.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
typedef long long i64;

template <int C, bool NT>
__global__ __launch_bounds__(32 * C) void k_cols(const u32 *__restrict__ in, u32 *__restrict__ out, int tiles_per_batch, int xcd)
{ // pass-1 pattern: element (row t, column c) at t * 1024 + c
    u32 vb = blockIdx.x;
    if (xcd) {
        const u32 x = vb & 7u, i = vb >> 3, per = (u32)tiles_per_batch >> 3;
        vb = (i / per) * (u32)tiles_per_batch + x * per + (i % per);
    }
    const u32 batch = vb / tiles_per_batch, tile = vb % tiles_per_batch;
    const u32 *gi = in + (i64)batch * (1 << 20) + tile * C;
    u32 *go = out + (i64)batch * (1 << 20) + tile * C;
    const int ca = threadIdx.x & (C - 1), r = threadIdx.x / C;
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void *)gi, 0, 0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)go, 0, 0xffffffffu, 0x00020000);
    const int off = (r * 1024 + ca) * 4;
    u32 v[32];
#pragma unroll
    for (int a = 0; a < 32; a++) v[a] = __builtin_amdgcn_raw_buffer_load_b32(ri, off, a * 32 * 4096, NT ? 2 : 0);
#pragma unroll
    for (int a = 0; a < 32; a++) __builtin_amdgcn_raw_buffer_store_b32(v[a] + 1u, ro, off, a * 32 * 4096, NT ? 2 : 0);
}

template <int C, bool NT>
__global__ __launch_bounds__(32 * C) void k_rows(const u32 *__restrict__ in, u32 *__restrict__ out, int tiles_per_batch, int xcd)
{ // pass-2 pattern: rows in (row c, position t at c * 1024 + t), transposed out (position t, row c at t * 1024 + c)
    u32 vb = blockIdx.x;
    if (xcd) {
        const u32 x = vb & 7u, i = vb >> 3, per = (u32)tiles_per_batch >> 3;
        vb = (i / per) * (u32)tiles_per_batch + x * per + (i % per);
    }
    const u32 batch = vb / tiles_per_batch, tile = vb % tiles_per_batch;
    const u32 *gi = in + (i64)batch * (1 << 20) + (i64)tile * C * 1024;
    u32 *go = out + (i64)batch * (1 << 20) + tile * C;
    const int cl = threadIdx.x >> 5, r = threadIdx.x & 31;   // load: position runs fastest
    const int c = threadIdx.x & (C - 1), ka = threadIdx.x / C; // store: row runs fastest
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void *)gi, 0, 0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)go, 0, 0xffffffffu, 0x00020000);
    const int ioff = (cl * 1024 + r) * 4, ooff = (ka * 1024 + c) * 4;
    u32 v[32];
#pragma unroll
    for (int a = 0; a < 32; a++) v[a] = __builtin_amdgcn_raw_buffer_load_b32(ri, ioff, a * 32 * 4, NT ? 2 : 0);
#pragma unroll
    for (int a = 0; a < 32; a++) __builtin_amdgcn_raw_buffer_store_b32(v[a] + 1u, ro, ooff, a * 32 * 4096, NT ? 2 : 0);
}

__global__ void k_copy(const uint4 *a, uint4 *o, i64 nvec)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i < nvec) o[i] = a[i];
}

template <typename F>
float timeit(F f, int iters = 20)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; i++) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main(int argc, char **argv)
{
    const int batch = argc > 1 ? atoi(argv[1]) : 64;
    const i64 n = (i64)batch << 20;
    u32 *a, *b, *c;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4);
    hipMemset(a, 1, n * 4); hipMemset(b, 2, n * 4); hipMemset(c, 3, n * 4);
    const double gb = 2.0 * n * 4 / 1e9;
    auto rep = [&](const char *name, float ms) { printf("%-58s %8.4f ms  %6.2f TB/s (read + write)\n", name, ms, gb / ms); fflush(stdout); };
    rep("flat 16-byte copy", timeit([&] { hipLaunchKernelGGL(k_copy, dim3((unsigned)(n / 4 / 256)), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, n / 4); }));
#define COLS(C, NT, X) rep("pass 1 (columns) C=" #C " nt=" #NT " xcd=" #X, timeit([&] { hipLaunchKernelGGL((k_cols<C, NT>), dim3(batch * (1024 / C)), dim3(32 * C), 0, 0, a, b, 1024 / C, X); }))
#define ROWS(C, NT, X) rep("pass 2 (rows -> transposed) C=" #C " nt=" #NT " xcd=" #X, timeit([&] { hipLaunchKernelGGL((k_rows<C, NT>), dim3(batch * (1024 / C)), dim3(32 * C), 0, 0, b, c, 1024 / C, X); }))
    COLS(8, false, 1); COLS(16, false, 0); COLS(16, false, 1); COLS(16, true, 1); COLS(32, false, 0); COLS(32, false, 1); COLS(32, true, 1);
    ROWS(8, false, 1); ROWS(16, false, 0); ROWS(16, false, 1); ROWS(16, true, 1); ROWS(32, false, 0); ROWS(32, false, 1); ROWS(32, true, 1);
    // the pair on sub-batches: intermediate `b` re-used by every sub-batch (stays in the Infinity Cache)
    for (int sub : {64, 16, 8, 4, 2}) {
        if (sub > batch) continue;
        const float ms = timeit([&] {
            for (int b0 = 0; b0 < batch; b0 += sub) {
                hipLaunchKernelGGL((k_cols<16, false>), dim3(sub * 64), dim3(512), 0, 0, a + (i64)b0 * (1 << 20), b, 64, 1);
                hipLaunchKernelGGL((k_rows<16, false>), dim3(sub * 64), dim3(512), 0, 0, b, c + (i64)b0 * (1 << 20), 64, 1);
            }
        });
        printf("pair, sub-batches of %2d transforms (intermediate %3d MiB)    %8.4f ms  = %5.3f of the one-round-trip roofline at 8 TB/s\n", sub, sub * 4, ms,
               (gb / 8.0) / ms); // gb GB at 8000 GB/s = gb / 8 ms
    }
    return 0;
}
