// Go / no-go micro-benchmark for VERDICT r04 item 3: a traffic-reducing INTERMEDIATE for the two-pass 2^20-point transform
// (1024 x 1024, batch 64), p < 2^24.  Memory-access skeletons only (no arithmetic beyond the packing itself), next to
// tools/ubench/ntt_access.hip whose two kernels are repeated here as the baseline:
//   V0  today's layout: 32-bit intermediate at [k1][column]; pass 1 stores and pass 2 loads/stores in 64-byte pieces / whole rows
//   V1  32-bit intermediate, TILE-MAJOR [column tile][k1][16 columns]: a pass-1 workgroup writes ONE contiguous 64 KiB block
//       (every store a full 256-byte wave access), a pass-2 workgroup (16 rows) gathers 64 chunks of 1 KiB
//   V2  24-bit PACKED tile-major intermediate (3 bytes per point: 14 instead of 16 B/point through the fabric): pass 1 repacks four
//       values into three dwords inside each quad of lanes (DPP) and stores 192 contiguous bytes per wave and step; pass 2 reads
//       each value with one UNALIGNED dword load (byte offset 3e) and masks it
//   V3  as V2 but pass 1 stores each value as a 16-bit + an 8-bit store (no repacking arithmetic, twice the store instructions)
// Each thread moves 32 elements, 512-thread workgroups, XCD-aware tile order, non-temporal last pass as in the product.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/ntt_packed tools/ubench/ntt_packed.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
typedef long long i64;

constexpr int C = 16, T = 512, N1 = 1024;

__device__ __forceinline__ u32 remap(u32 vb, int tiles_per_batch)
{ // XCD x owns a contiguous range of tiles of every batch item (as ntt_m32_kernel, tile_order 1)
    const u32 x = vb & 7u, i = vb >> 3, per = (u32)tiles_per_batch >> 3;
    return (i / per) * (u32)tiles_per_batch + x * per + (i % per);
}

// ---- pass 1: C adjacent columns x 1024 rows in (row stride 4 KiB) ----
// LAYOUT 0: out at [row][column] (64-byte pieces); 1: tile-major 32-bit; 2: tile-major packed, quad repack; 3: packed, short + byte stores
template <int LAYOUT>
__global__ __launch_bounds__(T) void k_pass1(const u32 *__restrict__ in, unsigned char *__restrict__ out)
{
    const u32 vb = remap(blockIdx.x, N1 / C);
    const u32 batch = vb / (N1 / C), tile = vb % (N1 / C);
    const u32 *gi = in + (i64)batch * (1 << 20) + tile * C;
    const int ca = threadIdx.x & (C - 1), r = threadIdx.x / C;
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void *)gi, 0, 0xffffffffu, 0x00020000);
    const int off = (r * 1024 + ca) * 4;
    u32 v[32];
#pragma unroll
    for (int a = 0; a < 32; a++) v[a] = __builtin_amdgcn_raw_buffer_load_b32(ri, off, a * 32 * 4096, 0) & 0xffffffu;
    if (LAYOUT == 0) {
        u32 *go = (u32 *)out + (i64)batch * (1 << 20) + tile * C;
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)go, 0, 0xffffffffu, 0x00020000);
#pragma unroll
        for (int a = 0; a < 32; a++) __builtin_amdgcn_raw_buffer_store_b32(v[a] + 1u, ro, off, a * 32 * 4096, 0);
    } else if (LAYOUT == 1) {
        // element (row t = r + 32 a, column ca) of the tile at word t * 16 + ca = tid + 512 a of the tile's 64 KiB block
        u32 *go = (u32 *)out + ((i64)batch * (N1 / C) + tile) * (N1 * C);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)go, 0, N1 * C * 4, 0x00020000);
#pragma unroll
        for (int a = 0; a < 32; a++) __builtin_amdgcn_raw_buffer_store_b32(v[a] + 1u, ro, (int)threadIdx.x * 4, a * T * 4, 0);
    } else if (LAYOUT == 2) {
        // quad q = lanes 4q .. 4q+3 holds elements e0..e3 (24 bits each) -> dwords d0 = e0 | e1 << 24, d1 = e1 >> 8 | e2 << 16,
        // d2 = e2 >> 16 | e3 << 8; lane j < 3 of the quad forms and stores d_j: one DPP move (the neighbour's value), one shift, one
        // shift-or
        unsigned char *go = out + ((i64)batch * (N1 / C) + tile) * (N1 * C * 3);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)go, 0, N1 * C * 3, 0x00020000);
        const int j = threadIdx.x & 3;
        const int doff = ((int)(threadIdx.x >> 2) * 3 + j) * 4; // dword index 3 * quad + j inside the step's 1536 bytes
        const u32 sh_lo = 8u * j, sh_hi = 24u - 8u * j;
#pragma unroll
        for (int a = 0; a < 32; a++) {
            const u32 e = v[a] + 1u;
            const u32 nb = (u32)__builtin_amdgcn_mov_dpp((int)e, 0x39, 0xf, 0xf, true); // quad_perm [1,2,3,0]: lane j gets lane j+1's value
            const u32 d = (e >> sh_lo) | (nb << sh_hi);
            if (j < 3) __builtin_amdgcn_raw_buffer_store_b32(d, ro, doff, a * (T * 3), 0);
        }
    } else {
        unsigned char *go = out + ((i64)batch * (N1 / C) + tile) * (N1 * C * 3);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)go, 0, N1 * C * 3, 0x00020000);
#pragma unroll
        for (int a = 0; a < 32; a++) {
            const u32 e = v[a] + 1u;
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)e, ro, (int)threadIdx.x * 3, a * (T * 3), 0);
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(e >> 16), ro, (int)threadIdx.x * 3 + 2, a * (T * 3), 0);
        }
    }
}

// ---- pass 2: C adjacent rows, stored transposed (64-byte pieces at 4 KiB stride, non-temporal) ----
template <int LAYOUT>
__global__ __launch_bounds__(T) void k_pass2(const unsigned char *__restrict__ in, u32 *__restrict__ out)
{
    const u32 vb = remap(blockIdx.x, N1 / C);
    const u32 batch = vb / (N1 / C), tile = vb % (N1 / C);
    u32 *go = out + (i64)batch * (1 << 20) + tile * C;
    const int cl = threadIdx.x >> 5, r = threadIdx.x & 31;   // load: position runs fastest
    const int c = threadIdx.x & (C - 1), ka = threadIdx.x / C; // store: row runs fastest
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)go, 0, 0xffffffffu, 0x00020000);
    const int ooff = (ka * 1024 + c) * 4;
    u32 v[32];
    if (LAYOUT == 0) {
        const u32 *gi = (const u32 *)in + (i64)batch * (1 << 20) + (i64)tile * C * 1024;
        const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void *)gi, 0, 0xffffffffu, 0x00020000);
        const int ioff = (cl * 1024 + r) * 4;
#pragma unroll
        for (int a = 0; a < 32; a++) v[a] = __builtin_amdgcn_raw_buffer_load_b32(ri, ioff, a * 32 * 4, 2);
    } else if (LAYOUT == 1) {
        // row k1 = tile * 16 + cl, column r + 32 a: column tile 2 a + (r >> 4), inside it word k1 * 16 + (r & 15)
        const u32 *gi = (const u32 *)in + (i64)batch * (1 << 20);
        const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void *)gi, 0, 1 << 22, 0x00020000);
        const int ioff = ((r >> 4) * (N1 * C) + ((int)tile * 16 + cl) * 16 + (r & 15)) * 4;
#pragma unroll
        for (int a = 0; a < 32; a++) v[a] = __builtin_amdgcn_raw_buffer_load_b32(ri, ioff, a * 2 * (N1 * C) * 4, 2);
    } else {
        const unsigned char *gi = in + (i64)batch * (3 << 20);
        const int ioff = ((r >> 4) * (N1 * C) + ((int)tile * 16 + cl) * 16 + (r & 15)) * 3;
#pragma unroll
        for (int a = 0; a < 32; a++) {
            u32 w;
            __builtin_memcpy(&w, gi + ioff + a * 2 * (N1 * C) * 3, 4); // one unaligned global_load_dword
            v[a] = w & 0xffffffu;
        }
    }
#pragma unroll
    for (int a = 0; a < 32; a++) __builtin_amdgcn_raw_buffer_store_b32(v[a] + 1u, ro, ooff, a * 32 * 4096, 2);
}

template <typename F>
float timeit(F f, int iters = 30)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 200; i++) f(); // ~50 ms: the clocks come back up (profiles/r04_bench_clock_ramp.txt)
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
    u32 *a, *c;
    unsigned char *b;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4 + 64); hipMalloc(&c, n * 4);
    hipMemset(a, 1, n * 4); hipMemset(b, 2, n * 4 + 64); hipMemset(c, 3, n * 4);
    const dim3 grid(batch * (N1 / C)), blk(T);
    const double alg_ms = 2.0 * n * 4 / 8e12 * 1e3; // one round trip of the data at 8 TB/s
    const char *names[4] = {"V0 32-bit [k1][column] (today)", "V1 32-bit tile-major", "V2 24-bit packed tile-major, quad repack", "V3 24-bit packed tile-major, short+byte stores"};
    auto run = [&](int v, float &m1, float &m2, float &pair) {
        auto p1 = [&] {
            if (v == 0) hipLaunchKernelGGL((k_pass1<0>), grid, blk, 0, 0, a, b);
            else if (v == 1) hipLaunchKernelGGL((k_pass1<1>), grid, blk, 0, 0, a, b);
            else if (v == 2) hipLaunchKernelGGL((k_pass1<2>), grid, blk, 0, 0, a, b);
            else hipLaunchKernelGGL((k_pass1<3>), grid, blk, 0, 0, a, b);
        };
        auto p2 = [&] {
            if (v == 0) hipLaunchKernelGGL((k_pass2<0>), grid, blk, 0, 0, b, c);
            else if (v == 1) hipLaunchKernelGGL((k_pass2<1>), grid, blk, 0, 0, b, c);
            else hipLaunchKernelGGL((k_pass2<2>), grid, blk, 0, 0, b, c);
        };
        m1 = timeit(p1); m2 = timeit(p2);
        pair = timeit([&] { p1(); p2(); });
    };
    for (int rep = 0; rep < 2; rep++)
        for (int v = 0; v < 4; v++) {
            float m1, m2, pr;
            run(v, m1, m2, pr);
            printf("%-52s pass 1 %7.4f ms  pass 2 %7.4f ms  pair %7.4f ms = %5.3f of the one-round-trip roofline\n", names[v], m1, m2, pr, alg_ms / pr);
            fflush(stdout);
        }
    return 0;
}
