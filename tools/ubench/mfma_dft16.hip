// Radix-16 sub-DFT over GF(p), p < 2^23, two ways, register-resident (no memory traffic): is the matrix core worth it?
//   (a) MATRIX CORE: D = W (16 x 16) * X (16 points x 16 independent transforms) as four v_mfma_f64_16x16x4_f64; products are
//       below 2^46 and 16-term sums below 2^50, exact in fp64.  Each output is then reduced mod p in fp64 (x - floor(x / p) p:
//       multiply, floor, fma), multiplied by the inter-stage twiddle (exact: < 2^46) and reduced again -- the work a real
//       transform needs between two radix-16 stages.  The output layout (column per lane) equals the B-operand layout, so a
//       chain of LEFT multiplications needs no data movement; a real transform changes the digit between stages and would add an
//       LDS exchange per stage on top of what is timed here.
//   (b) VECTOR ALU: the radix-16 decimation-in-frequency network of gfa_ntt_m32.hip (signed Montgomery, 17 products + 64
//       add / sub per 16 points) plus the same inter-stage twiddle product.
// Prints time per point and stage.   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_dft16 tools/ubench/mfma_dft16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef int i32;
typedef unsigned int u32;

constexpr int ITERS = 2048;

__device__ __forceinline__ double redp(double x, double p, double pinv) { return __builtin_fma(-__builtin_floor(x * pinv), p, x); }

__global__ __launch_bounds__(256) void k_mfma(double *out, const double *wmat, const double *tw, double p, double pinv)
{
    const int lane = threadIdx.x & 63;
    // A operand of chunk j: W[row = lane % 16][col = 4 * (lane / 16) + j]  (matches the D layout: lane holds rows 4*(lane/16)+v)
    double a[4], t[4];
    d4 x;
    for (int j = 0; j < 4; j++) {
        a[j] = wmat[(lane & 15) * 16 + 4 * (lane >> 4) + j];
        t[j] = tw[(lane + 64 * j) & 255];
        x[j] = (double)((lane * 7 + j * 13 + blockIdx.x) % 8388593);
    }
    for (int it = 0; it < ITERS; it++) {
        d4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], x[j], acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; v++) x[v] = redp(redp(acc[v], p, pinv) * t[v], p, pinv);
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
}

__device__ __forceinline__ i32 mulhi_i(i32 a, i32 b) { return (i32)(((long long)a * b) >> 32); }
__device__ __forceinline__ i32 mulm(i32 x, i32 wm, i32 wp, i32 p)
{
    const i32 m = (i32)((u32)x * (u32)wp);
    return mulhi_i(x, wm) - mulhi_i(m, p);
}

__global__ __launch_bounds__(256) void k_valu(i32 *out, const i32 *net, const i32 *tw, i32 p)
{
    i32 v[16], t[16];
    for (int j = 0; j < 16; j++) { v[j] = (threadIdx.x * 7 + j * 13 + blockIdx.x) % 8388593; t[j] = tw[(threadIdx.x + j) & 255]; }
    for (int it = 0; it < ITERS / 4; it++) { // one iteration transforms 16 points per LANE: 4x the points of an MFMA iteration per wave
#pragma unroll
        for (int s = 3; s >= 0; s--) {
            const int half = 1 << s;
#pragma unroll
            for (int b = 0; b < 16; b += 2 * half)
#pragma unroll
                for (int j = 0; j < half; j++) {
                    const i32 u = v[b + j], x = v[b + j + half];
                    v[b + j] = u + x;
                    const int tj = j << (3 - s);
                    v[b + j + half] = tj ? mulm(u - x, net[2 * tj], net[2 * tj + 1], p) : u - x;
                }
        }
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = mulm(v[j], t[j], t[j] * 3, p);
    }
    i32 s = 0;
    for (int j = 0; j < 16; j++) s += v[j];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    const int blocks = 256 * 8;
    double *dout, *dw, *dtw;
    i32 *iout, *inet, *itw;
    hipMalloc(&dout, blocks * 256 * 8); hipMalloc(&dw, 256 * 8); hipMalloc(&dtw, 256 * 8);
    hipMalloc(&iout, blocks * 256 * 4); hipMalloc(&inet, 64 * 4); hipMalloc(&itw, 256 * 4);
    double hw[256];
    int hi[256];
    for (int i = 0; i < 256; i++) { hw[i] = (double)((i * 2654435761u) % 7340033u); hi[i] = (int)((i * 2654435761u) % 7340033u) - 3670016; }
    hipMemcpy(dw, hw, sizeof hw, hipMemcpyHostToDevice); hipMemcpy(dtw, hw, sizeof hw, hipMemcpyHostToDevice);
    hipMemcpy(inet, hi, 64 * 4, hipMemcpyHostToDevice); hipMemcpy(itw, hi, 256 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, dout, dw, dtw, 7340033.0, 1.0 / 7340033.0);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double pts_m = (double)blocks * 4 * 256 * ITERS; // 256 points per wave and iteration
    const double rate_m = pts_m / ms / 1e6;
    printf("matrix core  (4 x v_mfma_f64_16x16x4 + 2 fp64 reductions + twiddle): %8.3f ms  %7.2f Gpoint-stages/s\n", ms, rate_m);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, iout, inet, itw, 7340033);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double pts_v = (double)blocks * 256 * 16 * (ITERS / 4);
    const double rate_v = pts_v / ms / 1e6;
    printf("vector ALU   (radix-16 network, signed Montgomery + twiddle):          %8.3f ms  %7.2f Gpoint-stages/s\n", ms, rate_v);
    printf("a 2^20-point transform is 5 such stages per point: 2^20 x 64 points = %.0f us (matrix core) / %.0f us (vector ALU) of arithmetic;\n"
           "the two memory passes of the same transform take 200-240 us (tools/ubench/ntt_access.hip, profiles/r03_m32_time.txt)\n",
           67108864.0 * 5 / rate_m / 1e3, 67108864.0 * 5 / rate_v / 1e3);
    return 0;
}
